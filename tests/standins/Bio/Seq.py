_COMP = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")


class Seq(str):
    """TEST-ONLY stand-in: `str(Seq(s).reverse_complement())`."""

    def reverse_complement(self):
        return Seq(str(self).translate(_COMP)[::-1])
