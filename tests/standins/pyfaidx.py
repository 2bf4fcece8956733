class Fasta:  # TEST-ONLY stand-in, import target only
    def __init__(self, *a, **k):
        raise NotImplementedError("pyfaidx stand-in")
