"""TEST-ONLY stand-in for Biopython (only `Bio.Seq.Seq.reverse_complement`)."""
