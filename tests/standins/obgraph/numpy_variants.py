class NumpyVariants:  # TEST-ONLY stand-in, import target only
    pass
