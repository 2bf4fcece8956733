class VariantToNodes:  # TEST-ONLY stand-in, import target only
    pass
