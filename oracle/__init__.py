"""CPU oracle of the reference hot path: TEST INFRASTRUCTURE, not product code (see oracle/gki_oracle.c)."""
