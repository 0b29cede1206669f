"""CPU oracle -- test infrastructure only (see lbm_oracle.c)."""
