"""CPU oracle (test infrastructure only) - see hpvg_oracle.py."""
