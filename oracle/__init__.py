"""CPU oracle (test infrastructure only -- see swinir_oracle.py header)."""
