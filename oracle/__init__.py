"""CPU oracle (test infrastructure only; see oracle/hero_oracle.py for who may import it)."""
