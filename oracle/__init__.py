"""CPU checkers for the tiny-llm hot path.  TEST INFRASTRUCTURE ONLY — see tiny_oracle.py's header."""
