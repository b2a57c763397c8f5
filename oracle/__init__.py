"""CPU restatement of the reference path: TEST INFRASTRUCTURE ONLY (see oracle/tvts_oracle.py header)."""
