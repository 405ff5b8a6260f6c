"""CPU oracle of the MDE hot path -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py)."""
