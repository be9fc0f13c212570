"""CPU oracle for the minppo hot path.  TEST INFRASTRUCTURE — see the module headers.

Importable only from tests/, `__graft_entry__.smoke()` and bench.py's cpu_baseline leg.
"""
