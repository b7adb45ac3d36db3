"""CPU oracle for the COBS classic-index query path.  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
