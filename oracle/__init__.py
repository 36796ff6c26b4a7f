"""CPU oracle of the reference's learning path.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
