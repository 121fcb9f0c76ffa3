"""TEST INFRASTRUCTURE ONLY: CPU oracle of the robust_cvd geometric-consistency optimizer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Parity unpinned (no reference tests / golden vectors exist; the reference cannot be built here) --
see oracle/README.md.
"""
