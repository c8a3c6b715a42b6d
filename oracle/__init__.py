"""ORACLE — CPU restatement of the reference's algorithm for the hot path (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
omg_amd never does (tests/test_layout.py enforces it).  See each module's header for the
reference file:line it follows and for what pins it.
"""
