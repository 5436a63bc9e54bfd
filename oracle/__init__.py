"""ORACLE — CPU restatements of the reference's hot-path algorithms.  Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path."""
