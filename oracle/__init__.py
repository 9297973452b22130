"""ORACLE — test infrastructure only (see oracle/orb_extractor_oracle.cpp header).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package.
"""
