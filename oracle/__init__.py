"""CPU oracle -- TEST INFRASTRUCTURE ONLY ("parity unpinned" by the reference; see aslam_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
