"""CPU oracle (test infrastructure only; PARITY UNPINNED -- see attn_pool_oracle.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
