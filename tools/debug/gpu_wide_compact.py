"""A/B of the compact LDS carve-up of the wide kernel (ALTRO_WIDE_COMPACT = 0 / 32 / 48, read at create)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tools")); sys.path.insert(0, R)
import bench_wide as bw
for n, B in ((32, 8192), (24, 8192), (48, 2048), (48, 8192)):
    for c in ("0", "48"):
        os.environ["ALTRO_WIDE_COMPACT"] = c
        print("compact", c, end=" ")
        bw.sweep(n, 4, 50, B, 10)
