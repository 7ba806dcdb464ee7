"""Per-(kernel, launch grid) average of the PMC counters in a rocprofv3 rocpd database.
The same templated GEMM kernel serves several stages of a forward; the grid separates them
(e.g. the fusion GEMM + per-proposal max is k_gemm_nt<64,64,32> on a (157*256) x 16 grid at cfg 2)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
grid = {r[0]: (r[1], r[2]) for r in c.execute("select dispatch_id, grid_x, grid_y from kernels")}
agg = {}
for name, did, cn, v in c.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
    name = re.sub(r"\(.*$", "", str(name).replace("(anonymous namespace)::", "")).replace("void ", "")[:80]
    gx, gy = grid.get(did, (0, 0))
    a = agg.setdefault((name, gx, gy, cn), [0.0, 0])
    a[0] += float(v)
    a[1] += 1
keys = sorted({k[:3] for k in agg})
ctrs = sorted({k[3] for k in agg})
print("%-82s %9s %5s %6s " % ("kernel", "grid_x", "gy", "calls") + " ".join("%16s" % x[:16] for x in ctrs))
for k in keys:
    calls = max(agg[k + (x,)][1] for x in ctrs if k + (x,) in agg)
    print("%-82s %9d %5d %6d " % (k[0], k[1], k[2], calls) +
          " ".join("%16.1f" % (agg[k + (x,)][0] / agg[k + (x,)][1]) if k + (x,) in agg else " " * 16 for x in ctrs))
