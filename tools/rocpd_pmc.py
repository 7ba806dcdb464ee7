"""Per-kernel average of the PMC counters in a rocprofv3 rocpd database."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
rows = c.execute("select * from pmc_events").fetchall()
idx = {n: i for i, n in enumerate(cols)}
agg = {}
for r in rows:
    name = re.sub(r"\(.*$", "", str(r[idx.get("name", idx.get("kernel_name", 0))])).replace("void ", "")[:90]
    cn = r[idx["counter_name"]] if "counter_name" in idx else r[idx["pmc_name"]]
    v = r[idx["value"]] if "value" in idx else r[idx["counter_value"]]
    a = agg.setdefault((name, cn), [0.0, 0])
    a[0] += float(v); a[1] += 1
names = sorted({k[0] for k in agg})
ctrs = sorted({k[1] for k in agg})
print("%-92s %6s " % ("kernel", "calls") + " ".join("%16s" % x[:16] for x in ctrs))
for n in names:
    calls = max(agg[(n, x)][1] for x in ctrs if (n, x) in agg)
    print("%-92s %6d " % (n, calls) + " ".join("%16.1f" % (agg[(n, x)][0] / agg[(n, x)][1]) if (n, x) in agg else " " * 16 for x in ctrs))
