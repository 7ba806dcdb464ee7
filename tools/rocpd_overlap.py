"""Per training step of a rocprofv3 rocpd database (steps end at k_adam): start / end of the collective's kernels (names with
nccl / rccl) relative to the conv layers' backward (first k_bn_csr* launch behind the first collective .. last launch in
front of k_adam)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select k.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k "
                 "on d.kernel_id = k.id order by d.start").fetchall()
names = [re.sub(r"\(.*$", "", str(r[0]).replace("(anonymous namespace)::", "")) for r in rows]
adam = [i for i, n in enumerate(names) if "k_adam" in n]
coll = lambda n: ("nccl" in n.lower()) or ("rccl" in n.lower())
print("# %d dispatches, %d steps; collective kernel names: %s" % (len(rows), len(adam), sorted({n[:60] for n in names if coll(n)})))
ok = tot = 0
for a, b in zip([-1] + adam[:-1], adam):
    seg = list(range(a + 1, b + 1))
    cs = [i for i in seg if coll(names[i])]
    if not cs:
        continue
    first = cs[0]
    conv = [i for i in seg if i > first and not coll(names[i]) and "k_adam" not in names[i]]
    bwd_end = max(rows[i][2] for i in conv) if conv else rows[first][2]
    bwd_start = min(rows[i][1] for i in conv) if conv else rows[first][1]
    tot += 1
    under = rows[first][1] < bwd_end
    ok += under
    print("step: head all-reduce start %+8.1f us, end %+8.1f us relative to the conv backward's first launch; conv backward lasts %7.1f us "
          "(%d launches); %d collective kernels; starts under the conv backward: %s"
          % ((rows[first][1] - bwd_start) / 1e3, (rows[first][2] - bwd_start) / 1e3, (bwd_end - bwd_start) / 1e3, len(conv), len(cs), under))
print("# head bucket's all-reduce started before the end of the conv backward in %d of %d steps" % (ok, tot))
