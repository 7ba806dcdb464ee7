"""kernel durations and the idle time in front of each kernel from a rocprofv3 --kernel-trace database of
tools/exp/handover_trace.py: python tools/exp/handover_gaps.py <results.db> [launches per forward = 7]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 7
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute("select k.kernel_name, d.start, d.end from %s d join %s k on d.kernel_id=k.id order by d.start" % (kd, ks)).fetchall()
rows = rows[per * 60:-per * 5]
dur, gap = collections.OrderedDict(), collections.OrderedDict()
for i, (nm, s, e) in enumerate(rows):
    key = (i % per, nm[:44])
    dur.setdefault(key, []).append(e - s)
    if i:
        gap.setdefault(key, []).append(s - rows[i - 1][2])
print("span per forward %.1f us over %d forwards" % ((rows[-1][2] - rows[0][1]) / (len(rows) / per) / 1e3, len(rows) // per))
for key in dur:
    g = gap.get(key, [0])
    print("  %-46s dur %6.2f us   idle before %6.2f us" % (key[1], sum(dur[key]) / len(dur[key]) / 1e3, sum(g) / len(g) / 1e3))
