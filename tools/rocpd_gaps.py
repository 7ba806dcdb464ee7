"""Idle time between consecutive kernel dispatches in the steady-state tail of a rocprofv3 rocpd trace."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 26       # dispatches per forward
rows = c.execute("select start, end, name from kernels order by start").fetchall()
tail = rows[-per * 50:]
busy = sum(e - s for s, e, _ in tail)
gaps = [max(0, tail[i + 1][0] - tail[i][1]) for i in range(len(tail) - 1)]
span = tail[-1][1] - tail[0][0]
print("last %d dispatches: span %.1f us, busy %.1f us (%.1f%%), idle gaps %.1f us; per forward: span %.1f busy %.1f gap %.1f"
      % (len(tail), span / 1e3, busy / 1e3, 100.0 * busy / span, sum(gaps) / 1e3, span / 50e3, busy / 50e3, sum(gaps) / 50e3))
g = sorted(gaps)
print("gap percentiles (us): p50 %.2f p90 %.2f p99 %.2f max %.2f" % (g[len(g) // 2] / 1e3, g[int(len(g) * .9)] / 1e3,
                                                                       g[int(len(g) * .99)] / 1e3, g[-1] / 1e3))
