"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel table
(calls, total, avg, min, max, share) — the text committed under profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
    name = name.replace("void ", "")
    return name[:110]


def main(path, skip_first=0):
    c = sqlite3.connect(path)
    rows = c.execute("select k.display_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, k.arch_vgpr_count, "
                     "k.accum_vgpr_count, d.group_segment_size, d.grid_size_y from rocpd_kernel_dispatch d join "
                     "rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start").fetchall()
    agg = {}
    t0, t1 = rows[0][1], rows[-1][2]
    for name, s, e, gx, wx, vg, ag, lds, gy in rows:
        # one line per (kernel, launch grid): the templated GEMM serves several stages of a forward
        a = agg.setdefault((short(name), gx, gy), [0, 0, 10 ** 18, 0, gx, wx, vg, ag, lds])
        dur = e - s
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur); a[4] = gx
    tot = sum(a[1] for a in agg.values())
    print("# %d dispatches, %d distinct kernels, GPU-busy %.3f ms over a %.3f ms window" %
          (len(rows), len(agg), tot / 1e6, (t1 - t0) / 1e6))
    print("%-100s %7s %10s %9s %9s %9s %6s %9s %6s %5s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                                "pct", "grid_x", "grid_y", "vgpr", "lds"))
    for (name, gx, gy), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-100s %7d %10.1f %9.2f %9.2f %9.2f %6.2f %9d %6d %5d %6d" % (name[:100], a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                                           a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot, gx, gy,
                                                                           a[6] + a[7], a[8]))


if __name__ == "__main__":
    main(sys.argv[1])
