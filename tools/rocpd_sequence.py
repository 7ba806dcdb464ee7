"""Ordered kernel sequence of the LAST step of a rocprofv3 rocpd database: the launches between the last two occurrences
of a marker kernel (default k_adam), with start offset, duration and the gap to the previous kernel's end."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
    return name.replace("void ", "")[:90]


def main(path, marker="k_adam"):
    c = sqlite3.connect(path)
    rows = c.execute("select k.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch d "
                     "join rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start").fetchall()
    idx = [i for i, r in enumerate(rows) if short(r[0]).startswith(marker)]
    lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
    t0, prev = rows[lo][1], rows[lo][1]
    print("# %d launches, %.1f us wall, %.1f us busy" % (hi - lo, (rows[hi - 1][2] - t0) / 1e3,
                                                         sum(r[2] - r[1] for r in rows[lo:hi]) / 1e3))
    for name, s, e, gx, gy in rows[lo:hi]:
        print("%9.1f %8.2f %6.2f  %-90s %8d %5d" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(name), gx, gy))
        prev = e


if __name__ == "__main__":
    main(*sys.argv[1:])
