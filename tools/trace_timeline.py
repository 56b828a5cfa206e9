"""Print a window of the kernel timeline of a rocprofv3 --kernel-trace run (rocpd sqlite output).
Usage: python tools/trace_timeline.py <results.db> <kernel-substring> <nth-from-end> <window_ms>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.grid_size_y from {kd} d join {ks} s "
                       f"on d.kernel_id=s.id order by d.start"))


def short(n):
    m = re.search(r'feasthip\d+(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:24]


t0 = rows[0][1]
anchor = [s - t0 for n, s, e, q, gx, gy in rows if sys.argv[2] in n][-int(sys.argv[3])]
win = float(sys.argv[4]) * 1e6
for n, s, e, q, gx, gy in rows:
    if anchor <= s - t0 < anchor + win:
        print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f}us q{q} {short(n):16s} {gx}x{gy}")
