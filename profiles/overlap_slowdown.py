"""Per build kernel: average duration when running alone vs while a k_track launch is in flight
(rocprofv3 --kernel-trace rocpd database of the double-buffered bench).  usage: overlap_slowdown.py <db>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
tr = [(s, e) for n, s, e in rows if "k_track" in n]
acc = collections.OrderedDict()
for n, s, e in rows:
    if "k_track" in n or "rocclr" in n or "at::" in n:
        continue
    key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    ov = any(s < te and e > ts for ts, te in tr)
    acc.setdefault(key, [[], []])[1 if ov else 0].append((e - s) / 1e3)
print("k_track avg %.0f us over %d launches" % (sum(e - s for s, e in tr) / len(tr) / 1e3, len(tr)))
fmt = lambda v: ("%7.1f (n=%d)" % (sum(v) / len(v), len(v))) if v else "      -"
for k, (iso, ov) in acc.items():
    print("  %-24s alone %s   overlapped %s" % (k, fmt(iso), fmt(ov)))
