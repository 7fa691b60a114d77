"""Prints the kernel timeline of the last few bench steps from a rocprofv3 rocpd database
(--kernel-trace): per kernel start offset, duration and queue, so that the overlap of the
build stream and the tracker stream can be read off.  usage: timeline_rocpd.py <db> [n_tail_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ntail = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(db.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
# anchor: the k_track launches
tr = [i for i, r in enumerate(rows) if "k_track" in r[0]]
print("kernels %d, k_track launches %d" % (len(rows), len(tr)))
if len(tr) > 8:
    d = [(rows[i][2] - rows[i][1]) / 1e3 for i in tr]
    gaps = [(rows[tr[j + 1]][1] - rows[tr[j]][1]) / 1e3 for j in range(len(tr) - 1)]
    print("k_track durations (us), last 24:", " ".join("%.0f" % x for x in d[-24:]))
    print("k_track start-to-start (us), last 24:", " ".join("%.0f" % x for x in gaps[-24:]))
sel = rows[-ntail:]
t0 = sel[0][1]
for name, st, en, q, sid in sel:
    short = name.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[-28:]
    print("%9.1f %8.1f  q%-3s s%-3s %s" % ((st - t0) / 1e3, (en - st) / 1e3, q, sid, short))
