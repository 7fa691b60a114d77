"""Do consecutive k_track launches overlap in time?  usage: track_overlap.py <rocpd db>  (rocprofv3 --kernel-trace)"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
tr = [(st, en, q, sid) for n, st, en, q, sid in rows if "k_track" in n and "gate" not in n]
print("k_track launches:", len(tr))
ov = 0
for i in range(1, len(tr)):
    gap = (tr[i][0] - tr[i - 1][1]) / 1e3
    if gap < 0:
        ov += 1
    if 8 <= i < 24:
        print("launch %3d: start +%8.1f us after prev start, dur %7.1f us, starts %8.1f us %s prev end, queue %s stream %s"
              % (i, (tr[i][0] - tr[i - 1][0]) / 1e3, (tr[i][1] - tr[i][0]) / 1e3, abs(gap), "BEFORE" if gap < 0 else "after", tr[i][2], tr[i][3]))
print("overlapping consecutive launches: %d of %d" % (ov, len(tr) - 1))
