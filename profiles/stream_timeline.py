"""Timeline of one steady-state step from a rocprofv3 --kernel-trace rocpd database: for every kernel of the step
its start (relative), duration and the idle gap on its stream before it.  usage: stream_timeline.py <db> [step_index]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(db.execute("select name, start, end, %s from kernels order by start" % (sid or "0")))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
gray = [i for i, r in enumerate(rows) if "k_gray_depth" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(gray) - 3
i0, i1 = gray[k], gray[k + 1]
t0 = rows[i0][1]
last_end = {}
for n, s, e, q in rows[max(0, i0 - 30):i0]:
    last_end[q] = e
print("step %d (between two k_gray_depth launches): %.1f us" % (k, (rows[i1][1] - t0) / 1e3))
for n, s, e, q in rows[i0:i1]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    print("  %-26s stream %-4s start %8.1f  dur %7.1f  gap before %6.1f" % (short(n)[:26], q, (s - t0) / 1e3, (e - s) / 1e3, gap))
