"""Steady-state view of the pipelined step from a rocprofv3 --kernel-trace rocpd database of bench.py: busy time per stream and
step, every kernel's average duration IN the step, the idle gap in front of each kernel on its stream.
usage: stream_busy.py <db> [first tracker launch] [launches]"""
import collections
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
sid = "stream_id" if "stream_id" in cols else "queue_id"
rows = list(db.execute("select name, start, end, %s from kernels order by start" % sid))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
trk = [i for i, r in enumerate(rows) if "k_track<false>" in r[0]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 14
count = int(sys.argv[3]) if len(sys.argv) > 3 else 30
t_idx = trk[first:first + count]
t0, t1 = rows[t_idx[0]][1], rows[t_idx[-1]][1]
nsteps = len(t_idx) - 1
print("%d steps between tracker launches %d and %d: %.1f us per step" % (nsteps, first, first + count - 1, (t1 - t0) / 1e3 / nsteps))
busy = collections.defaultdict(float)
dur = collections.defaultdict(list)
by = collections.defaultdict(list)
for n, s, e, q in rows:
    if t0 <= s < t1:
        busy[q] += (e - s) / 1e3
        dur[short(n)].append((e - s) / 1e3)
        by[q].append((s, e, short(n)))
for q in sorted(busy):
    names = collections.Counter(x[2] for x in by[q]).most_common(2)
    print("stream %-3s busy %6.1f us per step  (%s)" % (q, busy[q] / nsteps, ", ".join(n for n, _ in names)))
print("kernel                         per step   avg us   min     max")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-30s %6.2f   %7.1f %7.1f %7.1f" % (k, len(v) / nsteps, statistics.mean(v), min(v), max(v)))
print("idle gap in front of a kernel on its own stream (mean / max us):")
for q in sorted(by):
    gaps = collections.defaultdict(list)
    for a, b in zip(by[q], by[q][1:]):
        gaps[b[2]].append((b[0] - a[1]) / 1e3)
    print("  stream %-3s %s" % (q, {k: (round(statistics.mean(v), 1), round(max(v), 1)) for k, v in gaps.items()}))
