"""Device timeline of the sequential stream from a rocprofv3 --kernel-trace rocpd database of profiles/single_stream_profile.py:
per frame (k_track launch to k_track launch) the kernels on the consumer's stream, their durations and the idle gaps between
them -- what separates the stream from one trackFrames per frame.  usage: seq_timeline.py <db> [first_frame] [n_frames]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = list(db.execute("select name, start, end, %s from kernels order by start" % sid))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
trk = [i for i, r in enumerate(rows) if "k_track<true>" in r[0] or "k_trackILb1" in r[0]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(trk) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
count = int(sys.argv[4]) if len(sys.argv) > 4 else 100  # launches the statistics cover (stay inside ONE run of the driver)
q_trk = rows[trk[first]][3]
per = []
for k in range(first, min(first + count, len(trk) - 1)):
    per.append((rows[trk[k + 1]][1] - rows[trk[k]][1]) / 1e3)
import statistics
print("frames %d..%d: k_track start to next k_track start: median %.1f us (min %.1f, max %.1f); k_track duration median %.1f us"
      % (first, first + len(per), statistics.median(per), min(per), max(per),
         statistics.median([(rows[i][2] - rows[i][1]) / 1e3 for i in trk[first:first + len(per)]])))
for k in range(first, first + n):
    i0, i1 = trk[k], trk[k + 1]
    t0 = rows[i0][1]
    print("frame %d: %.1f us" % (k, (rows[i1][1] - t0) / 1e3))
    last = None
    for nme, s, e, q in rows[i0:i1]:
        tag = "consumer" if q == q_trk else "other   "
        gap = (s - last) / 1e3 if (last is not None and q == q_trk) else float("nan")
        if q == q_trk:
            last = e
        print("  %-22s %s stream %-3s start %7.1f dur %7.1f gap before (same stream) %6.1f" % (short(nme)[:22], tag, q, (s - t0) / 1e3, (e - s) / 1e3, gap))
