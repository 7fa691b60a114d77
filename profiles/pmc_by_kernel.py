"""Average PMC counter value per launch, per kernel, from a rocprofv3 --pmc rocpd database.
usage: pmc_by_kernel.py <db> [kernel-substring ...]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2:]
cur = db.execute("select * from counters_collection limit 1")
cols = [d[0] for d in cur.description]
kn = "kernel_name" if "kernel_name" in cols else "name"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for k, c, v, d in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % kn):
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if want and not any(w in short for w in want):
        continue
    acc[short][c] += v
    cnt[short].add(d)
for k in acc:
    n = len(cnt[k])
    print(k, "launches", n, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))
