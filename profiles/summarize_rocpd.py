"""Turns a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel summary CSV kept under profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgprs,lds_bytes")
for r in rows:
    print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.1f,%s,%s' % (r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                     100.0 * r[2] / tot, r[6], r[7]))
