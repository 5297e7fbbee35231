"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel stats and PMC counter sums per dispatch."""
import sqlite3, sys, json
db=sys.argv[1]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(kernels)")]
rows=c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows) or 1
print("KERNEL_STATS (ns)")
print("%-60s %6s %14s %12s %12s %12s %6s"%("name","calls","total","avg","min","max","%"))
for r in rows:
    print("%-60s %6d %14d %12.0f %12d %12d %6.2f"%(r[0][:60],r[1],r[2],r[3],r[4],r[5],100*r[2]/tot))
try:
    ccols=[r[1] for r in c.execute("pragma table_info(counters_collection)")]
    q=c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    if q:
        print("\nPMC (per kernel, per counter): calls, sum, avg-per-dispatch")
        for r in q: print("%-50s %-24s %6d %20.0f %20.1f"%(r[0][:50],r[1],r[2],r[3],r[4]))
except Exception as e:
    print("no counters:",e)
