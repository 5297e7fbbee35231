"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel stats and PMC counter sums per dispatch.
  python scripts/rocpd_summary.py x_results.db [--last N]
--last N: only the last N dispatches of every kernel (a run that has to march INTO the state it measures — the SPH developed
state, 1 520 sub-steps after the lattice — is summarised over its timed tail, not over the transient)."""
import sqlite3, sys, json
argv = sys.argv[1:]
last = None
if "--last" in argv:
    i = argv.index("--last"); last = int(argv[i + 1]); del argv[i:i + 2]
db=argv[0]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(kernels)")]
if last:
    rows=c.execute("select name, count(*), sum(d), avg(d), min(d), max(d) from (select name, end-start as d, row_number() over "
                   "(partition by name order by start desc) as rn from kernels) where rn <= ? group by name order by 3 desc", (last,)).fetchall()
else:
    rows=c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows) or 1
print("KERNEL_STATS (ns)" + (f" — the last {last} dispatches of every kernel" if last else ""))
print("%-60s %6s %14s %12s %12s %12s %6s"%("name","calls","total","avg","min","max","%"))
for r in rows:
    print("%-60s %6d %14d %12.0f %12d %12d %6.2f"%(r[0][:60],r[1],r[2],r[3],r[4],r[5],100*r[2]/tot))
try:
    ccols=[r[1] for r in c.execute("pragma table_info(counters_collection)")]
    if last:
        order=next((k for k in ("dispatch_id","start","id") if k in ccols), None)
        if order is None: raise RuntimeError("counters_collection has no column to order dispatches by: %s" % ccols)
        q=c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from (select kernel_name, counter_name, value, "
                    "dense_rank() over (partition by kernel_name, counter_name order by %s desc) as rn from counters_collection) where rn <= ? "
                    "group by kernel_name, counter_name" % order, (last,)).fetchall()
    else:
        q=c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    if q:
        print("\nPMC (per kernel, per counter): calls, sum, avg-per-dispatch")
        for r in q: print("%-50s %-24s %6d %20.0f %20.1f"%(r[0][:50],r[1],r[2],r[3],r[4]))
except Exception as e:
    print("no counters:",e)
