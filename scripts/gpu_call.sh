#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
W=${W:-cfg2}
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/g5_prof -o p -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-shaded --no-concurrent --workload $W > $R/gpurun_out/g5_prof.log 2>&1)
python - <<'PY'
import sqlite3, glob
db=glob.glob("gpurun_out/g5_prof/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows=c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx=[i for i,r in enumerate(rows) if "preprocess" in r[0]]
k=len(idx)//2
i0=idx[k]; i1=idx[k+1]
t0=rows[i0][1]
prev_end=None
for r in rows[i0:i1+1]:
    gap = (r[1]-prev_end)/1e3 if prev_end else 0.0
    print("%8.1f %7.1f gap %5.1f  %s"%((r[1]-t0)/1e3,(r[2]-r[1])/1e3,gap,r[0][:60]))
    prev_end=max(prev_end or 0, r[2])
PY
rm -rf gpurun_out/g5_prof
