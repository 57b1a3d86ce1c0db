#!/bin/bash
# L2 request / hit / miss counters of the composite kernels (separate --pmc pass, kernel-trace only).
# usage: scripts/pmc_l2.sh <workload> <tag>   -> gpurun_out/<tag>_l2/ + a per-kernel summary on stdout
set -u
export TMPDIR=/tmp
R=$PWD
W=${1:-cfg2}
TAG=${2:-l2}
cd /tmp
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${TAG}_l2 -o p -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-shaded --no-concurrent --workload $W > $R/gpurun_out/${TAG}_l2.log 2>&1
cd $R
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if not any(s in k for s in ("render_", "cull", "grad_reduce", "geom_bwd")):
        continue
    c = {n: sum(x) / len(x) for n, x in v.items()}
    name = k.replace("svgir::(anonymous namespace)::", "").split("(")[0][:40]
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print(f"{name:40s} req {c.get('TCC_REQ_sum', 0):12.0f} read {c.get('TCC_READ_sum', 0):12.0f} write {c.get('TCC_WRITE_sum', 0):11.0f} "
          f"atomic {c.get('TCC_ATOMIC_sum', 0):11.0f} hit {hit:12.0f} miss {miss:11.0f} hit-rate {hit / max(1.0, hit + miss):.3f}")
PY
