#!/bin/bash
# Memory-side PMC counters of the tracer kernels (separate --pmc passes, kernel-trace only).  usage: scripts/pmc_tracer_mem.sh <tag>
set -u
export TMPDIR=/tmp
R=$PWD
TAG=${1:-pmctm}
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TCP|TA|TD|TCC)_[A-Z0-9_]+(\[[0-9]+\])?\b" | sort -u > $R/gpurun_out/${TAG}_avail.txt
pass() {  # <suffix> <counters...>
  local sfx=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}_$sfx -o p -- python3 $R/scripts/tracer_pmc.py > $R/gpurun_out/${TAG}_$sfx.log 2>&1
}
pass a TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass b TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass c TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass d TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
cd $R
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob("gpurun_out/${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/${TAG}_*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    if "trace_kernel" not in k:
        continue
    print(k.split("(")[0][-40:], "us", sum(dur[k]) / max(1, len(dur[k])))
    for n, x in sorted(v.items()):
        print(f"   {n:44s} {sum(x) / len(x):16.0f}  (n={len(x)})")
PY
