#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c3_prof -o p -- python3 $R/scripts/tracer_cfg3_probe.py 200000 shell > $R/gpurun_out/c3_prof.log 2>&1)
tail -5 gpurun_out/c3_prof.log
DB=$(find gpurun_out/c3_prof -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB gpurun_out/c3_tracer_kernel_stats.txt "python3 scripts/tracer_cfg3_probe.py 200000 shell" | head -30
