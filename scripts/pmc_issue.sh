#!/bin/bash
# Instruction-issue counters of the composite / shading kernels (one --pmc pass, kernel-trace only) -> what bench.py reports as
# roofline.issue_frac.  usage: scripts/pmc_issue.sh <workload> <tag>   -> gpurun_out/<tag>_issue.json
set -u
export TMPDIR=/tmp
R=$PWD
W=${1:-cfg2}
TAG=${2:-issue}
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/gpurun_out/${TAG}_iss -o p -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-shaded --no-concurrent --workload $W > $R/gpurun_out/${TAG}_iss.log 2>&1
cd $R
python3 scripts/pmc_issue_summary.py gpurun_out/${TAG}_iss $W > gpurun_out/${TAG}_issue.json
cat gpurun_out/${TAG}_issue.json
