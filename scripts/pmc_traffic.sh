#!/bin/bash
# HBM traffic of the composite kernels from the L2 memory-side request counters (separate --pmc passes, kernel-trace
# only).  usage: scripts/pmc_traffic.sh <workload> <tag>   -> gpurun_out/<tag>_traffic.json
set -u
export TMPDIR=/tmp
R=$PWD
W=${1:-cfg2}
TAG=${2:-traffic}
cd /tmp
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/${TAG}_rd -o p -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-shaded --no-concurrent --workload $W > $R/gpurun_out/${TAG}_rd.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $R/gpurun_out/${TAG}_wr -o p -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-shaded --no-concurrent --workload $W > $R/gpurun_out/${TAG}_wr.log 2>&1
cd $R
python3 scripts/pmc_traffic_summary.py gpurun_out/${TAG}_rd gpurun_out/${TAG}_wr $W > gpurun_out/${TAG}_traffic.json
cat gpurun_out/${TAG}_traffic.json
