#!/bin/bash
# SQ PMC counters of the tracer kernels (separate --pmc passes, kernel-trace only).  usage: scripts/pmc_tracer.sh <tag>
set -u
export TMPDIR=/tmp
R=$PWD
TAG=${1:-pmct}
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_a -o p -- python3 $R/scripts/tracer_pmc.py > $R/gpurun_out/${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_b -o p -- python3 $R/scripts/tracer_pmc.py > $R/gpurun_out/${TAG}_b.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b trace_kernel | tee gpurun_out/${TAG}_summary.txt
