#!/bin/bash
# One GPU session: tests, PMC traffic, bench lines and rocprofv3 kernel-trace summaries.  usage: scripts/gpu_round.sh <tag> [quick]
set -u
export TMPDIR=/tmp
R=$PWD
TAG=${1:-x}
mkdir -p gpurun_out
if [ "${2:-}" != "quick" ]; then
  python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/${TAG}_tests.log
  cat gpurun_out/${TAG}_tests.log
fi
for W in ${PMC_WORKLOADS:-cfg2 cfg3_train cfg3_eval cfg4 cfg5}; do   # PMC passes first: bench.py reports them as roofline.traffic / roofline.issue_frac (stamped with the kernel-source hash)
  scripts/pmc_traffic.sh $W ${TAG}_$W > /dev/null 2>&1 && cp gpurun_out/${TAG}_${W}_traffic.json profiles/traffic_$W.json && cp gpurun_out/${TAG}_${W}_traffic.json gpurun_out/traffic_$W.json
  scripts/pmc_issue.sh $W ${TAG}_$W > /dev/null 2>&1 && cp gpurun_out/${TAG}_${W}_issue.json profiles/issue_$W.json && cp gpurun_out/${TAG}_${W}_issue.json gpurun_out/issue_$W.json
done
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 6000 gpurun_out/${TAG}_bench_default.json; tail -3 gpurun_out/${TAG}_bench_default.err
for W in cfg3_train cfg3_eval cfg4 cfg5 cfg5_dense; do
  python bench.py --workload $W --no-cpu-baseline --repeats 5 > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err
  tail -c 3000 gpurun_out/${TAG}_bench_$W.json; tail -3 gpurun_out/${TAG}_bench_$W.err
done
for W in train_step tracers; do   # the "next" rows of SURVEY 8: whole training iteration (f2 + f4), cache producers (f3)
  python bench.py --workload $W > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err
  tail -c 3000 gpurun_out/${TAG}_bench_$W.json; tail -3 gpurun_out/${TAG}_bench_$W.err
done
for W in cfg2 cfg3_train cfg5 train_step tracers; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$W -o p -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-shaded --no-concurrent --workload $W > $R/gpurun_out/${TAG}_prof_$W.log 2>&1)
  DB=$(find gpurun_out/${TAG}_prof_$W -name "*.db" | head -1)
  python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_${W}_kernel_stats.txt "python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-shaded --no-concurrent --workload $W" | head -30
  rm -rf gpurun_out/${TAG}_prof_$W   # (the summary is what is kept: gpurun_out/ travels back only while it is below 64 MiB)
done
# tracers: traversal statistics (dev build, if present) and SQ counters of the two trace kernels
if [ -f build/variants/dev/libsvgir_raster.so ]; then
  SVGIR_RASTER_LIB=$PWD/build/variants/dev/libsvgir_raster.so timeout 900 python scripts/tracer_cfg3_probe.py 200000 shell > gpurun_out/${TAG}_tracer_stats.txt 2>&1
  tail -12 gpurun_out/${TAG}_tracer_stats.txt
fi
bash scripts/pmc_tracer.sh ${TAG}_pmct > gpurun_out/${TAG}_tracer_pmc.txt 2>&1; tail -6 gpurun_out/${TAG}_tracer_pmc.txt
rm -rf gpurun_out/${TAG}_pmct_a gpurun_out/${TAG}_pmct_b gpurun_out/${TAG}_*_rd gpurun_out/${TAG}_*_wr gpurun_out/${TAG}_*_iss
python scripts/sort_probe.py cfg2 cfg5 cfg5_dense 2>&1 | grep '^cfg' > gpurun_out/${TAG}_sort_probe.txt; cat gpurun_out/${TAG}_sort_probe.txt
timeout 600 python scripts/blob_sizes.py cfg2 cfg3_train cfg3_eval cfg5 cfg5_dense 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_blob_sizes.txt; cat gpurun_out/${TAG}_blob_sizes.txt
python scripts/parity_report.py ${PARITY_TAG:-r06} cfg1 cfg2 cfg3_train cfg3_eval cfg4 cfg5 cfg5_dense > gpurun_out/${TAG}_parity.log 2>&1; tail -30 gpurun_out/${TAG}_parity.log
