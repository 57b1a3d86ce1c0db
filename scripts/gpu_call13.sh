#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/c13_tests.log
timeout 900 python scripts/stress.py 11 40 2>&1 | tail -4 | tee gpurun_out/c13_stress.log
python bench.py --workload tracers > gpurun_out/r04_b_bench_tracers.json 2> gpurun_out/r04_b_bench_tracers.err; tail -c 1800 gpurun_out/r04_b_bench_tracers.json
SVGIR_RASTER_LIB=$PWD/build/variants/dev/libsvgir_raster.so timeout 900 python scripts/tracer_cfg3_probe.py 200000 shell > gpurun_out/r04_b_tracer_stats.txt 2>&1; tail -9 gpurun_out/r04_b_tracer_stats.txt
bash scripts/pmc_tracer.sh r04_b_pmct > gpurun_out/r04_b_tracer_pmc.txt 2>&1; tail -4 gpurun_out/r04_b_tracer_pmc.txt | cut -c1-700
rm -rf gpurun_out/r04_b_pmct_a gpurun_out/r04_b_pmct_b
