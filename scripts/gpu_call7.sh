#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
W=cfg2 REPEATS=5 bash scripts/gpu_variants.sh stream
SVGIR_RASTER_LIB=$PWD/build/variants/stream/libsvgir_raster.so python -m pytest tests/test_gpu_parity.py -q -x -k "rgss or cfg2" 2>&1 | tail -3 | tee gpurun_out/c7_stream_parity.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/var_stream_cfg2.json").read().strip().splitlines()[-1]); print("stream stage_ms", d["stage_ms"])
PY
/usr/bin/time -v python -m pytest tests/test_gpu_parity.py -q -x -k "cfg5_dense" 2>&1 | grep -E "passed|failed|Elapsed|Maximum resident|Error|assert" | tee gpurun_out/c7_dense.log
python bench.py --workload cfg5_dense --no-cpu-baseline --repeats 3 --steps 10 --no-shaded --no-concurrent > gpurun_out/c7_bench_cfg5_dense.json 2> gpurun_out/c7_bench_cfg5_dense.err; tail -c 2500 gpurun_out/c7_bench_cfg5_dense.json; tail -3 gpurun_out/c7_bench_cfg5_dense.err
