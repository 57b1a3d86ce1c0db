#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_shading.py tests/test_gpu_render_view.py -q -x 2>&1 | tail -8 | tee gpurun_out/c4_tests.log
for W in cfg3_train cfg3_eval; do
python bench.py --workload $W --steps 30 --warmup 5 --repeats 5 --no-cpu-baseline --no-shaded --no-concurrent > gpurun_out/c4_$W.json 2> gpurun_out/c4_$W.err
python - <<PY
import json
d=json.loads(open("gpurun_out/c4_$W.json").read().strip().splitlines()[-1])
print("$W", "ms/step %.4f"%d["ms_per_step"], {k:round(v*1e3,1) for k,v in d["stage_ms"].items() if k.startswith("shade") or k in ("render","render_bwd")})
PY
done
