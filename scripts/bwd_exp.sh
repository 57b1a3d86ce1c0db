#!/bin/bash
# GPU-side: time render_bwd with extra -D flags ($RENDER_DEFS)
cd svg-ir_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I../../include ${RENDER_DEFS:-} -c render_bwd.hip -o render_bwd.o 2>&1 | grep -E "error" -A5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvgir_raster.so api.o binning.o geom_bwd.o grad_reduce.o image_ops.o preprocess.o render_bwd.o render_fwd.o shade.o
for w in cfg2 cfg3_train cfg4 cfg5; do
(cd ../.. && python bench.py --no-cpu-baseline --steps 20 --workload $w --no-shade | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('${RENDER_DEFS:-}', '$w', r['stage_ms']['render_bwd'])")
done
