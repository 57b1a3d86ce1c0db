#!/bin/bash
# GPU-side sweep of the shading-backward launch shape (waves per workgroup / waves per SIMD)
cd svg-ir_amd/csrc
for v in "8 4" "4 3" "4 4" "8 2" "4 2"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I../../include -DSHADE_BWAVES=$1 -DSHADE_BWPE=$2 -c shade.hip -o shade.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvgir_raster.so api.o binning.o geom_bwd.o grad_reduce.o image_ops.o preprocess.o render_bwd.o render_fwd.o shade.o
  (cd ../.. && python bench.py --no-cpu-baseline --steps 20 --workload cfg3_train | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['stage_ms']['shade_fwd'], r['stage_ms']['shade_bwd'], r['ms_per_step'])")
done
