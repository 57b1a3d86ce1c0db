#!/bin/bash
# GPU-side sweep of the backward composite's register budget (waves per SIMD for svgss / rgss instantiations)
cd svg-ir_amd/csrc
for v in "2 5" "3 5" "4 5"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I../../include -DBWD_WPE_V=$1 -DBWD_WPE_P=$2 -c render_bwd.hip -o render_bwd.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvgir_raster.so api.o binning.o geom_bwd.o grad_reduce.o image_ops.o preprocess.o render_bwd.o render_fwd.o shade.o
  for w in cfg3_train cfg4; do
  (cd ../.. && python bench.py --no-cpu-baseline --steps 20 --workload $w --no-shade | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', '$w', r['stage_ms']['render_bwd'])")
  done
done
