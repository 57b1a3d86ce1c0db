#!/bin/bash
# GPU-side sweep of the backward segment length (common.hpp SEG): rebuilds the whole library per value
cd svg-ir_amd/csrc
for seg in 64 128 256; do
  sed -i "s/^constexpr int SEG = [0-9]*;/constexpr int SEG = $seg;/" common.hpp
  make -s -j8 > /dev/null 2>&1
  for w in cfg2 cfg3_train cfg5; do
  (cd ../.. && python bench.py --no-cpu-baseline --steps 20 --workload $w --no-shade | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('SEG=$seg', '$w', r['ms_per_step'], r['stage_ms']['render'], r['stage_ms']['render_bwd'])")
  done
done
sed -i "s/^constexpr int SEG = [0-9]*;/constexpr int SEG = 128;/" common.hpp
