#!/bin/bash
# GPU-side sweep of the backward segment length (common.hpp SEG) and forward batch size: rebuilds the library per value
cd svg-ir_amd/csrc
cp common.hpp /tmp/common.hpp.bak; cp stage.hpp /tmp/stage.hpp.bak
for v in "64 64" "64 32" "32 32"; do
  set -- $v
  sed -i "s/^constexpr int SEG = [0-9]*;/constexpr int SEG = $1;/" common.hpp
  sed -i "s/static constexpr int CH = NF <= 48 ? [0-9]* : 32;/static constexpr int CH = NF <= 48 ? $2 : 32;/" stage.hpp
  make -s -j8 > /dev/null 2>&1
  for w in cfg2 cfg3_train cfg5; do
  (cd ../.. && python bench.py --no-cpu-baseline --steps 20 --workload $w --no-shade | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('SEG=$1 CH=$2', '$w', r['ms_per_step'], r['stage_ms']['render'], r['stage_ms']['render_bwd'])")
  done
done
cp /tmp/common.hpp.bak common.hpp; cp /tmp/stage.hpp.bak stage.hpp
