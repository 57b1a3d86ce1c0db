#!/bin/bash
# Build an experiment variant of libsvgir_raster.so WITHOUT touching the product objects / library:
#   scripts/build_variant.sh <name> [-DFLAG ...]   ->  build/variants/<name>/libsvgir_raster.so
# Select it at run time with SVGIR_RASTER_LIB=build/variants/<name>/libsvgir_raster.so (gaussian_renderer/_native.py).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants/$NAME
mkdir -p "$OUT"
FLAGS=$(make -s -C "$ROOT/svg-ir_amd/csrc" print-hipflags)
pids=()
for f in api preprocess binning render_fwd render_bwd render_bwd_plain render_generic geom_bwd grad_reduce image_ops shade subset epilogue loss optim bvh pbgi; do
  [ -f "$ROOT/svg-ir_amd/csrc/$f.hip" ] || continue
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$ROOT/svg-ir_amd/csrc/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgir_raster.so" "$OUT"/*.o
rm -f "$OUT"/*.o
echo "$OUT/libsvgir_raster.so"
