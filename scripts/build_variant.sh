#!/bin/bash
# Build an experiment variant of libsvgir_raster.so WITHOUT touching the product objects / library:
#   [FILES="render_fwd binning"] scripts/build_variant.sh <name> [-DFLAG ...]   ->  build/variants/<name>/libsvgir_raster.so
# FILES: recompile only these translation units with the extra flags and link the product's objects for the rest (the flags must not
# change anything the other units see).  Select the result at run time with SVGIR_RASTER_LIB=build/variants/<name>/libsvgir_raster.so
# (gaussian_renderer/_native.py).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants/$NAME
mkdir -p "$OUT"
FLAGS=$(make -s -C "$ROOT/svg-ir_amd/csrc" print-hipflags)
ALL="api preprocess binning render_fwd render_bwd render_bwd_plain render_generic geom_bwd grad_reduce image_ops shade subset epilogue loss optim bvh pbgi"
pids=()
OBJS=()
for f in $ALL; do
  [ -f "$ROOT/svg-ir_amd/csrc/$f.hip" ] || continue
  if [ -n "${FILES:-}" ] && ! echo " $FILES " | grep -q " $f "; then
    OBJS+=("$ROOT/svg-ir_amd/csrc/$f.o")   # product object (make -C svg-ir_amd/csrc first)
    continue
  fi
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$ROOT/svg-ir_amd/csrc/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
  OBJS+=("$OUT/$f.o")
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsvgir_raster.so" "${OBJS[@]}"
rm -f "$OUT"/*.o
echo "$OUT/libsvgir_raster.so"
