#!/bin/bash
# GPU-side: rebuild shade.o with in-kernel s_memtime phase counters and print the phase split of shade_bwd
cd svg-ir_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I../../include -DSHADE_TIMING ${SHADE_DEFS:-} -c shade.hip -o shade.o 2>&1 | grep -E "error" -A5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvgir_raster.so api.o binning.o geom_bwd.o grad_reduce.o image_ops.o preprocess.o render_bwd.o render_fwd.o shade.o
cd ../..
python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, "svg-ir_amd"); sys.path.insert(0, ".")
import torch
from gaussian_renderer import shading, _native as N
from svgir_harness import shade_inputs
dev = torch.device("cuda:0")
P, Ns = 200000, 64
d = shade_inputs.make(P, Ns, seed=2, device=dev)
vm = torch.eye(4, device=dev)
leaves = {k: d[k].clone().requires_grad_(True) for k in ("base_color", "roughness", "normals", "radiance")}
env = d["env"].clone().requires_grad_(True)
def fb():
    f, vf, red = shading.shade_and_pack(leaves["base_color"], leaves["roughness"], leaves["normals"], d["viewdirs"],
                                        leaves["radiance"], shade_inputs.Light(env), d["visibility"], d["dirs"], d["areas"], vm, True)
    (f.sum() + vf.sum()).backward()
for _ in range(3): fb()
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
N.lib.svgir_debug_shade_timing(out, 1)
n = 5
for _ in range(n): fb()
torch.cuda.synchronize()
N.lib.svgir_debug_shade_timing(out, 1)
names = ["loop-top(prev epilogue stores)", "prologue loads+math", "stage (math, env gather, LDS write)", "prefetch issue + sync", "phase 2/3 (4 its)", "epilogue reduce+store", "final barrier wait", "env flush"]
tot = sum(out)
for nm, v in zip(names, out):
    print(f"{nm:40s} {v / n / P:10.1f} ticks/Gaussian  {100.0 * v / tot:5.1f}%")
PY
