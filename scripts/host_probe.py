"""GPU probe: host-side (Python + launch) time of the forward / backward binding calls vs the GPU time of a step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from svgir_harness import runner, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
variant = scenes.CONFIGS[name][1]["variant"]
sc = scenes.make(name)
grads = scenes.upstream_grads(sc, variant)
sct = runner.to_torch(sc, dev)
gt = {k: torch.from_numpy(v).to(dev) for k, v in grads.items()}
st = runner.settings(sct, variant)
from gaussian_renderer.rgss_rasterization import _C
empty = torch.empty(0, dtype=torch.float32, device=dev)

def fwd():
    return _C.rasterize_gaussians(st.bg, sct["means3D"], sct["features"], empty, sct["opacities"], sct["scales"],
                                  sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx,
                                  st.tanfovy, st.cx, st.cy, st.image_height, st.image_width, sct["shs"], st.sh_degree,
                                  st.campos, False, False, False)
def bwd(out):
    (R, ncontrib, color, normal, opac, depth, feat, pn, sx, weights, radii, gb, bb, ib) = out
    return _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], radii, empty, sct["scales"], sct["rotations"],
                                           st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy,
                                           gt["color"], gt["normal"], gt["opacity"], gt["depth"], gt["feature"], sct["shs"],
                                           st.sh_degree, st.campos, gb, R, bb, ib, True, False)
for _ in range(5):
    bwd(fwd())
torch.cuda.synchronize()
n = 50
tf = tb = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); out = fwd(); b = time.perf_counter(); g = bwd(out); c = time.perf_counter()
    tf += b - a; tb += c - b
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"{name}: step {tot / n * 1e3:.3f} ms; host time in forward call {tf / n * 1e3:.3f} ms (includes the R sync), in backward call {tb / n * 1e3:.3f} ms")
# forward only, back to back
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): out = fwd()
torch.cuda.synchronize(); print(f"forward only: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): bwd(fwd())
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# ---- forward-only variants: does keeping the previous step's blobs alive matter? ----
from gaussian_renderer import _native as N
for label, keep in (("previous blobs alive during the next forward", True), ("previous blobs freed first", False)):
    out = None
    for _ in range(5):
        if not keep: out = None
        out = fwd()
    torch.cuda.synchronize(); N.set_profiling(True); N.ALLOC_STATS.update(calls=0, seconds=0.0, max_seconds=0.0)
    t0 = time.perf_counter()
    for _ in range(n):
        if not keep: out = None
        out = fwd()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{label}: {el / n * 1e3:.3f} ms/forward; alloc callbacks {N.ALLOC_STATS}; stages", {k: round(v, 4) for k, v in N.last_timings()})
    N.set_profiling(False)
