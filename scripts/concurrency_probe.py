"""GPU probe: does the chip have head-room for more concurrent composite work?  Runs the same fwd+bwd view on N HIP
streams at once and reports wall time per round (N views)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from svgir_harness import cameras, runner, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
variant = scenes.CONFIGS[name][1]["variant"]
sc = scenes.make(name)
grads = scenes.upstream_grads(sc, variant)
sct = runner.to_torch(sc, dev)
gt = {k: torch.from_numpy(v).to(dev) for k, v in grads.items()}
st = runner.settings(sct, variant)
if variant == "svgss":
    from gaussian_renderer.svgss_rasterization import _C
else:
    from gaussian_renderer.rgss_rasterization import _C
empty = torch.empty(0, dtype=torch.float32, device=dev)


def step():
    if variant == "svgss":
        out = _C.rasterize_gaussians(st.bg, sct["means3D"], sct["features"], sct["vfeatures"], empty, sct["opacities"],
                                     sct["scales"], sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix,
                                     st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height, st.image_width,
                                     sct["shs"], st.sh_degree, st.campos, False, False, st.config)
        (R, color, normal, depth, opac, feat, vfeat, weights, radii, gb, bb, ib) = out
        _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], sct["vfeatures"], radii, empty, sct["scales"],
                                        sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint,
                                        st.patch_bbox, st.tanfovx, st.tanfovy, gt["color"], gt["normal"], gt["depth"],
                                        gt["opacity"], gt["feature"], gt["vfeature"], sct["shs"], st.sh_degree, st.campos, gb,
                                        R, bb, ib, False, st.config)
    else:
        out = _C.rasterize_gaussians(st.bg, sct["means3D"], sct["features"], empty, sct["opacities"], sct["scales"],
                                     sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx,
                                     st.tanfovy, st.cx, st.cy, st.image_height, st.image_width, sct["shs"], st.sh_degree,
                                     st.campos, False, False, False)
        (R, ncontrib, color, normal, opac, depth, feat, pn, sx, weights, radii, gb, bb, ib) = out
        _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], radii, empty, sct["scales"], sct["rotations"],
                                        st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy,
                                        gt["color"], gt["normal"], gt["opacity"], gt["depth"], gt["feature"], sct["shs"],
                                        st.sh_degree, st.campos, gb, R, bb, ib, True, False)


import threading
for n in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    rounds = 20

    def work(s, k):
        with torch.cuda.stream(s):
            for _ in range(k):
                step()

    for s in streams:
        work(s, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(s, rounds)) for s in streams]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name}: {n} streams: {el / rounds * 1e3:.3f} ms per round of {n} views -> {el / rounds / n * 1e3:.3f} ms/view", flush=True)
