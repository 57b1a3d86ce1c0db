"""Error table of the shading kernels on the reference-generated fixtures (tests/golden/shading.npz): per tensor, err(HIP vs reference
fp64) next to err(reference fp32 vs reference fp64), errors = max abs / tensor scale.  Run on the GPU box:
    [SVGIR_RASTER_LIB=build/variants/<name>/libsvgir_raster.so] python scripts/shade_budget_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd"))
from gaussian_renderer import shading  # noqa: E402


class Light:
    def __init__(self, env):
        self.env = env


def err(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


g = np.load(os.path.join(ROOT, "tests", "golden", "shading.npz"))
dev = torch.device("cuda:0")
worst = 0.0
for tag in "abc":
    pre = f"shade_{tag}_"
    d = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    names = ("base", "rough", "normals", "radiance", "env")
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    pbr, ex = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], Light(lg["env"]),
                                          visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                          incident_areas_precompute=c["areas"])
    w = lambda k: d[k].float().to(dev)  # noqa: E731
    loss = (pbr * w("w_pbr")).sum() + sum((ex[k] * w("w_" + k)).sum() for k in ("diffuse_light", "specular", "direct", "indirect")) \
        + (ex["incident_lights"].mean(-2) * w("w_inc")).sum() + (ex["global_incident_lights"].mean(-2) * w("w_glob")).sum()
    loss.backward()
    got = dict(pbr=pbr, specular=ex["specular"], direct=ex["direct"], indirect=ex["indirect"], diffuse_light=ex["diffuse_light"])
    got.update({"g_" + k: lg[k].grad for k in names})
    for k, v in got.items():
        eh, er = err(v, d[k]), err(d["f32_" + k], d[k])
        ratio = eh / (1.5 * er + 1e-4)
        worst = max(worst, ratio)
        print(f"{tag} {k:14s} HIP {eh:.2e}   reference fp32 {er:.2e}   HIP / budget {ratio:.2f}")
print(f"worst HIP / budget: {worst:.2f}")
