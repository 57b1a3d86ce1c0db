"""Measured HIP-vs-oracle error per tensor and config -> gpurun_out/parity_<tag>.json (copied to profiles/ by hand).
    python scripts/parity_report.py [tag] [configs...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import parity_util as pu
from oracle import oracle as orc
from svgir_harness import runner, scenes

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
cfgs = sys.argv[2:] or ["cfg1", "cfg2", "cfg3_train", "cfg3_eval", "cfg4", "cfg5"]
dev = torch.device("cuda:0")
rep = {}
for cfg in cfgs:
    variant = scenes.CONFIGS[cfg][1]["variant"]
    sc = scenes.make(cfg)
    train = cfg in ("cfg2", "cfg3_train", "cfg4", "cfg5", "cfg5_dense")
    grads = scenes.upstream_grads(sc, variant) if train else None
    sct = runner.to_torch(sc, dev)
    t0 = time.time()
    raw = runner.forward_raw(sct, variant)
    out, leaves = runner.render(sct, variant, requires_grad=train)
    if train:
        runner.backward(out, grads, variant)
    torch.cuda.synchronize()
    o = orc.OracleRun(sc, orc.SVGSS if variant == "svgss" else orc.RGSS)
    R = o.forward()
    if train:
        o.backward(grads["color"], grads["normal"], grads["depth"], grads["opacity"], grads["feature"], grads.get("vfeature"))
    im = o.images()
    r = {"num_rendered": [int(out["num_rendered"]), int(R)], "radii_equal": bool(np.array_equal(out["radii"].cpu().numpy(), im["radii"])),
         "point_list_equal": bool(np.array_equal(raw["point_list"], o.get("point_list")[:R])),
         "ranges_equal": bool(np.array_equal(raw["ranges"].reshape(-1), o.get("ranges").reshape(-1))),
         "n_contrib_mismatch_frac": float((raw["n_contrib"] != o.get("n_contrib").reshape(raw["n_contrib"].shape)).mean()),
         "forward": {}, "backward": {}}
    for k in ["color", "normal", "depth", "opacity", "feature", "weights"] + (["vfeature"] if variant == "svgss" else []):
        r["forward"][k] = pu.stats(out[k].detach().cpu().numpy(), im[k])
    if train:
        gr = o.grads()
        pairs = [("means3D", "means3D"), ("scales", "scales"), ("rotations", "rotations"), ("opacities", "opacity"),
                 ("shs", "sh"), ("features", "features"), ("means2D", "means2D")] + ([("vfeatures", "vfeatures")] if variant == "svgss" else [])
        for lk, ok in pairs:
            if leaves[lk].grad is not None:
                r["backward"][lk] = pu.stats(leaves[lk].grad.cpu().numpy(), gr[ok])
        # the fp64 anchor: how far the product and the fp32 oracle (= the reference's own arithmetic) are from the exact gradient
        o64 = orc.OracleRun(sc, orc.SVGSS if variant == "svgss" else orc.RGSS, fp64=True)
        o64.forward()
        o64.backward(grads["color"], grads["normal"], grads["depth"], grads["opacity"], grads["feature"], grads.get("vfeature"))
        ex = o64.grads()
        r["backward_vs_fp64"] = {}
        for lk, ok in pairs:
            if leaves[lk].grad is not None:
                r["backward_vs_fp64"][lk] = {"hip": pu.stats(leaves[lk].grad.cpu().numpy(), ex[ok]), "oracle_fp32": pu.stats(gr[ok], ex[ok])}
        del o64
    r["seconds"] = time.time() - t0
    rep[cfg] = r
    print(cfg, "R", r["num_rendered"], "point_list", r["point_list_equal"], "ranges", r["ranges_equal"], "ncontrib mismatch %.2e" % r["n_contrib_mismatch_frac"])
    for ph in ("forward", "backward"):
        for k, v in r[ph].items():
            print("   %-8s %-10s max_norm %.2e p99.99 %.2e flip_frac %.2e rel_frac %.2e p99.99_rel %.2e" % (ph, k, v["max_norm"], v["p9999_norm"], v["flip_frac"], v["rel_frac"], v.get("p9999_rel", 0)))
# ---- shading: distance of the HIP kernels from the reference's OWN outputs (tests/golden/shading.npz = rendering_equation4 run by the
# reference's code in fp64 and in fp32, forward and autograd): what an integrator who swaps the kernels in will see per tensor
def _err(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = (b if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).detach().double().cpu().reshape(-1)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


class _Light:
    def __init__(self, env):
        self.env = env


from gaussian_renderer import shading
gold = np.load(os.path.join(ROOT, "tests", "golden", "shading.npz"))
rep["shading_vs_reference"] = {}
for tagc in ("a", "b", "c"):
    pre = "shade_" + tagc + "_"
    d = {k[len(pre):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(pre)}
    names = ("base", "rough", "normals", "radiance", "env")
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    pbr, exr = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], _Light(lg["env"]),
                                           visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"], incident_areas_precompute=c["areas"])
    w = lambda k: d[k].float().to(dev)  # noqa: E731
    loss = (pbr * w("w_pbr")).sum() + sum((exr[k] * w("w_" + k)).sum() for k in ("diffuse_light", "specular", "direct", "indirect")) \
        + (exr["incident_lights"].mean(-2) * w("w_inc")).sum() + (exr["global_incident_lights"].mean(-2) * w("w_glob")).sum()
    loss.backward()
    got = dict(pbr=pbr, mean_incident=exr["incident_lights"].mean(-2), mean_global=exr["global_incident_lights"].mean(-2),
               **{k: exr[k] for k in ("diffuse_light", "specular", "direct", "indirect")})
    rr = {}
    for k, v in got.items():
        rr[k] = {"hip_vs_ref_fp32": _err(v, d["f32_" + k]), "hip_vs_ref_fp64": _err(v, d[k]), "ref_fp32_vs_ref_fp64": _err(d["f32_" + k], d[k])}
    for k in names:
        if "g_" + k in d and "f32_g_" + k in d:
            rr["grad_" + k] = {"hip_vs_ref_fp32": _err(lg[k].grad, d["f32_g_" + k]), "hip_vs_ref_fp64": _err(lg[k].grad, d["g_" + k]),
                               "ref_fp32_vs_ref_fp64": _err(d["f32_g_" + k], d["g_" + k])}
    rep["shading_vs_reference"]["fixture_" + tagc] = rr
    print("shading fixture", tagc, {k: "%.1e/%.1e/%.1e" % (v["hip_vs_ref_fp32"], v["hip_vs_ref_fp64"], v["ref_fp32_vs_ref_fp64"]) for k, v in rr.items()})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"parity_{tag}.json"), "w") as f:
    json.dump(rep, f, indent=1)
