"""End-to-end svgss view on the GPU (shading -> packing -> rasterizer -> unpacking, svgir_harness/render_view.py)
against the same chain assembled from the two CPU oracles in fp64.  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import shading_oracle as so
from svgir_harness import render_view, runner, scenes, shade_inputs

pytestmark = pytest.mark.gpu


def _cmp(name, a, b, tol=2e-4, flip_frac=5e-4):
    a = a.detach().double().cpu().numpy().reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    scale = max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > tol * (scale + np.abs(b))
    assert bad.mean() <= flip_frac, f"{name}: {bad.sum()}/{bad.size} entries off (max {np.abs(a - b).max():.3e}, scale {scale:.3e})"


@pytest.mark.parametrize("is_training", [True, False])
def test_svgss_view_end_to_end(built, is_training):
    dev = torch.device("cuda:0")
    S, VS = (4, 52) if is_training else (7, 64)
    sc = scenes.surface_scene(P=4000, W=160, H=120, seed=61, sh_degree=2, variant="svgss", S=S, VS=VS, scale_lo=0.02,
                              scale_hi=0.07)
    P = sc["means3D"].shape[0]
    d = shade_inputs.make(P, 32, seed=4)
    d["roughness"] = d["roughness"].clamp_min(0.3)
    # ---- GPU ----
    sct = runner.to_torch(sc, dev)
    mat = {k: v.to(dev) for k, v in d.items() if k != "env"}
    with torch.no_grad():
        res, _ = render_view.render_svgss_view(sct, mat, shade_inputs.Light(d["env"].to(dev)), is_training)
    # ---- oracles, fp64 ----
    dd = {k: v.double() for k, v in d.items()}
    ref = so.shade(dd["base_color"], dd["roughness"], dd["normals"], dd["viewdirs"], dd["radiance"], dd["visibility"],
                   dd["dirs"], dd["areas"], dd["env"])
    view3 = torch.from_numpy(sc["viewmatrix"]).double()[:3, :3]
    f, vf = so.pack(ref, dd["base_color"], dd["roughness"], dd["normals"], view3, is_training)
    sc_o = dict(sc)
    sc_o["features"] = f.float().numpy()
    sc_o["vfeatures"] = vf.float().numpy()
    o = orc.OracleRun(sc_o, orc.SVGSS)
    R = o.forward()
    im = o.images()
    t = lambda k: torch.from_numpy(np.asarray(im[k], dtype=np.float64))  # noqa: E731
    rendered = (R, t("color"), t("normal"), t("opacity"), t("depth"), t("feature"), t("vfeature"),
                torch.from_numpy(np.asarray(im["weights"], dtype=np.float64)), torch.from_numpy(im["radii"]))
    exp = render_view.unpack(rendered, torch.from_numpy(sc["bg"]).double(), is_training)
    assert res["num_rendered"] == R
    assert torch.equal(res["radii"].cpu(), exp["radii"])
    keys = ["render", "depth", "opacity", "pbr", "normal", "base_color", "roughness", "local_lights", "visibility"]
    keys += ["diffuse"] if is_training else ["lights", "direct", "indirect"]
    for k in keys:
        assert res[k].shape == exp[k].shape, k
        _cmp(k, res[k], exp[k].numpy())
