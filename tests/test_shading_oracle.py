"""Pins the shading oracle (oracle/shading_oracle.py) against fixtures produced by the REFERENCE's own
rendering_equation4 / GGX_specular4 (tests/golden/shading.npz, scripts/make_golden.py): forward outputs and the
autograd gradients w.r.t. base colour, roughness, normals, radiance and the env texels."""
import os

import numpy as np
import pytest
import torch

from oracle import shading_oracle as so

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shading.npz")


def _load(tag):
    g = np.load(GOLD)
    pre = "shade_" + tag + "_"
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_env_lookup_matches_grid_sample(tag):
    d = _load(tag)
    got = so.env_lookup(d["env"], d["dirs"])
    torch.testing.assert_close(got, d["env_lookup"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_forward_matches_reference_rendering_equation4(tag):
    d = _load(tag)
    out = so.shade(d["base"], d["rough"], d["normals"], d["viewdirs"], d["radiance"], d["vis"], d["dirs"], d["areas"], d["env"])
    for k in ("pbr", "diffuse_light", "specular", "direct", "indirect"):
        torch.testing.assert_close(out[k], d[k], rtol=1e-9, atol=1e-12, msg=k)
    torch.testing.assert_close(out["mean_incident"], d["mean_incident"], rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(out["mean_global"], d["mean_global"], rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(out["mean_local"], d["mean_local"], rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(out["mean_vis"], d["mean_vis"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_backward_matches_reference_autograd(tag):
    d = _load(tag)
    leaves = {k: d[k].clone().requires_grad_(True) for k in ("base", "rough", "normals", "radiance", "env")}
    out = so.shade(leaves["base"], leaves["rough"], leaves["normals"], d["viewdirs"], leaves["radiance"], d["vis"],
                   d["dirs"], d["areas"], leaves["env"])
    loss = (out["pbr"] * d["w_pbr"]).sum() + sum((out[k] * d["w_" + k]).sum() for k in ("diffuse_light", "specular", "direct", "indirect"))
    loss = loss + (out["mean_incident"] * d["w_inc"]).sum() + (out["mean_global"] * d["w_glob"]).sum()
    loss.backward()
    for k in leaves:
        torch.testing.assert_close(leaves[k].grad, d["g_" + k], rtol=1e-8, atol=1e-11, msg=k)


def test_packing_layout():
    d = _load("a")
    out = so.shade(d["base"], d["rough"], d["normals"], d["viewdirs"], d["radiance"], d["vis"], d["dirs"], d["areas"], d["env"])
    view = torch.eye(3, dtype=torch.float64)
    f, vf = so.pack(out, d["base"], d["rough"], d["normals"], view, training=True)
    assert f.shape[1] == 4 and vf.shape[1] == 52
    # normals: [n,4,3] -> c*4+k
    torch.testing.assert_close(vf[:, 24:36].reshape(-1, 3, 4), d["normals"].transpose(1, 2))
    f, vf = so.pack(out, d["base"], d["rough"], d["normals"], view, training=False)
    assert f.shape[1] == 7 and vf.shape[1] == 64
