"""GPU parity of the HIP linear BVH + visibility tracer (csrc/bvh.hip, drop-in `submodules.bvh.RayTracer`) against the CPU
oracle (oracle/bvh_oracle.cpp: a tree-free restatement of submodules/bvh/src/trace.cu) and the fixture the reference's own
`RayTracer` Python produced (tests/golden/bvh.npz).

Tolerance.  fp32 on both sides, but different summation orders and rcp-based slabs: visibility within 2e-4 absolute (flat
surfels carry inverse covariances ~1e6: the fp32 evaluation of t and power loses ~3 digits to cancellation -- in the
reference's own code as here, tests/test_bvh_oracle.py), and at most 1e-3 of the rays may take the other side of a
threshold (0.9 cut-off, box boundary, t >= 0.01, alpha facing tests): those are counted, not compared."""
import os

import numpy as np
import pytest
import torch

from oracle import bvh_oracle as bo
from tests.test_bvh_oracle import _scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bvh.npz")
DEV = "cuda:0"


def _tracer(sc):
    from submodules.bvh import RayTracer
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return RayTracer(t(sc["means"]), t(sc["scales"]), t(sc["rots"])), t


def _compare(vis, contrib, ovis, ocnt, flip_frac=1e-3, tol=2e-4):
    vis, contrib = vis.reshape(-1), contrib.reshape(-1)
    ovis, ocnt = ovis.reshape(-1), ocnt.reshape(-1)
    flip = (contrib != ocnt) | ((vis > 0) != (ovis > 0))
    assert flip.mean() <= flip_frac, f"{flip.sum()} of {flip.size} rays took another branch"
    ok = ~flip
    err = np.abs(vis[ok] - ovis[ok])
    assert err.max() <= tol, f"visibility differs by {err.max():.3e}"


def test_small_scene_every_ray(built):
    sc = _scene(500, 21)
    rt, t = _tracer(sc)
    rng = np.random.default_rng(22)
    S = 24
    d = rng.normal(size=(500, S, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(sc["means"][:, None], d.shape)
    out = rt.trace_visibility(t(sc["means"])[:, None].expand(500, S, 3), t(d), t(sc["means"]), t(sc["symm"]), t(sc["opacity"]), t(sc["normals"]))
    assert out["visibility"].shape == (500, S, 1) and out["contribute"].shape == (500, S, 1)
    assert out["contribute"].dtype == torch.int32
    boxes = bo.leaf_boxes(sc["means"], sc["scales"], sc["rots"])
    ocnt, ovis = bo.trace_visibility(boxes, o, d, sc["means"], sc["symm"], sc["opacity"], sc["normals"])
    _compare(out["visibility"].cpu().numpy(), out["contribute"].cpu().numpy(), ovis, ocnt)
    assert 0.05 < (ovis > 0).mean() < 0.999


def test_reference_fixture_leaf_boxes_through_the_tracer(built):
    """A ray that starts inside a leaf box and leaves it hits that box and (for these well separated probes) is attenuated by
    exactly that surfel: checks the leaf boxes / Morton order / refit of the builder with the reference's own scene."""
    g = np.load(GOLD)
    sc = dict(means=g["means3D"], scales=g["scales"], rots=g["rotations"])
    P = sc["means"].shape[0]
    from submodules.bvh import RayTracer
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rt = RayTracer(t(sc["means"]), t(sc["scales"]), t(sc["rots"]))
    rng = np.random.default_rng(5)
    symm = np.tile(np.array([4.0, 0, 0, 4.0, 0, 4.0], np.float32), (P, 1))       # isotropic blobs, sigma = 0.5
    opacity = rng.uniform(0.01, 0.04, size=P).astype(np.float32)
    normals = np.zeros((P, 3), np.float32)                                          # (never back-facing)
    d = rng.normal(size=(P, 16, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(sc["means"][:, None], d.shape).copy()
    out = rt.trace_visibility(t(o), t(d), t(sc["means"]), t(symm), t(opacity), t(normals))
    boxes = g["aabbs"][P - 1:]                                                     # the REFERENCE's leaf boxes
    ocnt, ovis = bo.trace_visibility(boxes, o, d, sc["means"], symm, opacity, normals)
    _compare(out["visibility"].cpu().numpy(), out["contribute"].cpu().numpy(), ovis, ocnt, flip_frac=2e-3)
    assert ocnt.max() > 3


@pytest.mark.parametrize("P", [1, 2, 3, 65])
def test_tiny_trees_and_degenerate_inputs(built, P):
    sc = _scene(P, 40 + P)
    sc["means"][: P // 2] = sc["means"][0]            # duplicates: equal Morton codes, the index breaks the tie
    rt, t = _tracer(sc)
    rng = np.random.default_rng(41)
    d = rng.normal(size=(200, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:5, 0] = 0.0                                      # axis-parallel components: infinite slab planes
    d[:5] /= np.linalg.norm(d[:5], axis=-1, keepdims=True)
    o = rng.uniform(-1, 1, size=(200, 3)).astype(np.float32)
    out = rt.trace_visibility(t(o), t(d), t(sc["means"]), t(sc["symm"]), t(sc["opacity"]), t(sc["normals"]))
    boxes = bo.leaf_boxes(sc["means"], sc["scales"], sc["rots"])
    ocnt, ovis = bo.trace_visibility(boxes, o, d, sc["means"], sc["symm"], sc["opacity"], sc["normals"])
    _compare(out["visibility"].cpu().numpy(), out["contribute"].cpu().numpy(), ovis, ocnt, flip_frac=0.01)


def test_empty_inputs(built):
    from submodules.bvh import RayTracer
    z = lambda *s: torch.zeros(*s, device=DEV)
    rt = RayTracer(z(0, 3), z(0, 3), z(0, 4))
    out = rt.trace_visibility(z(7, 3), torch.ones(7, 3, device=DEV), z(0, 3), z(0, 6), z(0), z(0, 3))
    assert torch.all(out["visibility"] == 1) and torch.all(out["contribute"] == 0)
    sc = _scene(10, 3)
    rt, t = _tracer(sc)
    out = rt.trace_visibility(z(0, 3), z(0, 3), t(sc["means"]), t(sc["symm"]), t(sc["opacity"]), t(sc["normals"]))
    assert out["visibility"].shape == (0, 1)


def test_cfg3_scale_update_visibility_sample(built):
    """BASELINE configs[2] scale: P = 200 000 surfels, Ns = 64 incident directions per surfel (scene/gaussian_model.py:435-464
    `update_visibility`): all 12.8 M rays are traced on the GPU; a random subset of 3 000 rays is checked against the
    brute-force oracle (200 000 surfels each)."""
    from svgir_harness import scenes
    sc0 = scenes.make("cfg3_train")
    P = sc0["means3D"].shape[0]
    scales = sc0["scales"].astype(np.float32).copy()
    scales[:, 2] = np.minimum(scales[:, 2], 1e-3)
    q = sc0["rotations"] / np.linalg.norm(sc0["rotations"], axis=1, keepdims=True)
    sc = _scene(1, 0)
    r, x, y, z = q.T.astype(np.float64)
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                  np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                  np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    Linv = R * (1.0 / scales.astype(np.float64))[:, None, :]
    Cinv = Linv @ Linv.transpose(0, 2, 1)
    symm = np.stack([Cinv[:, 0, 0], Cinv[:, 0, 1], Cinv[:, 0, 2], Cinv[:, 1, 1], Cinv[:, 1, 2], Cinv[:, 2, 2]], -1).astype(np.float32)
    normals = R[:, :, 2].astype(np.float32)
    opacity = sc0["opacities"].reshape(-1).astype(np.float32)
    means = sc0["means3D"].astype(np.float32)
    from submodules.bvh import RayTracer
    from gaussian_renderer import shading
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rt = RayTracer(t(means), t(scales), t(q.astype(np.float32)))
    Ns = 64
    dirs, _areas = shading.sample_incident_rays(t(normals), False, Ns)            # [P, Ns, 3]
    out = rt.trace_visibility(t(means)[:, None].expand_as(dirs), dirs, t(means), t(symm), t(opacity), t(normals))
    torch.cuda.synchronize()
    vis = out["visibility"].reshape(-1).cpu().numpy()
    cnt = out["contribute"].reshape(-1).cpu().numpy()
    assert vis.shape[0] == P * Ns and np.isfinite(vis).all() and ((vis == 0) | (vis >= 0.9 - 1e-6)).all()
    rng = np.random.default_rng(3)
    pick = rng.choice(P * Ns, size=3000, replace=False)
    dn = dirs.reshape(-1, 3).cpu().numpy()[pick]
    on = np.repeat(means, Ns, axis=0)[pick]
    boxes = bo.leaf_boxes(means, scales, q.astype(np.float32))
    ocnt, ovis = bo.trace_visibility(boxes, on, dn, means, symm, opacity, normals)
    _compare(vis[pick], cnt[pick], ovis, ocnt, flip_frac=3e-3)
    assert 0.02 < (ovis > 0).mean() < 0.98
