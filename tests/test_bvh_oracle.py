"""CPU tests of the visibility-tracer oracle (oracle/bvh_oracle.cpp) against the fixture the REFERENCE's own `RayTracer`
Python produced (scripts/make_golden_bvh.py -> tests/golden/bvh.npz) and against an independent numpy construction."""
import os

import numpy as np

from oracle import bvh_oracle as bo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bvh.npz")


def test_leaf_boxes_match_the_reference_python():
    g = np.load(GOLD)
    P = g["means3D"].shape[0]
    boxes = bo.leaf_boxes(g["means3D"], g["scales"], g["rotations"])
    ref = g["aabbs"][P - 1:]                     # the reference stores the leaves behind the P - 1 internal nodes
    assert ref.shape == boxes.shape
    np.testing.assert_allclose(boxes, ref, rtol=0, atol=2e-6)
    # internal boxes start as the reduce's identity, node flags: internal 0 / leaf 1 (submodules/bvh/__init__.py:32-37)
    assert np.all(g["aabbs"][:P - 1, :3] == 100000) and np.all(g["aabbs"][:P - 1, 3:] == -100000)
    assert np.all(g["nodes"][:P - 1, 4] == 0) and np.all(g["nodes"][P - 1:, 4] == 1) and np.all(g["nodes"][:, :4] == -1)


def test_origin_offset_and_result_keys_of_the_reference():
    g = np.load(GOLD)
    np.testing.assert_allclose(g["rays_o_passed"], g["rays_o_in"] + g["rays_d_in"] * np.float32(0.05), rtol=0, atol=1e-7)
    np.testing.assert_array_equal(g["rays_d_passed"], g["rays_d_in"])
    assert list(g["out_keys"]) == ["contribute", "visibility"]
    assert tuple(g["out_vis_shape"]) == g["rays_d_in"].shape[:-1] + (1,)


def _scene(P, seed, flat=True):
    rng = np.random.default_rng(seed)
    means = rng.uniform(-1, 1, size=(P, 3)).astype(np.float32)
    scales = np.exp(rng.uniform(np.log(0.02), np.log(0.15), size=(P, 3))).astype(np.float32)
    if flat:
        scales[:, 2] = 1e-3
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                  np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                  np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1).astype(np.float64)
    Linv = R * (1.0 / scales.astype(np.float64))[:, None, :]          # build_scaling_rotation(1 / s, q) = R diag(1 / s)
    Cinv = Linv @ Linv.transpose(0, 2, 1)
    symm = np.stack([Cinv[:, 0, 0], Cinv[:, 0, 1], Cinv[:, 0, 2], Cinv[:, 1, 1], Cinv[:, 1, 2], Cinv[:, 2, 2]], -1).astype(np.float32)
    opacity = rng.uniform(0.0, 1.0, size=P).astype(np.float32)
    normals = R[:, :, 2].astype(np.float32)
    return dict(means=means, scales=scales, rots=q, symm=symm, opacity=opacity, normals=normals, R=R, Cinv=Cinv)


def test_trace_against_independent_numpy_fp64():
    sc = _scene(300, 5)
    rng = np.random.default_rng(6)
    S = 16
    d = rng.normal(size=(40, S, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(sc["means"][:40, None].astype(np.float64), d.shape)
    boxes = bo.leaf_boxes(sc["means"], sc["scales"], sc["rots"])
    contrib, vis = bo.trace_visibility(boxes, o, d, sc["means"], sc["symm"], sc["opacity"], sc["normals"], fp64=True)
    # independent: vectorised fp64 over all (ray, surfel) pairs in matrix notation
    oo = (o + 0.05 * d).reshape(-1, 1, 3)
    dd = d.reshape(-1, 1, 3)
    lo, hi = boxes[None, :, :3].astype(np.float64), boxes[None, :, 3:].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - oo) / dd, (hi - oo) / dd
    tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
    enter, leave = tn.max(-1), tf.min(-1)
    hit = (enter <= leave) & (leave > 0)
    s6 = sc["symm"].astype(np.float64)     # (the same fp32-rounded inverse covariances the oracle is given)
    Cinv = np.stack([np.stack([s6[:, 0], s6[:, 1], s6[:, 2]], -1), np.stack([s6[:, 1], s6[:, 3], s6[:, 4]], -1),
                     np.stack([s6[:, 2], s6[:, 4], s6[:, 5]], -1)], 1)[None]
    m = sc["means"].astype(np.float64)[None] - oo
    Cd = np.einsum("rpij,rqj->rpi", Cinv, dd)
    t = np.einsum("rpi,rpi->rp", m, Cd) / np.einsum("rqi,rpi->rp", dd, Cd)
    diff = m - t[..., None] * dd
    power = -0.5 * np.einsum("rpi,rpij,rpj->rp", diff, Cinv, diff)
    facing = np.einsum("pi,rqi->rp", sc["normals"].astype(np.float64), dd) <= 0
    ok = hit & (sc["opacity"][None] >= 1 / 255) & facing & (t >= 0.01) & (power <= 0)
    alpha = np.where(ok, sc["opacity"][None].astype(np.float64) * np.exp(np.minimum(power, 0)), 0.0)
    prod = np.prod(1 - alpha, axis=1)
    ref_vis = np.where(prod < 0.9, 0.0, prod).reshape(vis.shape)
    ref_cnt = np.where(prod < 0.9, 0, ok.sum(1)).reshape(vis.shape)
    sure = np.abs(prod - 0.9).reshape(vis.shape) > 1e-6
    np.testing.assert_allclose(vis[sure], ref_vis[sure], rtol=0, atol=2e-6)
    assert np.array_equal(contrib[sure], ref_cnt[sure])
    assert 0.1 < (ref_vis > 0).mean() < 0.999     # the case exercises both outcomes


def test_fp32_and_fp64_modes_agree_and_edge_cases():
    sc = _scene(200, 9)
    rng = np.random.default_rng(10)
    d = rng.normal(size=(500, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = rng.uniform(-1, 1, size=(500, 3)).astype(np.float32)
    boxes = bo.leaf_boxes(sc["means"], sc["scales"], sc["rots"])
    c32, v32 = bo.trace_visibility(boxes, o, d, sc["means"], sc["symm"], sc["opacity"], sc["normals"])
    c64, v64 = bo.trace_visibility(boxes, o, d, sc["means"], sc["symm"], sc["opacity"], sc["normals"], fp64=True)
    same = (v32 > 0) == (v64 > 0)
    assert same.mean() > 0.995
    # (flat surfels: inverse covariances ~1e6, the fp32 evaluation of t and power loses ~3 digits to cancellation)
    np.testing.assert_allclose(v32[same], v64[same], rtol=0, atol=2e-4)
    # one surfel: the root is its leaf and is entered without a box test; transparent or back-facing surfels never count
    one = _scene(1, 3)
    b1 = bo.leaf_boxes(one["means"], one["scales"], one["rots"])
    dd = -one["normals"][0][None] + 0.0
    oo = (one["means"][0] + one["normals"][0] * 0.5)[None].astype(np.float32)
    c, v = bo.trace_visibility(b1, oo, dd, one["means"], one["symm"], np.array([0.5], np.float32), one["normals"])
    assert c[0] == 0 and v[0] == 0.0          # alpha = 0.5: the product 0.5 is below the 0.9 cut-off -> blocked, count stays 0
    c, v = bo.trace_visibility(b1, oo, dd, one["means"], one["symm"], np.array([0.05], np.float32), one["normals"])
    assert c[0] == 1 and abs(v[0] - 0.95) < 1e-3
    c, v = bo.trace_visibility(b1, oo, dd, one["means"], one["symm"], np.array([0.001], np.float32), one["normals"])
    assert c[0] == 0 and v[0] == 1.0
    c, v = bo.trace_visibility(b1, oo, -dd, one["means"], one["symm"], np.array([0.5], np.float32), one["normals"])
    assert c[0] == 0 and v[0] == 1.0
