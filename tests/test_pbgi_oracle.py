"""CPU tests of oracle/pbgi_oracle.cpp (restatement of the reference's pbgi LBVH build + radiance tracer) and of the
reference-generated glue fixture tests/golden/pbgi_glue.npz."""
import json
import os

import numpy as np
import pytest

from oracle import pbgi_oracle as po
from tests import pbgi_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check_tree(info, aabb, srt, P):
    L = P - 1
    assert (np.diff(srt[:, 0].astype(np.int64)) >= 0).all()                       # sorted by code
    assert sorted(srt[:, 1].tolist()) == list(range(P))                              # a permutation
    assert (info[L:, 0] == 0).all() and (info[L:, 1] == 0).all() and (info[L:, 2] == srt[:, 1]).all()
    seen = np.zeros(2 * P - 1, dtype=np.int32)
    lo, hi = {}, {}

    def walk(n):   # returns the range of sorted positions below n
        seen[n] += 1
        if n >= L:
            return n - L, n - L
        a0, a1 = walk(int(info[n, 0])); b0, b1 = walk(int(info[n, 1]))
        assert a1 + 1 == b0                                                         # children cover adjacent ranges, left first
        for c in (int(info[n, 0]), int(info[n, 1])):
            assert (aabb[n, :3] <= aabb[c, :3]).all() and (aabb[n, 3:] >= aabb[c, 3:]).all()
        assert (aabb[n, :3] == np.minimum(aabb[info[n, 0], :3], aabb[info[n, 1], :3])).all()
        assert (aabb[n, 3:] == np.maximum(aabb[info[n, 0], 3:], aabb[info[n, 1], 3:])).all()
        return a0, b1
    import sys
    sys.setrecursionlimit(10000)
    assert walk(0) == (0, P - 1)
    assert (seen == 1).all()


@pytest.mark.parametrize("P,dups", [(1, 0), (2, 0), (3, 0), (100, 0), (100, 37), (2000, 0)])
def test_build_is_a_valid_lbvh(P, dups):
    sc = pbgi_scene.make(P=max(P, 20), shells=1 if P < 20 else 20, seed=P, duplicates=dups)
    xyz, scales = sc["xyz"][:P], sc["scales"][:P]
    info, aabb, srt = po.build(xyz, scales)
    _check_tree(info, aabb, srt, P)
    # leaf boxes: centre +- 3 max|scale| (get_elements.slang:86-105)
    ext = 3.0 * np.abs(scales).max(1, keepdims=True)
    order = srt[:, 1]
    assert np.array_equal(aabb[P - 1:, :3], (xyz - ext)[order]) and np.array_equal(aabb[P - 1:, 3:], (xyz + ext)[order])


def test_equal_codes_are_split_by_position():
    xyz = np.zeros((9, 3), dtype=np.float32); xyz[8] = 1.0        # eight identical centres + one to span an extent
    info, aabb, srt = po.build(xyz, np.full((9, 3), 0.01, dtype=np.float32))
    assert len(set(srt[:8, 0].tolist())) == 1 and srt[:8, 1].tolist() == list(range(8))   # stable
    _check_tree(info, aabb, srt, 9)


def _brute_first_hit(sc, row, s):
    """Tree-free closest accepted primitive of the FIRST query of one ray (t in [0.042, 0.2)), or None if no primitive is
    accepted below 0.2 (then the reference's answer depends on which far leaves its traversal happens to visit)."""
    o = sc["xyz"][row].astype(np.float64)
    d = sc["ray_d"][row, s].astype(np.float64); d /= np.linalg.norm(d)
    R = pbgi_scene.rotmat(sc["rot"].astype(np.float64))
    nw = R[:, :, 2]
    denom = nw @ d
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((sc["xyz"] - o) * nw).sum(1) / denom
    pos = o + t[:, None] * d
    w = pos - sc["xyz"]
    pm = np.einsum("pji,pj->pi", R, w)                                 # R^T w
    sx, sy = sc["scales"][:, 0].astype(np.float64), sc["scales"][:, 1].astype(np.float64)
    dis = (pm[:, 0] / sx) ** 2 + (pm[:, 1] / sy) ** 2
    ci = sc["cov_inv"].astype(np.float64)
    dd = -w
    power = -0.5 * (dd[:, 0] ** 2 * ci[:, 0] + dd[:, 1] ** 2 * ci[:, 3] + dd[:, 2] ** 2 * ci[:, 5] + 2 * dd[:, 0] * dd[:, 1] * ci[:, 1] +
                    2 * dd[:, 0] * dd[:, 2] * ci[:, 2] + 2 * dd[:, 1] * dd[:, 2] * ci[:, 4])
    alpha = np.minimum(0.99, sc["opacity"][:, 0] * np.exp(np.minimum(power, 0)))
    n = sc["normals"] / np.linalg.norm(sc["normals"], axis=1, keepdims=True)
    ok = (np.abs(denom) >= 1e-6) & (t >= 0.042) & (power <= 0) & (alpha >= 1 / 255) & (dis <= 9) & ((n @ d) < 0)
    margin = np.abs(t - 0.2).min() if ok.any() else 1.0
    near = ok & (t < 0.2)
    if not near.any():
        return None, margin
    order = np.argsort(np.where(near, t, np.inf))
    gap = t[order[1]] - t[order[0]] if near.sum() > 1 else 1.0
    return int(order[0]), min(margin, gap)


def test_trace_first_hits_agree_with_a_tree_free_search():
    sc = pbgi_scene.make(P=600, shells=6, S=8, seed=11)
    info, aabb, _ = po.build(sc["xyz"], sc["scales"])
    rad, vis, hit, uvs = po.trace(info, aabb, sc["xyz"], sc["ray_d"], sc["xyz"], sc["scales"], sc["rot"], sc["normals"], sc["opacity"],
                                  sc["cov_inv"], sc["shs"])
    assert rad.shape == (600, 8, 3) and vis.shape == (600, 8, 1) and hit.dtype == np.int32 and uvs.shape == (600, 8, 2)
    assert np.isfinite(rad).all() and (rad >= 0).all() and (rad <= 10).all() and (vis >= 0).all() and (vis <= 1).all()
    checked = hits = 0
    for row in range(0, 600, 7):
        for s in range(8):
            g, margin = _brute_first_hit(sc, row, s)
            if margin < 1e-4:          # too close to a threshold for an fp32-vs-fp64 comparison
                continue
            h = int(hit[row, s, 0])
            if g is None:
                assert h in (-1, 0)    # (0: the reference's "accepted beyond t_max" answer, see the oracle's header Q-b)
            elif g == row:
                assert h == -1         # a self hit ends the ray (Q-d)
            else:
                assert h == g
                hits += 1
            checked += 1
    assert checked > 300 and hits > 50
    # rays that hit nothing keep full visibility and no radiance
    none = hit[..., 0] == -1
    assert (vis[none] == 1).all() and (rad[none] == 0).all() and (uvs[none] == 0).all()
    # every hit lowers the transmittance
    assert (vis[hit[..., 0] > 0] < 1).all()


def test_trace_is_deterministic_and_rows_are_independent():
    sc = pbgi_scene.make(P=300, shells=3, S=4, seed=5)
    info, aabb, _ = po.build(sc["xyz"], sc["scales"])
    args = (sc["xyz"], sc["scales"], sc["rot"], sc["normals"], sc["opacity"], sc["cov_inv"], sc["shs"])
    a = po.trace(info, aabb, sc["xyz"], sc["ray_d"], *args)
    b = po.trace(info, aabb, sc["xyz"], sc["ray_d"], *args)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # a chunk that starts at row 100: the self-hit test uses the row INSIDE the chunk (Q-d), everything else is per ray
    c = po.trace(info, aabb, sc["xyz"][100:200], sc["ray_d"][100:200], *args)
    same = np.array_equal(c[2], a[2][100:200])
    assert c[0].shape == (100, 4, 3) and (same or (c[2] != a[2][100:200]).mean() < 0.2)


def test_reference_glue_fixture():
    """What the reference's own Python binds to which kernel parameter (scripts/make_golden_pbgi.py ran pbgi/renderer.py and
    pbgi/bvhhelpers.py with a recording slangtorch): the contract svg-ir_amd/pbgi mirrors."""
    doc = json.loads(str(np.load(os.path.join(GOLD, "pbgi_glue.npz"))["doc"]))
    P, N, S = doc["P"], doc["N"], doc["S"]
    kernels = [c["kernel"] for c in doc["build_calls"]]
    assert kernels[:4] == ["generateGaussianElements", "morton_codes", "radix_sort", "hierarchy"] and kernels[-1] == "set_root"
    gen = doc["build_calls"][0]["args"]
    assert gen["centers"]["source"] == "xyz" and gen["scales"]["source"] == "scaling" and gen["ele_aabb"]["shape"] == [P, 6]
    assert doc["build_calls"][2]["grid"] == [1, 1, 1]                      # the single-workgroup sort
    assert doc["info_shape"] == [2 * P - 1, 3] and doc["info_dtype"] == "torch.int32" and doc["aabb_shape"] == [2 * P - 1, 6]
    (tr,) = doc["trace_calls"]
    assert tr["kernel"] == "render_radiance_with_sampling_SH" and tr["grid"] == [(N + 255) // 256, S, 1] and tr["block"] == [256, 1, 1]
    a = tr["args"]
    assert a["N"]["scalar"] == N and a["S"]["scalar"] == S
    src = {k: v.get("source") for k, v in a.items() if "shape" in v}
    assert src == {"ray_origins": "ray_o", "ray_directions": "ray_d", "g_lbvh_info": "LBVHNode_info", "g_lbvh_aabb": "LBVHNode_aabb",
                   "centers": "xyz", "scales": "scaling", "rotates": "rotation", "colors": "geo_normal", "opacity": "opacity",
                   "cov3D_inverse": "cov3D_inv", "SHs": "features", "Le": "radiance", "visibility": "visibility",
                   "hit_indices": "hit_indices", "uvs": "uvs"}
    assert a["Le"]["fill"] == 0.0 and a["visibility"]["fill"] == 1.0 and a["hit_indices"]["fill"] == 0.0 and a["uvs"]["fill"] == 0.0
    assert [o["shape"] for o in doc["outputs"]] == [[N, S, 3], [N, S, 1], [N, S, 1], [N, S, 2]]
    assert [o["dtype"] for o in doc["outputs"]] == ["float32", "float32", "int32", "float32"]
