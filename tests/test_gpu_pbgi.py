"""GPU parity of the HIP pbgi LBVH + radiance tracer (csrc/pbgi.hip, drop-in `pbgi.renderer.Renderer` /
`pbgi.bvhhelpers.get_gs_bvh`) against the CPU oracle (oracle/pbgi_oracle.cpp).

The tree is integer work on top of a few exactly specified fp32 operations: node tables, boxes and the sorted (code, primitive)
pairs must be IDENTICAL.  The tracer is fp32 on both sides with the same operation order; expf differs in the last bit, so
hit indices / uvs may differ on a ray that sits on a threshold (at most 1e-3 of the rays, counted, not compared) and radiance /
visibility agree to 2e-5 elsewhere."""
import numpy as np
import pytest
import torch

from oracle import pbgi_oracle as po
from tests import pbgi_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _renderer(sc):
    from pbgi.renderer import Renderer
    R = Renderer()
    R.set_proxy(t(sc["xyz"]), t(sc["scales"]), t(sc["rot"]), t(sc["normals"]), t(sc["opacity"]), t(sc["shs"]))
    R.build_bvh()
    return R


def _compare(gpu, orc, flip_frac=1e-3, tol=2e-5):
    rad, vis, hit, uvs = (x.cpu().numpy() for x in gpu)
    orad, ovis, ohit, ouvs = orc
    assert rad.shape == orad.shape and vis.shape == ovis.shape and hit.shape == ohit.shape and uvs.shape == ouvs.shape
    flip = (hit[..., 0] != ohit[..., 0]) | ((vis[..., 0] > 0) != (ovis[..., 0] > 0))
    assert flip.mean() <= flip_frac, f"{flip.sum()} of {flip.size} rays took another branch"
    ok = ~flip
    for name, a, b in (("radiance", rad, orad), ("visibility", vis, ovis), ("uv", uvs, ouvs)):
        err = np.abs(a[ok] - b[ok]).max() if ok.any() else 0.0
        assert err <= tol, f"{name} differs by {err:.3e}"
    return flip.mean()


@pytest.mark.parametrize("P,dups", [(1, 0), (2, 0), (3, 0), (100, 37), (2000, 0), (200000, 0)])
def test_tree_is_the_oracles_tree(built, P, dups):
    from pbgi.bvhhelpers import GsBvh, get_gs_bvh
    sc = pbgi_scene.make(P=max(P, 20), shells=max(1, min(P // 100, 2000)), seed=P + 1, duplicates=dups)
    xyz, scales = sc["xyz"][:P], sc["scales"][:P]
    info, aabb, srt = GsBvh(t(xyz), t(scales)).tensors(with_sorted=True)
    oinfo, oaabb, osrt = po.build(xyz, scales)
    assert info.dtype == torch.int32 and tuple(info.shape) == (2 * P - 1, 3) and tuple(aabb.shape) == (2 * P - 1, 6)
    assert np.array_equal(srt.cpu().numpy(), osrt)
    assert np.array_equal(info.cpu().numpy(), oinfo)
    assert np.array_equal(aabb.cpu().numpy(), oaabb)
    i2, a2 = get_gs_bvh(t(xyz), t(scales), None, "m_gen_ele", "m_morton_codes", "m_radixsort", "m_hierarchy", "m_bounding_box")
    assert torch.equal(i2, info) and torch.equal(a2, aabb)          # the reference's call signature; deterministic


def test_trace_small_scene_every_ray(built):
    sc = pbgi_scene.make(P=2000, shells=20, S=16, seed=3)
    R = _renderer(sc)
    out = R.render_radiance_with_sampling_SH(t(sc["xyz"]), t(sc["ray_d"]), t(sc["cov_inv"]), 16)
    assert [tuple(o.shape) for o in out] == [(2000, 16, 3), (2000, 16, 1), (2000, 16, 1), (2000, 16, 2)]
    assert out[2].dtype == torch.int32 and out[0].dtype == torch.float32
    info, aabb, _ = po.build(sc["xyz"], sc["scales"])
    assert np.array_equal(R.LBVHNode_info.cpu().numpy(), info) and np.array_equal(R.LBVHNode_aabb.cpu().numpy(), aabb)
    orc = po.trace(info, aabb, sc["xyz"], sc["ray_d"], sc["xyz"], sc["scales"], sc["rot"], sc["normals"], sc["opacity"], sc["cov_inv"], sc["shs"])
    _compare(out, orc)
    hit = orc[2][..., 0]
    assert (hit >= 0).mean() > 0.1 and (hit == -1).any() and (orc[1] < 0.9).mean() > 0.05      # the scene exercises hits and multi-hit rays


def test_chunks_use_the_row_inside_the_chunk(built):
    """GaussianModel.update_radiace traces chunk by chunk (scene/gaussian_model.py:488-502): rows restart at 0 in every chunk
    and the self-hit test compares with that row (oracle header, Q-d)."""
    sc = pbgi_scene.make(P=600, shells=6, S=8, seed=9)
    R = _renderer(sc)
    info, aabb, _ = po.build(sc["xyz"], sc["scales"])
    for lo, hi in ((0, 200), (200, 600)):
        out = R.render_radiance_with_sampling_SH(t(sc["xyz"][lo:hi]), t(sc["ray_d"][lo:hi]), t(sc["cov_inv"]), 8)
        orc = po.trace(info, aabb, sc["xyz"][lo:hi], sc["ray_d"][lo:hi], sc["xyz"], sc["scales"], sc["rot"], sc["normals"], sc["opacity"],
                       sc["cov_inv"], sc["shs"])
        _compare(out, orc, flip_frac=2e-3)


def test_degenerate_inputs(built):
    sc = pbgi_scene.make(P=100, shells=1, S=4, seed=2)
    sc["ray_d"][0, 0] = 0.0                       # zero direction: NaN after normalisation, the ray must still terminate
    sc["ray_d"][1, 1] = [1.0, 0.0, 0.0]           # axis-aligned: zero components take the 1e-6 substitute (Q-e)
    sc["rot"][5] = 0.0                            # zero quaternion: the 1e-8 under the square root keeps it finite
    sc["opacity"][7] = 0.0
    R = _renderer(sc)
    out = R.render_radiance_with_sampling_SH(t(sc["xyz"]), t(sc["ray_d"]), t(sc["cov_inv"]), 4)
    torch.cuda.synchronize()
    info, aabb, _ = po.build(sc["xyz"], sc["scales"])
    orc = po.trace(info, aabb, sc["xyz"], sc["ray_d"], sc["xyz"], sc["scales"], sc["rot"], sc["normals"], sc["opacity"], sc["cov_inv"], sc["shs"])
    rad = out[0].cpu().numpy()
    assert np.isfinite(rad).all()
    keep = np.ones((100, 4), dtype=bool); keep[0, 0] = False
    _compare([o[torch.from_numpy(keep).to(DEV)].unsqueeze(0) for o in out], [o[keep][None] for o in orc], flip_frac=0.02)
    from pbgi.renderer import Renderer
    with pytest.raises(RuntimeError):
        Renderer().render_radiance_with_sampling_SH(t(sc["xyz"]), t(sc["ray_d"]), t(sc["cov_inv"]), 4)
    with pytest.raises(RuntimeError):
        from pbgi.bvhhelpers import GsBvh
        GsBvh(torch.zeros(4, 3), torch.ones(4, 3))                  # CPU tensors: no fallback


def test_full_size_against_an_oracle_subset(built):
    """P = 200 k surfels x 64 rays (BASELINE cfg3's surfel and sample counts); the oracle traces 400 rows drawn at random
    from the whole scene (every region of the tree, not one Morton neighbourhood) plus the first and the last 8."""
    sc = pbgi_scene.make(P=200000, shells=2000, S=64, seed=13)
    R = _renderer(sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = R.render_radiance_with_sampling_SH(t(sc["xyz"]), t(sc["ray_d"]), t(sc["cov_inv"]), 64)
    e1.record(); torch.cuda.synchronize()
    print(f"pbgi trace 200k x 64 rays: {e0.elapsed_time(e1):.2f} ms")
    info, aabb, _ = po.build(sc["xyz"], sc["scales"])
    assert np.array_equal(R.LBVHNode_info.cpu().numpy(), info)
    P = sc["xyz"].shape[0]
    rows = np.unique(np.concatenate([np.random.default_rng(5).choice(P, 400, replace=False), np.arange(8), np.arange(P - 8, P)]))
    orc = po.trace(info, aabb, sc["xyz"][rows], sc["ray_d"][rows], sc["xyz"], sc["scales"], sc["rot"], sc["normals"], sc["opacity"], sc["cov_inv"],
                   sc["shs"])
    ri = torch.from_numpy(rows).to(DEV)
    _compare([o[ri] for o in out], orc, flip_frac=2e-3)
    hit = out[2][..., 0]
    assert (hit >= 0).float().mean().item() > 0.05
    vis = out[1]
    assert torch.isfinite(out[0]).all() and (vis >= 0).all() and (vis <= 1).all()
