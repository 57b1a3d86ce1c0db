"""Whole stage-2 training iteration (svgir_harness.workloads.TrainStep: shade -> rasterize -> unpack -> L1 + SSIM -> backward ->
densification statistics -> Adam) against the CHAIN of the CPU oracles:

    incident lattice (oracle/epilogue_oracle.py, fp64)  ->  shading + packing (oracle/shading_oracle.py, torch fp64, autograd)
    ->  rasterizer (oracle/svgir_oracle.cpp, fp32, hand-derived backward)  ->  unpack + L1 / SSIM (epilogue_oracle, torch fp64, autograd)
    ->  add_densification_stats restated  ->  torch.optim.Adam (fp32)

i.e. the sequence of the reference's train.py:133-143, 221 / gaussian_renderer/svgss.py:15-262 / scene/gaussian_model.py:775-813,
1270-1276.  Checked: the loss, every parameter's gradient, xyz_gradient_accum / denom / weights_accum, and the parameters after one
and after two optimizer steps -- at a reduced size and on the cfg3_train scene (BASELINE.json configs[2]) at full size.  The same
chain forward-only at the evaluation widths (cfg3_eval, Ns = 384) through render_svgss_view."""
import math

import numpy as np
import pytest
import torch

from oracle import epilogue_oracle as eo
from oracle import oracle as orc
from oracle import shading_oracle as so
from svgir_harness import render_view, runner, scenes, shade_inputs, workloads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAMBDA_DSSIM = 0.2


def _cmp(name, a, b, tol=2e-4, flip_frac=5e-4):
    a = torch.as_tensor(a).detach().double().cpu().numpy().reshape(-1)
    b = np.asarray(torch.as_tensor(b).detach().double().cpu().numpy(), dtype=np.float64).reshape(-1)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > tol * (scale + np.abs(b))
    assert bad.mean() <= flip_frac, f"{name}: {bad.sum()}/{bad.size} entries off (max {np.abs(a - b).max():.3e}, scale {scale:.3e})"


def _raster_oracle(sc, par, feats, vfeats):
    sc_o = dict(sc)
    sc_o.update(means3D=par["xyz"], scales=par["scaling"], rotations=par["rotation"], opacities=par["opacity"], shs=par["shs"],
                features=feats, vfeatures=vfeats)
    return orc.OracleRun(sc_o, orc.SVGSS)


def _shade_chunks(par, ids, vis, dirs, campos, view3, training, g_feat=None, g_vfeat=None, chunk=4096):
    """Shading + packing oracle over the surfels `ids` in chunks.  Forward: packed rows.  With upstream gradients of the packed rows:
    also the gradients of base_color, roughness, normal, radiance (rows `ids`) and env (summed), by torch.autograd per chunk."""
    S, VS = (4, 52) if training else (7, 64)
    F_, VF = np.zeros((len(ids), S)), np.zeros((len(ids), VS))
    grads = None
    # the reference's get_radiances (scene/gaussian_model.py:323-324): the cache is detached, the scalar ratio learns
    ratio_form = "radiance_ratio" in par
    mats = ("base_color", "roughness", "normal") + (() if ratio_form else ("radiance",))
    if g_feat is not None:
        grads = {k: np.zeros(par[k][ids].shape) for k in mats}
        grads["env"] = np.zeros(par["env"].shape)
        if ratio_form:
            grads["radiance_ratio"] = np.zeros(())
    for c0 in range(0, len(ids), chunk):
        ii = ids[c0:c0 + chunk]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()  # noqa: E731
        lv = {k: t(par[k][ii]).requires_grad_(g_feat is not None) for k in mats}
        env = t(par["env"]).requires_grad_(g_feat is not None)
        if ratio_form:   # (the product is an fp32 product in the reference and in the kernels: rounded to fp32 here too)
            ratio = torch.tensor(float(par["radiance_ratio"]), dtype=torch.float64, requires_grad=g_feat is not None)
            raw = t(par["radiance"][ii])
            prod32 = torch.from_numpy((par["radiance"][ii].astype(np.float32) * np.float32(par["radiance_ratio"])).astype(np.float64))
            incident = torch.nan_to_num(prod32 + (raw * ratio - (raw * ratio).detach()), nan=0.0)   # fp32 value, fp64 gradient path
        else:
            incident = lv["radiance"]
        xyz = t(par["xyz"][ii])
        viewdirs = torch.nn.functional.normalize(t(campos)[None] - xyz, dim=-1)
        ref = so.shade(lv["base_color"], lv["roughness"], lv["normal"], viewdirs, incident, t(vis[ii]), t(dirs[c0:c0 + chunk]),
                       torch.full((len(ii), dirs.shape[1], 1), 2 * math.pi, dtype=torch.float64), env)
        f, vf = so.pack(ref, lv["base_color"], lv["roughness"], lv["normal"], t(view3), training)
        F_[c0:c0 + chunk], VF[c0:c0 + chunk] = f.detach().numpy(), vf.detach().numpy()
        if g_feat is not None:
            ((f * t(g_feat[ii])).sum() + (vf * t(g_vfeat[ii])).sum()).backward()
            for k in lv:
                grads[k][c0:c0 + chunk] = lv[k].grad.numpy()
            grads["env"] += env.grad.numpy()
            if ratio_form:
                grads["radiance_ratio"] += ratio.grad.numpy()
    return F_, VF, grads


def oracle_iteration(sc, par, vis, geo_n, offsets, gt, Ns, want_grads=True):
    """One iteration of the chain on host arrays `par` (fp32 numpy, the parameter block).  Returns loss, images, gradients (fp32
    numpy, dense), and what add_densification_stats adds."""
    P = par["xyz"].shape[0]
    S, VS = 4, 52
    view3 = np.asarray(sc["viewmatrix"], dtype=np.float64)[:3, :3]
    # which surfels can the composite read at all?  (the packed rows of the others are never touched)
    o0 = _raster_oracle(sc, par, np.zeros((P, S), np.float32), np.zeros((P, VS), np.float32))
    o0.forward()
    ids = np.nonzero(o0.images()["radii"] > 0)[0]
    dirs, _ = eo.fibonacci_dirs(geo_n[ids], Ns, None if offsets is None else offsets[ids])
    F_, VF, _ = _shade_chunks(par, ids, vis, dirs, sc["campos"], view3, True)
    feats, vfeats = np.zeros((P, S), np.float32), np.zeros((P, VS), np.float32)
    feats[ids], vfeats[ids] = F_, VF
    o = _raster_oracle(sc, par, feats, vfeats)
    R = o.forward()
    im = o.images()
    t = lambda k: torch.from_numpy(np.asarray(im[k], dtype=np.float64)).requires_grad_(True)  # noqa: E731
    opac, feat, vfeat = t("opacity"), t("feature"), t("vfeature")
    res = eo.unpack_svgss_torch(opac, feat, vfeat, torch.from_numpy(np.asarray(sc["bg"])).double(), True)
    l1, ssim = eo.l1_ssim_torch(res["pbr"], torch.from_numpy(gt).double())
    loss = (1.0 - LAMBDA_DSSIM) * l1 + LAMBDA_DSSIM * (1.0 - ssim)
    out = dict(R=R, loss=float(loss.detach()), pbr=res["pbr"].detach().numpy(), radii=im["radii"], weights=im["weights"])
    if not want_grads:
        return out
    loss.backward()
    z = lambda ch: np.zeros((ch, sc["H"], sc["W"]))  # noqa: E731
    gz = lambda t_, ch: z(ch) if t_.grad is None else t_.grad.numpy()  # noqa: E731  (the training loss reads pbr only: no gradient
    o.backward(z(3), z(3), z(1), gz(opac, 1), gz(feat, S), gz(vfeat, VS // 4))   # reaches the plain feature planes)
    gr = o.grads()
    _, _, sg = _shade_chunks(par, ids, vis, dirs, sc["campos"], view3, True, gr["features"].astype(np.float64), gr["vfeatures"].astype(np.float64))
    g = {"xyz": gr["means3D"], "scaling": gr["scales"], "rotation": gr["rotations"], "opacity": gr["opacity"], "shs": gr["sh"],
         "env": sg["env"]}
    for k in ("base_color", "roughness", "normal", "radiance"):
        if k in sg:
            g[k] = np.zeros(par[k].shape)
            g[k][ids] = sg[k]
    if "radiance_ratio" in sg:
        g["radiance_ratio"] = sg["radiance_ratio"]
    out["grads"] = {k: np.asarray(v, dtype=np.float32).reshape(par[k].shape) for k, v in g.items()}
    vf_ = im["radii"] > 0   # add_densification_stats (scene/gaussian_model.py:1270-1276)
    out["stat_grad"] = np.where(vf_, np.linalg.norm(gr["means2D"][:, :2].astype(np.float64), axis=-1), 0.0)[:, None]
    out["stat_denom"] = vf_.astype(np.float64)[:, None]
    return out


def _run_and_compare(ts, sc, steps, tol=2e-4, flip=5e-4):
    P, Ns = ts.P, ts.Ns
    par0 = {k: v.detach().cpu().numpy().copy() for k, v in ts.params.items()}
    vis = ts.visibility.cpu().numpy()
    geo_n = ts.geo_n.cpu().numpy()
    gt = ts.gt.cpu().numpy()
    # the reference optimizer on the oracle's gradients
    ref_p = {k: torch.nn.Parameter(torch.from_numpy(v.copy())) for k, v in par0.items()}
    lrs = {g["name"]: g["lr"] for g in ts.optimizer.param_groups}
    adam = torch.optim.Adam([{"params": [ref_p[k]], "lr": lrs[k]} for k in ref_p], lr=0.0, eps=1e-15)
    acc_g, acc_d, acc_w = np.zeros((P, 1)), np.zeros((P, 1)), np.zeros((P, 1))
    gen = torch.Generator(DEV).manual_seed(77)
    for it in range(steps):
        offs = torch.rand(P, device=DEV, generator=gen) * (2 * math.pi)
        par = {k: v.detach().numpy() for k, v in ref_p.items()}
        ref = oracle_iteration(sc, par, vis, geo_n, offs.cpu().numpy().reshape(-1, 1), gt, Ns)
        R, pbr, loss = ts.step(offsets=offs, keep_grads=True)
        torch.cuda.synchronize()
        assert R == ref["R"], (it, R, ref["R"])
        assert abs(float(loss) - ref["loss"]) <= 2e-5 * abs(ref["loss"]), (it, float(loss), ref["loss"])
        _cmp(f"pbr[{it}]", pbr, ref["pbr"], tol, flip)
        for k, p in ts.params.items():
            if k not in ref["grads"]:   # the detached radiance cache: in the optimizer (gaussian_model.py:527), never a gradient
                assert k == "radiance" and p.grad is None
                continue
            if p.dim() == 0:
                assert abs(float(p.grad) - float(ref["grads"][k])) <= 1e-3 * abs(float(ref["grads"][k])) + 1e-7, (k, float(p.grad), float(ref["grads"][k]))
                continue
            _cmp(f"grad {k}[{it}]", p.grad, ref["grads"][k], tol=5e-4, flip_frac=2e-3)
        for k, p in ref_p.items():
            p.grad = torch.from_numpy(ref["grads"][k].copy()) if k in ref["grads"] else None
        adam.step()
        acc_g += ref["stat_grad"]; acc_d += ref["stat_denom"]; acc_w += ref["weights"]
        _cmp(f"xyz_gradient_accum[{it}]", ts.xyz_gradient_accum, acc_g, tol=5e-4, flip_frac=2e-3)
        assert np.array_equal(ts.denom.cpu().numpy(), acc_d.astype(np.float32)), it
        _cmp(f"weights_accum[{it}]", ts.weights_accum, acc_w, tol, flip)
        # Parameters after the step, as the update in units of the learning rate ((p_new - p_old) / lr lies in [-1, 1]; the first Adam
        # step is -lr * sign(g) wherever |g| >> eps: entries whose gradient is noise may take either sign, hence the fraction)
        for k, p in ts.params.items():
            upd = (p.detach().cpu().double() - torch.from_numpy(par0[k]).double()) / lrs[k]
            upd_ref = (ref_p[k].detach().double() - torch.from_numpy(par0[k]).double()) / lrs[k]
            if k not in ref["grads"]:
                assert float(upd.abs().max()) == 0.0, k   # untouched
                continue
            if p.dim() == 0:
                assert abs(float(upd) - float(upd_ref)) <= 2e-3, (k, float(upd), float(upd_ref))
                continue
            big = torch.from_numpy(np.abs(ref["grads"][k]) > 1e-3 * np.abs(ref["grads"][k]).max())   # gradients above the noise
            if int(big.sum()):
                _cmp(f"update {k}[{it}]", upd[big], upd_ref[big], tol=2e-3, flip_frac=5e-3)
        for k, p in ts.params.items():
            p.grad = None


def test_train_step_matches_the_oracle_chain_small(built):
    dev = torch.device(DEV)
    sc = scenes.surface_scene(P=4000, W=160, H=128, seed=71, sh_degree=3, variant="svgss", S=4, VS=52, scale_lo=0.02, scale_hi=0.07)
    for fused, rad_grad in ((True, False), (False, False), (True, True)):
        ts = workloads.TrainStep(dev, seed=9, Ns=32, fused=fused, scene=sc, radiance_grad=rad_grad)
        _run_and_compare(ts, sc, steps=2)


def test_train_step_matches_the_oracle_chain_cfg3_train(built):
    """BASELINE.json configs[2] end to end: P = 200 000, 800 x 800, S = 4, VS = 52, Ns = 64, in-kernel lattice with random azimuths."""
    dev = torch.device(DEV)
    sc = scenes.make("cfg3_train")
    ts = workloads.TrainStep(dev, seed=5, name="cfg3_train", Ns=64, fused=True, scene=sc)
    _run_and_compare(ts, sc, steps=1)


def test_eval_view_matches_the_oracle_chain_cfg3_eval(built):
    """The same chain forward only at the evaluation widths (S = 7, VS = 64, Ns = 384, evaluation lattice) on the cfg3 scene at full
    size, through render_svgss_view (fused: the contribution pre-pass selects the shaded surfels)."""
    dev = torch.device(DEV)
    sc = scenes.make("cfg3_eval")
    sct = runner.to_torch(sc, dev)
    P, Ns = sc["means3D"].shape[0], 384
    q = torch.nn.functional.normalize(sct["rotations"], dim=-1)
    r, x, y, z = q.unbind(-1)
    geo_n = torch.nn.functional.normalize(torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], dim=-1), dim=-1)
    d = shade_inputs.make(P, Ns, seed=6, device=dev, geo_normals=geo_n, with_dirs=False)
    from gaussian_renderer import shading
    st = runner.settings(sct, "svgss")
    mat = dict(d, viewdirs=torch.nn.functional.normalize(st.campos[None, :] - sct["means3D"], dim=-1),
               dirs=shading.FibonacciLattice(geo_n, Ns, None), areas=None)
    with torch.no_grad():
        res, _ = render_view.render_svgss_view(sct, mat, shade_inputs.Light(d["env"]), False, fused=True)
    torch.cuda.synchronize()
    # ---- the oracles ----
    par = dict(xyz=sc["means3D"], scaling=sc["scales"], rotation=sc["rotations"], opacity=sc["opacities"], shs=sc["shs"],
               base_color=d["base_color"].cpu().numpy(), roughness=d["roughness"].cpu().numpy(), normal=d["normals"].cpu().numpy(),
               radiance=d["radiance"].cpu().numpy(), env=d["env"].cpu().numpy())
    vis = d["visibility"].cpu().numpy()
    o0 = _raster_oracle(sc, par, np.zeros((P, 7), np.float32), np.zeros((P, 64), np.float32))
    o0.forward()
    ids = np.nonzero(o0.images()["weights"][:, 0] > 0)[0]     # the surfels that are blended at all (the images depend on no other)
    dirs, _ = eo.fibonacci_dirs(geo_n.cpu().numpy()[ids], Ns, None)
    view3 = np.asarray(sc["viewmatrix"], dtype=np.float64)[:3, :3]
    F_, VF, _ = _shade_chunks(par, ids, vis, dirs, sc["campos"], view3, False, chunk=1024)
    feats, vfeats = np.zeros((P, 7), np.float32), np.zeros((P, 64), np.float32)
    feats[ids], vfeats[ids] = F_, VF
    o = _raster_oracle(sc, par, feats, vfeats)
    R = o.forward()
    im = o.images()
    t = lambda k: torch.from_numpy(np.asarray(im[k], dtype=np.float64))  # noqa: E731
    exp = eo.unpack_svgss_torch(t("opacity"), t("feature"), t("vfeature"), torch.from_numpy(sc["bg"]).double(), False)
    exp.update(render=t("color"), depth=t("depth"), opacity=t("opacity"))
    assert res["num_rendered"] == R and np.array_equal(res["radii"].cpu().numpy(), im["radii"])
    for k in ("render", "depth", "opacity", "pbr", "normal", "base_color", "roughness", "local_lights", "visibility", "lights", "direct", "indirect"):
        _cmp(k, res[k], exp[k].numpy())
