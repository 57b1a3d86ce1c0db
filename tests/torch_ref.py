"""Independent, differentiable fp64 PyTorch restatement of the rasterizer FORWARD (dense: every pixel x every
Gaussian), written in ordinary matrix notation rather than following the reference's code layout.

Used to pin the oracle (oracle/svgir_oracle.cpp):
  * forward outputs of the oracle (fp64 mode) must agree with this to ~1e-10;
  * the oracle's hand-derived backward must equal torch.autograd of this forward wherever the reference's
    backward is a true derivative.  Where the reference deliberately is NOT a derivative, this module mirrors the
    same stop-gradients (bilinear corner weights and the depth-differencing offset are constants; the local
    homography is a constant) and the tests account for the explicit extra terms (Q4 x10 on normals, Q5).

Integer decisions (visibility, tile rectangles, depth order) are taken from the oracle run and are not
differentiated.
"""
import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_rgb(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] unit -> rgb before the +0.5 / clamp."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def quat_to_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    return R


def forward(leaves, sc, consts, variant):
    """leaves: dict of float64 torch tensors (means3D, scales, rotations, opacities, shs, features, vfeatures,
    means2D[P,3] zero offset in NDC).  sc: the numpy scene (camera, sizes, config).  consts: dict from the oracle
    fp64 run: radii[P], rect[P,4] (x0,y0,x1,y1 tiles), order (stable depth order of all P), Jinv[P,10],
    lambda[P,2].  Returns dict of CHW outputs."""
    dt = torch.float64
    W, H = sc["W"], sc["H"]
    svgss = variant == "svgss"
    cfg = [float(c) for c in sc["config"]] if svgss else [1.0, 1.0, 1.0]
    surface, normalize_depth, pix_depth = cfg[0] > 0, cfg[1] > 0, cfg[2] > 0
    V = torch.tensor(np.asarray(sc["viewmatrix"], dtype=np.float64))    # = W2C^T
    PM = torch.tensor(np.asarray(sc["projmatrix"], dtype=np.float64))   # full projection, row-vector convention
    campos = torch.tensor(np.asarray(sc["campos"], dtype=np.float64))
    bg = torch.tensor(np.asarray(sc["bg"], dtype=np.float64))
    fx = W / (2.0 * sc["tanfovx"])
    fy = H / (2.0 * sc["tanfovy"])
    mod = float(sc.get("scale_modifier", 1.0))

    m = leaves["means3D"]
    P = m.shape[0]
    ones = torch.ones((P, 1), dtype=dt)
    hom = torch.cat([m, ones], dim=-1) @ PM                      # [P,4]
    pw = 1.0 / (hom[:, 3] + 0.0000001)
    ndc = hom[:, :2] * pw[:, None] + leaves["means2D"][:, :2]     # NDC offset leaf carries dL/dNDC
    pview = (torch.cat([m, ones], dim=-1) @ V)[:, :3]
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=-1)
    Wrot = V[:3, :3].T                                            # W2C rotation
    Rq = quat_to_R(leaves["rotations"])
    n_view = (Wrot @ Rq[:, :, 2].T).T                             # view-space normal [P,3]
    # Q4: the reference scales the gradient that reaches the per-Gaussian normal by 10 (value unchanged)
    n_blend = n_view * 10.0 - (n_view * 9.0).detach()
    s = leaves["scales"] * mod
    sz = torch.zeros_like(s[:, 2]) if (mod * (1.0 if surface else 0.0)) != 0 else leaves["scales"][:, 2]
    Sd = torch.stack([s[:, 0], s[:, 1], sz], dim=-1)
    Sigma = Rq @ torch.diag_embed(Sd * Sd) @ Rq.transpose(1, 2)
    tx, ty, tz = pview[:, 0], pview[:, 1], pview[:, 2]
    limx, limy = 1.3 * sc["tanfovx"], 1.3 * sc["tanfovy"]
    vis = torch.tensor(consts["radii"] > 0)
    assert bool(((tx / tz).abs()[vis] < limx).all()) and bool(((ty / tz).abs()[vis] < limy).all()), \
        "test scene must not hit the frustum clamp (not a true derivative there)"
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], dim=-1),
                     torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], dim=-1)], dim=1)   # [P,2,3]
    Tm = J @ Wrot                                                                          # [P,2,3]
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], dim=-1)
    d = m - campos[None]
    dirs = d / d.norm(dim=-1, keepdim=True)
    rgb = torch.clamp_min(sh_rgb(int(sc["sh_degree"]), leaves["shs"], dirs) + 0.5, 0.0)
    opac = leaves["opacities"].reshape(-1)

    # ---- dense compositing in depth order ----
    order = torch.tensor(consts["order"].astype(np.int64))
    rect = consts["rect"][consts["order"]]
    visible = (consts["radii"] > 0)[consts["order"]]
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pxs, pys = xs.reshape(-1), ys.reshape(-1)
    tix, tiy = pxs // 16, pys // 16
    valid = (visible[None, :] & (tix[:, None] >= rect[None, :, 0]) & (tix[:, None] < rect[None, :, 2])
             & (tiy[:, None] >= rect[None, :, 1]) & (tiy[:, None] < rect[None, :, 3]))
    valid = torch.tensor(valid)
    pxf = torch.tensor(pxs, dtype=dt)[:, None]
    pyf = torch.tensor(pys, dtype=dt)[:, None]
    po = pix[order]
    co = conic[order]
    dx = po[None, :, 0] - pxf
    dy = po[None, :, 1] - pyf
    power = -0.5 * (co[None, :, 0] * dx * dx + co[None, :, 2] * dy * dy) - co[None, :, 1] * dx * dy
    alpha = torch.clamp_max(opac[order][None, :] * torch.exp(power), 0.99)
    contrib = valid & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
    keep = 1.0 - a_eff
    T_before = torch.cumprod(torch.cat([torch.ones_like(keep[:, :1]), keep[:, :-1]], dim=1), dim=1)
    stop = contrib & (T_before * (1.0 - alpha) < 0.0001)
    done = torch.cumsum(stop.to(torch.int64), dim=1) > 0
    blended = contrib & ~done
    w = torch.where(blended, alpha * T_before, torch.zeros_like(alpha))
    T_final = torch.prod(torch.where(blended, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
    T_final = torch.clamp_max(T_final, 1 - 0.000001)

    Jc = torch.tensor(consts["Jinv"])[order]
    dep = pview[order][:, 2][None, :].expand(dx.shape)
    if surface and pix_depth:
        du = (dx * Jc[None, :, 0] + dy * Jc[None, :, 1]).detach()
        dv = (dx * Jc[None, :, 2] + dy * Jc[None, :, 3]).detach()
        dep = dep - (du * Jc[None, :, 6] + dv * Jc[None, :, 9])
    D = (w * dep).sum(1)
    out = {}
    out["color"] = (w @ rgb[order]) + T_final[:, None] * bg[None]
    out["normal"] = (w.detach() @ n_blend[order] + (w @ n_view[order].detach()) - (w.detach() @ n_view[order].detach())) if surface else torch.zeros((W * H, 3), dtype=dt)
    out["depth"] = (D / (1.0 - T_final) if normalize_depth else D + T_final * 10.0)[:, None]
    out["opacity"] = (1.0 - T_final)[:, None]
    out["feature"] = w @ leaves["features"][order]
    if svgss:
        VS = leaves["vfeatures"].shape[1]
        if surface and pix_depth:
            lam = torch.tensor(consts["lambda"])[order]
            u = torch.clamp(du / (0.5 * lam[None, :, 0] + 0.1) * 0.5 + 0.5, 0.001, 0.999)
            v = torch.clamp(dv / (0.5 * lam[None, :, 1] + 0.1) * 0.5 + 0.5, 0.001, 0.999)
            cw = torch.stack([(1 - u) * (1 - v), u * (1 - v), (1 - u) * v, u * v], dim=-1).detach()   # [N,P,4]
        else:
            cw = torch.zeros(dx.shape + (4,), dtype=dt)
        vf = leaves["vfeatures"][order].reshape(P, VS // 4, 4)
        out["vfeature"] = torch.einsum("np,npk,pck->nc", w, cw, vf)
    out["weights"] = torch.zeros(P, dtype=dt).index_add(0, order, w.sum(0))
    res = {k: (v.T.reshape(-1, H, W) if k != "weights" else v) for k, v in out.items()}
    res["_blended"] = blended
    res["_order"] = order
    res["_n_view_normal_scale"] = 1.0
    return res


def consts_from_oracle(o, sc):
    """Integer decisions + non-differentiated per-Gaussian fields of an oracle run (fp64)."""
    radii = o.get("radii")
    P = radii.shape[0]
    means2D = o.get("means2D").reshape(P, 2)
    depths = o.get("depths")
    gx, gy = (sc["W"] + 15) // 16, (sc["H"] + 15) // 16
    rect = np.zeros((P, 4), dtype=np.int64)
    r = radii.astype(np.float64)
    rect[:, 0] = np.clip(np.trunc((means2D[:, 0] - r) / 16), 0, gx)
    rect[:, 1] = np.clip(np.trunc((means2D[:, 1] - r) / 16), 0, gy)
    rect[:, 2] = np.clip(np.trunc((means2D[:, 0] + r + 15) / 16), 0, gx)
    rect[:, 3] = np.clip(np.trunc((means2D[:, 1] + r + 15) / 16), 0, gy)
    key = depths.astype(np.float32).view(np.uint32).astype(np.uint64)
    key[radii <= 0] = 0xFFFFFFFF
    order = np.argsort(key, kind="stable")
    return dict(radii=radii, rect=rect, order=order, Jinv=o.get("Jinv").reshape(P, 10).astype(np.float64),
                **{"lambda": o.get("lambda").reshape(P, 2).astype(np.float64)})
