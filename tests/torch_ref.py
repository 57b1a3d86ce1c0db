"""Independent, differentiable fp64 PyTorch restatement of the rasterizer FORWARD (dense: every pixel x every
Gaussian), written in ordinary matrix notation rather than following the reference's code layout.

Used to pin the oracle (oracle/svgir_oracle.cpp):
  * forward outputs of the oracle (fp64 mode) must agree with this to ~1e-10;
  * the oracle's hand-derived backward must equal torch.autograd of this forward wherever the reference's
    backward is a true derivative.  Where the reference deliberately is NOT a derivative, this module mirrors the
    same stop-gradients (bilinear corner weights and the depth-differencing offset are constants; the local
    homography is a constant) and the tests account for the explicit extra terms (Q4 x10 on normals, Q5).

Integer decisions (visibility, tile rectangles, depth order) and the local homography are constants of the
differentiation.  They come either from the oracle run (`consts_from_oracle`) or -- oracle-free -- from
`consts_independent`, an fp64 construction from the geometry itself (ray / tangent-plane intersection for the
homography, eigenvalues of the 2D covariance for the radius, floor arithmetic for the tile rectangle, the cull
conditions of auxiliary.h / forward.cu stated as geometric predicates), which is what pins the oracle's `local_homo`,
`getRect`, radius formula and cull thresholds by test instead of by inspection.
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


def consts_independent(sc, variant):
    """The same constants as consts_from_oracle, derived WITHOUT the oracle in fp64 numpy.

    Geometry (camera space: x right, y down, z forward; V = W2C^T as stored by the reference):
      * a surfel is a planar Gaussian in the plane through its centre c (camera space) spanned by its local axes
        a0, a1 with normal n = a0 x a1 (third column of R(q), rotated into camera space);
      * culls: behind the near plane (svgss: c.z < 0 or projected centre outside the patch box grown by 20 %; rgss:
        c.z <= 0.2), back-facing (c . n > -0.01), grazing (the two probe rays below make |cos| < 0.01 with n),
        singular 2D covariance, empty tile rectangle;
      * local homography: probe rays through (qx + 1/1000, qy) and (qx, qy + 1/1000) on the z = 1 plane, q = c.xy / c.z,
        are intersected with the tangent plane; the offsets of the two hit points from c, expressed in (a0, a1) and
        divided by (mean focal length / 1000), are the columns of the 2x2 screen -> tangent map J; the depth offset of a
        tangent displacement (du, dv) is du a0.z + dv a1.z;
      * radius = ceil(3 sqrt(larger eigenvalue of the low-pass-filtered 2D covariance)), eigenvalues from the trace /
        determinant with the reference's floor of 0.1 under the root;
      * tile rectangle = tiles whose 16 px cells meet [centre - radius, centre + radius + 15] (truncating division).
    """
    W, H = sc["W"], sc["H"]
    svgss = variant == "svgss"
    cfg = [float(c) for c in sc["config"]] if svgss else [1.0, 1.0, 1.0]
    surface, pix_depth = cfg[0] > 0, cfg[2] > 0
    f64 = lambda k: np.asarray(sc[k], dtype=np.float64)  # noqa: E731
    V, PM = f64("viewmatrix"), f64("projmatrix")
    m, q, sc3 = f64("means3D"), f64("rotations"), f64("scales")
    P = m.shape[0]
    fx, fy = W / (2.0 * sc["tanfovx"]), H / (2.0 * sc["tanfovy"])
    mod = float(sc.get("scale_modifier", 1.0))
    hom = np.concatenate([m, np.ones((P, 1))], -1)
    c = (hom @ V)[:, :3]                                      # centre in camera space
    clip = hom @ PM
    ndc = clip[:, :2] / (clip[:, 3:4] + 0.0000001)
    pix = np.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], -1)
    r_, x, y, z = q.T
    Rq = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r_ * z), 2 * (x * z + r_ * y),
                   2 * (x * y + r_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r_ * x),
                   2 * (x * z - r_ * y), 2 * (y * z + r_ * x), 1 - 2 * (x * x + y * y)], -1).reshape(P, 3, 3)
    Wrot = V[:3, :3].T
    axes = np.einsum("ij,pjk->pik", Wrot, Rq)                 # columns = a0, a1, n in camera space
    a0, a1, n = axes[:, :, 0], axes[:, :, 1], axes[:, :, 2]
    vis = np.ones(P, dtype=bool)
    if svgss:
        y0, x0, y1, x1 = [float(v) for v in np.asarray(sc["patch_bbox"]).reshape(-1)[:4]]
        gw, gh = (x1 - x0) * 0.2, (y1 - y0) * 0.2
        vis &= ~((c[:, 2] < 0) | (pix[:, 0] < x0 - gw) | (pix[:, 0] >= x1 + gw) | (pix[:, 1] < y0 - gh) | (pix[:, 1] >= y1 + gh))
    else:
        vis &= c[:, 2] > 0.2
    Jinv = np.zeros((P, 10))
    if surface:
        vis &= ~(np.einsum("pi,pi->p", c, n) > -0.01)
        if pix_depth:
            qxy = c[:, :2] / c[:, 2:3]
            k = ((fx + fy) / 2) / 1000.0
            hits = []
            for e in (np.array([1 / 1000.0, 0.0]), np.array([0.0, 1 / 1000.0])):
                ray = np.concatenate([qxy + e[None], np.ones((P, 1))], -1)
                length = np.linalg.norm(ray, axis=-1)
                ray = ray / length[:, None]
                cosang = np.einsum("pi,pi->p", ray, n)
                vis &= ~(np.abs(cosang / length) < 0.01)
                t = np.einsum("pi,pi->p", c, n) / cosang        # ray parameter of the plane hit
                hits.append(ray * t[:, None] - c)
            Jinv[:, 0] = np.einsum("pi,pi->p", hits[0], a0) / k
            Jinv[:, 1] = np.einsum("pi,pi->p", hits[1], a0) / k
            Jinv[:, 2] = np.einsum("pi,pi->p", hits[0], a1) / k
            Jinv[:, 3] = np.einsum("pi,pi->p", hits[1], a1) / k
            Jinv[:, 4:7], Jinv[:, 7:10] = a0, a1
    # 2D covariance of the (flattened) Gaussian under the local affine approximation of the projection
    s = sc3 * mod
    sz = np.zeros(P) if (mod * (1.0 if surface else 0.0)) != 0 else sc3[:, 2]
    Sd = np.stack([s[:, 0], s[:, 1], sz], -1)
    Sigma = np.einsum("pij,pj,pkj->pik", Rq, Sd * Sd, Rq)
    limx, limy = 1.3 * sc["tanfovx"], 1.3 * sc["tanfovy"]
    tz = c[:, 2]
    txc = np.clip(c[:, 0] / tz, -limx, limx) * tz
    tyc = np.clip(c[:, 1] / tz, -limy, limy) * tz
    Jp = np.zeros((P, 2, 3))
    Jp[:, 0, 0], Jp[:, 0, 2] = fx / tz, -fx * txc / (tz * tz)
    Jp[:, 1, 1], Jp[:, 1, 2] = fy / tz, -fy * tyc / (tz * tz)
    Tm = Jp @ Wrot
    cov = Tm @ Sigma @ Tm.transpose(0, 2, 1)
    ca, cb, cc = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = ca * cc - cb * cb
    vis &= det != 0
    mid = 0.5 * (ca + cc)
    lam_max = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    radius = np.ceil(3.0 * np.sqrt(np.maximum(lam_max, mid - np.sqrt(np.maximum(0.1, mid * mid - det)))))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rect = np.zeros((P, 4), dtype=np.int64)
    with np.errstate(invalid="ignore"):
        rect[:, 0] = np.clip(np.trunc((pix[:, 0] - radius) / 16), 0, gx)
        rect[:, 1] = np.clip(np.trunc((pix[:, 1] - radius) / 16), 0, gy)
        rect[:, 2] = np.clip(np.trunc((pix[:, 0] + radius + 15) / 16), 0, gx)
        rect[:, 3] = np.clip(np.trunc((pix[:, 1] + radius + 15) / 16), 0, gy)
    vis &= (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1]) > 0
    radii = np.where(vis, radius, 0).astype(np.int32)
    rect[~vis] = 0
    key = c[:, 2].astype(np.float32).view(np.uint32).astype(np.uint64)
    key[~vis] = 0xFFFFFFFF
    order = np.argsort(key, kind="stable")
    return dict(radii=radii, rect=rect, order=order, Jinv=np.where(vis[:, None], Jinv, 0.0), pix=pix, depth=c[:, 2],
                margin=np.abs(3.0 * np.sqrt(lam_max) - np.round(3.0 * np.sqrt(lam_max))),
                **{"lambda": sc3[:, :2].copy()})
