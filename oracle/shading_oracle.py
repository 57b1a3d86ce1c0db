"""CPU oracle of the per-splat spatially-varying BRDF shading.

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); the product never imports it.

Restates, in plain per-element math, what the reference computes with broadcasting PyTorch ops in
  gaussian_renderer/svgss.py:537-593 (rendering_equation4), :595-631 (GGX_specular4),
  scene/direct_light_map.py:70-83,103-106 (DirectLightMap.direct_light on softplus(env), x2),
  gaussian_renderer/svgss.py:143-166 (feature / vfeature packing).
Parity: PINNED -- tests/golden/shading.npz holds inputs, outputs and autograd gradients produced by importing the
reference's own rendering_equation4 in the authoring container (scripts/make_golden.py); tests/test_shading_oracle.py
checks this restatement (forward, and torch.autograd of it for the backward) against them.

Conventions: base_color [n,12] and every [n,12] output are channel-major / corner-minor (index c*4+k); normals
[n,4,3]; incident arrays [n,Ns,*]; env [1,He,We,3] raw (softplus applied inside).
"""
import math

import torch
import torch.nn.functional as F


def env_lookup(env, dirs, softplus=True, scale=2.0):
    """scale*bilinear(f(env)) at lat-long coordinates of `dirs` [..,3] (grid_sample, align_corners=True, zero pad);
    f = softplus, scale 2 for DirectLightMap (scene/direct_light_map.py:70-83); identity, scale 1 for EnvLight
    (scene/envmap.py:53-72)."""
    sp = (F.softplus(env) if softplus else env).reshape((-1,) + tuple(env.shape[-3:]))[0]  # [He,We,3]
    He, We = sp.shape[0], sp.shape[1]
    d = dirs.reshape(-1, 3)
    phi = torch.arccos(d[:, 2]) - 1e-6
    theta = torch.atan2(d[:, 1], d[:, 0])
    gy = phi / math.pi * 2 - 1
    gx = -theta / math.pi
    x = (gx + 1) * 0.5 * (We - 1)
    y = (gy + 1) * 0.5 * (He - 1)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    fx, fy = x - x0, y - y0
    out = torch.zeros((d.shape[0], 3), dtype=sp.dtype)
    for dy_, wy in ((0, 1 - fy), (1, fy)):
        for dx_, wx in ((0, 1 - fx), (1, fx)):
            xi = (x0 + dx_).long()
            yi = (y0 + dy_).long()
            ok = (xi >= 0) & (xi < We) & (yi >= 0) & (yi < He)
            tex = sp[yi.clamp(0, He - 1), xi.clamp(0, We - 1)]
            out = out + torch.where(ok[:, None], tex * (wx * wy)[:, None], torch.zeros_like(tex))
    return (out * scale).reshape(dirs.shape)


def _normalize(v):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def ggx(normals, viewdirs, dirs, rough, fresnel=0.04):
    """[n,Ns,4] specular term per (sample, corner)."""
    L = _normalize(dirs)[:, :, None, :]               # [n,Ns,1,3]
    V = _normalize(viewdirs)[:, None, None, :]         # [n,1,1,3]
    H = _normalize((L + V) / 2.0)                      # [n,Ns,1,3]
    N = _normalize(normals)                            # [n,4,3]
    N = N * (V[:, 0] * N).sum(-1, keepdim=True).sign()
    Nb = N[:, None]                                    # [n,1,4,3]
    NoL = (Nb * L).sum(-1).clamp(1e-6, 1)              # [n,Ns,4]
    NoV = (Nb * V).sum(-1).clamp(1e-6, 1)              # [n,1,4]
    NoH = (Nb * H).sum(-1).clamp(1e-6, 1)              # [n,Ns,4]
    VoH = (V * H).sum(-1).clamp(1e-6, 1)               # [n,Ns,1]
    r = rough[:, None, :]                              # [n,1,4]
    a = r * r
    a2 = a * a
    k = (a + 2 * r + 1.0) / 8.0
    frac = (fresnel + (1 - fresnel) * torch.pow(torch.full_like(VoH, 2.0), (-5.55473 * VoH - 6.98316) * VoH)) * a2
    nom0 = NoH * NoH * (a2 - 1) + 1
    nom = (4 * math.pi * nom0 * nom0 * (NoV * (1 - k) + k) * (NoL * (1 - k) + k)).clamp(1e-6, 4 * math.pi)
    return frac / nom


def shade(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, softplus=True, scale=2.0,
          transform=None):
    """Returns dict of reduced outputs: pbr, diffuse_light, specular, direct, indirect [n,12];
    mean_incident, mean_local, mean_global [n,3]; mean_vis [n,1].  `transform` [3,3]: the env lookup uses
    dirs @ transform.T (EnvLight.transform, scene/envmap.py:57-60), everything else the untransformed dirs."""
    n, Ns = dirs.shape[0], dirs.shape[1]
    ldirs = dirs if transform is None else dirs @ transform.T
    glob = env_lookup(env, ldirs, softplus, scale).clamp(0, 64) * visibility          # [n,Ns,3]
    loc = radiance
    inc = loc + glob
    ndi = (normals[:, None] * dirs[:, :, None]).sum(-1).clamp(min=0)  # [n,Ns,4]
    fs = ggx(normals, viewdirs, dirs, roughness)                      # [n,Ns,4]
    fd = base_color.reshape(n, 3, 4) / math.pi                        # [n,3,4] (c,k)
    geo = (areas * ndi)[:, :, None, :]                                # [n,Ns,1,4]

    def transport(Lc):   # [n,Ns,3] -> [n,Ns,3,4]
        return Lc[:, :, :, None] * geo

    t, td, ti = transport(inc), transport(glob), transport(loc)
    f = fd[:, None] + fs[:, :, None, :]
    out = {
        "pbr": (f * t).mean(1).reshape(n, 12),
        "diffuse_light": t.mean(1).reshape(n, 12),
        "specular": (fs[:, :, None, :] * t).mean(1).reshape(n, 12),
        "direct": (f * td).mean(1).reshape(n, 12),
        "indirect": (f * ti).mean(1).reshape(n, 12),
        "mean_incident": inc.mean(1), "mean_local": loc.mean(1), "mean_global": glob.mean(1),
        "mean_vis": visibility.mean(1),
    }
    return out


def pack(out, base_color, roughness, normals, view3x3, training):
    """svgss.py:143-166: (features [n,S], vfeatures [n,VS]); S,VS = 4,52 (training) or 7,64 (eval)."""
    n = base_color.shape[0]
    nv = (normals @ view3x3).transpose(1, 2).reshape(n, -1)  # view-space shading normals, c*4+k
    if training:
        feats = torch.cat([out["mean_vis"], out["mean_local"]], dim=-1)
        vfeats = torch.cat([out["pbr"], base_color, nv, roughness, out["diffuse_light"]], dim=-1)
    else:
        feats = torch.cat([out["mean_incident"], out["mean_local"], out["mean_vis"]], dim=-1)
        vfeats = torch.cat([out["pbr"], base_color, nv, roughness, out["direct"], out["indirect"]], dim=-1)
    return feats, vfeats
