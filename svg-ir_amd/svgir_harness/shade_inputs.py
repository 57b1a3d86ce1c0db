"""Synthetic inputs of the SV-BRDF shading stage in the reference's shapes (base_color sigma(N)*0.77+0.03
[P,12], roughness sigma(N)*0.9+0.09 [P,4], shading normals = geometric normal + N(0,0.1) offsets [P,4,3], Ns Fibonacci
directions around the normal (utils/graphics_utils.py:9-37 without the random rotation), areas 2*pi, visibility
U(0,1) > 0.3, radiance |N(0,0.2)|, env = U(0,3) raw texels (softplus applied by the light) at 32x64)."""
import math

import torch


class Light:
    """Minimal stand-in with the attribute the shading binding reads from scene.direct_light_map.DirectLightMap."""

    def __init__(self, env):
        self.env = env


def fibonacci_dirs(normals, Ns):
    """Hemisphere directions around `normals` [P,3] -> [P,Ns,3] (z clamped to sin(10 deg) like the reference)."""
    dev = normals.device
    idx = torch.arange(Ns, dtype=torch.float32, device=dev)
    z = (1 - 2 * idx / (2 * Ns - 1)).clamp_min(math.sin(10 / 180 * math.pi))
    rad = torch.sqrt(1 - z * z)
    theta = math.pi * (3.0 - math.sqrt(5.0)) * idx
    local = torch.stack([torch.sin(theta) * rad, torch.cos(theta) * rad, z], dim=-1)      # [Ns,3]
    n = torch.nn.functional.normalize(normals, dim=-1)
    a = torch.where(n[:, 2:3].abs() < 0.9, torch.tensor([0.0, 0.0, 1.0], device=dev), torch.tensor([1.0, 0.0, 0.0], device=dev))
    t0 = torch.nn.functional.normalize(torch.cross(a.expand_as(n), n, dim=-1), dim=-1)
    t1 = torch.cross(n, t0, dim=-1)
    return local[None, :, 0:1] * t0[:, None] + local[None, :, 1:2] * t1[:, None] + local[None, :, 2:3] * n[:, None]


def make(P, Ns, seed=2, device="cpu", geo_normals=None, env_res=32, with_dirs=True):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    if geo_normals is None:
        geo_normals = torch.nn.functional.normalize(rnd(P, 3), dim=-1)
    geo_normals = geo_normals.float().cpu()
    d = {
        "base_color": torch.sigmoid(rnd(P, 12)) * 0.77 + 0.03,
        "roughness": torch.sigmoid(rnd(P, 4)) * 0.9 + 0.09,
        "normals": torch.nn.functional.normalize(geo_normals[:, None] + 0.1 * rnd(P, 4, 3), dim=-1),
        "viewdirs": torch.nn.functional.normalize(geo_normals + 0.5 * rnd(P, 3), dim=-1),
        "visibility": (torch.rand(P, Ns, 1, generator=g) > 0.3).float(),
        "radiance": (0.2 * rnd(P, Ns, 3)).abs(),
        "env": 3.0 * torch.rand(1, env_res, 2 * env_res, 3, generator=g),
    }
    d = {k: v.to(device) for k, v in d.items()}
    if with_dirs:   # (omitted when the shading kernels generate the lattice themselves: 16 bytes per sample less)
        d["areas"] = torch.full((P, Ns, 1), 2 * math.pi, device=device)
        d["dirs"] = fibonacci_dirs(geo_normals.to(device), Ns).contiguous()
    return d
