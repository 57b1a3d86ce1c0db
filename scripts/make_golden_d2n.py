"""Generates tests/golden/depth2normal_grad.npz: the REFERENCE's differentiable `depth2normal` (utils/image_utils.py:61-125) run in
the authoring container with autograd -- depth, mask, camera numbers, an upstream gradient, the normal map and d(loss)/d(depth)
(the depth-normal consistency term of the stage-1 loss differentiates through it, gaussian_renderer/render.py:158-160).

    python scripts/make_golden_d2n.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_golden as mg             # noqa: E402
import make_golden_view as mgv       # noqa: E402


def main():
    mgv.setup_reference()
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: mg._Stub("_C")
    from utils.image_utils import depth2normal
    g = torch.Generator().manual_seed(99)
    H, W = 37, 53
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = (3.0 + 0.01 * xx - 0.02 * yy + 0.3 * torch.sin(xx * 0.3) * torch.cos(yy * 0.2) + 0.05 * torch.rand(H, W, generator=g))[None]
    mask = (torch.rand(1, H, W, generator=g) > 0.1)
    cam = types.SimpleNamespace(prcppoint=torch.tensor([0.47, 0.55]), image_width=W, image_height=H, FoVx=0.69, FoVy=0.52)
    depth = depth.clone().requires_grad_(True)
    n = depth2normal(depth, mask, cam)
    up = torch.randn(3, H, W, generator=g)
    (n * up).sum().backward()
    np.savez(os.path.join(ROOT, "tests", "golden", "depth2normal_grad.npz"), depth=depth.detach().numpy(), mask=mask.numpy(),
             prcppoint=cam.prcppoint.numpy(), fovx=np.float32(cam.FoVx), fovy=np.float32(cam.FoVy), upstream=up.numpy(),
             normal=n.detach().numpy(), depth_grad=depth.grad.numpy())
    print("wrote depth2normal_grad.npz", n.shape, float(depth.grad.abs().max()))


if __name__ == "__main__":
    main()
