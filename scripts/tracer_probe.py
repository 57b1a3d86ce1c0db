"""GPU probe: build and trace times of the two LBVH tracers at P = 200k surfels x 64 rays (tests/pbgi_scene.py scene)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pbgi_scene
from pbgi.renderer import Renderer
from submodules.bvh import RayTracer

dev = "cuda:0"
sc = pbgi_scene.make(P=200000, shells=2000, S=64, seed=13)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xyz, scales, rot, nrm, op, cov, shs, rd = (t(sc[k]) for k in ("xyz", "scales", "rot", "normals", "opacity", "cov_inv", "shs", "ray_d"))
def timed(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
ms, rt = timed(lambda: RayTracer(xyz, scales, rot)); print(f"submodules.bvh build  {ms:8.3f} ms")
und = torch.nn.functional.normalize(rd, dim=-1)
ms, out = timed(lambda: rt.trace_visibility(xyz[:, None].expand(-1, 64, -1), und, xyz, cov, op, nrm)); print(f"trace_visibility     {ms:8.3f} ms   ({200000 * 64 / ms / 1e3:.1f} M rays/s), mean visibility {out['visibility'].mean().item():.3f}")
R = Renderer(); R.set_proxy(xyz, scales, rot, nrm, op, shs)
ms, _ = timed(lambda: R.build_bvh()); print(f"pbgi build_bvh        {ms:8.3f} ms")
ms, out = timed(lambda: R.render_radiance_with_sampling_SH(xyz, rd, cov, 64), n=2); print(f"render_radiance       {ms:8.3f} ms   ({200000 * 64 / ms / 1e3:.1f} M rays/s), hit fraction {(out[2] >= 0).float().mean().item():.3f}")
