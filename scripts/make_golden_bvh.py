"""Generates tests/golden/bvh.npz by running the REFERENCE's `RayTracer` Python (submodules/bvh/__init__.py) in the authoring
container: the leaf / node box tensors it hands to `_C.create_bvh`, and the arguments (origin offset!) it hands to
`_C.trace_bvh_opacity`, with a recording stub in place of the CUDA extension `_C` and `device="cuda"` redirected to the CPU.
The fixture is data only (inputs + the tensors the reference computed); no reference source is copied.

    python scripts/make_golden_bvh.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_golden as mg                      # its stub finder for the heavy / CUDA-only dependencies
    sys.meta_path.insert(0, mg._Finder())
    mg.STUBS.discard("submodules")                # we import the real submodules/bvh/__init__.py below
    sys.path.insert(0, REF)
    rec = {}

    class _C:                                     # recording stand-in for the CUDA extension (bvh_tracing._C)
        @staticmethod
        def create_bvh(means3D, scales, rotations, nodes, aabbs):
            rec["create"] = dict(means3D=means3D.clone(), scales=scales.clone(), rotations=rotations.clone(), nodes=nodes.clone(),
                                 aabbs=aabbs.clone())
            return nodes, aabbs, torch.zeros(means3D.shape[0], dtype=torch.long)

        @staticmethod
        def trace_bvh_opacity(tree, aabb, rays_o, rays_d, means3D, symm_inv, opacity, normals):
            rec["trace"] = dict(rays_o=rays_o.clone(), rays_d=rays_d.clone(), means3D=means3D.clone(), symm_inv=symm_inv.clone(),
                                opacity=opacity.clone(), normals=normals.clone())
            shape = rays_o.shape[:-1]
            return torch.zeros(shape, dtype=torch.int32), torch.ones(shape)

    mod_bt = types.ModuleType("bvh_tracing")
    mod_bt._C = _C
    sys.modules["bvh_tracing"] = mod_bt
    # device="cuda" -> CPU for the tensor constructors the module and build_rotation call
    real = {n: getattr(torch, n) for n in ("zeros", "full", "ones")}

    def redirect(fn):
        def f(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return fn(*a, **k)
        return f
    for n, fn in real.items():
        setattr(torch, n, redirect(fn))
    try:
        spec = importlib.util.spec_from_file_location("ref_bvh", os.path.join(REF, "submodules", "bvh", "__init__.py"))
        ref_bvh = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_bvh)
        rng = np.random.default_rng(77)
        P, S = 96, 8
        means = rng.uniform(-1, 1, size=(P, 3)).astype(np.float32)
        scales = np.exp(rng.uniform(np.log(0.004), np.log(0.08), size=(P, 3))).astype(np.float32)
        rots = rng.normal(size=(P, 4)).astype(np.float32)          # un-normalised on purpose
        rt = ref_bvh.RayTracer(torch.from_numpy(means), torch.from_numpy(scales), torch.from_numpy(rots))
        rays_o = torch.from_numpy(means)[:, None].expand(P, S, 3)
        rays_d = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(P, S, 3)).astype(np.float32)), dim=-1)
        symm = torch.from_numpy(rng.normal(size=(P, 6)).astype(np.float32))
        opac = torch.from_numpy(rng.uniform(0, 1, size=(P,)).astype(np.float32))
        nrm = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(P, 3)).astype(np.float32)), dim=-1)
        out = rt.trace_visibility(rays_o, rays_d, torch.from_numpy(means), symm, opac, nrm)
    finally:
        for n, fn in real.items():
            setattr(torch, n, fn)
    c, t = rec["create"], rec["trace"]
    np.savez(os.path.join(OUT, "bvh.npz"), means3D=means, scales=scales, rotations=rots,
             aabbs=c["aabbs"].numpy(), nodes=c["nodes"].numpy(),
             rays_o_in=rays_o.contiguous().numpy(), rays_d_in=rays_d.numpy(), rays_o_passed=t["rays_o"].numpy(),
             rays_d_passed=t["rays_d"].numpy(), out_keys=np.array(sorted(out.keys())),
             out_vis_shape=np.array(out["visibility"].shape), out_contrib_shape=np.array(out["contribute"].shape))
    print("wrote bvh.npz:", c["aabbs"].shape, list(out.keys()))


if __name__ == "__main__":
    main()
