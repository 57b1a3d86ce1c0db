"""Generates tests/golden/pbgi_glue.npz by running the REFERENCE's pbgi Python (pbgi/bvhhelpers.py `get_gs_bvh`,
pbgi/renderer.py `Renderer.build_bvh` / `render_radiance_with_sampling_SH`) in the authoring container with a RECORDING
stand-in for `slangtorch` (the slang compiler is not available, so the kernels themselves cannot run): which kernels are
launched in which order, with which tensors (shape, dtype, initial fill) bound to which parameter names, and with which
launch geometry.  The fixture is data only; no reference source is copied.

    python scripts/make_golden_pbgi.py
"""
import importlib
import json
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
    import make_golden as mg
    calls = []

    def describe(v):
        if isinstance(v, torch.Tensor):
            u = torch.unique(v)
            return {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "fill": float(u[0]) if u.numel() == 1 else None,
                    "id": id(v)}
        if isinstance(v, (int, float)):
            return {"scalar": v}
        return {"other": type(v).__name__}

    class _Kernel:
        def __init__(self, module, name):
            self.module, self.name = module, name

        def __call__(self, **kw):
            rec = {"module": self.module, "kernel": self.name, "args": {k: describe(v) for k, v in kw.items()}, "order": list(kw.keys())}
            calls.append(rec)

            class _Launch:
                @staticmethod
                def launchRaw(blockSize, gridSize):
                    rec["block"], rec["grid"] = [int(x) for x in blockSize], [int(x) for x in gridSize]
            return _Launch()

    class _Module:
        def __init__(self, path):
            self._name = os.path.basename(path)

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            if name == "pushConstantsMortonCodes":   # a struct constructor, not a kernel
                return lambda **kw: {"struct": name, **{k: (float(v) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}}
            return _Kernel(self._name, name)

    st = types.ModuleType("slangtorch")
    st.loadModule = lambda path, *a, **k: _Module(path)
    sys.modules["slangtorch"] = st
    mg.STUBS.discard("slangtorch"); mg.STUBS.discard("pbgi")
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, REF)
    real_cuda, real_sync = torch.Tensor.cuda, torch.cuda.synchronize
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        renderer = importlib.import_module("pbgi.renderer")
        R = renderer.Renderer()
        rng = np.random.default_rng(3)
        P, N, S = 40, 24, 8
        t = lambda a: torch.from_numpy(a.astype(np.float32))
        R.proxy_xyzs, R.proxy_scales, R.proxy_rotates = t(rng.normal(size=(P, 3))), t(rng.uniform(0.01, 0.1, size=(P, 3))), t(rng.normal(size=(P, 4)))
        R.proxy_normals, R.proxy_opacity, R.proxy_features = t(rng.normal(size=(P, 3))), t(rng.uniform(size=(P, 1))), t(rng.normal(size=(P, 16, 3)))
        R.proxy_idx = torch.arange(P)
        named = {id(R.proxy_xyzs): "xyz", id(R.proxy_scales): "scaling", id(R.proxy_rotates): "rotation", id(R.proxy_normals): "geo_normal",
                 id(R.proxy_opacity): "opacity", id(R.proxy_features): "features"}
        R.build_bvh()
        n_build = len(calls)
        ray_o, ray_d, cov = t(rng.normal(size=(N, 3))), t(rng.normal(size=(N, S, 3))), t(rng.normal(size=(P, 6)))
        named.update({id(ray_o): "ray_o", id(ray_d): "ray_d", id(cov): "cov3D_inv", id(R.LBVHNode_info): "LBVHNode_info", id(R.LBVHNode_aabb): "LBVHNode_aabb"})
        out = R.render_radiance_with_sampling_SH(ray_o, ray_d, cov, S)
        for i, o in enumerate(out):
            named[id(o)] = ("radiance", "visibility", "hit_indices", "uvs")[i]
    finally:
        torch.Tensor.cuda, torch.cuda.synchronize = real_cuda, real_sync
    for c in calls:
        for a in c["args"].values():
            if "id" in a:
                a["source"] = named.get(a.pop("id"))
    doc = {"P": P, "N": N, "S": S, "build_calls": calls[:n_build], "trace_calls": calls[n_build:],
           "info_shape": list(R.LBVHNode_info.shape), "info_dtype": str(R.LBVHNode_info.dtype), "aabb_shape": list(R.LBVHNode_aabb.shape),
           "outputs": [{"shape": list(o.shape), "dtype": str(o.dtype).replace("torch.", "")} for o in out]}
    np.savez(os.path.join(OUT, "pbgi_glue.npz"), doc=np.array(json.dumps(doc)))
    print("wrote pbgi_glue.npz:", [c["kernel"] for c in calls])
    print(json.dumps(doc["trace_calls"], indent=1)[:3000])


if __name__ == "__main__":
    main()
