"""Drop-in replacement of the reference's `submodules.bvh` (submodules/bvh/__init__.py:28-71): `RayTracer` -- a linear BVH
over the surfels and the visibility tracer that produces `_visibility_tracing` for the shading kernels
(scene/gaussian_model.py:435-464 `update_visibility`, :396-433 `finetune_visibility`).

Same class, constructor and method names and the same result dictionary; the CUDA / thrust extension `_C`
(`create_bvh`, `trace_bvh_opacity`) is replaced by the HIP kernels behind the C ABI (`svgir_bvh_build`,
`svgir_bvh_trace_visibility`, include/svgir_raster.h -> svg-ir_amd/csrc/bvh.hip).  No CPU / PyTorch fallback."""
import ctypes as C

import torch

from gaussian_renderer import _native

_lib = _native.lib
_lib.svgir_bvh_bytes.restype = C.c_size_t
_lib.svgir_bvh_bytes.argtypes = [C.c_int32]
_lib.svgir_bvh_build.restype = C.c_int
_lib.svgir_bvh_build.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.svgir_bvh_trace_visibility.restype = C.c_int
_lib.svgir_bvh_trace_visibility.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

ORIGIN_OFFSET = 0.05   # trace_visibility starts every ray at rays_o + 0.05 * rays_d (submodules/bvh/__init__.py:62)


class RayTracer:
    def __init__(self, means3D, scales, rotations):
        """means3D [P,3], scales [P,3] (activated), rotations [P,4] (r,x,y,z; normalised like build_rotation does)."""
        if not means3D.is_cuda:
            raise RuntimeError("RayTracer needs CUDA/HIP tensors (there is no CPU path)")
        dev = means3D.device
        self.device = dev
        self.P = int(means3D.shape[0])
        with torch.cuda.device(dev):
            m, s, r = (_native.f32c(t.detach(), dev) for t in (means3D, scales, rotations))
            self.bvh = torch.empty(int(_lib.svgir_bvh_bytes(self.P)), dtype=torch.uint8, device=dev)
            _native.check(_lib.svgir_bvh_build(self.P, _native.ptr(m), _native.ptr(s), _native.ptr(r), self.bvh.data_ptr(),
                                               _native.stream_ptr(dev)), "bvh_build")

    @torch.no_grad()
    def trace_visibility(self, rays_o, rays_d, means3D, symm_inv, opacity, normals):
        """rays_o / rays_d [..., 3]; means3D [P,3], symm_inv [P,6] (inverse covariance, upper triangle), opacity [P],
        normals [P,3] -> {"visibility": [..., 1] float, "contribute": [..., 1] int32}."""
        dev = self.device
        shape = tuple(rays_d.shape[:-1])
        with torch.cuda.device(dev):
            ro = _native.f32c(rays_o.expand(*shape, 3), dev).reshape(-1, 3)
            rd = _native.f32c(rays_d, dev).reshape(-1, 3)
            n = int(rd.shape[0])
            vis = _native.out_tensor((n,), torch.float32, dev)
            contrib = _native.out_tensor((n,), torch.int32, dev)
            args = [_native.f32c(t, dev) for t in (means3D, symm_inv, opacity.reshape(-1), normals)]
            _native.check(_lib.svgir_bvh_trace_visibility(self.P, self.bvh.data_ptr(), n, _native.ptr(ro), _native.ptr(rd),
                                                          ORIGIN_OFFSET, *[_native.ptr(t) for t in args], contrib.data_ptr(),
                                                          vis.data_ptr(), _native.stream_ptr(dev)), "bvh_trace_visibility")
        return {
            "visibility": vis.reshape(*shape).unsqueeze(-1),
            "contribute": contrib.reshape(*shape).unsqueeze(-1),
        }
