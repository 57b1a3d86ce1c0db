"""`get_gs_bvh` of the reference (pbgi/bvhhelpers.py:96-156): LBVH over Gaussian surfels -> (LBVHNode_info [2P-1,3] int32 =
{left, right, primitive}, LBVHNode_aabb [2P-1,6]).  The five slang modules it drives (element boxes, Morton codes,
single-workgroup radix sort, hierarchy, one box launch per tree level) are `svgir_pbgi_bvh_build`; the tensors come from
`svgir_pbgi_bvh_export`.  No CPU / PyTorch fallback."""
import ctypes as C

import torch

from gaussian_renderer import _native

_lib = _native.lib
_lib.svgir_pbgi_bvh_bytes.restype = C.c_size_t
_lib.svgir_pbgi_bvh_bytes.argtypes = [C.c_int32]
_lib.svgir_pbgi_bvh_build.restype = C.c_int
_lib.svgir_pbgi_bvh_build.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.svgir_pbgi_bvh_export.restype = C.c_int
_lib.svgir_pbgi_bvh_export.argtypes = [C.c_int32] + [C.c_void_p] * 5
_lib.svgir_pbgi_trace_radiance.restype = C.c_int
_lib.svgir_pbgi_trace_radiance.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 14


class GsBvh:
    """The opaque device blob of one build + the element count; `tensors()` gives the reference's two tensors."""

    def __init__(self, centers, scales):
        if not centers.is_cuda:
            raise RuntimeError("the pbgi BVH needs CUDA/HIP tensors (there is no CPU path)")
        dev = centers.device
        self.device, self.P = dev, int(centers.shape[0])
        if self.P < 1:
            raise ValueError("get_gs_bvh needs at least one primitive")
        with torch.cuda.device(dev):
            c, s = _native.f32c(centers.detach(), dev), _native.f32c(scales.detach(), dev)
            self.blob = torch.empty(int(_lib.svgir_pbgi_bvh_bytes(self.P)), dtype=torch.uint8, device=dev)
            _native.check(_lib.svgir_pbgi_bvh_build(self.P, _native.ptr(c), _native.ptr(s), self.blob.data_ptr(), _native.stream_ptr(dev)),
                          "pbgi_bvh_build")

    def tensors(self, with_sorted=False):
        dev, n = self.device, 2 * self.P - 1
        with torch.cuda.device(dev):
            info = _native.out_tensor((n, 3), torch.int32, dev)
            aabb = _native.out_tensor((n, 6), torch.float32, dev)
            srt = _native.out_tensor((self.P, 2), torch.int32, dev) if with_sorted else None
            _native.check(_lib.svgir_pbgi_bvh_export(self.P, self.blob.data_ptr(), info.data_ptr(), aabb.data_ptr(),
                                                    srt.data_ptr() if with_sorted else None, _native.stream_ptr(dev)), "pbgi_bvh_export")
        return (info, aabb, srt) if with_sorted else (info, aabb)


def get_gs_bvh(centers, scales, rotates=None, *slang_modules):
    """Same leading arguments as the reference (its trailing slang-module arguments are accepted and ignored; `rotates` is
    unused there too).  Returns (LBVHNode_info, LBVHNode_aabb)."""
    return GsBvh(centers, scales).tensors()
