"""ctypes front-end of oracle/pbgi_oracle.cpp (CPU restatement of the reference's pbgi LBVH build and radiance tracer).

TEST INFRASTRUCTURE ONLY: importable from tests/; the product never imports it.  Parity status: UNPINNED for the kernels
(slang sources that cannot be compiled here) -- see the header of pbgi_oracle.cpp; the Python glue around them is pinned by
tests/golden/pbgi_glue.npz (scripts/make_golden_pbgi.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvgir_pbgi_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "pbgi_oracle.cpp")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libsvgir_pbgi_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_pbgi_build.argtypes = [C.c_int] + [C.c_void_p] * 5
        _lib.orc_pbgi_trace.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 13
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build(centers, scales):
    """(info [2P-1,3] int32, aabb [2P-1,6] float32, sorted [P,2] int32) -- get_gs_bvh (pbgi/bvhhelpers.py:96-156)."""
    centers, scales = _f(centers), _f(scales)
    P = centers.shape[0]
    info = np.zeros((2 * P - 1, 3), dtype=np.int32)
    aabb = np.zeros((2 * P - 1, 6), dtype=np.float32)
    srt = np.zeros((P, 2), dtype=np.int32)
    if lib().orc_pbgi_build(P, centers.ctypes.data, scales.ctypes.data, info.ctypes.data, aabb.ctypes.data, srt.ctypes.data) != 0:
        raise ValueError("orc_pbgi_build")
    return info, aabb, srt


def trace(info, aabb, ray_o, ray_d, centers, scales, rotations, normals, opacity, cov_inv, shs):
    """(radiance [N,S,3], visibility [N,S,1], hit_indices [N,S,1] int32, uvs [N,S,2]) -- render_radiance_with_sampling_SH
    (pbgi/renderer.py:596-615, intersect_test.slang:1879-1990)."""
    ray_o, ray_d = _f(ray_o), _f(ray_d)
    N, S = ray_d.shape[0], ray_d.shape[1]
    centers, scales, rotations, normals, cov_inv = _f(centers), _f(scales), _f(rotations), _f(normals), _f(cov_inv)
    opacity, shs = _f(opacity).reshape(-1), _f(shs)
    info, aabb = np.ascontiguousarray(info, dtype=np.int32), _f(aabb)
    rad = np.zeros((N, S, 3), dtype=np.float32)
    vis = np.ones((N, S, 1), dtype=np.float32)
    hit = np.zeros((N, S, 1), dtype=np.int32)
    uvs = np.zeros((N, S, 2), dtype=np.float32)
    if lib().orc_pbgi_trace(centers.shape[0], info.ctypes.data, aabb.ctypes.data, N, S, ray_o.ctypes.data, ray_d.ctypes.data,
                            centers.ctypes.data, scales.ctypes.data, rotations.ctypes.data, normals.ctypes.data, opacity.ctypes.data,
                            cov_inv.ctypes.data, shs.ctypes.data, rad.ctypes.data, vis.ctypes.data, hit.ctypes.data, uvs.ctypes.data) != 0:
        raise ValueError("orc_pbgi_trace")
    return rad, vis, hit, uvs
