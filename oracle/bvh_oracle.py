"""ctypes front-end of oracle/bvh_oracle.cpp (CPU restatement of the reference's surfel visibility tracer).

TEST INFRASTRUCTURE ONLY: importable from tests/ and __graft_entry__.smoke(); the product never imports it.
Parity status: see the header of bvh_oracle.cpp (leaf boxes / argument order / origin offset pinned by
tests/golden/bvh.npz, generated from the reference's own `RayTracer` Python; the per-ray arithmetic is a restatement of
CUDA code that cannot be built or run here: parity unpinned)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvgir_bvh_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "bvh_oracle.cpp")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libsvgir_bvh_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_bvh_leaf_boxes.argtypes = [C.c_int] + [C.c_void_p] * 4
        _lib.orc_bvh_trace.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 6 + [C.c_int]
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def leaf_boxes(means3D, scales, rotations):
    """[P,6] (lower xyz, upper xyz) -- RayTracer.__init__ (submodules/bvh/__init__.py:30-58)."""
    means3D, scales, rotations = _f(means3D), _f(scales), _f(rotations)
    P = means3D.shape[0]
    out = np.empty((P, 6), dtype=np.float32)
    lib().orc_bvh_leaf_boxes(P, means3D.ctypes.data, scales.ctypes.data, rotations.ctypes.data, out.ctypes.data)
    return out


def trace_visibility(boxes, rays_o, rays_d, means3D, cov_inv, opacity, normals, t_offset=0.05, fp64=False):
    """(contribute int32 [...], visibility float32 [...]) for rays of shape [..., 3] -- RayTracer.trace_visibility +
    trace_bvh_opacity (submodules/bvh/__init__.py:60-71, src/trace.cu:186-262)."""
    ro, rd = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    boxes, means3D, cov_inv, opacity, normals = _f(boxes), _f(means3D), _f(cov_inv), _f(opacity).reshape(-1), _f(normals)
    n = ro.shape[0]
    contrib = np.zeros(n, dtype=np.int32)
    vis = np.ones(n, dtype=np.float32)
    lib().orc_bvh_trace(means3D.shape[0], boxes.ctypes.data, n, ro.ctypes.data, rd.ctypes.data, float(t_offset),
                        means3D.ctypes.data, cov_inv.ctypes.data, opacity.ctypes.data, normals.ctypes.data,
                        contrib.ctypes.data, vis.ctypes.data, 1 if fp64 else 0)
    shape = np.asarray(rays_o).shape[:-1]
    return contrib.reshape(shape), vis.reshape(shape)
