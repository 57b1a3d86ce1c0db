"""ctypes front-end of the CPU parity oracle (oracle/svgir_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (svg-ir_amd/) must never import this module.

Parity status: see the header of svgir_oracle.cpp (the reference has no tests / golden vectors for the rasterizer and
is CUDA-only; the restatement is pinned by fixtures generated from the reference's importable Python helpers, by an
oracle-free fp64 construction of every per-Gaussian decision and of the instance list, by torch.autograd of an
independent forward, and by finite differences -- tests/test_oracle.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvgir_oracle.so")
_FAST_PATH = os.path.join(_HERE, "libsvgir_oracle_fast.so")
_lib = None
_lib_fast = None

RGSS, SVGSS = 0, 1


class _Params(C.Structure):
    _fields_ = [
        ("variant", C.c_int), ("fp64", C.c_int),
        ("P", C.c_int), ("S", C.c_int), ("VS", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("bg", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("features", C.c_void_p), ("vfeatures", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p), ("prcppoint", C.c_void_p), ("patchbbox", C.c_void_p), ("campos", C.c_void_p),
        ("config", C.c_void_p), ("config_len", C.c_int),
        ("scale_modifier", C.c_double), ("tan_fovx", C.c_double), ("tan_fovy", C.c_double), ("cx", C.c_double),
        ("cy", C.c_double),
        ("prefiltered", C.c_int), ("computer_pseudo_normal", C.c_int), ("backward_geometry", C.c_int),
        ("num_threads", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with g++ (seconds)."""
    src = os.path.join(_HERE, "svgir_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsvgir_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def build_fast():
    """`-O3 -march=native -fopenmp` build for the CPU-baseline timing: compiled on the machine that runs
    it, never used for parity.  Returns False when it cannot be built here."""
    src = os.path.join(_HERE, "svgir_oracle.cpp")
    try:
        if not os.path.exists(_FAST_PATH) or os.path.getmtime(_FAST_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libsvgir_oracle_fast.so"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
        lib(fast=True)
        return True
    except Exception:
        return False


def lib(fast=False):
    global _lib, _lib_fast
    if fast:
        if _lib_fast is None:
            _lib_fast = _bind(C.CDLL(_FAST_PATH))
        return _lib_fast
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(_lib):
    if True:
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_create.argtypes = [C.POINTER(_Params)]
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_forward.restype = C.c_int
        _lib.orc_forward.argtypes = [C.c_void_p]
        _lib.orc_backward.argtypes = [C.c_void_p] * 7
        _lib.orc_mark_visible.argtypes = [C.c_void_p]
        _lib.orc_get.restype = C.c_int
        _lib.orc_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_int)]
        _lib.orc_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        _lib.orc_max_threads.restype = C.c_int
    return _lib


_DT = {0: np.float32, 1: np.float64, 2: np.int32, 3: np.uint32, 4: np.uint64, 5: np.uint8}


class OracleRun:
    """One forward (+ optional backward) evaluation of the restated reference rasterizer on host arrays.

    `scene` is a dict with the reference's tensor names (see svgir_harness.scenes): means3D[P,3], scales[P,3],
    rotations[P,4], opacities[P,1], shs[P,M,3] | colors_precomp[P,3], features[P,S], vfeatures[P,VS],
    viewmatrix[4,4], projmatrix[4,4], campos[3], bg[3], plus scalars W, H, tanfovx, tanfovy, sh_degree,
    scale_modifier and, per variant, patch_bbox/prcppoint/config (svgss) or cx/cy/backward_geometry/
    computer_pseudo_normal (rgss).
    """

    def __init__(self, scene, variant, fp64=False, num_threads=0, fast=False):
        self.L = lib(fast=fast)
        self.dt = np.float64 if fp64 else np.float32
        self.variant = variant
        self._keep = {}
        P = int(scene["means3D"].shape[0])
        p = _Params()
        p.variant, p.fp64, p.P = variant, int(fp64), P
        p.W, p.H = int(scene["W"]), int(scene["H"])
        p.D = int(scene.get("sh_degree", 0))
        feats = scene.get("features")
        vfeats = scene.get("vfeatures")
        p.S = 0 if feats is None else int(feats.shape[1])
        p.VS = 0 if (vfeats is None or variant == RGSS) else int(vfeats.shape[1])
        shs = scene.get("shs")
        p.M = 0 if shs is None else int(shs.shape[1])

        def arr(name, a):
            if a is None:
                return None
            a = np.ascontiguousarray(np.asarray(a, dtype=self.dt))
            self._keep[name] = a
            return a.ctypes.data if a.size else None

        p.bg = arr("bg", scene["bg"])
        p.means3D = arr("means3D", scene["means3D"])
        p.shs = arr("shs", shs)
        p.colors_precomp = arr("colors_precomp", scene.get("colors_precomp"))
        p.features = arr("features", feats)
        p.vfeatures = arr("vfeatures", vfeats if variant == SVGSS else None)
        p.opacities = arr("opacities", scene["opacities"])
        p.scales = arr("scales", scene.get("scales"))
        p.rotations = arr("rotations", scene.get("rotations"))
        p.cov3D_precomp = arr("cov3D_precomp", scene.get("cov3D_precomp"))
        p.viewmatrix = arr("viewmatrix", scene["viewmatrix"])
        p.projmatrix = arr("projmatrix", scene["projmatrix"])
        p.campos = arr("campos", scene["campos"])
        if variant == SVGSS:
            p.prcppoint = arr("prcppoint", scene.get("prcppoint", np.array([0.5, 0.5])))
            p.patchbbox = arr("patchbbox", scene.get("patch_bbox", np.array([0, 0, p.H, p.W])))
            cfg = np.asarray(scene.get("config", [1.0, 1.0, 1.0]), dtype=self.dt)
            p.config = arr("config", cfg)
            p.config_len = int(cfg.size)
        p.scale_modifier = float(scene.get("scale_modifier", 1.0))
        p.tan_fovx, p.tan_fovy = float(scene["tanfovx"]), float(scene["tanfovy"])
        p.cx = float(scene.get("cx", p.W / 2.0))
        p.cy = float(scene.get("cy", p.H / 2.0))
        p.prefiltered = 0
        p.computer_pseudo_normal = int(bool(scene.get("computer_pseudo_normal", False)))
        p.backward_geometry = int(bool(scene.get("backward_geometry", True)))
        p.num_threads = int(num_threads)
        self.p = p
        self.h = C.c_void_p(self.L.orc_create(C.byref(p)))
        self.num_rendered = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def forward(self):
        self.num_rendered = self.L.orc_forward(self.h)
        return self.num_rendered

    def backward(self, g_color, g_normal, g_depth, g_opac, g_feature=None, g_vfeature=None):
        N = self.p.W * self.p.H

        def g(a, ch):
            if a is None:
                a = np.zeros((ch, N), dtype=self.dt)
            a = np.ascontiguousarray(np.asarray(a, dtype=self.dt))
            assert a.size == ch * N, (a.shape, ch, N)
            return a

        gs = [g(g_color, 3), g(g_normal, 3), g(g_depth, 1), g(g_opac, 1), g(g_feature, self.p.S),
              g(g_vfeature, self.p.VS // 4)]
        self._keep["grads"] = gs
        self.L.orc_backward(self.h, *[a.ctypes.data if a.size else None for a in gs])

    def mark_visible(self):
        self.L.orc_mark_visible(self.h)
        return self.get("present") > 0

    def get(self, name):
        ptr, n, dt = C.c_void_p(), C.c_longlong(), C.c_int()
        rc = self.L.orc_get(self.h, name.encode(), C.byref(ptr), C.byref(n), C.byref(dt))
        if rc != 0:
            raise KeyError(name)
        dtype = np.dtype(_DT[dt.value])
        if n.value == 0:
            return np.zeros((0,), dtype=dtype)
        buf = (C.c_char * (n.value * dtype.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).copy()

    def timings(self):
        t = (C.c_double * 5)()
        self.L.orc_timings(self.h, t)
        return dict(zip(["preprocess", "binning", "render", "render_bwd", "preprocess_bwd"], list(t)))

    # convenience: images reshaped CHW like the reference's return values
    def images(self):
        H, W, S, VC = self.p.H, self.p.W, self.p.S, self.p.VS // 4
        out = {
            "color": self.get("out_color").reshape(3, H, W),
            "normal": self.get("out_normal").reshape(3, H, W),
            "depth": self.get("out_depth").reshape(1, H, W),
            "opacity": self.get("out_opac").reshape(1, H, W),
            "feature": self.get("out_feature").reshape(S, H, W),
            "weights": self.get("out_weights").reshape(-1, 1),
            "radii": self.get("radii"),
            "n_contrib": self.get("n_contrib").reshape(H, W),
        }
        if self.variant == SVGSS:
            out["vfeature"] = self.get("out_vfeature").reshape(VC, H, W)
        else:
            out["pseudo_normal"] = self.get("out_pseudo_normal").reshape(3, H, W)
            out["surface_xyz"] = self.get("out_surface_xyz").reshape(3, H, W)
        return out

    def grads(self):
        P, S, VS, M = self.p.P, self.p.S, self.p.VS, self.p.M
        out = {
            "means2D": self.get("dL_dmean2D").reshape(P, 3),
            "colors": self.get("dL_dcolor").reshape(P, 3),
            "opacity": self.get("dL_dopacity").reshape(P, 1),
            "means3D": self.get("dL_dmean3D").reshape(P, 3),
            "features": self.get("dL_dfeature").reshape(P, S),
            "cov3D": self.get("dL_dcov3D").reshape(P, 6),
            "sh": self.get("dL_dsh").reshape(P, M, 3),
            "scales": self.get("dL_dscale").reshape(P, 3),
            "rotations": self.get("dL_drot").reshape(P, 4),
            "conic": self.get("dL_dconic").reshape(P, 4),
            "normal": self.get("dL_dnormal").reshape(P, 3),
            "depth": self.get("dL_ddepth").reshape(P, 1),
        }
        if self.variant == SVGSS:
            out["vfeatures"] = self.get("dL_dvfeature").reshape(P, VS)
            out["viewmat"] = self.get("dL_dviewmat").reshape(4, 4)
            out["projmat"] = self.get("dL_dprojmat").reshape(4, 4)
            out["campos"] = self.get("dL_dcampos").reshape(3)
        return out


def max_threads():
    return lib().orc_max_threads()
