"""ctypes bridge to libsvgir_raster.so (the C ABI declared in include/svgir_raster.h).

PyTorch is used only for device memory and the current HIP stream; the library sees raw device pointers.
There is NO fallback: if the shared library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os
import threading
import time

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SVGIR_RASTER_LIB", os.path.join(_PKG, "libsvgir_raster.so"))

RGSS, SVGSS = 0, 1
ABI_VERSION = 14

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)


class Params(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("P", C.c_int32), ("S", C.c_int32), ("VS", C.c_int32), ("D", C.c_int32),
        ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("features", C.c_void_p), ("vfeatures", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p), ("prcppoint", C.c_void_p), ("patchbbox", C.c_void_p),
        ("config", C.c_void_p), ("config_len", C.c_int32),
        ("scale_modifier", C.c_float), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("cx", C.c_float),
        ("cy", C.c_float),
        ("prefiltered", C.c_int32), ("computer_pseudo_normal", C.c_int32), ("backward_geometry", C.c_int32),
        ("debug", C.c_int32), ("features_ready", C.c_void_p), ("forward_only", C.c_int32), ("shade", C.c_void_p),
        ("workload_scope", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "out_color", "out_normal", "out_depth", "out_opacity", "out_feature", "out_vfeature", "out_pseudo_normal",
        "out_surface_xyz", "out_weights", "radii")]


class Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dout_color", "dL_dout_normal", "dL_dout_depth", "dL_dout_opacity", "dL_dout_feature", "dL_dout_vfeature",
        "dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dfeatures", "dL_dvfeatures", "dL_dnormal",
        "dL_ddepth", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dviewmat",
        "dL_dprojmat", "dL_dcampos", "clear_base")] + [("clear_bytes", C.c_size_t)] + [(n, C.c_void_p) for n in (
        "dL_dbase_color", "dL_droughness", "dL_dshade_normals", "dL_dradiance", "dL_denv", "env_grad_work", "dL_dreduced",
        "out_weights", "dL_dradiance_ratio")]


class ViewCall(C.Structure):
    """svgir_view_call: one view of svgir_forward_batch."""
    _fields_ = [("params", C.POINTER(Params)), ("outputs", C.POINTER(Outputs)), ("geom", ALLOC_FN), ("geom_ctx", C.c_void_p),
                ("binning", ALLOC_FN), ("binning_ctx", C.c_void_p), ("image", ALLOC_FN), ("image_ctx", C.c_void_p),
                ("stream", C.c_void_p), ("num_rendered", C.c_int32)]


class ShadeParams(C.Structure):
    """svgir_shade_params (include/svgir_raster.h)."""
    _fields_ = [("P", C.c_int32), ("Ns", C.c_int32), ("env_h", C.c_int32), ("env_w", C.c_int32),
                ("env_softplus", C.c_int32), ("training", C.c_int32), ("env_scale", C.c_float),
                ("base_color", C.c_void_p), ("roughness", C.c_void_p), ("normals", C.c_void_p),
                ("viewdirs", C.c_void_p), ("radiance", C.c_void_p), ("visibility", C.c_void_p),
                ("incident_dirs", C.c_void_p), ("incident_areas", C.c_void_p), ("env", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("env_work", C.c_void_p), ("env_transform", C.c_void_p),
                ("lattice_normals", C.c_void_p), ("lattice_offsets", C.c_void_p), ("lattice_work", C.c_void_p),
                ("subset", C.c_void_p), ("subset_count", C.c_void_p), ("radiance_ratio", C.c_void_p)]


class FusedShade(C.Structure):
    """svgir_fused_shade: the shading of a view run inside svgir_forward / svgir_backward for the view's working set."""
    _fields_ = [("sp", ShadeParams), ("reduced", C.c_void_p), ("all_surfels", C.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C svg-ir_amd/csrc). There is no CPU / PyTorch fallback for the rasterizer.")
    lib = C.CDLL(LIB_PATH)
    lib.svgir_abi_version.restype = C.c_int
    lib.svgir_geom_bytes.restype = C.c_size_t
    lib.svgir_geom_bytes.argtypes = [C.c_int32]
    lib.svgir_image_bytes.restype = C.c_size_t
    lib.svgir_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.svgir_binning_bytes.restype = C.c_size_t
    lib.svgir_binning_bytes.argtypes = [C.c_int32] * 5
    lib.svgir_image_ncontrib_offset.restype = C.c_size_t
    lib.svgir_image_ncontrib_offset.argtypes = [C.c_int32, C.c_int32]
    lib.svgir_image_ranges_offset.restype = C.c_size_t
    lib.svgir_image_ranges_offset.argtypes = [C.c_int32, C.c_int32]
    lib.svgir_binning_point_list_offset.restype = C.c_size_t
    lib.svgir_binning_point_list_offset.argtypes = [C.c_size_t, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.svgir_forward.restype = C.c_int
    lib.svgir_forward.argtypes = [C.POINTER(Params), C.POINTER(Outputs), ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p,
                                  ALLOC_FN, C.c_void_p, C.c_void_p]
    lib.svgir_forward_batch.restype = C.c_int
    lib.svgir_forward_batch.argtypes = [C.POINTER(ViewCall), C.c_int32]
    lib.svgir_backward.restype = C.c_int
    lib.svgir_backward.argtypes = [C.POINTER(Params), C.POINTER(Grads), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.svgir_backward_scratch_bytes.restype = C.c_size_t
    lib.svgir_backward_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.svgir_backward_scratch_bytes_for.restype = C.c_size_t
    lib.svgir_backward_scratch_bytes_for.argtypes = [C.c_int32, C.c_int32, C.c_size_t, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.svgir_speculation_stats.restype = None
    lib.svgir_speculation_stats.argtypes = [C.POINTER(C.c_int64)]
    lib.svgir_reset_workload_history.restype = None
    lib.svgir_reset_workload_history.argtypes = [C.c_int32]
    lib.svgir_mark_visible.restype = C.c_int
    lib.svgir_mark_visible.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svgir_set_profiling.argtypes = [C.c_int]
    lib.svgir_last_timings.restype = C.c_int
    lib.svgir_last_timings.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]
    lib.svgir_last_error.restype = C.c_char_p
    if lib.svgir_abi_version() != ABI_VERSION:
        raise ImportError("libsvgir_raster.so ABI version mismatch")
    return lib


lib = _load()

EXPORTS = ("svgir_abi_version", "svgir_geom_bytes", "svgir_image_bytes", "svgir_binning_bytes",
           "svgir_image_ncontrib_offset", "svgir_image_ranges_offset", "svgir_binning_point_list_offset", "svgir_forward", "svgir_forward_batch", "svgir_backward", "svgir_mark_visible",
           "svgir_backward_scratch_bytes", "svgir_backward_scratch_bytes_for", "svgir_speculation_stats", "svgir_reset_workload_history", "svgir_set_profiling", "svgir_last_timings", "svgir_last_error", "svgir_shade_forward",
           "svgir_shade_backward", "svgir_incident_dirs", "svgir_resample_bilinear", "svgir_unpack_planes",
           "svgir_unpack_forward", "svgir_unpack_backward", "svgir_depth2normal", "svgir_depth2normal_backward", "svgir_pack_rgss_forward",
           "svgir_pack_rgss_backward", "svgir_unpack_rgss_forward", "svgir_unpack_rgss_backward", "svgir_l1_ssim_partials",
           "svgir_l1_ssim_forward", "svgir_l1_ssim_backward", "svgir_adam_step", "svgir_densify_stats",
           "svgir_mask_scan_work_words", "svgir_mask_scan", "svgir_gather_rows", "svgir_densify_masks", "svgir_append_rows",
           "svgir_split_transform", "svgir_bvh_bytes", "svgir_bvh_build",
           "svgir_bvh_trace_visibility", "svgir_pbgi_bvh_bytes", "svgir_pbgi_bvh_build", "svgir_pbgi_bvh_export",
           "svgir_pbgi_trace_radiance")


_scope = threading.local()


class workload_scope:
    """`with workload_scope(7): ...` -- every rasterizer call of this thread inside the block carries svgir_params.workload_scope = 7.
    The reference's settings tuples are fixed (GaussianRasterizationSettings), so the scope travels next to them: a caller that keeps
    several models with equal image sizes / widths in one process gives each its own id, and their speculation histories (instance
    capacity, state slots, depth-key byte: include/svgir_raster.h) stay apart."""

    def __init__(self, scope):
        self.scope, self.prev = int(scope), 0

    def __enter__(self):
        self.prev = getattr(_scope, "id", 0)
        _scope.id = self.scope
        return self

    def __exit__(self, *exc):
        _scope.id = self.prev
        return False


def new_params():
    """A zeroed svgir_params carrying the calling thread's workload scope."""
    p = Params()
    p.workload_scope = getattr(_scope, "id", 0)
    return p


def reset_workload_history(scope=-1):
    """svgir_reset_workload_history: forget the speculation history of one scope (or all, scope < 0)."""
    lib.svgir_reset_workload_history(int(scope))


def speculation_stats():
    """{forwards, re-runs: instance capacity / state slots / depth-key byte, three-pass depth sorts} of this process (include/svgir_raster.h)."""
    out = (C.c_int64 * 5)()
    lib.svgir_speculation_stats(out)
    return dict(zip(("forwards", "rerun_capacity", "rerun_slots", "rerun_depth_key", "three_pass"), [int(v) for v in out]))


def last_error():
    return (lib.svgir_last_error() or b"").decode()


def check(rc, what):
    if rc < 0:
        raise RuntimeError(f"svgir {what} failed ({rc}): {last_error()}")
    return rc


def guarded(dev, what, fn, *args):
    """Calls a C-ABI entry point with `dev` as the current device (the library creates its side stream / events on the device
    that owns the stream; tensors on a non-current device must not end up with helpers on another GPU)."""
    with torch.cuda.device(dev):
        return check(fn(*args), what)


def ptr(t):
    """Device pointer of a contiguous float/int tensor, or None for empty/absent tensors (the reference passes
    `torch.Tensor([])` whose data pointer is null, rasterize_points.cu:111-119)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def f32c(t, device):
    """Contiguous fp32 copy/view on `device` (the reference calls .contiguous() on every argument)."""
    if t is None:
        return None
    if t.numel() == 0:
        return t
    if t.device != device or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t.contiguous()


ALLOC_STATS = {"calls": 0, "seconds": 0.0, "max_seconds": 0.0}   # wall time spent inside the allocation callbacks


class BlobAllocator:
    """The three resizable byte blobs of the reference glue (rasterize_points.cu:27-33) as torch uint8 tensors.

    The ctypes thunks must not keep the allocator (and with it the blobs) alive through a reference cycle -- a cycle is
    only collected by Python's cyclic GC, i.e. several generations of multi-GB blobs would pile up and every forward
    would fall through the caching allocator to hipMalloc.  The closures therefore capture a plain dict, and `take()`
    hands the tensors over and drops the thunks."""

    def __init__(self, device, stream=None):
        self.device = device
        self.stream = stream    # the stream the forward runs on when it is not the caller's current one (svgir_forward_batch)
        self.tensors = {}
        self._fns = []

    def fn(self, name):
        tensors, device, stream = self.tensors, self.device, self.stream

        def alloc(nbytes, _ctx):
            t0 = time.perf_counter()
            tensors[name] = None          # a speculative blob that turned out too small is released first
            if stream is None:
                t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            else:   # (the caching allocator ties a block to the stream that is current when it is allocated)
                with torch.cuda.stream(stream):
                    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            tensors[name] = t
            dt = time.perf_counter() - t0
            ALLOC_STATS["calls"] += 1
            ALLOC_STATS["seconds"] += dt
            ALLOC_STATS["max_seconds"] = max(ALLOC_STATS["max_seconds"], dt)
            return t.data_ptr()

        f = ALLOC_FN(alloc)
        self._fns.append(f)  # keep the thunk alive for the duration of the call
        return f

    def get(self, name):
        t = self.tensors.get(name)
        if t is None:
            t = torch.empty(0, dtype=torch.uint8, device=self.device)
        return t

    def take(self, *names):
        out = tuple(self.get(n) for n in names)
        self._fns = []
        self.tensors = {}
        return out


def run_forward(steps):
    """Drives one `_forward_steps` generator of a binding (it yields (device, Params, Outputs, BlobAllocator) where the C call belongs
    and returns the binding's tuple): ONE svgir_forward on the current stream."""
    try:
        dev, p, o, blobs = next(steps)
    except StopIteration as e:   # (P == 0: nothing to launch)
        return e.value
    rendered = guarded(dev, "forward", lib.svgir_forward, p, o, blobs.fn("geom"), None, blobs.fn("binning"), None, blobs.fn("image"), None,
                       stream_ptr(dev))
    try:
        steps.send(rendered)
    except StopIteration as e:
        return e.value
    raise RuntimeError("binding generator did not finish")


def run_forward_batch(make_steps, device, streams):
    """svgir_forward_batch: `make_steps[v]()` creates the `_forward_steps` generator of view v (called with streams[v] current, so the
    view's outputs and blobs belong to that stream); all views are launched before the first one's instance count is awaited -- one host
    thread, len(streams) views in flight.  Returns the bindings' tuples in order."""
    n = len(make_steps)
    gens, reqs, out = [None] * n, [None] * n, [None] * n
    for v in range(n):
        with torch.cuda.stream(streams[v]):
            g = make_steps[v]()
            try:
                reqs[v] = next(g)
                gens[v] = g
            except StopIteration as e:
                out[v] = e.value
    live = [v for v in range(n) if gens[v] is not None]
    if live:
        calls = (ViewCall * len(live))()
        for i, v in enumerate(live):
            dev, p, o, blobs = reqs[v]
            blobs.stream = streams[v]
            c = calls[i]
            c.params, c.outputs = C.pointer(p), C.pointer(o)
            c.geom, c.binning, c.image = blobs.fn("geom"), blobs.fn("binning"), blobs.fn("image")
            c.stream = streams[v].cuda_stream
        with torch.cuda.device(device):
            rc = lib.svgir_forward_batch(calls, len(live))
        for i, v in enumerate(live):
            if calls[i].num_rendered < 0:
                raise RuntimeError(f"svgir forward_batch failed for view {v} ({calls[i].num_rendered}): {last_error()}")
        check(rc, "forward_batch")
        for i, v in enumerate(live):
            with torch.cuda.stream(streams[v]):
                try:
                    gens[v].send(int(calls[i].num_rendered))
                except StopIteration as e:
                    out[v] = e.value
    return out


CLEAR_HINT = True   # pass the gradient blob as svgir_grads.clear_base (tests switch it off to cover the per-tensor clears)
POISON = os.environ.get("SVGIR_POISON", "") not in ("", "0")   # tests: NaN-fill every buffer the library must overwrite


def out_tensor(shape, dtype, device):
    """An output buffer the library writes completely (ABI 5: nothing needs clearing by the caller)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    if POISON and t.numel():
        t.fill_(float("nan") if dtype.is_floating_point else -7)
    return t


def grad_blob(device, shapes, zero=False):
    """fp32 gradient tensors of `shapes` carved out of ONE allocation; every view starts on a 256-byte boundary.  Returns
    (views, blob).  Not cleared (unless `zero`: the P = 0 case, where nothing runs): svgir_backward clears the blob itself
    (Grads.clear_base / clear_bytes) next to its first kernel."""
    sizes = [1] * len(shapes)
    for i, sh in enumerate(shapes):
        for d in sh:
            sizes[i] *= int(d)
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 63) // 64 * 64
    blob = torch.zeros(tot, dtype=torch.float32, device=device) if zero else out_tensor((tot,), torch.float32, device)
    return [blob[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)], blob


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def set_profiling(on):
    lib.svgir_set_profiling(1 if on else 0)


def last_timings(with_counts=False):
    """Average milliseconds per stage since profiling was enabled (resolves the recorded HIP events)."""
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    cnt = (C.c_int * 32)()
    n = lib.svgir_last_timings(names, ms, cnt, 32)
    if with_counts:
        return [(names[i].decode(), float(ms[i]), int(cnt[i])) for i in range(n)]
    return [(names[i].decode(), float(ms[i])) for i in range(n)]
