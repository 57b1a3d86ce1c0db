"""The consumers of the rasterizer's gradients (csrc/optim.hip):

  FusedAdam                 -- torch.optim.Adam as the reference configures it (scene/gaussian_model.py:737-773: one
                               parameter per group, per-group lr, eps = 1e-15), same `param_groups` / `state` layout
                               (state[p] = {"step", "exp_avg", "exp_avg_sq"}), so the reference's `_prune_optimizer`,
                               `cat_tensors_to_optimizer` and `replace_tensor_to_optimizer` keep working on it; `step()` is
                               ONE kernel launch for all parameters
  add_densification_stats   -- scene/gaussian_model.py:1270-1276, one launch
  prune_rows                -- `t[mask]` for many per-Gaussian tensors at once (`_prune_optimizer` + `prune_points`,
                               scene/gaussian_model.py:1020-1062): one scan of the mask, one gather launch
"""
import ctypes as C

import torch

from gaussian_renderer import _native as N

MAX_TENSORS = 32


class _AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_double), ("step", C.c_int32)]


class _RowTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int32)]


N.lib.svgir_adam_step.restype = C.c_int
N.lib.svgir_adam_step.argtypes = [C.POINTER(_AdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p]
N.lib.svgir_densify_stats.restype = C.c_int
N.lib.svgir_densify_stats.argtypes = [C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 6
N.lib.svgir_mask_scan_work_words.restype = C.c_size_t
N.lib.svgir_mask_scan_work_words.argtypes = [C.c_int32]
N.lib.svgir_mask_scan.restype = C.c_int
N.lib.svgir_mask_scan.argtypes = [C.c_int32] + [C.c_void_p] * 5
N.lib.svgir_gather_rows.restype = C.c_int
N.lib.svgir_gather_rows.argtypes = [C.POINTER(_RowTensor), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for `torch.optim.Adam(params, lr=..., eps=...)` (betas, eps, per-group lr; no weight decay / amsgrad)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        batches = {}   # (device, beta1, beta2, eps) -> [entries]
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: contiguous fp32 parameters on the GPU only (libsvgir_raster.so has no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)   # torch.optim.Adam's layout (capturable=False: a CPU scalar tensor)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                batches.setdefault((p.device, float(b1), float(b2), float(group["eps"])), []).append(
                    (p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), int(st["step"])))
        for (dev, b1, b2, eps), ents in batches.items():
            for i in range(0, len(ents), MAX_TENSORS):
                chunk = ents[i:i + MAX_TENSORS]
                arr = (_AdamTensor * len(chunk))()
                for a, (p, g, m, v, lr, step) in zip(arr, chunk):
                    a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                    a.n, a.lr, a.step = p.numel(), lr, step
                with torch.cuda.device(dev):
                    N.check(N.lib.svgir_adam_step(arr, len(chunk), b1, b2, eps, N.stream_ptr(dev)), "adam_step")
        return loss


@torch.no_grad()
def add_densification_stats(viewspace_grad, update_filter, weights, weights_accum, xyz_gradient_accum, denom):
    """GaussianModel.add_densification_stats on its four arrays, in place.  viewspace_grad [P,>=2], update_filter bool [P],
    weights [P,1] or None."""
    dev = viewspace_grad.device
    P = viewspace_grad.shape[0]
    vg = N.f32c(viewspace_grad, dev)
    flt = update_filter.to(torch.uint8) if update_filter.dtype != torch.uint8 else update_filter
    flt = flt.contiguous()
    w = N.f32c(weights, dev) if weights is not None else None
    N.check(N.lib.svgir_densify_stats(P, vg.data_ptr(), vg.shape[1], flt.data_ptr(), N.ptr(w), N.ptr(weights_accum),
                                      xyz_gradient_accum.data_ptr(), denom.data_ptr(), N.stream_ptr(dev)), "densify_stats")


@torch.no_grad()
def prune_rows(tensors, keep):
    """[t[keep] for t in tensors] for per-row tensors sharing dim 0 (fp32 / int32 / any dtype whose row is a multiple of
    4 bytes), with ONE mask scan and one gather launch per 32 tensors.  Like boolean indexing it reads the kept count back
    to size the results."""
    dev = keep.device
    P = keep.shape[0]
    k8 = keep.to(torch.uint8).contiguous() if keep.dtype != torch.uint8 else keep.contiguous()
    kept = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    work = torch.empty(N.lib.svgir_mask_scan_work_words(P), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    N.check(N.lib.svgir_mask_scan(P, k8.data_ptr(), kept.data_ptr(), work.data_ptr(), count.data_ptr(), N.stream_ptr(dev)), "mask_scan")
    n = int(count.item())
    srcs = [t.contiguous() for t in tensors]
    outs = []
    for t in srcs:
        if t.shape[0] != P:
            raise ValueError("prune_rows: every tensor must have one row per mask entry")
        outs.append(torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev))
    for i in range(0, len(srcs), MAX_TENSORS):
        chunk = list(zip(srcs[i:i + MAX_TENSORS], outs[i:i + MAX_TENSORS]))
        arr = (_RowTensor * len(chunk))()
        for a, (s, d) in zip(arr, chunk):
            rb = (s.numel() // max(P, 1)) * s.element_size() if P else 0
            if rb % 4:
                raise ValueError("prune_rows: rows must be a multiple of 4 bytes")
            a.src, a.dst, a.row_bytes = s.data_ptr(), d.data_ptr(), rb
        if n:
            N.check(N.lib.svgir_gather_rows(arr, len(chunk), kept.data_ptr(), count.data_ptr(), n, N.stream_ptr(dev)), "gather_rows")
    return outs
