"""The consumers of the rasterizer's gradients (csrc/optim.hip):

  FusedAdam                 -- torch.optim.Adam as the reference configures it (scene/gaussian_model.py:737-773: one
                               parameter per group, per-group lr, eps = 1e-15), same `param_groups` / `state` layout
                               (state[p] = {"step", "exp_avg", "exp_avg_sq"}), so the reference's `_prune_optimizer`,
                               `cat_tensors_to_optimizer` and `replace_tensor_to_optimizer` keep working on it; `step()` is
                               ONE kernel launch for all parameters
  add_densification_stats   -- scene/gaussian_model.py:1270-1276, one launch
  prune_rows                -- `t[mask]` for many per-Gaussian tensors at once (`_prune_optimizer` + `prune_points`,
                               scene/gaussian_model.py:1020-1062): one scan of the mask, one gather launch
  DensifyState              -- the densification half of GaussianModel on the parameter block (same method names):
                               `step` (NaN-gradient scrub + Adam + zero_grad in ONE launch, :775-813), `densify_and_clone`,
                               `densify_and_split`, `densify_and_prune`, `prune`, `prune_points`, `cat_tensors_to_optimizer`,
                               `densification_postfix` (:1064-1268): selection masks in one kernel, every parameter / moment /
                               bookkeeping array appended in one launch, split transform in one launch
"""
import ctypes as C

import torch

from gaussian_renderer import _native as N

MAX_TENSORS = 32


class _AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_double), ("step", C.c_int32), ("flags", C.c_int32), ("nan_value", C.c_float)]


ADAM_SCRUB_NAN, ADAM_ZERO_GRAD, APPEND_ZERO_NEW = 1, 2, 1


class _AppendTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int32), ("flags", C.c_int32)]


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
N.lib.svgir_densify_masks.restype = C.c_int
N.lib.svgir_densify_masks.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_float] * 3 + [C.c_void_p] * 3
N.lib.svgir_append_rows.restype = C.c_int
N.lib.svgir_append_rows.argtypes = [C.POINTER(_AppendTensor), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
N.lib.svgir_split_transform.restype = C.c_int
N.lib.svgir_split_transform.argtypes = [C.c_int64, C.c_int32] + [C.c_void_p] * 5


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for `torch.optim.Adam(params, lr=..., eps=...)` (betas, eps, per-group lr; no weight decay / amsgrad)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, nan_values=None, zero_grad=False):
        """`nan_values`: {group name: replacement} -- NaN gradient entries of those groups are replaced (in the gradient
        tensor too) before the update; `zero_grad`: the reference's `optimizer.zero_grad()` behind the step -- on torch >= 2 that is
        `set_to_none=True`, so `p.grad` is None afterwards (a parameter that receives no gradient in a later iteration is then
        SKIPPED by the next step, exactly as with torch.optim.Adam, instead of being stepped with g = 0).  The kernel still
        scrubs NaNs in its one pass over the gradients (GaussianModel.step = replace_nangrad_to_zero + optimizer.step +
        optimizer.zero_grad); `zero_grad="fill"` keeps the round-3 behaviour (gradient tensors kept and zero-filled by the kernel)."""
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
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                g = p.grad
                flags = ADAM_ZERO_GRAD if zero_grad == "fill" else 0
                nanv = 0.0
                if nan_values is not None and group.get("name") in nan_values:
                    flags |= ADAM_SCRUB_NAN
                    nanv = float(nan_values[group["name"]])
                batches.setdefault((p.device, float(b1), float(b2), float(group["eps"])), []).append(
                    (p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), int(st["step"]), flags, nanv))
        for (dev, b1, b2, eps), ents in batches.items():
            for i in range(0, len(ents), MAX_TENSORS):
                chunk = ents[i:i + MAX_TENSORS]
                arr = (_AdamTensor * len(chunk))()
                for a, (p, g, m, v, lr, step, flags, nanv) in zip(arr, chunk):
                    a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                    a.n, a.lr, a.step, a.flags, a.nan_value = p.numel(), lr, step, flags, nanv
                with torch.cuda.device(dev):
                    N.check(N.lib.svgir_adam_step(arr, len(chunk), b1, b2, eps, N.stream_ptr(dev)), "adam_step")
        if zero_grad and zero_grad != "fill":   # (stream-ordered allocator: the gradient storage outlives the launch that reads it)
            for ents in batches.values():
                for ent in ents:
                    ent[0].grad = None
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


# replacement values of GaussianModel.replace_nangrad_to_zero (scene/gaussian_model.py:775-797); groups not listed keep their NaNs
NANGRAD_VALUES = {"xyz": 0.0, "f_dc": 0.0, "f_rest": 0.0, "scaling": 1e-6, "rotation": 1e-6, "opacity": 0.0}
NANGRAD_VALUES_PBR = {"roughness": 1e-6, "base_color": 0.0, "normal": 0.0}


def _scan(mask):
    """(list int32 [P], count int32 [1] on the device, number of set entries) of a bool mask -- one scan, one 4-byte read."""
    dev = mask.device
    P = mask.shape[0]
    k8 = mask.to(torch.uint8).contiguous()
    kept = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    work = torch.empty(N.lib.svgir_mask_scan_work_words(P), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    N.check(N.lib.svgir_mask_scan(P, k8.data_ptr(), kept.data_ptr(), work.data_ptr(), count.data_ptr(), N.stream_ptr(dev)), "mask_scan")
    return kept, count, int(count.item())


@torch.no_grad()
def append_rows(tensors, sel_list, sel_count, n_sel, repeat=1, zero_new=()):
    """[cat(t, t[sel].repeat(repeat, 1...)) for t in tensors] in one launch per 32 tensors; tensors whose index is in
    `zero_new` get zero rows appended instead (the Adam moments in cat_tensors_to_optimizer)."""
    if not tensors:
        return []
    dev = tensors[0].device
    P = tensors[0].shape[0]
    srcs = [t.contiguous() for t in tensors]
    outs = [torch.empty((P + n_sel * repeat,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in srcs]
    for i in range(0, len(srcs), MAX_TENSORS):
        chunk = list(zip(range(i, min(i + MAX_TENSORS, len(srcs))), srcs[i:i + MAX_TENSORS], outs[i:i + MAX_TENSORS]))
        arr = (_AppendTensor * len(chunk))()
        for a, (j, s, d) in zip(arr, chunk):
            rb = (d.numel() // max(d.shape[0], 1)) * d.element_size() if d.shape[0] else 0
            if rb % 4:
                raise ValueError("append_rows: rows must be a multiple of 4 bytes")
            a.src, a.dst, a.row_bytes, a.flags = s.data_ptr(), d.data_ptr(), rb, (APPEND_ZERO_NEW if j in zero_new else 0)
        N.check(N.lib.svgir_append_rows(arr, len(chunk), P, sel_list.data_ptr(), sel_count.data_ptr(), n_sel, repeat,
                                        N.stream_ptr(dev)), "append_rows")
    return outs


class DensifyState:
    """The optimisation / densification half of the reference's GaussianModel on the per-Gaussian parameter block.

    `params`: {group name: nn.Parameter [P, ...]} with the reference's group names (xyz, normal, rotation, scaling, opacity,
    f_dc, f_rest [, base_color, roughness, incidents_dc, incidents_rest, visibility_dc, visibility_rest]); `optimizer`: a
    FusedAdam (or torch.optim.Adam) with one parameter per group, named like the reference's.  The bookkeeping arrays are
    the reference's: weights_accum, xyz_gradient_accum, normal_gradient_accum, denom [P,1], max_radii2D [P]."""

    def __init__(self, params, optimizer, percent_dense=0.01, use_pbr=False):
        self.params, self.optimizer, self.percent_dense, self.use_pbr = dict(params), optimizer, percent_dense, use_pbr
        dev = self.params["xyz"].device
        P = self.params["xyz"].shape[0]
        z = lambda *s: torch.zeros(*s, device=dev)
        self.weights_accum, self.xyz_gradient_accum, self.normal_gradient_accum, self.denom = z(P, 1), z(P, 1), z(P, 1), z(P, 1)
        self.max_radii2D = z(P)

    # -- GaussianModel.step (scene/gaussian_model.py:809-813) --
    def step(self):
        nv = dict(NANGRAD_VALUES)
        if self.use_pbr:
            nv.update(NANGRAD_VALUES_PBR)
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.step(nan_values=nv, zero_grad=True)
        else:
            raise RuntimeError("DensifyState.step needs a FusedAdam")

    def add_densification_stats(self, viewspace_point_tensor, update_filter, weights):
        add_densification_stats(viewspace_point_tensor.grad, update_filter, weights, self.weights_accum, self.xyz_gradient_accum,
                                self.denom)

    # -- helpers --
    def _groups(self):
        out = []
        for group in self.optimizer.param_groups:
            assert len(group["params"]) == 1
            out.append((group, self.optimizer.state.get(group["params"][0], None)))
        return out

    def _install(self, group, stored_state, new_param, new_m, new_v):
        """the parameter swap of _prune_optimizer / cat_tensors_to_optimizer"""
        old = group["params"][0]
        if stored_state is not None:
            stored_state["exp_avg"], stored_state["exp_avg_sq"] = new_m, new_v
            del self.optimizer.state[old]
        group["params"][0] = torch.nn.Parameter(new_param.requires_grad_(True))
        if stored_state is not None:
            self.optimizer.state[group["params"][0]] = stored_state
        self.params[group["name"]] = group["params"][0]

    # -- cat_tensors_to_optimizer + densification_postfix, fused: new rows = rows `sel` of every parameter, `repeat` times --
    def _append_selected(self, sel_mask, repeat):
        sel_list, sel_count, n = _scan(sel_mask)
        groups = self._groups()
        tensors, zero_new, slots = [], set(), []
        for group, st in groups:
            p = group["params"][0].detach()
            slots.append(len(tensors)); tensors.append(p)
            if st is not None:
                zero_new.update((len(tensors), len(tensors) + 1))
                tensors += [st["exp_avg"], st["exp_avg_sq"]]
        outs = append_rows(tensors, sel_list, sel_count, n, repeat, zero_new)
        for (group, st), k in zip(groups, slots):
            self._install(group, st, outs[k], outs[k + 1] if st is not None else None, outs[k + 2] if st is not None else None)
        P_new = self.params["xyz"].shape[0]
        dev = sel_mask.device
        # densification_postfix: weights_accum gets ones for the new points, the other statistics restart from zero
        self.weights_accum = torch.cat([self.weights_accum, torch.ones((n * repeat, 1), device=dev)], dim=0)
        self.xyz_gradient_accum = torch.zeros((P_new, 1), device=dev)
        self.normal_gradient_accum = torch.zeros((P_new, 1), device=dev)
        self.denom = torch.zeros((P_new, 1), device=dev)
        self.max_radii2D = torch.zeros((P_new,), device=dev)
        return n

    def cat_tensors_to_optimizer(self, tensors_dict):
        """The reference's method for caller-supplied extension tensors (new moments are zero)."""
        optimizable = {}
        for group, st in self._groups():
            ext = tensors_dict[group["name"]]
            new_p = torch.cat((group["params"][0].detach(), ext), dim=0)
            new_m = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0) if st is not None else None
            new_v = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0) if st is not None else None
            self._install(group, st, new_p, new_m, new_v)
            optimizable[group["name"]] = group["params"][0]
        return optimizable

    def prune_points(self, mask):
        """prune_points + _prune_optimizer: one mask scan + one gather launch for every parameter, moment and statistic."""
        keep = ~mask
        groups = self._groups()
        tensors, slots = [], []
        for group, st in groups:
            slots.append(len(tensors)); tensors.append(group["params"][0].detach())
            if st is not None:
                tensors += [st["exp_avg"], st["exp_avg_sq"]]
        book = [self.weights_accum, self.xyz_gradient_accum, self.normal_gradient_accum, self.denom, self.max_radii2D]
        outs = prune_rows(tensors + book, keep)
        for (group, st), k in zip(groups, slots):
            self._install(group, st, outs[k], outs[k + 1] if st is not None else None, outs[k + 2] if st is not None else None)
        self.weights_accum, self.xyz_gradient_accum, self.normal_gradient_accum, self.denom, self.max_radii2D = outs[len(tensors):]

    def _masks(self, grad_threshold, scene_extent, grad_normal_threshold):
        dev = self.params["xyz"].device
        P = self.params["xyz"].shape[0]
        clone = torch.empty(P, dtype=torch.uint8, device=dev)
        split = torch.empty(P, dtype=torch.uint8, device=dev)
        N.check(N.lib.svgir_densify_masks(P, self.xyz_gradient_accum.data_ptr(), self.normal_gradient_accum.data_ptr(),
                                          self.denom.data_ptr(), self.params["scaling"].detach().contiguous().data_ptr(),
                                          float(grad_threshold), float(grad_normal_threshold), float(self.percent_dense * scene_extent),
                                          clone.data_ptr(), split.data_ptr(), N.stream_ptr(dev)), "densify_masks")
        return clone.bool(), split.bool()

    def densify_and_clone(self, clone_mask):
        return self._append_selected(clone_mask, 1)

    def densify_and_split(self, split_mask, N_split=2, z=None):
        """`split_mask` over the CURRENT points (pad with False for points cloned since the statistics were taken, like the
        reference's padded_grad); `z`: standard-normal draws [N_split * selected, 3] (drawn here when None)."""
        P = self.params["xyz"].shape[0]
        dev = split_mask.device
        if split_mask.shape[0] < P:
            split_mask = torch.cat([split_mask, torch.zeros(P - split_mask.shape[0], dtype=torch.bool, device=dev)])
        n = self._append_selected(split_mask, N_split)
        n_new = n * N_split
        if n_new:
            if z is None:
                z = torch.randn(n_new, 3, device=dev)
            z = N.f32c(z, dev)
            xyz, scaling, rot = self.params["xyz"].data, self.params["scaling"].data, self.params["rotation"].data
            N.check(N.lib.svgir_split_transform(n_new, N_split, z.data_ptr(), xyz[P:].data_ptr(), scaling[P:].data_ptr(),
                                                rot[P:].data_ptr(), N.stream_ptr(dev)), "split_transform")
        prune_filter = torch.cat((split_mask, torch.zeros(n_new, device=dev, dtype=torch.bool)))
        self.prune_points(prune_filter)
        return n

    def _prune_mask(self, min_opacity, extent, max_screen_size, weights_threshold):
        opacity = torch.sigmoid(self.params["opacity"].detach())
        prune_mask = (opacity < min_opacity).squeeze(-1)
        prune_mask = torch.logical_or(self.weights_accum[:, 0] < weights_threshold, prune_mask)
        if max_screen_size:
            big_points_vs = self.max_radii2D > max_screen_size
            big_points_ws = torch.nan_to_num(torch.exp(self.params["scaling"].detach()), nan=1e-6).max(dim=1).values > 0.1 * extent   # get_scaling
            prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
        return prune_mask

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, max_grad_normal, weights_threshold=1e-5, z=None):
        clone_mask, split_mask = self._masks(max_grad, extent, max_grad_normal)
        self.densify_and_clone(clone_mask)
        self.densify_and_split(split_mask, 2, z)
        self.prune_points(self._prune_mask(min_opacity, extent, max_screen_size, weights_threshold))
        self.weights_accum.data[:] = 0.0

    def prune(self, min_opacity, extent, max_screen_size, weights_threshold=1e-4):
        self.prune_points(self._prune_mask(min_opacity, extent, max_screen_size, weights_threshold))
        self.weights_accum.data[:] = 0.0
