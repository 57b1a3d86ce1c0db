"""Fused per-splat spatially-varying BRDF shading -- the MI355X counterpart of the PyTorch code the reference runs
right before the svgss rasterizer call:

  rendering_equation4 / GGX_specular4            gaussian_renderer/svgss.py:537-631
  DirectLightMap.direct_light / EnvLight.direct_light   scene/direct_light_map.py:70-83, scene/envmap.py:53-72
  feature / vfeature packing                      gaussian_renderer/svgss.py:143-166

`rendering_equation4(...)` keeps the reference's name, argument list and return convention `(pbr, extra_results)`;
`shade_and_pack(...)` additionally fuses the packing and returns the rasterizer's `features` / `vfeatures`.
The compute is `svgir_shade_forward` / `svgir_shade_backward` of libsvgir_raster.so (csrc/shade.hip); there is no
PyTorch fallback.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _native as N

NRED = 70


ShadeParams = N.ShadeParams


N.lib.svgir_shade_forward.restype = C.c_int
N.lib.svgir_shade_forward.argtypes = [C.POINTER(ShadeParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
N.lib.svgir_shade_backward.restype = C.c_int
N.lib.svgir_shade_backward.argtypes = [C.POINTER(ShadeParams)] + [C.c_void_p] * 11
N.lib.svgir_incident_dirs.restype = C.c_int
N.lib.svgir_incident_dirs.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 6
N.lib.svgir_resample_bilinear.restype = C.c_int
N.lib.svgir_resample_bilinear.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]


class FibonacciLattice:
    """The incident directions of `sample_incident_rays` (scene/gaussian_model.py:23-31) WITHOUT the [P,Ns,3] tensor:
    the Fibonacci hemisphere lattice of utils/graphics_utils.py:9-37 around `normals` [P,3], with the per-surfel random
    azimuth `offsets` [P] of the training branch (None: evaluation lattice).  Passed as `incident_dirs_precompute` the
    shading kernels build the directions in registers (12 + 4 bytes per SURFEL instead of 16 bytes per sample);
    `.dirs()` / `.areas()` materialise the reference's tensors for other consumers (ray tracer, visibility baking)."""

    def __init__(self, normals, sample_num, offsets=None):
        self.normals = N.f32c(normals.detach(), normals.device)
        self.offsets = None if offsets is None else N.f32c(offsets.detach().reshape(-1), normals.device)
        self.sample_num = int(sample_num)

    @property
    def shape(self):
        return (self.normals.shape[0], self.sample_num, 3)

    def __getitem__(self, idx):   # chunking along the surfel axis (svgss.py:121-136)
        return FibonacciLattice(self.normals[idx], self.sample_num, None if self.offsets is None else self.offsets[idx])

    def _materialise(self, want_dirs, want_areas):
        dev = self.normals.device
        if dev.type != "cuda":
            raise RuntimeError("FibonacciLattice: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
        P, Ns = self.normals.shape[0], self.sample_num
        dirs = torch.empty((P, Ns, 3), dtype=torch.float32, device=dev) if want_dirs else None
        areas = torch.empty((P, Ns, 1), dtype=torch.float32, device=dev) if want_areas else None
        work = torch.empty(4 * Ns, dtype=torch.float32, device=dev)
        if P:
            N.check(N.lib.svgir_incident_dirs(P, Ns, N.ptr(self.normals), N.ptr(self.offsets), work.data_ptr(), N.ptr(dirs),
                                              N.ptr(areas), N.stream_ptr(dev)), "incident_dirs")
        return dirs, areas

    def dirs(self):
        return self._materialise(True, False)[0]

    def areas(self):
        return self._materialise(False, True)[1]


def sample_incident_rays(normals, is_training=False, sample_num=24, materialize=True):
    """Drop-in for scene/gaussian_model.py:23-31.  Returns (incident_dirs [N,S,3], incident_areas [N,S,1]) like the
    reference, or -- materialize=False -- (FibonacciLattice, None) for the fused shading kernels."""
    offsets = torch.rand(normals.shape[0], device=normals.device) * (2 * torch.pi) if is_training else None
    lat = FibonacciLattice(normals, sample_num, offsets)
    if not materialize:
        return lat, None
    return lat._materialise(True, True)


RATIO_WORK = 256   # SVGIR_SHADE_RATIO_WORK: floats behind the env-gradient scratch when the ratio's gradient is asked for


def _params(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, softplus, scale,
            viewmatrix=None, training=True, env_transform=None, ratio=None):
    dev = base_color.device
    if dev.type != "cuda":
        raise RuntimeError("shading: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
    lattice = dirs if isinstance(dirs, FibonacciLattice) else None
    keep = [N.f32c(t, dev) for t in (base_color, roughness, normals, viewdirs, radiance, visibility,
                                      None if lattice else dirs, areas, env, viewmatrix, env_transform)]
    bc, ro, nr, vd, ra, vi, di, ar, en, vm, et = keep
    P, Ns = ra.shape[0], ra.shape[1]
    env_h, env_w = en.shape[-3], en.shape[-2]
    work = torch.empty(env_h * env_w * 4, dtype=torch.float32, device=dev)   # f(env), one float4 per texel
    p = ShadeParams()
    p.P, p.Ns, p.env_h, p.env_w = P, Ns, env_h, env_w
    p.env_softplus, p.training, p.env_scale = int(bool(softplus)), int(bool(training)), float(scale)
    p.base_color, p.roughness, p.normals, p.viewdirs = N.ptr(bc), N.ptr(ro), N.ptr(nr), N.ptr(vd)
    p.radiance, p.visibility, p.incident_dirs, p.incident_areas = N.ptr(ra), N.ptr(vi), N.ptr(di), N.ptr(ar)
    p.env, p.viewmatrix, p.env_work, p.env_transform = N.ptr(en), N.ptr(vm), work.data_ptr(), N.ptr(et)
    keep.append(work)
    if ratio is not None:   # get_radiances = nan_to_num(_radiances.detach() * _radiance_ratio) inside the kernels (gaussian_model.py:323-324)
        if ratio.numel() != 1:
            raise RuntimeError("radiance_ratio must hold one value")
        rt = N.f32c(ratio.detach().reshape(1), dev)
        p.radiance_ratio = rt.data_ptr()
        keep.append(rt)
    if lattice is not None:
        if lattice.sample_num != Ns or lattice.normals.shape[0] != P:
            raise RuntimeError("FibonacciLattice does not match the radiance tensor's [P, Ns]")
        lwork = torch.empty(4 * Ns, dtype=torch.float32, device=dev)
        p.lattice_normals, p.lattice_offsets, p.lattice_work = N.ptr(lattice.normals), N.ptr(lattice.offsets), lwork.data_ptr()
        keep += [lattice, lwork]
    return p, keep, dev, P, Ns, env_h, env_w


def _radiance_grads(ctx, i_rad, i_ratio, rad, ratio, env_h, env_w, dev):
    """(dL_dradiance or None, dL_dradiance_ratio or None, env-gradient scratch) of a shading backward.  Without a ratio the library
    always writes dL_dradiance; with one the cache is usually detached (the reference's get_radiances) and only the scalar's gradient
    is produced -- no [P, Ns, 3] tensor at all."""
    want_ratio = ratio is not None and ctx.needs_input_grad[i_ratio]
    want_rad = ratio is None or ctx.needs_input_grad[i_rad]
    d_rad = torch.empty_like(rad) if want_rad else None
    d_ratio = torch.empty(1, dtype=torch.float32, device=dev) if want_ratio else None
    gwork = torch.empty(env_h * env_w * 3 + (RATIO_WORK if want_ratio else 0), dtype=torch.float32, device=dev)
    return d_rad, d_ratio, gwork


class _Shade(torch.autograd.Function):
    """reduced[P,70] = [pbr12, diffuse_light12, specular12, direct12, indirect12, mean_incident3, mean_local3,
    mean_global3, mean_visibility1]; differentiable w.r.t. base_color, roughness, normals, radiance, env."""

    @staticmethod
    def forward(ctx, base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, softplus, scale,
                env_transform=None, ratio=None):
        p, keep, dev, P, Ns, _, _ = _params(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas,
                                            env, softplus, scale, env_transform=env_transform, ratio=ratio)
        reduced = torch.empty((P, NRED), dtype=torch.float32, device=dev)
        if P:
            N.check(N.lib.svgir_shade_forward(p, reduced.data_ptr(), None, None, N.stream_ptr(dev)), "shade_forward")
        lat = dirs if isinstance(dirs, FibonacciLattice) else None
        ctx.save_for_backward(base_color, roughness, normals, viewdirs, radiance, visibility, None if lat else dirs, areas, env, ratio)
        ctx.cfg = (softplus, scale, env_transform, lat)
        return reduced

    @staticmethod
    def backward(ctx, g_reduced):
        base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, ratio = ctx.saved_tensors
        softplus, scale, env_transform, lat = ctx.cfg
        p, keep, dev, P, Ns, env_h, env_w = _params(base_color, roughness, normals, viewdirs, radiance, visibility,
                                                    lat if lat is not None else dirs, areas, env, softplus, scale,
                                                    env_transform=env_transform, ratio=ratio)
        g = N.f32c(g_reduced, dev)
        d_base, d_rough, d_norm, d_env = (torch.empty_like(keep[i]) for i in (0, 1, 2, 8))  # all overwritten
        d_rad, d_ratio, gwork = _radiance_grads(ctx, 4, 12, keep[4], ratio, env_h, env_w, dev)
        if P:
            N.check(N.lib.svgir_shade_backward(p, g.data_ptr(), None, None, d_base.data_ptr(), d_rough.data_ptr(), d_norm.data_ptr(),
                                               N.ptr(d_rad), d_env.data_ptr(), gwork.data_ptr(), N.ptr(d_ratio), N.stream_ptr(dev)),
                    "shade_backward")
        else:
            d_env.zero_()
            if d_ratio is not None:
                d_ratio.zero_()
        return (d_base.reshape(base_color.shape), d_rough.reshape(roughness.shape), d_norm.reshape(normals.shape), None,
                None if d_rad is None else d_rad.reshape(radiance.shape), None, None, None, d_env.reshape(env.shape), None, None, None,
                None if d_ratio is None else d_ratio.reshape(ratio.shape))


class _ShadePack(torch.autograd.Function):
    """Shading + packing in one kernel: (features [P,S], vfeatures [P,VS], reduced [P,70]); the backward feeds the
    rasterizer's dL_dfeatures / dL_dvfeatures (and dL_dreduced, if used) straight into svgir_shade_backward."""

    @staticmethod
    def forward(ctx, base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, viewmatrix,
                softplus, scale, training, env_transform=None, ratio=None):
        p, keep, dev, P, Ns, _, _ = _params(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas,
                                            env, softplus, scale, viewmatrix=viewmatrix, training=training,
                                            env_transform=env_transform, ratio=ratio)
        S, VS = (4, 52) if training else (7, 64)
        red = torch.empty((P, NRED), dtype=torch.float32, device=dev)
        feats = torch.empty((P, S), dtype=torch.float32, device=dev)
        vfeats = torch.empty((P, VS), dtype=torch.float32, device=dev)
        if P:
            N.check(N.lib.svgir_shade_forward(p, red.data_ptr(), feats.data_ptr(), vfeats.data_ptr(), N.stream_ptr(dev)),
                    "shade_forward")
        lat = dirs if isinstance(dirs, FibonacciLattice) else None
        ctx.save_for_backward(base_color, roughness, normals, viewdirs, radiance, visibility, None if lat else dirs, areas, env,
                              viewmatrix, ratio)
        ctx.cfg = (softplus, scale, training, env_transform, lat)
        ctx.set_materialize_grads(False)
        return feats, vfeats, red

    @staticmethod
    def backward(ctx, g_feat, g_vfeat, g_red):
        base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, viewmatrix, ratio = ctx.saved_tensors
        softplus, scale, training, env_transform, lat = ctx.cfg
        p, keep, dev, P, Ns, env_h, env_w = _params(base_color, roughness, normals, viewdirs, radiance, visibility,
                                                    lat if lat is not None else dirs, areas, env, softplus, scale,
                                                    viewmatrix=viewmatrix, training=training, env_transform=env_transform, ratio=ratio)
        if g_feat is None and g_vfeat is None and g_red is None:
            return (None,) * 15
        gf, gv, gr = (N.f32c(t, dev) for t in (g_feat, g_vfeat, g_red))
        d_base, d_rough, d_norm, d_env = (torch.empty_like(keep[i]) for i in (0, 1, 2, 8))
        d_rad, d_ratio, gwork = _radiance_grads(ctx, 4, 14, keep[4], ratio, env_h, env_w, dev)
        if P:
            N.check(N.lib.svgir_shade_backward(p, N.ptr(gr), N.ptr(gf), N.ptr(gv), d_base.data_ptr(), d_rough.data_ptr(),
                                               d_norm.data_ptr(), N.ptr(d_rad), d_env.data_ptr(), gwork.data_ptr(), N.ptr(d_ratio),
                                               N.stream_ptr(dev)), "shade_backward")
        else:
            d_env.zero_()
            if d_ratio is not None:
                d_ratio.zero_()
        return (d_base.reshape(base_color.shape), d_rough.reshape(roughness.shape), d_norm.reshape(normals.shape), None,
                None if d_rad is None else d_rad.reshape(radiance.shape), None, None, None, d_env.reshape(env.shape), None, None, None,
                None, None, None if d_ratio is None else d_ratio.reshape(ratio.shape))


def _env_of(light):
    """(env texture [.., He, We, 3], softplus flag, scale, lookup transform [3,3] or None) for the reference's two
    light classes."""
    if hasattr(light, "env"):       # scene/direct_light_map.py: learnable map, softplus, x2
        return light.env, True, 2.0, None
    if hasattr(light, "envmap"):    # scene/envmap.py: HDR map, bilinear down-sample to 32x64, optional rotation
        src = N.f32c(light.envmap, light.envmap.device)
        env = torch.empty((32, 64, src.shape[2]), dtype=torch.float32, device=src.device)
        N.check(N.lib.svgir_resample_bilinear(src.data_ptr(), src.shape[0], src.shape[1], src.shape[2], env.data_ptr(), 32, 64,
                                              N.stream_ptr(src.device)), "resample_bilinear")
        return env, False, 1.0, getattr(light, "transform", None)   # lookup direction = dirs @ transform.T
    raise TypeError("direct_light_env_light must expose .env (DirectLightMap) or .envmap (EnvLight)")


def rendering_equation4(base_color, roughness, normals, viewdirs, radiance, direct_light_env_light=None,
                        visibility_precompute=None, incident_dirs_precompute=None, incident_areas_precompute=None, *,
                        radiance_ratio=None):
    """Drop-in for gaussian_renderer/svgss.py:537-593.  Returns (pbr [n,12], extra_results).
    `radiance_ratio` (extension, keyword only): the reference hands in `pc.get_radiances` = nan_to_num(_radiances.detach() *
    _radiance_ratio) (gaussian_model.py:323-324); pass the RAW cache as `radiance` and the scalar here and the kernels form the product
    themselves -- no [n, Ns, 3] temporaries, and the backward returns the scalar's gradient directly."""
    dirs, areas = incident_dirs_precompute, incident_areas_precompute
    env, softplus, scale, transform = _env_of(direct_light_env_light)
    red = _Shade.apply(base_color, roughness, normals, viewdirs, radiance, visibility_precompute, dirs, areas, env,
                       softplus, scale, transform, radiance_ratio)
    extra_results = {
        # (with a FibonacciLattice the [n, Ns, 3] tensor is never built: an empty stand-in keeps the chunk loop's
        #  torch.cat over every key working, `incident_dirs_precompute.dirs()` materialises the real thing)
        "incident_dirs": dirs if torch.is_tensor(dirs) else radiance.new_empty((radiance.shape[0], 0, 3)),
        # The reference returns the per-sample [n, Ns, 3] light here; its callers only ever take `.mean(-2)` of it
        # (svgss.py:143-151), possibly after `torch.cat(..., dim=0)` over 100k-surfel chunks (svgss.py:121-136).  The fused
        # kernel never materialises the per-sample values: the tensor carries the mean as its single "sample", so both
        # the concatenation and the mean (exactly) keep working.
        "incident_lights": red[:, 60:63].unsqueeze(-2),
        "local_incident_lights": radiance if radiance_ratio is None else torch.nan_to_num(radiance.detach() * radiance_ratio, nan=0.0),
        "global_incident_lights": red[:, 66:69].unsqueeze(-2),
        "incident_visibility": visibility_precompute,
        "diffuse_light": red[:, 12:24],
        "specular": red[:, 24:36],
        "direct": red[:, 36:48],
        "indirect": red[:, 48:60],
    }
    return red[:, 0:12], extra_results


def shade_and_pack(base_color, roughness, normals, viewdirs, radiance, direct_light_env_light, visibility, dirs, areas,
                   viewmatrix, is_training, radiance_ratio=None):
    """Shading + the packing of svgss.py:143-166: returns (features [n,S], vfeatures [n,VS], reduced [n,70]).
    The packing and its adjoint are done inside the kernels in both the no-grad and the autograd path."""
    env, softplus, scale, transform = _env_of(direct_light_env_light)
    return _ShadePack.apply(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, viewmatrix,
                            softplus, scale, bool(is_training), transform, radiance_ratio)


# ---- shading fused into the rasterizer calls: "shade only what the view reads" ---------------------------------------------------

def _fused_struct(p, reduced, all_surfels):
    fs = N.FusedShade()
    C.memmove(C.byref(fs, N.FusedShade.sp.offset), C.byref(p), C.sizeof(ShadeParams))
    fs.reduced = N.ptr(reduced)
    fs.all_surfels = int(bool(all_surfels))
    return fs


class _ShadedRasterize(torch.autograd.Function):
    """svgss.py:116-182 as ONE pair of library calls: svgir_forward runs its preprocess / sorts / per-tile cull, shades exactly the
    surfels that are a candidate of some 8x8 sub-tile of THIS view (packing included) and composites; svgir_backward differentiates
    the composite, then the shading of the surfels that received a blend weight.  Same results as shade_and_pack(...) followed by
    GaussianRasterizer(...) -- bit for bit, except dL/d(env), whose float atomics are summed in another order."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, base_color, roughness, normals, viewdirs, radiance,
                visibility, dirs, areas, env, raster_settings, softplus, scale, training, env_transform, all_surfels, want_reduced,
                ratio=None):
        from .svgss_rasterization import _C
        st = raster_settings
        sp, keep, dev, P, Ns, _, _ = _params(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env,
                                             softplus, scale, viewmatrix=st.viewmatrix, training=training,
                                             env_transform=env_transform, ratio=ratio)
        if P != means3D.shape[0]:
            raise RuntimeError("render_shaded: the material tensors and means3D disagree on the number of surfels")
        S, VS = (4, 52) if training else (7, 64)
        feats, vfeats = N.out_tensor((P, S), torch.float32, dev), N.out_tensor((P, VS), torch.float32, dev)
        red = N.out_tensor((P, NRED), torch.float32, dev) if want_reduced else None
        fs = _fused_struct(sp, red, all_surfels)
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        out = _C.rasterize_gaussians(st.bg, means3D, feats, vfeats, empty, opacities, scales, rotations, st.scale_modifier, empty,
                                     st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy,
                                     st.image_height, st.image_width, sh, st.sh_degree, st.campos, st.prefiltered, st.debug, st.config,
                                     shade=fs, forward_only=not any(ctx.needs_input_grad))
        (R, color, normal, depth, opacity, feature, vfeature, weights, radii, gb, bb, ib) = out
        lat = dirs if isinstance(dirs, FibonacciLattice) else None
        ctx.save_for_backward(means3D, sh, scales, rotations, base_color, roughness, normals, viewdirs, radiance, visibility,
                              None if lat else dirs, areas, env, feats, vfeats, weights, radii, gb, bb, ib, ratio)
        ctx.cfg = (st, R, softplus, scale, training, env_transform, lat, all_surfels)
        ctx.mark_non_differentiable(weights, radii)
        ctx.set_materialize_grads(False)
        if red is None:
            red = empty
        return R, color, normal, opacity, depth, feature, vfeature, weights, radii, red

    @staticmethod
    def backward(ctx, _gR, g_color, g_normal, g_opacity, g_depth, g_feature, g_vfeature, _gw, _gr, g_red):
        from .svgss_rasterization import _C
        (means3D, sh, scales, rotations, base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, feats,
         vfeats, weights, radii, gb, bb, ib, ratio) = ctx.saved_tensors
        st, R, softplus, scale, training, env_transform, lat, all_surfels = ctx.cfg
        dev = means3D.device
        H, W = st.image_height, st.image_width
        sp, keep, _, P, Ns, env_h, env_w = _params(base_color, roughness, normals, viewdirs, radiance, visibility,
                                                   lat if lat is not None else dirs, areas, env, softplus, scale,
                                                   viewmatrix=st.viewmatrix, training=training, env_transform=env_transform, ratio=ratio)
        if g_red is not None and not all_surfels:
            raise RuntimeError("render_shaded: a loss on `reduced` needs all_surfels=True (rows of unshaded surfels are zero)")
        def _g(g, ch):  # autograd hands None for outputs that did not take part in the loss: an empty tensor = all zero for the library
            return g if g is not None else torch.empty(0, dtype=torch.float32, device=dev)

        empty = torch.empty(0, dtype=torch.float32, device=dev)
        args = (st.bg, means3D, feats, vfeats, radii, empty, scales, rotations, st.scale_modifier, empty, st.viewmatrix, st.projmatrix,
                st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, _g(g_color, 3), _g(g_normal, 3), _g(g_depth, 1), _g(g_opacity, 1),
                _g(g_feature, feats.shape[1]), _g(g_vfeature, vfeats.shape[1] // 4), sh, st.sh_degree, st.campos, gb, R, bb, ib,
                st.debug, st.config)
        if not any(ctx.needs_input_grad[i] for i in (6, 7, 8, 10, 14, 22)):   # frozen materials (evaluation): the rasterizer's backward alone
            res = _C.rasterize_gaussians_backward(*args)
            return (res[3], res[0], res[7], res[2], res[8], res[9]) + (None,) * 17
        fs = _fused_struct(sp, None, all_surfels)
        d_env = N.out_tensor(keep[8].shape, torch.float32, dev)
        want_ratio = ratio is not None and ctx.needs_input_grad[22]
        want_rad = ratio is None or ctx.needs_input_grad[10]   # (with a ratio the cache is normally detached: no [P, Ns, 3] gradient)
        d_ratio = N.out_tensor((1,), torch.float32, dev) if want_ratio else None
        gwork = torch.empty(env_h * env_w * 3 + (RATIO_WORK if want_ratio else 0), dtype=torch.float32, device=dev)
        # (the four per-surfel gradient tensors are carved out of the rasterizer's gradient allocation by the binding: the composite
        # backward clears that region in passing, the shading backward then writes the rows of the surfels that were blended.  Measured
        # at cfg3_train: +28 us in render_bwd / grad_reduce for the 160 MB; a zero-fill launch costs 37 us, zero-fill stores from the
        # shading backward's own waves 40 us -- its per-chunk vmcnt(0) waits for them)
        sg = dict(dL_denv=d_env, env_grad_work=gwork, dL_dreduced=None if g_red is None else N.f32c(g_red, dev), out_weights=weights,
                  _shapes=dict(dL_dbase_color=keep[0].shape, dL_droughness=keep[1].shape, dL_dshade_normals=keep[2].shape))
        if want_rad:
            sg["_shapes"]["dL_dradiance"] = keep[4].shape
        if want_ratio:
            sg["dL_dradiance_ratio"] = d_ratio
        if P == 0:
            d_env.zero_()
            if want_ratio:
                d_ratio.zero_()
        res = _C.rasterize_gaussians_backward(*args, shade=fs, shade_grads=sg, scratch_feature_grads=not all_surfels)
        d_base, d_rough, d_norm, d_rad = sg["dL_dbase_color"], sg["dL_droughness"], sg["dL_dshade_normals"], sg.get("dL_dradiance")
        (g_means2D, _gc, g_opac, g_means3D, _gf, _gvf, _gcov, g_sh, g_scales, g_rot, _gv, _gp, _gcp) = res
        return (g_means3D, g_means2D, g_sh, g_opac, g_scales, g_rot, d_base.reshape(base_color.shape),
                d_rough.reshape(roughness.shape), d_norm.reshape(normals.shape), None,
                None if d_rad is None else d_rad.reshape(radiance.shape), None, None,
                None, d_env.reshape(env.shape), None, None, None, None, None, None, None,
                None if d_ratio is None else d_ratio.reshape(ratio.shape))


def fused_shade(base_color, roughness, normals, viewdirs, radiance, direct_light_env_light, visibility, dirs, areas, viewmatrix,
                is_training, reduced=None, all_surfels=False, radiance_ratio=None):
    """(`_native.FusedShade`, keep-alive list) for the `shade=` keyword of `_C.rasterize_gaussians{,_backward}`: the binding-level
    form of render_shaded (no autograd)."""
    env, softplus, scale, transform = _env_of(direct_light_env_light)
    sp, keep, *_ = _params(base_color, roughness, normals, viewdirs, radiance, visibility, dirs, areas, env, softplus, scale,
                           viewmatrix=viewmatrix, training=bool(is_training), env_transform=transform, ratio=radiance_ratio)
    return _fused_struct(sp, reduced, all_surfels), keep + [reduced]


def render_shaded(raster_settings, means3D, means2D, opacities, shs, scales, rotations, base_color, roughness, normals, viewdirs,
                  radiance, direct_light_env_light, visibility, dirs, areas, is_training, all_surfels=False, want_reduced=False,
                  radiance_ratio=None):
    """Shading + packing + svgss rasterization of one view (the reference's svgss.py:116-182) with the shading restricted to the
    surfels the view reads.  Returns the rasterizer's 9-tuple (num_rendered, color, normal, opacity, depth, feature, vfeature,
    weights, radii) and `reduced` [P,70] (rows of unshaded surfels zero; an empty tensor unless want_reduced).
    all_surfels=True shades every surfel (needed when a loss reads `reduced` directly: the reference's lambda_light term,
    svgss.py:359-364)."""
    env, softplus, scale, transform = _env_of(direct_light_env_light)
    out = _ShadedRasterize.apply(means3D, means2D, shs, opacities, scales, rotations, base_color, roughness, normals, viewdirs,
                                 radiance, visibility, dirs, areas, env, raster_settings, softplus, scale, bool(is_training),
                                 transform, bool(all_surfels), bool(want_reduced), radiance_ratio)
    return out[:9], out[9]
