"""Drop-in for the reference's `gaussian_renderer.rgss_rasterization` (stage-1 Gaussian-surfel rasterizer).

Same public names, argument order, return tuples and gradient tuples as
/root/reference/gaussian_renderer/rgss_rasterization.py:
  * GaussianRasterizationSettings (16 fields, :189-205)
  * GaussianRasterizer(raster_settings).forward(...) -> 11-tuple (num_rendered, num_contrib[H,W] int32, color,
    normal, opacity, depth, feature, pseudo_normal, surface_xyz, weights, radii) (:224-263, :120)
  * _RasterizeGaussians: grads for (means3D, means2D, features, sh, colors_precomp, opacities, scales, rotations,
    cov3Ds_precomp, None) (:173-184); note the opacity/depth gradient order differs from svgss (Q13)
  * `_C` with the pybind argument order of rgss-rasterization/rasterize_points.h:18-73 (23 args -> 14-tuple,
    27 args -> 9-tuple); `num_contrib` is a non-owning int32 view into the image blob (Q10).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _native as N


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class _CBinding:
    """Same three entry points as the reference's pybind module (rgss-rasterization/ext.cpp:15-19)."""

    @staticmethod
    def rasterize_gaussians(*args, **kw):
        """The reference's `_C.rasterize_gaussians` (23 positional arguments, rasterize_points.h:18-43); keyword-only extension:
        `forward_only` (no backward will follow: the composite keeps no blend states)."""
        return N.run_forward(_CBinding._forward_steps(*args, **kw))

    @staticmethod
    def rasterize_gaussians_batch(calls, device, streams):
        """Extension: `calls` = [(args, kwargs)] of rasterize_gaussians, one view each, launched with ONE svgir_forward_batch --
        view v on streams[v], all views in flight before the first instance count is awaited (one host thread).  Returns the
        list of 14-tuples.  The caller orders `streams` against the producers / consumers of the tensors."""
        return N.run_forward_batch([(lambda a=a, k=k: _CBinding._forward_steps(*a, **k)) for a, k in calls], device, streams)

    @staticmethod
    def _forward_steps(background, means3D, features, colors, opacity, scales, rotations, scale_modifier,
                       cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, image_height,
                       image_width, sh, degree, campos, prefiltered, computer_pseudo_normal, debug, *, forward_only=False):
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:62-64
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("rgss rasterizer: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
        P = means3D.size(0)
        S = features.size(1) if features.dim() == 2 else 0
        H, W = int(image_height), int(image_width)
        f32 = dict(dtype=torch.float32, device=dev)
        if P == 0:  # nothing is launched: the reference returns its zero-initialised outputs (rasterize_points.cu:100)
            torch_empty = torch.zeros
        else:
            def torch_empty(shape, **kw):  # every element is written by the library
                return N.out_tensor(shape, kw["dtype"], kw["device"])
        out_color = torch_empty((3, H, W), **f32)
        out_normal = torch_empty((3, H, W), **f32)
        out_opacity = torch_empty((1, H, W), **f32)
        out_depth = torch_empty((1, H, W), **f32)
        out_feature = torch_empty((S, H, W), **f32)
        out_pseudo_normal = torch_empty((3, H, W), **f32)
        out_surface_xyz = torch_empty((3, H, W), **f32)
        out_weights = torch_empty((P, 1), **f32)
        radii = torch_empty((P,), dtype=torch.int32, device=dev)
        blobs = N.BlobAllocator(dev)
        rendered = 0
        if P != 0:
            keep = [N.f32c(t, dev) for t in (background, means3D, sh, colors, features, opacity, scales, rotations,
                                              cov3D_precomp, viewmatrix, projmatrix, campos)]
            (bg, m3, shc, col, fe, op, sc, ro, cv, vm, pm, cp) = keep
            p = N.new_params()
            p.variant, p.P, p.S, p.VS, p.D, p.W, p.H = N.RGSS, P, S, 0, int(degree), W, H
            p.M = shc.size(1) if (shc is not None and shc.numel() != 0) else 0
            p.background, p.means3D, p.shs, p.colors_precomp = N.ptr(bg), N.ptr(m3), N.ptr(shc), N.ptr(col)
            p.features, p.opacities = N.ptr(fe), N.ptr(op)
            p.scales, p.rotations, p.cov3D_precomp = N.ptr(sc), N.ptr(ro), N.ptr(cv)
            p.viewmatrix, p.projmatrix, p.cam_pos = N.ptr(vm), N.ptr(pm), N.ptr(cp)
            p.config_len = 0
            p.scale_modifier, p.tan_fovx, p.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
            p.cx, p.cy = float(cx), float(cy)
            p.prefiltered, p.debug = int(bool(prefiltered)), int(bool(debug))
            p.computer_pseudo_normal = int(bool(computer_pseudo_normal))
            o = N.Outputs()
            o.out_color, o.out_normal, o.out_depth, o.out_opacity = (out_color.data_ptr(), out_normal.data_ptr(),
                                                                     out_depth.data_ptr(), out_opacity.data_ptr())
            o.out_feature = N.ptr(out_feature)
            o.out_pseudo_normal, o.out_surface_xyz = out_pseudo_normal.data_ptr(), out_surface_xyz.data_ptr()
            o.out_weights, o.radii = out_weights.data_ptr(), radii.data_ptr()
            p.forward_only = int(bool(forward_only))
            rendered = yield (dev, p, o, blobs)          # <- svgir_forward / svgir_forward_batch (gaussian_renderer/_native.py)
            img = blobs.get("image")
            off = N.lib.svgir_image_ncontrib_offset(W, H)
            n_contrib = img[off:off + 4 * H * W].view(torch.int32).view(H, W)  # view into the blob (Q10)
        else:
            n_contrib = torch.zeros((H, W), dtype=torch.int32, device=dev)
        return (rendered, n_contrib, out_color, out_normal, out_opacity, out_depth, out_feature, out_pseudo_normal,
                out_surface_xyz, out_weights, radii, *blobs.take("geom", "binning", "image"))

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, features, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                     dL_dout_normal, dL_dout_opacity, dL_dout_depth, dL_dout_feature, sh, degree,
                                     campos, geomBuffer, R, binningBuffer, imageBuffer, backward_geometry, debug, *, out_weights=None):
        dev = means3D.device
        P = means3D.size(0)
        S = features.size(1) if features.dim() == 2 else 0
        # (an upstream gradient may be an EMPTY tensor = all zero: an output that took no part in the loss; nothing is read for it)
        _gs = [t for t in (dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opacity, dL_dout_feature) if t is not None and t.numel()]
        if not _gs:
            raise RuntimeError("rasterize_gaussians_backward: every upstream gradient is empty")
        H, W = _gs[0].size(1), _gs[0].size(2)
        M = sh.size(1) if sh.numel() != 0 else 0
        (dL_dmeans3D, dL_dmeans2D, dL_dfeatures, dL_dcolors, dL_dnormal, dL_ddepth, dL_dconic, dL_dopacity, dL_dcov3D,
         dL_dsh, dL_dscales, dL_drotations), gblob = N.grad_blob(
            dev, [(P, 3), (P, 3), (P, S), (P, 3), (P, 3), (P, 1), (P, 2, 2), (P, 1), (P, 6), (P, M, 3), (P, 3), (P, 4)], zero=(P == 0))
        if P != 0:
            keep = [N.f32c(t, dev) for t in (background, means3D, sh, colors, features, scales, rotations, cov3D_precomp,
                                              viewmatrix, projmatrix, campos, dL_dout_color, dL_dout_normal,
                                              dL_dout_depth, dL_dout_opacity, dL_dout_feature)]
            (bg, m3, shc, col, fe, sc, ro, cv, vm, pm, cp, gc, gn, gd, go, gf) = keep
            p = N.new_params()
            p.variant, p.P, p.S, p.VS, p.D, p.M, p.W, p.H = N.RGSS, P, S, 0, int(degree), M, W, H
            p.background, p.means3D, p.shs, p.colors_precomp = N.ptr(bg), N.ptr(m3), N.ptr(shc), N.ptr(col)
            p.features = N.ptr(fe)
            p.scales, p.rotations, p.cov3D_precomp = N.ptr(sc), N.ptr(ro), N.ptr(cv)
            p.viewmatrix, p.projmatrix, p.cam_pos = N.ptr(vm), N.ptr(pm), N.ptr(cp)
            p.config_len = 0
            p.scale_modifier, p.tan_fovx, p.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
            p.backward_geometry, p.debug = int(bool(backward_geometry)), int(bool(debug))
            g = N.Grads()
            g.dL_dout_color, g.dL_dout_normal, g.dL_dout_depth = N.ptr(gc), N.ptr(gn), N.ptr(gd)
            g.dL_dout_opacity, g.dL_dout_feature = N.ptr(go), N.ptr(gf)
            g.dL_dmeans2D, g.dL_dconic, g.dL_dopacity = dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr()
            g.dL_dcolors, g.dL_dfeatures = dL_dcolors.data_ptr(), N.ptr(dL_dfeatures)
            g.dL_dnormal, g.dL_ddepth, g.dL_dmeans3D = dL_dnormal.data_ptr(), dL_ddepth.data_ptr(), dL_dmeans3D.data_ptr()
            g.dL_dcov3D, g.dL_dsh, g.dL_dscales = dL_dcov3D.data_ptr(), N.ptr(dL_dsh), dL_dscales.data_ptr()
            g.dL_drotations = dL_drotations.data_ptr()
            if N.CLEAR_HINT:
                g.clear_base, g.clear_bytes = gblob.data_ptr(), gblob.numel() * 4
            if out_weights is not None:   # (extension: the per-Gaussian backward then walks the blended Gaussians only, from ~0.4 M surfels on)
                g.out_weights = N.ptr(N.f32c(out_weights, dev))
            rad = radii.contiguous()
            # scratch: one packed gradient row per Gaussian (include/svgir_raster.h)
            nscr = N.lib.svgir_backward_scratch_bytes(N.RGSS, P, binningBuffer.numel(), W, H, S, 0)
            scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
            N.guarded(dev, "backward", N.lib.svgir_backward, p, g, int(R), rad.data_ptr(), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                                         binningBuffer.numel(), imageBuffer.data_ptr(), scratch.data_ptr(), nscr,
                                         N.stream_ptr(dev))
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dfeatures, dL_dcov3D, dL_dsh, dL_dscales,
                dL_drotations)

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        P = means3D.size(0)
        present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
        if P != 0:
            m3, vm, pm = (N.f32c(t, means3D.device) for t in (means3D, viewmatrix, projmatrix))
            N.guarded(means3D.device, "mark_visible", N.lib.svgir_mark_visible, N.RGSS, P, m3.data_ptr(), vm.data_ptr(), pm.data_ptr(),
                                             present.data_ptr(), N.stream_ptr(means3D.device))
        return present


_C = _CBinding()


def rasterize_gaussians(
        means3D,
        means2D,
        features,
        sh,
        colors_precomp,
        opacities,
        scales,
        rotations,
        cov3Ds_precomp,
        raster_settings,
):
    # (every channel width the reference accepts runs through the C ABI: widths without a specialised composite kernel
    #  use the run-time-width kernels of csrc/render_generic.hip)
    return _RasterizeGaussians.apply(
        means3D, means2D, features, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, features, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        args = (
            raster_settings.bg, means3D, features, colors_precomp, opacities, scales, rotations,
            raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
            raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.cx, raster_settings.cy,
            raster_settings.image_height, raster_settings.image_width, sh, raster_settings.sh_degree,
            raster_settings.campos, raster_settings.prefiltered, raster_settings.computer_pseudo_normal,
            raster_settings.debug)
        fwd_only = not any(ctx.needs_input_grad)   # (evaluation / no_grad: the composite keeps no blend states for a backward)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                out = _C.rasterize_gaussians(*args, forward_only=fwd_only)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            out = _C.rasterize_gaussians(*args, forward_only=fwd_only)
        (num_rendered, num_contrib, color, normal, opacity, depth, feature, pseudo_normal, surface_xyz, weights, radii,
         geomBuffer, binningBuffer, imgBuffer) = out
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, features, scales, rotations, cov3Ds_precomp, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer, weights)
        ctx.mark_non_differentiable(num_contrib, pseudo_normal, surface_xyz, weights, radii)
        return (num_rendered, num_contrib, color, normal, opacity, depth, feature, pseudo_normal, surface_xyz, weights,
                radii)

    @staticmethod
    def backward(ctx, grad_num_rendered, grad_num_contrib, grad_out_color, grad_out_normal, grad_out_opacity,
                 grad_out_depth, grad_out_feature, grad_out_pseudo_normal, grad_out_surface_xyz, grad_out_weights,
                 grad_out_radii):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, features, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, weights) = ctx.saved_tensors
        H, W = raster_settings.image_height, raster_settings.image_width

        def _g(g, ch):  # autograd hands None for outputs that did not take part in the loss: an empty tensor = all zero for the library
            return g if g is not None else torch.empty(0, dtype=torch.float32, device=means3D.device)

        args = (raster_settings.bg, means3D, features, radii, colors_precomp, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy, _g(grad_out_color, 3),
                _g(grad_out_normal, 3), _g(grad_out_opacity, 1), _g(grad_out_depth, 1),
                _g(grad_out_feature, features.size(1) if features.dim() == 2 else 0), sh, raster_settings.sh_degree,
                raster_settings.campos, geomBuffer, num_rendered, binningBuffer, imgBuffer,
                raster_settings.backward_geometry, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                res = _C.rasterize_gaussians_backward(*args, out_weights=weights)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            res = _C.rasterize_gaussians_backward(*args, out_weights=weights)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_features, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations) = res

        def _m(g, like):
            return g if (like is not None and like.numel() != 0) else None

        grads = (
            grad_means3D,
            grad_means2D,
            _m(grad_features, features),
            _m(grad_sh, sh),
            _m(grad_colors_precomp, colors_precomp),
            grad_opacities,
            _m(grad_scales, scales),
            _m(grad_rotations, rotations),
            _m(grad_cov3Ds_precomp, cov3Ds_precomp),
            None,
        )
        return grads


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    cx: float
    cy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    backward_geometry: bool
    computer_pseudo_normal: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, features=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        if features is None:
            features = torch.empty_like(means3D[..., :0])

        return rasterize_gaussians(
            means3D, means2D, features, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
            raster_settings)
