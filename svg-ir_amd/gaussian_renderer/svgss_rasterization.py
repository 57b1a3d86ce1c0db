"""Drop-in for the reference's `gaussian_renderer.svgss_rasterization` (stage-2, spatially-varying surfels).

Same public names, argument order, return tuples and gradient tuples as
/root/reference/gaussian_renderer/svgss_rasterization.py:
  * GaussianRasterizationSettings (15 fields, :331-346)
  * GaussianRasterizer(raster_settings).forward(...) -> 9-tuple (:365-411, :183); .markVisible (:354-363)
  * _RasterizeGaussians autograd.Function: grads for (means3D, means2D, features, vfeatures, sh, colors_precomp,
    opacities, scales, rotations, cov3Ds_precomp, viewmatrix, projmatrix, campos, None) (:293-308)
  * `_C` with rasterize_gaussians / rasterize_gaussians_backward / mark_visible in the pybind argument order of
    svgss_rasterization/rasterize_points.h:18-82 (24 args -> 12-tuple, 31 args -> 13-tuple).
The compute runs in libsvgir_raster.so (hand-written HIP for gfx950) through the C ABI of include/svgir_raster.h.
"""
from typing import NamedTuple

import ctypes as C

import torch
import torch.nn as nn

from . import _native as N


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class _CBinding:
    """Same three entry points as the reference's pybind module (svgss_rasterization/ext.cpp:15-19)."""

    @staticmethod
    def rasterize_gaussians(*args, **kw):
        """The reference's `_C.rasterize_gaussians` (24 positional arguments, rasterize_points.h:18-46); keyword-only extensions:
        `features_ready`, `shade`, `forward_only` (see `_forward_steps`)."""
        return N.run_forward(_CBinding._forward_steps(*args, **kw))

    @staticmethod
    def rasterize_gaussians_batch(calls, device, streams):
        """Extension: `calls` = [(args, kwargs)] of rasterize_gaussians, one view each, launched with ONE svgir_forward_batch --
        view v on streams[v], all views in flight before the first instance count is awaited (one host thread).  Returns the
        list of 12-tuples.  The caller orders `streams` against the producers / consumers of the tensors."""
        return N.run_forward_batch([(lambda a=a, k=k: _CBinding._forward_steps(*a, **k)) for a, k in calls], device, streams)

    @staticmethod
    def _forward_steps(background, means3D, features, vfeatures, colors, opacity, scales, rotations,
                       scale_modifier, cov3D_precomp, viewmatrix, projmatrix, prcppoint, patchbbox, tan_fovx,
                       tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug, config, *,
                       features_ready=None, shade=None, forward_only=False):
        """`features_ready` (extension, keyword only): a torch.cuda.Event recorded on the stream that is still producing
        `features` / `vfeatures` (the shading kernels on a side stream); only the composite kernel waits for it, so the
        shading of a view overlaps its binning.  The caller keeps the tensors alive across streams (`record_stream`).
        `shade` (extension, keyword only): a `_native.FusedShade` -- the library shades the surfels this view's composite reads
        and WRITES `features` / `vfeatures` (pass uninitialised [P,S] / [P,VS] buffers; gaussian_renderer/shading.py).
        `forward_only` (extension, keyword only): no backward will follow -- the composite keeps no blend states."""
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:65-67
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("svgss rasterizer: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
        P = means3D.size(0)
        S = features.size(1) if features.dim() == 2 else 0
        VS = vfeatures.size(1) if vfeatures.dim() == 2 else 0
        H, W = int(image_height), int(image_width)
        f32 = dict(dtype=torch.float32, device=dev)
        if P == 0:  # nothing is launched: the reference returns its zero-initialised outputs (rasterize_points.cu:100)
            torch_empty = torch.zeros
        else:
            def torch_empty(shape, **kw):  # every element is written by the library
                return N.out_tensor(shape, kw["dtype"], kw["device"])
        out_color = torch_empty((3, H, W), **f32)
        out_normal = torch_empty((3, H, W), **f32)
        out_depth = torch_empty((1, H, W), **f32)
        out_opac = torch_empty((1, H, W), **f32)
        out_feature = torch_empty((S, H, W), **f32)
        out_vfeature = torch_empty((VS // 4, H, W), **f32)
        out_weights = torch_empty((P, 1), **f32)
        radii = torch_empty((P,), dtype=torch.int32, device=dev)
        blobs = N.BlobAllocator(dev)
        rendered = 0
        if P != 0:
            keep = [N.f32c(t, dev) for t in (background, means3D, sh, colors, features, vfeatures, opacity, scales,
                                              rotations, cov3D_precomp, viewmatrix, projmatrix, campos, prcppoint,
                                              patchbbox, config)]
            (bg, m3, shc, col, fe, vf, op, sc, ro, cv, vm, pm, cp, pr, pb, cfg) = keep
            p = N.new_params()
            p.variant, p.P, p.S, p.VS, p.D, p.W, p.H = N.SVGSS, P, S, VS, int(degree), W, H
            p.M = shc.size(1) if (shc is not None and shc.numel() != 0) else 0
            p.background, p.means3D, p.shs, p.colors_precomp = N.ptr(bg), N.ptr(m3), N.ptr(shc), N.ptr(col)
            p.features, p.vfeatures, p.opacities = N.ptr(fe), N.ptr(vf), N.ptr(op)
            p.scales, p.rotations, p.cov3D_precomp = N.ptr(sc), N.ptr(ro), N.ptr(cv)
            p.viewmatrix, p.projmatrix, p.cam_pos = N.ptr(vm), N.ptr(pm), N.ptr(cp)
            p.prcppoint, p.patchbbox = N.ptr(pr), N.ptr(pb)
            p.config, p.config_len = N.ptr(cfg), (cfg.numel() if cfg is not None else 0)
            p.scale_modifier, p.tan_fovx, p.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
            p.cx, p.cy = W / 2.0, H / 2.0
            p.prefiltered, p.debug = int(bool(prefiltered)), int(bool(debug))
            o = N.Outputs()
            o.out_color, o.out_normal, o.out_depth, o.out_opacity = (out_color.data_ptr(), out_normal.data_ptr(),
                                                                     out_depth.data_ptr(), out_opac.data_ptr())
            o.out_feature, o.out_vfeature = N.ptr(out_feature), N.ptr(out_vfeature)
            o.out_weights, o.radii = out_weights.data_ptr(), radii.data_ptr()
            if features_ready is not None:   # (a struct field of this call: nothing survives if anything below raises)
                p.features_ready = features_ready.cuda_event
            if shade is not None:
                p.shade = C.addressof(shade)
            p.forward_only = int(bool(forward_only))   # evaluation: no blend states are kept for a backward
            rendered = yield (dev, p, o, blobs)          # <- svgir_forward / svgir_forward_batch (gaussian_renderer/_native.py)
        # note: C++ order is (..., depth, opac, ...) -- the Python wrapper re-orders (svgss_rasterization.py:175,183)
        return (rendered, out_color, out_normal, out_depth, out_opac, out_feature, out_vfeature, out_weights, radii,
                *blobs.take("geom", "binning", "image"))

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, features, vfeatures, radii, colors, scales, rotations,
                                     scale_modifier, cov3D_precomp, viewmatrix, projmatrix, prcppoint, patchbbox,
                                     tan_fovx, tan_fovy, dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opac,
                                     dL_dout_feature, dL_dout_vfeature, sh, degree, campos, geomBuffer, R,
                                     binningBuffer, imageBuffer, debug, config, *, shade=None, shade_grads=None, out_weights=None,
                                     scratch_feature_grads=False):
        """`shade` / `shade_grads` (extension, keyword only): the `_native.FusedShade` of the forward and a dict of the shading's
        gradient outputs + `out_weights` (+ optional `dL_dreduced`), written by svgir_backward (gaussian_renderer/shading.py).
        `out_weights` (extension, keyword only): the forward's weights [P,1] -- the per-Gaussian kernels behind the composite then walk
        the blended Gaussians only (all others have zero gradients); worth it from a few hundred thousand surfels on.
        `scratch_feature_grads` (with `shade`): dL_dfeatures / dL_dvfeatures are intermediates of the fused call -- written for the blended
        surfels, read back by the shading's backward for exactly those -- so they need no zero rows: they are taken out of the cleared
        allocation (45 MB less to clear at P = 200 k); the rows of unblended surfels in the two returned tensors are then UNDEFINED."""
        dev = means3D.device
        P = means3D.size(0)
        S = features.size(1) if features.dim() == 2 else 0
        VS = vfeatures.size(1) if vfeatures.dim() == 2 else 0
        # (an upstream gradient may be an EMPTY tensor = all zero: an output that took no part in the loss; nothing is read for it)
        _gs = [t for t in (dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opac, dL_dout_feature, dL_dout_vfeature) if t is not None and t.numel()]
        if not _gs:
            raise RuntimeError("rasterize_gaussians_backward: every upstream gradient is empty")
        H, W = _gs[0].size(1), _gs[0].size(2)
        M = sh.size(1) if sh.numel() != 0 else 0
        # (fused shading: per-surfel gradient tensors the caller asks for by shape -- shade_grads["_shapes"] -- live in the same
        # allocation, so the composite backward's clearing sweep zeroes them too; svgir_backward then writes the differentiated rows)
        extra = list((shade_grads or {}).pop("_shapes", {}).items()) if shade is not None else []
        scratch_fg = bool(scratch_feature_grads) and shade is not None and P != 0
        views, gblob = N.grad_blob(dev, [(P, 3), (P, 3), (0 if scratch_fg else P, S), (0 if scratch_fg else P, VS), (P, 3), (P, 3), (P, 1),
                                         (P, 2, 2), (P, 1), (P, 6), (P, M, 3), (P, 3), (P, 4), (4, 4), (4, 4), (3,)] +
                                   [tuple(sh) for _, sh in extra], zero=(P == 0))
        (dL_dmeans3D, dL_dmeans2D, dL_dfeatures, dL_dvfeatures, dL_dcolors, dL_dnormal, dL_ddepth, dL_dconic,
         dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dviewmat, dL_dprojmat, dL_dcampos) = views[:16]
        if scratch_fg:   # (outside the cleared region: only the blended surfels' rows are ever written and read)
            dL_dfeatures = N.out_tensor((P, S), torch.float32, dev)      # (NaN-filled under SVGIR_POISON: the tests see a row that is
            dL_dvfeatures = N.out_tensor((P, VS), torch.float32, dev)    # read without having been written)
        for (name, _), v in zip(extra, views[16:]):
            shade_grads[name] = v
        if P != 0:
            keep = [N.f32c(t, dev) for t in (background, means3D, sh, colors, features, vfeatures, scales, rotations,
                                              cov3D_precomp, viewmatrix, projmatrix, campos, prcppoint, patchbbox,
                                              dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opac,
                                              dL_dout_feature, dL_dout_vfeature, config)]
            (bg, m3, shc, col, fe, vf, sc, ro, cv, vm, pm, cp, pr, pb, gc, gn, gd, go, gf, gvf, cfg) = keep
            p = N.new_params()
            p.variant, p.P, p.S, p.VS, p.D, p.M, p.W, p.H = N.SVGSS, P, S, VS, int(degree), M, W, H
            p.background, p.means3D, p.shs, p.colors_precomp = N.ptr(bg), N.ptr(m3), N.ptr(shc), N.ptr(col)
            p.features, p.vfeatures = N.ptr(fe), N.ptr(vf)
            p.scales, p.rotations, p.cov3D_precomp = N.ptr(sc), N.ptr(ro), N.ptr(cv)
            p.viewmatrix, p.projmatrix, p.cam_pos = N.ptr(vm), N.ptr(pm), N.ptr(cp)
            p.prcppoint, p.patchbbox = N.ptr(pr), N.ptr(pb)
            p.config, p.config_len = N.ptr(cfg), (cfg.numel() if cfg is not None else 0)
            p.scale_modifier, p.tan_fovx, p.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
            p.debug = int(bool(debug))
            g = N.Grads()
            g.dL_dout_color, g.dL_dout_normal, g.dL_dout_depth = N.ptr(gc), N.ptr(gn), N.ptr(gd)
            g.dL_dout_opacity, g.dL_dout_feature, g.dL_dout_vfeature = N.ptr(go), N.ptr(gf), N.ptr(gvf)
            g.dL_dmeans2D, g.dL_dconic, g.dL_dopacity = dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr()
            g.dL_dcolors, g.dL_dfeatures, g.dL_dvfeatures = dL_dcolors.data_ptr(), N.ptr(dL_dfeatures), N.ptr(dL_dvfeatures)
            g.dL_dnormal, g.dL_ddepth, g.dL_dmeans3D = dL_dnormal.data_ptr(), dL_ddepth.data_ptr(), dL_dmeans3D.data_ptr()
            g.dL_dcov3D, g.dL_dsh, g.dL_dscales = dL_dcov3D.data_ptr(), N.ptr(dL_dsh), dL_dscales.data_ptr()
            g.dL_drotations = dL_drotations.data_ptr()
            if N.CLEAR_HINT:
                g.clear_base, g.clear_bytes = gblob.data_ptr(), gblob.numel() * 4
            g.dL_dviewmat, g.dL_dprojmat, g.dL_dcampos = dL_dviewmat.data_ptr(), dL_dprojmat.data_ptr(), dL_dcampos.data_ptr()
            if out_weights is not None:
                g.out_weights = N.ptr(N.f32c(out_weights, dev))
            if shade is not None:
                p.shade = C.addressof(shade)
                for k, t in shade_grads.items():
                    setattr(g, k, N.ptr(t))
            rad = radii.contiguous()
            # scratch for the gradient accumulation: one row per (instance, sub-tile) pair that survived this view's cull (the
            # forward read the count back behind its cull), summed per Gaussian
            nscr = N.lib.svgir_backward_scratch_bytes_for(N.SVGSS, P, binningBuffer.numel(), imageBuffer.data_ptr(), W, H, S, VS)
            scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
            N.guarded(dev, "backward", N.lib.svgir_backward, p, g, int(R), rad.data_ptr(), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                                         binningBuffer.numel(), imageBuffer.data_ptr(), scratch.data_ptr(), nscr,
                                         N.stream_ptr(dev))
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dfeatures, dL_dvfeatures, dL_dcov3D, dL_dsh,
                dL_dscales, dL_drotations, dL_dviewmat, dL_dprojmat, dL_dcampos)

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        P = means3D.size(0)
        present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
        if P != 0:
            m3, vm, pm = (N.f32c(t, means3D.device) for t in (means3D, viewmatrix, projmatrix))
            N.guarded(means3D.device, "mark_visible", N.lib.svgir_mark_visible, N.SVGSS, P, m3.data_ptr(), vm.data_ptr(), pm.data_ptr(),
                                             present.data_ptr(), N.stream_ptr(means3D.device))
        return present


_C = _CBinding()


def rasterize_gaussians(
    means3D,
    means2D,
    sh,
    features,
    vfeatures,
    colors_precomp,
    opacities,
    scales,
    rotations,
    cov3Ds_precomp,
    viewmatrix,
    projmatrix,
    campos,
    raster_settings,
):
    # Parameter names are mislabelled in the reference too (Q13); the positional pass-through is what matters.
    # (positionally: sh <- features, features <- vfeatures, vfeatures <- shs; see GaussianRasterizer.forward)
    # (every channel width the reference accepts runs through the C ABI: widths without a specialised composite kernel
    #  use the run-time-width kernels of csrc/render_generic.hip)
    return _RasterizeGaussians.apply(
        means3D, means2D, sh, features, vfeatures, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
        viewmatrix, projmatrix, campos, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, features, vfeatures, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, viewmatrix, projmatrix, campos, raster_settings):
        args = (
            raster_settings.bg, means3D, features, vfeatures, colors_precomp, opacities, scales, rotations,
            raster_settings.scale_modifier, cov3Ds_precomp, viewmatrix, projmatrix, raster_settings.prcppoint,
            raster_settings.patch_bbox, raster_settings.tanfovx, raster_settings.tanfovy,
            raster_settings.image_height, raster_settings.image_width, sh, raster_settings.sh_degree, campos,
            raster_settings.prefiltered, raster_settings.debug, raster_settings.config)
        fwd_only = not any(ctx.needs_input_grad)   # (evaluation / no_grad: the composite keeps no blend states for a backward)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                out = _C.rasterize_gaussians(*args, forward_only=fwd_only)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            out = _C.rasterize_gaussians(*args, forward_only=fwd_only)
        (num_rendered, color, normal, depth, opacity, feature, vfeature, weights, radii, geomBuffer, binningBuffer,
         imgBuffer) = out
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, features, vfeatures, scales, rotations, cov3Ds_precomp, radii,
                              sh, geomBuffer, binningBuffer, imgBuffer, weights)
        ctx.mark_non_differentiable(weights, radii)
        return num_rendered, color, normal, opacity, depth, feature, vfeature, weights, radii

    @staticmethod
    def backward(ctx, grad_num_rendered, grad_out_color, grad_out_normal, grad_out_opacity, grad_out_depth,
                 grad_out_feature, grad_out_vfeature, grad_out_weights, grad_out_radii):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, features, vfeatures, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
         binningBuffer, imgBuffer, weights) = ctx.saved_tensors
        H, W = raster_settings.image_height, raster_settings.image_width

        def _g(g, ch):  # autograd hands None for outputs that did not take part in the loss: an empty tensor = all zero for the library
            return g if g is not None else torch.empty(0, dtype=torch.float32, device=means3D.device)

        args = (raster_settings.bg, means3D, features, vfeatures, radii, colors_precomp, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.prcppoint, raster_settings.patch_bbox,
                raster_settings.tanfovx, raster_settings.tanfovy, _g(grad_out_color, 3), _g(grad_out_normal, 3),
                _g(grad_out_depth, 1), _g(grad_out_opacity, 1),
                _g(grad_out_feature, features.size(1) if features.dim() == 2 else 0),
                _g(grad_out_vfeature, (vfeatures.size(1) if vfeatures.dim() == 2 else 0) // 4),
                sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered, binningBuffer,
                imgBuffer, raster_settings.debug, raster_settings.config)
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
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_features, grad_vfeatures,
         grad_cov3Ds_precomp, grad_sh, grad_scales, grad_rotations, grad_viewmat, grad_projmat, grad_campos) = res

        def _m(g, like):  # grads of inputs that were passed as empty placeholders
            return g if (like is not None and like.numel() != 0) else None

        grads = (
            grad_means3D,
            grad_means2D,
            _m(grad_features, features),
            _m(grad_vfeatures, vfeatures),
            _m(grad_sh, sh),
            _m(grad_colors_precomp, colors_precomp),
            grad_opacities,
            _m(grad_scales, scales),
            _m(grad_rotations, rotations),
            _m(grad_cov3Ds_precomp, cov3Ds_precomp),
            grad_viewmat,
            grad_projmat,
            grad_campos,
            None,
        )
        return grads


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    patch_bbox: torch.Tensor
    prcppoint: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    config: torch.Tensor


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
                cov3D_precomp=None, features=None, vfeatures=None):
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
        if vfeatures is None:
            vfeatures = torch.empty_like(means3D[..., :0])

        # Invoke the HIP rasterization routine
        return rasterize_gaussians(
            means3D, means2D, features, vfeatures, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
            raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.campos, raster_settings)
