"""L1 and SSIM of a rendered image against the ground truth, fused (csrc/loss.hip): the reference's
`F.l1_loss(image, gt)` + `ssim(image, gt)` (gaussian_renderer/svgss.py:281-289, render.py:150-151;
utils/loss_utils.py:21-64), one kernel forward, one backward."""
import ctypes as C

import torch

from gaussian_renderer import _native as N

N.lib.svgir_l1_ssim_partials.restype = C.c_size_t
N.lib.svgir_l1_ssim_partials.argtypes = [C.c_int32] * 3
N.lib.svgir_l1_ssim_forward.restype = C.c_int
N.lib.svgir_l1_ssim_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
N.lib.svgir_l1_ssim_backward.restype = C.c_int
N.lib.svgir_l1_ssim_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p]


class _L1Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt):
        dev = img.device
        if dev.type != "cuda":
            raise RuntimeError("l1_ssim: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
        a, b = N.f32c(img, dev), N.f32c(gt, dev)
        Cc, H, W = a.shape[-3], a.shape[-2], a.shape[-1]
        nblk = N.lib.svgir_l1_ssim_partials(Cc, H, W)
        partial = torch.empty((nblk, 2), dtype=torch.float32, device=dev)
        need = img.requires_grad
        dmaps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=dev) if need else None
        means = torch.empty(2, dtype=torch.float32, device=dev)   # {mean SSIM, mean L1}: reduced on the device
        N.check(N.lib.svgir_l1_ssim_forward(a.data_ptr(), b.data_ptr(), Cc, H, W, partial.data_ptr(), N.ptr(dmaps), means.data_ptr(),
                                            N.stream_ptr(dev)), "l1_ssim forward")
        ctx.save_for_backward(a, b, dmaps)
        return means[1], means[0]   # (l1, ssim)

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        a, b, dmaps = ctx.saved_tensors
        Cc, H, W = a.shape[-3], a.shape[-2], a.shape[-1]
        out = torch.empty_like(a)
        # the upstream scalars stay on the device: a 2-float buffer {g_ssim, g_l1} the kernel reads (no blocking read-back)
        zero = torch.zeros((), dtype=torch.float32, device=a.device)
        gdev = torch.stack([zero if g_ssim is None else g_ssim.to(torch.float32), zero if g_l1 is None else g_l1.to(torch.float32)])
        N.check(N.lib.svgir_l1_ssim_backward(a.data_ptr(), b.data_ptr(), dmaps.data_ptr(), Cc, H, W, 1.0, 1.0, gdev.data_ptr(),
                                             out.data_ptr(), N.stream_ptr(a.device)), "l1_ssim backward")
        return out, None


def l1_ssim(image, gt):
    """(F.l1_loss(image, gt), ssim(image, gt)) of the reference, [C,H,W] images; differentiable in `image`."""
    return _L1Ssim.apply(image, gt)


class _L1SsimLoss(torch.autograd.Function):
    """(1 - lambda) * L1 + lambda * (1 - SSIM) as ONE autograd node: the reference composes it from the two scalars with four
    elementwise launches forward and as many backward (train.py:133-134 / gaussian_renderer/svgss.py:281-289); here the weights go into
    the backward kernel as its two host scalars and the forward is one dot product of the two device means."""

    @staticmethod
    def forward(ctx, img, gt, lam):
        dev = img.device
        if dev.type != "cuda":
            raise RuntimeError("l1_ssim_loss: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
        a, b = N.f32c(img, dev), N.f32c(gt, dev)
        Cc, H, W = a.shape[-3], a.shape[-2], a.shape[-1]
        nblk = N.lib.svgir_l1_ssim_partials(Cc, H, W)
        partial = torch.empty((nblk, 2), dtype=torch.float32, device=dev)
        need = img.requires_grad
        dmaps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=dev) if need else None
        means = torch.empty(2, dtype=torch.float32, device=dev)   # {mean SSIM, mean L1}
        N.check(N.lib.svgir_l1_ssim_forward(a.data_ptr(), b.data_ptr(), Cc, H, W, partial.data_ptr(), N.ptr(dmaps), means.data_ptr(),
                                            N.stream_ptr(dev)), "l1_ssim forward")
        ctx.save_for_backward(a, b, dmaps)
        ctx.lam = float(lam)
        w = _weights(dev, ctx.lam)   # [-lambda, 1 - lambda, lambda]
        return torch.dot(means, w[:2]) + w[2]

    @staticmethod
    def backward(ctx, g):
        a, b, dmaps = ctx.saved_tensors
        Cc, H, W = a.shape[-3], a.shape[-2], a.shape[-1]
        out = torch.empty_like(a)
        gdev = g.to(torch.float32).reshape(1).expand(2).contiguous()   # the upstream scalar stays on the device
        N.check(N.lib.svgir_l1_ssim_backward(a.data_ptr(), b.data_ptr(), dmaps.data_ptr(), Cc, H, W, -ctx.lam, 1.0 - ctx.lam,
                                             gdev.data_ptr(), out.data_ptr(), N.stream_ptr(a.device)), "l1_ssim backward")
        return out, None, None


_W = {}


def _weights(dev, lam):
    key = (dev.index, lam)
    if key not in _W:
        _W[key] = torch.tensor([-lam, 1.0 - lam, lam], dtype=torch.float32, device=dev)
    return _W[key]


def l1_ssim_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda_dssim) * F.l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt)) -- the reference's photometric loss
    (train.py:133-134; arguments/__init__.py: lambda_dssim = 0.2) -- as one autograd node; differentiable in `image`."""
    return _L1SsimLoss.apply(image, gt, float(lambda_dssim))


def ssim(img1, img2, window_size=11, size_average=True):
    """Drop-in for utils/loss_utils.py:33 (window 11, size_average=True -- the only form the reference calls)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("only the reference's call form ssim(img1, img2) is implemented")
    return l1_ssim(img1, img2)[1]
