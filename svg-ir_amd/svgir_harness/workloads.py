"""Bench workloads of the rows SURVEY 8 marks "next" (f2, f3, f4), built from the drop-in modules only:

* `TrainStep`   -- one stage-2 training iteration on the cfg3_train scene, the sequence of the reference's
                   train.py:133-134 + gaussian_renderer/svgss.py:15-262 (render -> unpack -> L1 + SSIM) + loss.backward() +
                   scene/gaussian_model.py:1270-1276 (add_densification_stats) + :775-813 (GaussianModel.step: NaN scrub, Adam,
                   zero_grad): shade -> rasterize -> unpack -> L1 + SSIM -> backward -> fused Adam.
* `TracerCache` -- the visibility / radiance cache producers on the cfg3 GEOMETRY (not a synthetic shell scene):
                   scene/gaussian_model.py:435-466 `update_visibility` and :469-522 `update_radiace`, chunk loop included
                   (P // ((sample_num - 1) // 24 + 1) surfels per chunk): BVH builds + trace_visibility +
                   render_radiance_with_sampling_SH, P x sample_num rays each.

Both are test / bench scaffolding around product entry points; nothing here is imported by the product modules."""
import math

import numpy as np
import torch

from . import runner, scenes, shade_inputs


def _geo_normals(rotations):
    q = torch.nn.functional.normalize(rotations, dim=-1)
    r, x, y, z = q.unbind(-1)   # local z axis of the surfel = geometric normal
    return torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], dim=-1)


def inverse_covariance(scales, rotations):
    """GaussianModel.get_inverse_covariance (scene/gaussian_model.py:379-382): strip_symmetric(L L^T), L = R diag(1 / s)."""
    q = torch.nn.functional.normalize(rotations, dim=-1)
    r, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    L = R * (1.0 / scales)[:, None, :]
    M = L @ L.transpose(1, 2)
    return torch.stack([M[:, 0, 0], M[:, 0, 1], M[:, 0, 2], M[:, 1, 1], M[:, 1, 2], M[:, 2, 2]], dim=-1).contiguous()


class TrainStep:
    """One optimisation step of stage 2 on the cfg3_train scene (BASELINE.json configs[2]), everything on the device."""

    LAMBDA_DSSIM = 0.2   # arguments/__init__.py: lambda_dssim

    def __init__(self, dev, seed=5, name="cfg3_train", Ns=64, fused=True, scene=None, radiance_grad=False):
        from gaussian_renderer import shading
        from gaussian_renderer.svgss_rasterization import GaussianRasterizer
        from . import losses, optim, render_view
        self.shading, self.render_view, self.losses, self.optim = shading, render_view, losses, optim
        self.GaussianRasterizer = GaussianRasterizer
        self.dev, self.name, self.Ns, self.fused = dev, name, Ns, fused
        # The reference shades with pc.get_radiances = nan_to_num(_radiances.detach() * _radiance_ratio, nan=0) (gaussian_model.py:323-324):
        # the cache sits in the optimizer (group 'radiances', :527) but never receives a gradient; what learns is the scalar ratio
        # (lr 0.01, :526-528).  radiance_grad=True is the earlier rounds' form (the [P, Ns, 3] cache itself is a differentiated leaf).
        self.radiance_grad = radiance_grad
        sc = self.sc = scenes.make(name) if scene is None else scene
        sct = self.sct = runner.to_torch(sc, dev)
        self.P, self.W, self.H = int(sc["means3D"].shape[0]), sc["W"], sc["H"]
        self.S, self.VS = 4, 52
        self.st = runner.settings(sct, "svgss")
        geo_n = _geo_normals(sct["rotations"])
        sd = shade_inputs.make(self.P, Ns, seed=seed, device=dev, geo_normals=geo_n, with_dirs=False)
        self.visibility = sd["visibility"]
        self.geo_n = torch.nn.functional.normalize(geo_n, dim=-1)
        # the parameter block (one tensor per optimizer group, like GaussianModel.training_setup / training_setup_pbr)
        P = self.P
        par = {"xyz": sct["means3D"], "scaling": sct["scales"], "rotation": sct["rotations"], "opacity": sct["opacities"],
               "shs": sct["shs"], "base_color": sd["base_color"], "roughness": sd["roughness"], "normal": sd["normals"],
               "radiance": sd["radiance"], "env": sd["env"]}
        if not radiance_grad:
            par["radiance_ratio"] = torch.ones((), device=dev)
        self.params = {k: torch.nn.Parameter(v.detach().clone().contiguous()) for k, v in par.items()}
        lrs = {"xyz": 1.6e-6, "scaling": 5e-5, "rotation": 1e-5, "opacity": 5e-4, "shs": 2.5e-5, "base_color": 1e-4,
               "roughness": 1e-4, "normal": 1e-5, "radiance": 1e-4, "env": 1e-4, "radiance_ratio": 0.01}
        self.optimizer = optim.FusedAdam([{"params": [self.params[k]], "lr": lrs[k], "name": k} for k in par], lr=0.0, eps=1e-15)
        self.nan_values = {"base_color": 0.0, "roughness": 0.0, "normal": 0.0, "xyz": 0.0, "scaling": 0.0, "rotation": 0.0,
                           "opacity": 0.0, "shs": 0.0}
        # elements the optimizer actually updates per step (a group without a gradient is skipped, as torch.optim.Adam does)
        self.n_param_elems = sum(p.numel() for k, p in self.params.items() if radiance_grad or k != "radiance")
        self.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        self.weights_accum = torch.zeros(P, 1, device=dev)
        self.denom = torch.zeros(P, 1, device=dev)
        self.gt = torch.rand(3, self.H, self.W, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
        self.light = shade_inputs.Light(self.params["env"])
        self.marks = None   # optional: list of (name, torch.cuda.Event) filled by step() when set to []

    def _mark(self, name):
        if self.marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def step(self, offsets=None, keep_grads=False):
        """One iteration.  `offsets` [P]: the random azimuths of this iteration's incident lattices (drawn here when None);
        keep_grads: leave the parameters' .grad in place behind the optimizer step (tests)."""
        p, st = self.params, self.st
        self._mark("begin")
        campos = st.campos
        viewdirs = torch.nn.functional.normalize(campos[None, :] - p["xyz"].detach(), dim=-1)
        offs = torch.empty(self.P, device=self.dev).uniform_(0.0, 2 * math.pi) if offsets is None else offsets   # sample_incident_rays(training): random azimuths
        lattice = self.shading.FibonacciLattice(self.geo_n, self.Ns, offs)
        self.last_offsets = offs
        means2D = torch.zeros_like(p["xyz"], requires_grad=True)
        ratio = None if self.radiance_grad else p["radiance_ratio"]
        radiance = p["radiance"] if self.radiance_grad else p["radiance"].detach()
        if self.fused:   # the shading runs inside the rasterizer calls, for the surfels this view reads (include/svgir_raster.h: svgir_fused_shade)
            rendered, _ = self.shading.render_shaded(st, p["xyz"], means2D, p["opacity"], p["shs"], p["scaling"], p["rotation"],
                                                     p["base_color"], p["roughness"], p["normal"], viewdirs, radiance, self.light,
                                                     self.visibility, lattice, None, True, radiance_ratio=ratio)
        else:
            feats, vfeats, _ = self.shading.shade_and_pack(p["base_color"], p["roughness"], p["normal"], viewdirs, radiance,
                                                          self.light, self.visibility, lattice, None, st.viewmatrix, True,
                                                          radiance_ratio=ratio)
            self._mark("shade_fwd")
            rast = self.GaussianRasterizer(st)
            rendered = rast(means3D=p["xyz"], means2D=means2D, opacities=p["opacity"], shs=p["shs"], scales=p["scaling"],
                            rotations=p["rotation"], features=feats, vfeatures=vfeats)
        self._mark("raster_fwd")
        res = self.render_view.unpack(rendered, st.bg, True)
        loss = self.losses.l1_ssim_loss(res["pbr"], self.gt, self.LAMBDA_DSSIM)   # (1 - lambda) L1 + lambda (1 - SSIM), one node
        self._mark("unpack_loss_fwd")
        loss.backward()
        self._mark("backward")
        self.optim.add_densification_stats(means2D.grad, res["visibility_filter"], res["weights"], self.weights_accum,
                                           self.xyz_gradient_accum, self.denom)
        self.last = dict(means2D_grad=means2D.grad, visibility_filter=res["visibility_filter"], weights=res["weights"])
        self.optimizer.step(nan_values=self.nan_values, zero_grad=not keep_grads)
        self._mark("stats_adam")
        return int(rendered[0]), res["pbr"], loss

    def stage_table(self, steps):
        """Average milliseconds per phase over `steps` steps (HIP events on the current stream)."""
        acc = {}
        for _ in range(steps):
            self.marks = []
            self.step()
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
                acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
        self.marks = None
        return {k: v / steps for k, v in acc.items()}


class TracerCache:
    """update_visibility + update_radiace of the reference on the cfg3 geometry, through the drop-in RayTracer / Renderer."""

    def __init__(self, dev, name="cfg3_train", sample_num=64, P=None):
        from gaussian_renderer import shading
        from pbgi.renderer import Renderer
        from submodules.bvh import RayTracer
        self.shading, self.Renderer, self.RayTracer = shading, Renderer, RayTracer
        self.dev, self.sample_num = dev, sample_num
        sc = scenes.make(name) if P is None else scenes.make(name, P=P)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
        self.xyz = t(sc["means3D"])
        scales = t(sc["scales"]).clone()
        scales[:, 2] = 1e-3     # flat surfels (the reference's stage-2 models keep the third axis degenerate)
        self.scales = scales
        self.rot = torch.nn.functional.normalize(t(sc["rotations"]), dim=-1)
        self.opacity = t(sc["opacities"])
        self.shs = t(sc["shs"])
        self.normals = torch.nn.functional.normalize(_geo_normals(self.rot), dim=-1)
        self.cov_inv = inverse_covariance(self.scales, self.rot)
        self.P = int(self.xyz.shape[0])
        self.rays = self.P * sample_num

    def update_visibility(self):
        """scene/gaussian_model.py:435-466."""
        rt = self.RayTracer(self.xyz, self.scales, self.rot)
        P, Ns = self.P, self.sample_num
        chunk = P // ((Ns - 1) // 24 + 1)
        out = []
        for off in range(0, P, chunk):
            dirs, areas = self.shading.sample_incident_rays(self.normals[off:off + chunk], False, Ns)
            res = rt.trace_visibility(self.xyz[off:off + chunk, None].expand_as(dirs), dirs, self.xyz, self.cov_inv,
                                      self.opacity[:, 0], self.normals)
            out.append(res["visibility"])
        return torch.cat(out, dim=0)

    def update_radiance(self):
        """scene/gaussian_model.py:469-522."""
        R = self.Renderer()
        R.set_proxy(self.xyz, self.scales, self.rot, self.normals, self.opacity, self.shs)
        R.build_bvh()
        P, Ns = self.P, self.sample_num
        chunk = P // ((Ns - 1) // 24 + 1)
        rad, vis, idx = [], [], []
        for off in range(0, P, chunk):
            dirs, areas = self.shading.sample_incident_rays(self.normals[off:off + chunk], True, Ns)
            r, v, h, uv = R.render_radiance_with_sampling_SH(self.xyz[off:off + chunk], dirs, self.cov_inv, Ns)
            rad.append(r); vis.append(v); idx.append(h)
        return torch.cat(rad, dim=0), torch.cat(vis, dim=0), torch.cat(idx, dim=0)

    def step(self):
        vis = self.update_visibility()
        rad, vis2, idx = self.update_radiance()
        return vis, rad, idx

    def timed(self, fn, n=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, out
