"""Generates tests/golden/densify.npz by running the REFERENCE's own GaussianModel methods (scene/gaussian_model.py) in the
authoring container on the CPU: `step()` (replace_nangrad_to_zero + torch.optim.Adam.step + zero_grad, :775-813) and
`densify_and_prune` (-> densify_and_clone, densify_and_split, densification_postfix, cat_tensors_to_optimizer, prune_points,
_prune_optimizer, :1020-1268) on a GaussianModel created with __new__ and filled with seeded tensors.  CUDA device arguments
are redirected to the CPU, `torch.normal(mean, std)` is replaced by mean + std * Z with recorded draws Z (the product is fed the
same Z).  The fixture is data only: inputs, the gradients, and the tensors the reference ended up with.

    python scripts/make_golden_densify.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_golden as mg             # noqa: E402
import make_golden_view as mgv       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
# group name, per-Gaussian shape, learning rate (scene/gaussian_model.py:745-768 with the default OptimizationParams scale)
SPEC = [("xyz", (3,), 1.6e-4), ("normal", (12,), 1e-3), ("rotation", (4,), 1e-3), ("scaling", (3,), 5e-3), ("opacity", (1,), 5e-2),
        ("f_dc", (1, 3), 2.5e-3), ("f_rest", (15, 3), 1.25e-4), ("base_color", (12,), 1e-2), ("roughness", (4,), 1e-2),
        ("incidents_dc", (1, 3), 2e-3), ("incidents_rest", (15, 3), 1e-4), ("visibility_dc", (1, 1), 2.5e-3),
        ("visibility_rest", (15, 1), 1.25e-4)]
ATTR = {"xyz": "_xyz", "normal": "_normal", "rotation": "_rotation", "scaling": "_scaling", "opacity": "_opacity", "f_dc": "_shs_dc",
        "f_rest": "_shs_rest", "base_color": "_base_color", "roughness": "_roughness", "incidents_dc": "_incidents_dc",
        "incidents_rest": "_incidents_rest", "visibility_dc": "_visibility_dc", "visibility_rest": "_visibility_rest"}


POISON_ROWS = {5: (float("nan"),) * 3, 6: (float("nan"), None, None), 7: (100.0,) * 3, 8: (None, 100.0, None),
               9: (float("nan"), float("nan"), None), 10: (None, None, float("nan"))}   # row -> per-axis override of the log-scale


def main(fname="densify.npz", poison=False):
    """`poison`: a second fixture (densify_nan.npz) whose `_scaling` has NaN / overflowing rows -- the case the reference's
    get_scaling = nan_to_num(exp(.), nan=1e-6) exists for (scene/gaussian_model.py:270-272) -- all of them selected for
    densification."""
    mgv.setup_reference()
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: mg._Stub("_C")
    from scene.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(4242)
    P = 300
    out = {}
    with mgv.cpu_reference():
        gm = GaussianModel.__new__(GaussianModel)
        gm.use_pbr = True
        gm.percent_dense = 0.01
        gm.setup_functions()
        init = {}
        for name, shp, lr in SPEC:
            t = torch.randn((P,) + shp, generator=g)
            if name == "scaling":   # log-scales around the clone / split limit percent_dense * extent = 0.05
                t = torch.log(torch.exp(torch.empty(P, 3).uniform_(np.log(0.01), np.log(0.2), generator=g)))
            if name == "scaling" and poison:
                for r, ov in POISON_ROWS.items():
                    for ax, v in enumerate(ov):
                        if v is not None:
                            t[r, ax] = v
            if name == "opacity" and poison:
                t[sorted(POISON_ROWS)] = 2.0
            if name == "xyz":
                t = t * 0.8
            init[name] = t.clone()
            setattr(gm, ATTR[name], torch.nn.Parameter(t.clone().requires_grad_(True)))
            out["init_" + name] = t.numpy()
        groups = [{"params": [getattr(gm, ATTR[n])], "lr": lr, "name": n} for n, _, lr in SPEC]
        gm.optimizer = torch.optim.Adam(groups, lr=1e-4, eps=1e-15)
        # ---- three optimisation steps with NaN-poisoned gradients ----
        for it in range(3):
            for name, shp, lr in SPEC:
                gr = torch.randn((P,) + shp, generator=g) * 1e-2
                bad = torch.rand((P,) + shp, generator=g) < 0.01
                gr[bad] = float("nan")
                if it == 1 and name == "incidents_rest":
                    gr = None            # a group without a gradient this iteration
                getattr(gm, ATTR[name]).grad = gr
                out[f"grad{it}_{name}"] = gr.numpy() if gr is not None else np.zeros(0, np.float32)
            gm.step()
        for name, _, _ in SPEC:
            p = getattr(gm, ATTR[name])
            st = gm.optimizer.state[p]
            out["step_" + name] = p.detach().numpy().copy()
            out["step_m_" + name] = st["exp_avg"].numpy().copy()
            out["step_v_" + name] = st["exp_avg_sq"].numpy().copy()
            assert p.grad is None      # zero_grad(set_to_none=True)
        # ---- densify_and_prune ----
        gm.weights_accum = torch.rand(P, 1, generator=g) * 2e-5          # some below the 1e-5 threshold
        gm.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 6e-4
        gm.normal_gradient_accum = torch.rand(P, 1, generator=g) * 2e-4
        gm.denom = torch.randint(0, 3, (P, 1), generator=g).float()     # zeros -> NaN / inf grads
        gm.max_radii2D = torch.rand(P, generator=g) * 30
        if poison:   # the poisoned rows pass the gradient test and survive the weight / screen-size pruning (their opacity: set at init)
            rows = sorted(POISON_ROWS)
            gm.xyz_gradient_accum[rows] = 1e-3
            gm.denom[rows] = 1.0
            gm.weights_accum[rows] = 2e-5
            gm.max_radii2D[rows] = 1.0
        for k in ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D"):
            out["stat_" + k] = getattr(gm, k).numpy().copy()
        Z = []
        real_normal = torch.normal

        def fake_normal(mean, std, **kw):
            z = torch.randn(std.shape, generator=g)
            Z.append(z)
            return mean + std * z
        torch.normal = fake_normal
        try:
            args = dict(max_grad=2e-4, min_opacity=0.05, extent=5.0, max_screen_size=20, max_grad_normal=1.5e-4)
            gm.densify_and_prune(**args)
        finally:
            torch.normal = real_normal
        out["densify_args"] = np.array([args["max_grad"], args["min_opacity"], args["extent"], args["max_screen_size"], args["max_grad_normal"]])
        out["split_z"] = torch.cat(Z, 0).numpy() if Z else np.zeros((0, 3), np.float32)
        for name, _, _ in SPEC:
            p = getattr(gm, ATTR[name])
            st = gm.optimizer.state[p]
            out["dens_" + name] = p.detach().numpy().copy()
            out["dens_m_" + name] = st["exp_avg"].numpy().copy()
            out["dens_v_" + name] = st["exp_avg_sq"].numpy().copy()
        for k in ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D"):
            out["dens_" + k] = getattr(gm, k).numpy().copy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print("wrote", fname, ": P", P, "->", out["dens_xyz"].shape[0], "split draws", out["split_z"].shape,
          "NaN scaling entries after densify:", int(np.isnan(out["dens_scaling"]).sum()))


if __name__ == "__main__":
    main()
    main("densify_nan.npz", poison=True)
