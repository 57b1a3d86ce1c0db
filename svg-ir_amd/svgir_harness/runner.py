"""Feeds a synthetic scene (scenes.py) through the drop-in bindings, the way the reference's render_view()
functions do (gaussian_renderer/render.py:50-105, gaussian_renderer/svgss.py:51-182)."""
import numpy as np
import torch


def to_torch(scene, device):
    out = {}
    for k, v in scene.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
        else:
            out[k] = v
    return out


def settings(sc, variant, debug=False):
    if variant == "svgss":
        from gaussian_renderer.svgss_rasterization import GaussianRasterizationSettings
        return GaussianRasterizationSettings(
            image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=sc["bg"],
            scale_modifier=sc["scale_modifier"], viewmatrix=sc["viewmatrix"], projmatrix=sc["projmatrix"],
            patch_bbox=sc["patch_bbox"], prcppoint=sc["prcppoint"], sh_degree=sc["sh_degree"], campos=sc["campos"],
            prefiltered=False, debug=debug, config=sc["config"])
    from gaussian_renderer.rgss_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], cx=sc["cx"],
        cy=sc["cy"], bg=sc["bg"], scale_modifier=sc["scale_modifier"], viewmatrix=sc["viewmatrix"],
        projmatrix=sc["projmatrix"], sh_degree=sc["sh_degree"], campos=sc["campos"], prefiltered=False,
        backward_geometry=bool(sc.get("backward_geometry", True)),
        computer_pseudo_normal=bool(sc.get("computer_pseudo_normal", False)), debug=debug)


LEAVES = ("means3D", "shs", "opacities", "scales", "rotations", "features", "vfeatures")


def render(sc, variant, requires_grad=False, debug=False):
    """Returns (outputs dict, leaves dict).  `sc` is a to_torch() scene."""
    if variant == "svgss":
        from gaussian_renderer.svgss_rasterization import GaussianRasterizer
    else:
        from gaussian_renderer.rgss_rasterization import GaussianRasterizer
    leaves = {}
    for k in LEAVES:
        if k in sc and (variant == "svgss" or k != "vfeatures"):
            t = sc[k].detach().clone()
            t.requires_grad_(requires_grad)
            leaves[k] = t
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=requires_grad)
    leaves["means2D"] = means2D
    rast = GaussianRasterizer(settings(sc, variant, debug))
    kw = dict(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
              scales=leaves["scales"], rotations=leaves["rotations"], features=leaves["features"])
    if variant == "svgss":
        kw["vfeatures"] = leaves["vfeatures"]
        (R, color, normal, opacity, depth, feature, vfeature, weights, radii) = rast(**kw)
        out = dict(num_rendered=R, color=color, normal=normal, opacity=opacity, depth=depth, feature=feature,
                   vfeature=vfeature, weights=weights, radii=radii)
    else:
        (R, n_contrib, color, normal, opacity, depth, feature, pseudo_normal, surface_xyz, weights, radii) = rast(**kw)
        out = dict(num_rendered=R, n_contrib=n_contrib, color=color, normal=normal, opacity=opacity, depth=depth,
                   feature=feature, pseudo_normal=pseudo_normal, surface_xyz=surface_xyz, weights=weights, radii=radii)
    return out, leaves


def backward(out, grads, variant):
    """Back-propagates sum_k <out_k, grads_k> (grads: numpy or torch CHW arrays keyed like upstream_grads())."""
    dev = out["color"].device
    loss = 0
    for k in ("color", "normal", "depth", "opacity", "feature") + (("vfeature",) if variant == "svgss" else ()):
        g = grads[k]
        if not torch.is_tensor(g):
            g = torch.from_numpy(np.ascontiguousarray(g))
        if g.numel() == 0:
            continue
        loss = loss + (out[k] * g.to(dev)).sum()
    loss.backward()
