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


def forward_raw(sc, variant):
    """Forward through the `_C` binding (no autograd) keeping the three state blobs; also decodes the depth-sorted instance
    list and the tile ranges from them (include/svgir_raster.h introspection offsets).  `sc` is a to_torch() scene."""
    from gaussian_renderer import _native as N
    dev = sc["means3D"].device
    st = settings(sc, variant)
    empty = torch.empty(0, dtype=torch.float32, device=dev)
    S = sc["features"].shape[1]
    if variant == "svgss":
        from gaussian_renderer.svgss_rasterization import _C
        VS = sc["vfeatures"].shape[1]
        out = _C.rasterize_gaussians(st.bg, sc["means3D"], sc["features"], sc["vfeatures"], empty, sc["opacities"],
                                     sc["scales"], sc["rotations"], st.scale_modifier, empty, st.viewmatrix,
                                     st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy,
                                     st.image_height, st.image_width, sc["shs"], st.sh_degree, st.campos, False, False,
                                     st.config)
        (R, color, normal, depth, opac, feat, vfeat, weights, radii, gb, bb, ib) = out
        res = dict(num_rendered=R, color=color, normal=normal, depth=depth, opacity=opac, feature=feat, vfeature=vfeat,
                   weights=weights, radii=radii)
    else:
        from gaussian_renderer.rgss_rasterization import _C
        VS = 0
        out = _C.rasterize_gaussians(st.bg, sc["means3D"], sc["features"], empty, sc["opacities"], sc["scales"],
                                     sc["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix,
                                     st.tanfovx, st.tanfovy, st.cx, st.cy, st.image_height, st.image_width, sc["shs"],
                                     st.sh_degree, st.campos, False, False, False)
        (R, ncontrib, color, normal, opac, depth, feat, pn, sx, weights, radii, gb, bb, ib) = out
        res = dict(num_rendered=R, color=color, normal=normal, depth=depth, opacity=opac, feature=feat, weights=weights,
                   radii=radii)
    W, H = st.image_width, st.image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    off = N.lib.svgir_binning_point_list_offset(bb.numel(), ib.data_ptr(), W, H, S, VS)
    res["point_list"] = bb[off:off + 4 * R].view(torch.int32).cpu().numpy().astype("uint32")
    roff = N.lib.svgir_image_ranges_offset(W, H)
    res["ranges"] = ib[roff:roff + 8 * T].view(torch.int32).cpu().numpy().astype("uint32").reshape(T, 2)
    noff = N.lib.svgir_image_ncontrib_offset(W, H)
    res["n_contrib"] = ib[noff:noff + 4 * W * H].view(torch.int32).cpu().numpy().reshape(H, W)
    res["blobs"] = (gb, bb, ib)
    return res
