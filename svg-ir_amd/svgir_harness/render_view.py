"""End-to-end svgss view: shading -> packing -> rasterizer -> unpacking, the sequence of the reference's
`render_view` (gaussian_renderer/svgss.py:51-262) with explicit tensors instead of its scene / camera classes.

Only the parts that belong to the hot path and its immediate callers are reproduced: the SV-BRDF shading and packing
(svgss.py:125-166, fused: gaussian_renderer/shading.py), the rasterizer call (:170-182) and the image-space unpacking
(:188-246: division by the rendered opacity, channel split, sRGB, compositing over the background).  The
`depth2normal` pseudo normal and the environment backdrop of the eval branch need the reference's camera class and
are left to the caller."""
import torch

from gaussian_renderer import shading
from gaussian_renderer.svgss_rasterization import GaussianRasterizer

from . import runner


def rgb_to_srgb(img, clip=True):
    """utils/graphics_utils.py:198-221 (torch branch)."""
    out = torch.where(img > 0.0031308, torch.pow(torch.clamp(img, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * img)
    return out.clamp(0.0, 1.0) if clip else out


def unpack(rendered, bg_color, is_training):
    """svgss.py:188-246.  `rendered` = the rasterizer's 9-tuple; returns the result dict of the reference."""
    (num_rendered, image, normal, opacity, depth, feature, vfeature, weights, radii) = rendered
    feature = feature / opacity.clamp_min(1e-5)
    vfeature = vfeature / opacity.clamp_min(1e-5)
    bg = bg_color[:, None, None]

    def over_bg(r):
        return r * opacity + (1 - opacity) * bg

    res = {}
    if is_training:
        vis, local = feature.split([1, 3], dim=0)
        res.update(local_lights=over_bg(rgb_to_srgb(local)), visibility=over_bg(vis))
        pbr, base, shading_normal, rough, diffuse = vfeature.split([3, 3, 3, 1, 3], dim=0)
        res.update(base_color=over_bg(rgb_to_srgb(base)), diffuse=over_bg(rgb_to_srgb(diffuse)), roughness=over_bg(rough))
    else:
        light, local, vis = feature.split([3, 3, 1], dim=0)
        res.update(lights=over_bg(rgb_to_srgb(light)), local_lights=over_bg(rgb_to_srgb(local)), visibility=over_bg(vis))
        pbr, base, shading_normal, rough, direct, indirect = vfeature.split([3, 3, 3, 1, 3, 3], dim=0)
        res.update(base_color=over_bg(rgb_to_srgb(base)), direct=rgb_to_srgb(direct), indirect=rgb_to_srgb(indirect),
                   roughness=over_bg(rough))
    res.update(render=image, depth=depth, pbr=rgb_to_srgb(over_bg(pbr)), normal=shading_normal, opacity=opacity,
               visibility_filter=radii > 0, radii=radii, num_rendered=num_rendered, weights=weights)
    return res


def render_svgss_view(sc, mat, light, is_training):
    """sc: runner.to_torch() scene (geometry + camera); mat: dict with base_color [P,12], roughness [P,4], normals
    [P,4,3], viewdirs [P,3], radiance / dirs [P,Ns,3], visibility / areas [P,Ns,1]; light: DirectLightMap-like (.env).
    Returns (results dict, means2D gradient carrier)."""
    feats, vfeats, _ = shading.shade_and_pack(mat["base_color"], mat["roughness"], mat["normals"], mat["viewdirs"],
                                              mat["radiance"], light, mat["visibility"], mat["dirs"], mat["areas"],
                                              sc["viewmatrix"], is_training)
    means2D = torch.zeros_like(sc["means3D"], requires_grad=torch.is_grad_enabled())
    rast = GaussianRasterizer(runner.settings(sc, "svgss"))
    rendered = rast(means3D=sc["means3D"], means2D=means2D, opacities=sc["opacities"], shs=sc["shs"],
                    scales=sc["scales"], rotations=sc["rotations"], features=feats, vfeatures=vfeats)
    return unpack(rendered, sc["bg"], is_training), means2D
