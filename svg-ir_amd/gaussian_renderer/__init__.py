"""MI355X-native drop-in for the rasterizer bindings of learner-shx/SVG-IR.

Import layout mirrors the reference package (`gaussian_renderer.svgss_rasterization`,
`gaussian_renderer.rgss_rasterization`); only the rasterizer boundary is provided here -- render orchestration,
losses and scene handling stay with the caller.
"""
from . import rgss_rasterization, svgss_rasterization  # noqa: F401
