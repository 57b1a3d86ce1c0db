"""Synthetic scenes, cameras and the view-parallel driver used by bench.py and the tests.

Counterpart of the reference's callers (gaussian_renderer/render.py, gaussian_renderer/svgss.py, scene/cameras.py):
only what is needed to feed the rasterizer with inputs of the reference's shapes and conventions.
"""
