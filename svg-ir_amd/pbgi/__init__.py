"""Drop-in for the hot part of the reference's `pbgi` package (point-based GI): the linear BVH over the surfels and the
closest-hit radiance tracer behind `GaussianModel.update_radiace` (scene/gaussian_model.py:469-522).  The slang kernels of
pbgi/bvhworkers/ are replaced by HIP kernels behind the C ABI (svg-ir_amd/csrc/pbgi.hip); see `renderer.Renderer` and
`bvhhelpers.get_gs_bvh`.  The rest of the reference's pbgi renderer (index-buffer irradiance, mesh paths) is outside the
scope table of SURVEY 8."""
