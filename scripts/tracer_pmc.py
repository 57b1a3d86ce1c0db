"""Driver for PMC passes over the tracer kernels: ONE chunk (P // 3 rows x 64 rays) of each cache producer on the cfg3 geometry."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from svgir_harness import workloads
from gaussian_renderer import shading
from pbgi.renderer import Renderer
from submodules.bvh import RayTracer

dev = torch.device("cuda:0")
tc = workloads.TracerCache(dev)
n = tc.P // 3
rt = RayTracer(tc.xyz, tc.scales, tc.rot)
dirs, _ = shading.sample_incident_rays(tc.normals[:n], False, 64)
rt.trace_visibility(tc.xyz[:n, None].expand_as(dirs), dirs, tc.xyz, tc.cov_inv, tc.opacity[:, 0], tc.normals)
R = Renderer(); R.set_proxy(tc.xyz, tc.scales, tc.rot, tc.normals, tc.opacity, tc.shs); R.build_bvh()
dirs, _ = shading.sample_incident_rays(tc.normals[:n], True, 64)
R.render_radiance_with_sampling_SH(tc.xyz[:n], dirs, tc.cov_inv, 64)
torch.cuda.synchronize()
print("done")
