"""GPU probe: the two cache producers on the cfg3 GEOMETRY (svgir_harness.workloads.TracerCache = update_visibility +
update_radiace of the reference, P = 200 k surfels x 64 rays each).  With a -DSVGIR_DEV build (SVGIR_RASTER_LIB) also prints the
radiance tracer's traversal statistics.  usage: tracer_cfg3_probe.py [P] [shell]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussian_renderer import _native
from svgir_harness import workloads

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else None
tc = workloads.TracerCache(dev, P=P)
print(f"cfg3 geometry: P = {tc.P}, {tc.sample_num} rays per surfel, {tc.rays / 1e6:.1f} M rays per tracer")


def stats(tag):
    if not hasattr(_native.lib, "svgir_dev_pbgi_stats"):
        return
    out = (C.c_ulonglong * 24)()
    _native.lib.svgir_dev_pbgi_stats(out)
    q, pops, passed, leaves, acc, rays = (int(out[i]) for i in range(6))
    print(f"  [{tag}] max queries of one ray {int(out[6])}; rays by floor(log2 queries): {[int(out[8 + i]) for i in range(12)]}; "
          f"box tests of rays with >= 1 / 16 / 128 / 1024 queries: {[int(out[20 + i]) for i in range(4)]}")
    if rays:
        print(f"  [{tag}] per ray: {q / rays:.2f} queries, {pops / rays:.1f} popped nodes, {passed / rays:.1f} boxes passed, "
              f"{leaves / rays:.1f} leaves visited, {acc / rays:.2f} leaves accepted")


ms, vis = tc.timed(tc.update_visibility)
print(f"update_visibility (build + 3 chunks of trace_visibility): {ms:9.2f} ms  {tc.rays / ms / 1e3:8.1f} M rays/s   mean visibility {vis.mean().item():.3f}")
stats("warm-up etc."); 
ms, (rad, v2, idx) = tc.timed(tc.update_radiance, n=1)
stats("update_radiance x2")
print(f"update_radiance   (build + 3 chunks of render_radiance):  {ms:9.2f} ms  {tc.rays / ms / 1e3:8.1f} M rays/s   hit fraction {(idx >= 0).float().mean().item():.3f}, "
      f"mean radiance {rad.mean().item():.4f}, mean visibility {v2.mean().item():.3f}")
if len(sys.argv) > 2:   # the synthetic shell scene of tests/pbgi_scene.py for comparison
    from tests import pbgi_scene
    from pbgi.renderer import Renderer
    sc = pbgi_scene.make(P=200000, shells=2000, S=64, seed=13)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    R = Renderer(); R.set_proxy(t(sc["xyz"]), t(sc["scales"]), t(sc["rot"]), t(sc["normals"]), t(sc["opacity"]), t(sc["shs"])); R.build_bvh()
    f = lambda: R.render_radiance_with_sampling_SH(t(sc["xyz"]), t(sc["ray_d"]), t(sc["cov_inv"]), 64)
    ms, out = tc.timed(f, n=1)
    stats("shell scene x2")
    print(f"shell scene render_radiance: {ms:9.2f} ms  {200000 * 64 / ms / 1e3:8.1f} M rays/s   hit fraction {(out[2] >= 0).float().mean().item():.3f}")
