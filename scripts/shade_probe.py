"""GPU probe: timing of the fused shading kernels (forward / backward) vs the HBM roofline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from gaussian_renderer import shading
from svgir_harness import shade_inputs

dev = torch.device("cuda:0")
for P, Ns in ((200000, 64), (200000, 384)):
    d = shade_inputs.make(P, Ns, seed=2, device=dev)
    light = shade_inputs.Light(d["env"])
    vm = torch.eye(4, device=dev)
    def fwd():
        return shading.shade_and_pack(d["base_color"], d["roughness"], d["normals"], d["viewdirs"], d["radiance"], light,
                                      d["visibility"], d["dirs"], d["areas"], vm, Ns == 64)
    for _ in range(3): fwd()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fwd()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = P * Ns * 32 + P * (12 + 4 + 12 + 3 + 70 + 56) * 4
    print(f"shade fwd P={P} Ns={Ns}: {ms:.3f} ms  {byts/ms/1e6:.0f} GB/s ({byts/ms/1e6/8000*100:.1f}% of 8 TB/s)")
    if Ns == 64:
        leaves = {k: d[k].clone().requires_grad_(True) for k in ("base_color", "roughness", "normals", "radiance")}
        env = d["env"].clone().requires_grad_(True)
        def fb():
            f, vf, red = shading.shade_and_pack(leaves["base_color"], leaves["roughness"], leaves["normals"], d["viewdirs"],
                                                leaves["radiance"], shade_inputs.Light(env), d["visibility"], d["dirs"], d["areas"], vm, True)
            (f.sum() + vf.sum()).backward()
        for _ in range(3): fb()
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fb()
        e1.record(); torch.cuda.synchronize()
        print(f"shade fwd+bwd (autograd path) {e0.elapsed_time(e1)/10:.3f} ms")
