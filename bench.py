#!/usr/bin/env python3
"""Benchmark of the hot path: Gaussian-surfels/s, forward + backward, one 800x800 view per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3_train|...]

One process per GPU (launched by torchrun for N > 1).  A "step" is one forward+backward pass of the rasterizer
over one synthetic view (inputs resident in HBM before the timed region), through the drop-in binding layer
(`_C.rasterize_gaussians` + `_C.rasterize_gaussians_backward` -> C ABI -> HIP kernels).  For N > 1 the views are
sharded (each rank renders its own camera; weak scaling) and the only collective is one fused all_gather of a
3-float metrics vector per step (RCCL over xGMI).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (backward composite): algorithmic bytes (SURVEY 8d formula with the measured R)
                  / average kernel time from HIP events recorded on the launch stream DURING the timed steps,
                  against the 8 TB/s HBM peak;
  "cpu_baseline": the CPU oracle (port of the reference kernels; the reference has no CPU path) timed on the host
                  cores on a bounded number of the same steps (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(P, R, W, H, S, VS, svgss):
    """per-instance gather G and the forward/backward composite byte models."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    G = 8 + 16 + 12 + 4 + 12 + 40 + (8 if svgss else 0) + 4 * S + 4 * VS
    pix = W * H * (44 + 4 * S + VS)
    fwd = 8 * T + R * (4 + G) + pix + 4 * P
    bwd = 8 * T + R * (4 + G) + 2 * R * (52 + 4 * S + 4 * VS) + pix
    return dict(G=G, fwd=fwd, bwd=bwd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU-oracle steps (0 = as many as fit ~12 s)")
    ap.add_argument("--no-shade", action="store_true", help="svgss workloads: skip the SV-BRDF shading stage")
    ap.add_argument("--samples", type=int, default=0, help="incident samples per surfel (default 64 train / 384 eval)")
    args = ap.parse_args()

    from svgir_harness import cameras, runner, scenes, view_parallel as vp
    rank, world, local = vp.init_from_env()
    assert world == max(1, args.gpus) or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the rasterizer has no CPU path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from gaussian_renderer import _native

    gen_kw = dict(scenes.CONFIGS[args.workload][1])
    variant = gen_kw["variant"]
    sc = scenes.make(args.workload)
    # every rank renders its own view of the (replicated) scene: azimuth 30 + 45*rank degrees
    sc.update(cameras.make_camera(sc["W"], sc["H"], cameras.orbit_eye(4.0, 30.0 + 45.0 * rank, 25.0)))
    grads = scenes.upstream_grads(sc, variant)
    sct = runner.to_torch(sc, dev)
    per_gaussian = {k: sct[k] for k in ("means3D", "scales", "rotations", "opacities", "shs", "features") if k in sct}
    if variant == "svgss":
        per_gaussian["vfeatures"] = sct["vfeatures"]
    vp.broadcast_scene(per_gaussian)  # one-time replication (identical seeds already; exercised for N > 1)
    gt = {k: torch.from_numpy(v).to(dev) for k, v in grads.items()}
    P, W, H = int(sc["means3D"].shape[0]), sc["W"], sc["H"]
    S = int(sc["features"].shape[1])
    VS = int(sc["vfeatures"].shape[1]) if variant == "svgss" else 0

    if variant == "svgss":
        from gaussian_renderer.svgss_rasterization import _C
    else:
        from gaussian_renderer.rgss_rasterization import _C
    empty = torch.empty(0, dtype=torch.float32, device=dev)
    st = runner.settings(sct, variant)

    # svgss workloads: the per-surfel SV-BRDF shading (rendering_equation4 + feature packing) produces the rasterizer's
    # features / vfeatures every step, and its backward consumes the rasterizer's dL_dfeatures / dL_dvfeatures.
    shade = variant == "svgss" and not args.no_shade
    if shade:
        from gaussian_renderer import shading
        from svgir_harness import shade_inputs
        training = S == 4
        Ns = args.samples or (64 if training else 384)
        q = torch.nn.functional.normalize(sct["rotations"], dim=-1)
        r, x, y, z = q.unbind(-1)   # local z axis of the surfel = geometric normal
        geo_n = torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], dim=-1)
        sd = shade_inputs.make(P, Ns, seed=5 + rank, device=dev, geo_normals=geo_n)
        sd["viewdirs"] = torch.nn.functional.normalize(st.campos[None, :] - sct["means3D"], dim=-1)
        leaves = {k: sd[k].clone().requires_grad_(training) for k in ("base_color", "roughness", "normals", "radiance", "env")}
        light = shade_inputs.Light(leaves["env"])

    def step():
        """One forward + backward through the binding layer; returns (R, checksum tensor)."""
        if variant == "svgss":
            feats_in, vfeats_in = sct["features"], sct["vfeatures"]
            if shade:
                for v in leaves.values():
                    v.grad = None
                with torch.set_grad_enabled(training):
                    feats_in, vfeats_in, _ = shading.shade_and_pack(
                        leaves["base_color"], leaves["roughness"], leaves["normals"], sd["viewdirs"], leaves["radiance"],
                        light, sd["visibility"], sd["dirs"], sd["areas"], st.viewmatrix, training)
            out = _C.rasterize_gaussians(st.bg, sct["means3D"], feats_in.detach(), vfeats_in.detach(), empty,
                                         sct["opacities"], sct["scales"], sct["rotations"], st.scale_modifier, empty,
                                         st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx,
                                         st.tanfovy, st.image_height, st.image_width, sct["shs"], st.sh_degree,
                                         st.campos, False, False, st.config)
            (R, color, normal, depth, opac, feat, vfeat, weights, radii, gb, bb, ib) = out
            g = _C.rasterize_gaussians_backward(st.bg, sct["means3D"], feats_in.detach(), vfeats_in.detach(), radii, empty,
                                                sct["scales"], sct["rotations"], st.scale_modifier, empty,
                                                st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx,
                                                st.tanfovy, gt["color"], gt["normal"], gt["depth"], gt["opacity"],
                                                gt["feature"], gt["vfeature"], sct["shs"], st.sh_degree, st.campos,
                                                gb, R, bb, ib, False, st.config)
            if shade and training:
                torch.autograd.backward([feats_in, vfeats_in], [g[4], g[5]])
        else:
            out = _C.rasterize_gaussians(st.bg, sct["means3D"], sct["features"], empty, sct["opacities"],
                                         sct["scales"], sct["rotations"], st.scale_modifier, empty, st.viewmatrix,
                                         st.projmatrix, st.tanfovx, st.tanfovy, st.cx, st.cy, st.image_height,
                                         st.image_width, sct["shs"], st.sh_degree, st.campos, False, False, False)
            (R, ncontrib, color, normal, opac, depth, feat, pn, sx, weights, radii, gb, bb, ib) = out
            g = _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], radii, empty, sct["scales"],
                                                sct["rotations"], st.scale_modifier, empty, st.viewmatrix,
                                                st.projmatrix, st.tanfovx, st.tanfovy, gt["color"], gt["normal"],
                                                gt["opacity"], gt["depth"], gt["feature"], sct["shs"], st.sh_degree,
                                                st.campos, gb, R, bb, ib, True, False)
        return R, color, g[3]

    mvec = torch.zeros(3, dtype=torch.float32, device=dev)

    def metrics(R, color, gmean):
        # "loss"-like scalars gathered across ranks with ONE collective per step (device-side reductions only: an
        # H2D scalar copy per step would stall the launch queue)
        torch.sum(color.reshape(-1), dim=0, out=mvec[0])
        torch.sum(gmean.reshape(-1), dim=0, out=mvec[1])
        mvec[2].fill_(float(R))
        return mvec

    gatherer = vp.MetricsGatherer(3, dev)   # async: the collective of step i overlaps with step i+1
    for _ in range(args.warmup):
        R, color, gm = step()
        gatherer.submit(metrics(R, color, gm))
    gatherer.drain()
    vp.barrier()
    torch.cuda.synchronize()
    _native.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        R, color, gm = step()
        gatherer.submit(metrics(R, color, gm))
    allm = gatherer.results()   # inside the timed region: the last collective has completed
    vp.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stage = {n: (ms, cnt) for n, ms, cnt in _native.last_timings(with_counts=True)}
    if os.environ.get("SVGIR_BENCH_MEMSTATS"):
        ms_ = torch.cuda.memory_stats(dev)
        print("memstats:", {k: ms_[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries",
                                                "reserved_bytes.all.peak", "allocated_bytes.all.peak")},
              "blob callbacks:", _native.ALLOC_STATS, file=sys.stderr)
    _native.set_profiling(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank != 0:
        return
    value = world * P * args.steps / elapsed
    ab = algorithmic_bytes(P, R, W, H, S, VS, variant == "svgss")
    dom = "render_bwd"
    dom_ms = stage[dom][0]
    dom_name = "render_bwd_kernel (backward composite)"
    if "grad_reduce" in stage:
        # svgss: the gradient read-modify-write part of B_bwd is carried out by the row stores of render_bwd plus the
        # per-Gaussian reduce kernel; the roofline is taken over both so that the byte model stays comparable
        dom_ms += stage["grad_reduce"][0]
        dom_name = "render_bwd_kernel + grad_reduce_kernel (backward composite incl. its gradient accumulation)"
    achieved = ab["bwd"] / (dom_ms * 1e-3) / 1e9
    fwd_ms = stage["render"][0]
    # HBM bytes per launch of the dominant kernel from the PMC counters: a committed measurement of this workload
    # (scripts/pmc_traffic.sh -> profiles/traffic_<workload>.json: separate rocprofv3 --pmc passes, gfx950 correction)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            for kname, kv in json.load(f).get("kernels", {}).items():
                if kname.startswith("render_bwd_kernel") or kname.startswith("grad_reduce_kernel"):
                    traffic = (traffic or 0) + kv["read_bytes"] + kv["write_bytes"]
    res = {
        "metric": "Gaussian-surfels/sec fwd+bwd @800x800 (1 view)",
        "value": value, "unit": "surfels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {variant} path, P={P} surfels, {W}x{H}, SH degree {sc['sh_degree']}, "
                               f"S={S}, VS={VS}, fwd+bwd, one view per step per GPU (BASELINE.json configs[1] for cfg2)",
                   "num_rendered": int(R), "views_per_step": world, "parallelism": f"view-parallel x{world}"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": ab["bwd"], "avg_launch_ms": dom_ms, "launches": stage[dom][1],
                     "fwd_composite": {"achieved": ab["fwd"] / (fwd_ms * 1e-3) / 1e9, "avg_launch_ms": fwd_ms,
                                       "algorithmic_bytes_per_launch": ab["fwd"]}},
        "stage_ms": {k: round(v[0], 4) for k, v in stage.items()},
    }
    if shade:
        # shading kernels: HBM bytes = per-sample inputs (dirs 12 + area 4 + visibility 4 + radiance 12 = 32 B) x Ns
        # + per-surfel inputs (31 floats) + outputs (reduced 70 floats [+ features/vfeatures in the no-grad path])
        sf = P * Ns * 32 + P * (31 + 70 + (0 if training else S + VS)) * 4
        res["config"]["shading"] = f"rendering_equation4 + packing, Ns={Ns} incident samples/surfel, env 32x64, " \
                                   f"{'forward+backward' if training else 'forward only (eval)'}"
        res["shading"] = {"fwd": {"avg_launch_ms": stage["shade_fwd"][0], "algorithmic_bytes_per_launch": sf,
                                  "achieved": sf / (stage["shade_fwd"][0] * 1e-3) / 1e9, "unit": "GB/s",
                                  "frac": sf / (stage["shade_fwd"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        if training and "shade_bwd" in stage:
            sb = P * Ns * (32 + 12) + P * (31 + 70 + 28) * 4   # + dL_dradiance per sample, per-surfel gradients
            res["shading"]["bwd"] = {"avg_launch_ms": stage["shade_bwd"][0], "algorithmic_bytes_per_launch": sb,
                                     "achieved": sb / (stage["shade_bwd"][0] * 1e-3) / 1e9, "unit": "GB/s",
                                     "frac": sb / (stage["shade_bwd"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        var_id = orc.SVGSS if variant == "svgss" else orc.RGSS
        cores = orc.max_threads()
        o = orc.OracleRun(sc, var_id)

        def cpu_step():
            o.forward()
            o.backward(grads["color"], grads["normal"], grads["depth"], grads["opacity"], grads["feature"],
                       grads.get("vfeature"))

        tc = time.perf_counter()
        cpu_step()  # warm-up (thread pool, page faults); also sizes the sample
        one = time.perf_counter() - tc
        n_cpu = args.cpu_steps or max(2, min(200, int(12.0 / max(one, 1e-3))))
        tc = time.perf_counter()
        for _ in range(n_cpu):
            cpu_step()
        cpu_el = time.perf_counter() - tc
        per_step = cpu_el / n_cpu
        sample = f"{n_cpu} fwd+bwd rasterizer steps of the same {args.workload} view ({cpu_el:.1f} s, OpenMP " \
                 f"oracle/svgir_oracle.cpp, {cores} threads)"
        if shade:
            # shading oracle (torch fp64 restatement of the reference's PyTorch code) on the first P/10 surfels, x10
            from oracle import shading_oracle as so
            n = max(1, P // 10)
            cd = {k: v[:n].double().cpu() for k, v in sd.items() if k != "env"}
            cl = {k: leaves[k].detach()[:n].double().cpu().requires_grad_(training) for k in ("base_color", "roughness", "normals", "radiance")}
            cenv = leaves["env"].detach().double().cpu().requires_grad_(training)
            ts = time.perf_counter()
            r = so.shade(cl["base_color"], cl["roughness"], cl["normals"], cd["viewdirs"], cl["radiance"], cd["visibility"],
                         cd["dirs"], cd["areas"], cenv)
            if training:
                (r["pbr"].sum() + r["diffuse_light"].sum() + r["mean_local"].sum()).backward()
            shade_s = (time.perf_counter() - ts) * (P / n)
            per_step += shade_s
            sample += f" + shading oracle (torch fp64, {torch.get_num_threads()} threads) on {n} surfels scaled to P " \
                      f"({shade_s:.2f} s/step)"
        res["cpu_baseline"] = {"value": P / per_step, "unit": "surfels/s", "cores": cores, "kind": "port",
                               "sample": sample}
        # the same rasterizer step on ONE host thread (SURVEY 8d asks for both); 2 steps, a few seconds
        o1 = orc.OracleRun(sc, var_id, num_threads=1)
        tc = time.perf_counter()
        for _ in range(2):
            o1.forward()
            o1.backward(grads["color"], grads["normal"], grads["depth"], grads["opacity"], grads["feature"],
                        grads.get("vfeature"))
        res["cpu_baseline"]["single_thread"] = {"value": 2 * P / (time.perf_counter() - tc), "unit": "surfels/s",
                                                "cores": 1, "sample": "2 fwd+bwd rasterizer steps, 1 thread"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
