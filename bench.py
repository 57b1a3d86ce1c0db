#!/usr/bin/env python3
"""Benchmark of the hot path: Gaussian-surfels/s, forward + backward, one 800x800 view per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3_train|cfg3_eval|cfg4|cfg5]

One process per GPU.  Launched under torchrun (WORLD_SIZE set) every process is a rank; launched plainly with
`--gpus N > 1` the script starts N rank processes itself BEFORE anything touches the GPU (fresh child processes, no
re-exec) and relays rank 0's line.  A "step" is one forward+backward pass of the rasterizer over one synthetic view
(inputs resident in HBM before the timed region), through the drop-in binding layer (`_C.rasterize_gaussians` +
`_C.rasterize_gaussians_backward` -> C ABI -> HIP kernels); the svgss workloads run the per-surfel SV-BRDF shading in
every step too.  N = 1 times BASELINE.json configs[1] (cfg2); N > 1 times configs[3] (cfg4: the eight armadillo test
views, rank r renders view r, azimuth 45 r degrees) -- views are sharded, weak scaling, and the only collective is one
fused all_gather of a 3-float metrics vector per step (RCCL over xGMI).

Timing: W warm-up steps, then the K-step timed region (barrier + synchronize on both sides, max over ranks) is repeated
`--repeats` times; `ms_per_step` / `value` are the MEDIAN region, min / max are reported beside it.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (backward composite): algorithmic bytes (SURVEY 8d formula with the measured R)
                  / average kernel time from HIP events recorded on the launch stream DURING the timed steps,
                  against the 8 TB/s HBM peak; `traffic` = PMC bytes per launch from profiles/traffic_<workload>.json,
                  nulled when the kernel sources changed since that measurement (source hash);
  "shaded":       (N = 1, default workload only) the same measurement on cfg3_train -- shading + rasterizer forward +
                  backward, the "shaded + blended" number of north_star -- with its own roofline;
  "cpu_baseline": the CPU oracle (port of the reference kernels; the reference has no CPU path) timed on the host
                  cores on a bounded number of the same steps (rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(P, R, W, H, S, VS, svgss):
    """per-instance gather G and the forward/backward composite byte models."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    G = 8 + 16 + 12 + 4 + 12 + 40 + (8 if svgss else 0) + 4 * S + 4 * VS
    pix = W * H * (44 + 4 * S + VS)
    fwd = 8 * T + R * (4 + G) + pix + 4 * P
    bwd = 8 * T + R * (4 + G) + 2 * R * (52 + 4 * S + 4 * VS) + pix
    return dict(G=G, fwd=fwd, bwd=bwd)


# sources of the kernels whose HBM traffic is recorded in profiles/traffic_*.json (the composite and shading kernels)
TRAFFIC_SOURCES = ("common.hpp", "stage.hpp", "pairstage.hpp", "dev_trace.hpp", "render_fwd.hip", "render_bwd.hip", "render_bwd_plain.hip", "grad_reduce.hip",
                   "shade.hip")


def kernel_source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "svg-ir_amd", "csrc")
    for f in TRAFFIC_SOURCES:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def library_hash():
    """sha256 of the libsvgir_raster.so this process would load (a stale prebuilt library then cannot pass for the
    sources' measurement)."""
    lib = os.environ.get("SVGIR_RASTER_LIB") or os.path.join(ROOT, "svg-ir_amd", "libsvgir_raster.so")
    try:
        return hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def library_is_current():
    """True when the library this process loads is at least as new as every kernel source (it was built from them)."""
    lib = os.environ.get("SVGIR_RASTER_LIB") or os.path.join(ROOT, "svg-ir_amd", "libsvgir_raster.so")
    d = os.path.join(ROOT, "svg-ir_amd", "csrc")
    try:
        return os.path.getmtime(lib) >= max(os.path.getmtime(os.path.join(d, f)) for f in TRAFFIC_SOURCES)
    except OSError:
        return False


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=25, help="repetitions of the K-step timed region (median reported)")
    ap.add_argument("--workload", default=None, help="default: cfg2 (N = 1), cfg4 (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shaded", action="store_true", help="skip the extra cfg3_train (shaded + blended) record")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the supplementary two-views-on-two-streams record")
    ap.add_argument("--no-overlap", action="store_true", help="svgss workloads: run the shading forward on the rasterizer's stream "
                    "instead of a side stream that overlaps the binning (svgir_params.features_ready)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU-oracle steps (0 = as many as fit ~12 s)")
    ap.add_argument("--no-shade", action="store_true", help="svgss workloads: skip the SV-BRDF shading stage")
    ap.add_argument("--streamed-dirs", action="store_true", help="shading reads [P,Ns,3] incident directions from HBM "
                    "instead of generating the lattice in the kernels")
    ap.add_argument("--samples", type=int, default=0, help="incident samples per surfel (default 64 train / 384 eval)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise launch / collective / reporting logic with "
                    "gloo on the CPU (the rendering step is replaced by a token computation)")
    return ap.parse_args()


def visible_gpu_count():
    """Number of GPUs a child process will see, WITHOUT loading the HIP runtime in this (parent) process: KFD topology
    nodes with SIMDs, narrowed by the *_VISIBLE_DEVICES lists.  None when the topology is not readable (the children
    then check their own device and exit 2)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return 0    # no KFD topology: no AMD GPU driver in this container
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes.  Nothing in this process touches the
    GPU or the HIP runtime (no torch import): the children are plain child processes, never an exec of a GPU process."""
    n = args.gpus
    if not args.dry_run:
        have = visible_gpu_count()
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


class Workload:
    """One synthetic view of a BASELINE config resident on the device, and a step() that runs it once."""

    def __init__(self, name, dev, rank, world, args):
        import torch
        from svgir_harness import cameras, runner, scenes, view_parallel as vp
        self.name, self.dev = name, dev
        gen_kw = dict(scenes.CONFIGS[name][1])
        self.variant = variant = gen_kw["variant"]
        sc = scenes.make(name)
        if name == "cfg4":     # eight cameras at azimuth k * 45 degrees, elevation 30 degrees, one per rank
            sc.update(cameras.make_camera(sc["W"], sc["H"], cameras.orbit_eye(4.0, 45.0 * rank, 30.0)))
        elif world > 1:        # other workloads under N > 1: every rank its own view of the replicated scene
            sc.update(cameras.make_camera(sc["W"], sc["H"], cameras.orbit_eye(4.0, 30.0 + 45.0 * rank, 25.0)))
        self.sc = sc
        self.train = True   # the metric is fwd+bwd: the rasterizer backward is part of every workload's step (the shading
                            # backward only at the training widths)
        self.grads = scenes.upstream_grads(sc, variant)
        sct = self.sct = runner.to_torch(sc, dev)
        per_gaussian = {k: sct[k] for k in ("means3D", "scales", "rotations", "opacities", "shs", "features") if k in sct}
        if variant == "svgss":
            per_gaussian["vfeatures"] = sct["vfeatures"]
        vp.broadcast_scene(per_gaussian)  # one-time replication (identical seeds already; exercised for N > 1)
        self.gt = {k: torch.from_numpy(v).to(dev) for k, v in self.grads.items()}
        self.P, self.W, self.H = int(sc["means3D"].shape[0]), sc["W"], sc["H"]
        self.S = int(sc["features"].shape[1])
        self.VS = int(sc["vfeatures"].shape[1]) if variant == "svgss" else 0
        if variant == "svgss":
            from gaussian_renderer.svgss_rasterization import _C
        else:
            from gaussian_renderer.rgss_rasterization import _C
        self._C = _C
        self.empty = torch.empty(0, dtype=torch.float32, device=dev)
        self.st = runner.settings(sct, variant)
        # svgss workloads: the per-surfel SV-BRDF shading (rendering_equation4 + feature packing) produces the rasterizer's
        # features / vfeatures every step, and its backward consumes the rasterizer's dL_dfeatures / dL_dvfeatures.
        self.shade = variant == "svgss" and not args.no_shade
        if self.shade:
            from gaussian_renderer import shading
            from svgir_harness import shade_inputs
            self.shading = shading
            self.training = self.S == 4
            self.Ns = args.samples or (64 if self.training else 384)
            q = torch.nn.functional.normalize(sct["rotations"], dim=-1)
            r, x, y, z = q.unbind(-1)   # local z axis of the surfel = geometric normal
            geo_n = torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], dim=-1)
            self.sd = sd = shade_inputs.make(self.P, self.Ns, seed=5 + rank, device=dev, geo_normals=geo_n, with_dirs=args.streamed_dirs)
            sd["viewdirs"] = torch.nn.functional.normalize(self.st.campos[None, :] - sct["means3D"], dim=-1)
            if args.streamed_dirs:
                self.dirs, self.areas = sd["dirs"], sd["areas"]
            else:   # incident directions generated in the kernels; training lattices carry random offsets
                offs = torch.rand(self.P, device=dev) * 6.2831855 if self.training else None
                self.dirs, self.areas = shading.FibonacciLattice(torch.nn.functional.normalize(geo_n, dim=-1), self.Ns, offs), None
            self.leaves = {k: sd[k].clone().requires_grad_(self.training) for k in ("base_color", "roughness", "normals", "radiance", "env")}
            self.light = shade_inputs.Light(self.leaves["env"])
            # the shading forward does not depend on the binning of the view (and vice versa): it runs on a side stream and only
            # the composite kernel waits for it (include/svgir_raster.h: svgir_params.features_ready)
            self.side = None if getattr(args, "no_overlap", False) else torch.cuda.Stream(dev)
            self.feat_ev = torch.cuda.Event() if self.side is not None else None

    def step(self):
        """One forward (+ backward) through the binding layer; returns (R, colour image, a gradient tensor)."""
        import torch
        st, sct, gt, _C, empty = self.st, self.sct, self.gt, self._C, self.empty
        if self.variant == "svgss":
            feats_in, vfeats_in = sct["features"], sct["vfeatures"]
            if self.shade:
                for v in self.leaves.values():
                    v.grad = None
                lv, sd = self.leaves, self.sd
                main = torch.cuda.current_stream(self.dev)
                if self.side is not None:
                    self.side.wait_stream(main)
                with torch.cuda.stream(self.side if self.side is not None else main), torch.set_grad_enabled(self.training):
                    feats_in, vfeats_in, _ = self.shading.shade_and_pack(
                        lv["base_color"], lv["roughness"], lv["normals"], sd["viewdirs"], lv["radiance"], self.light,
                        sd["visibility"], self.dirs, self.areas, st.viewmatrix, self.training)
                if self.side is not None:
                    self.feat_ev.record(self.side)
                    feats_in.record_stream(main); vfeats_in.record_stream(main)
            ready = self.feat_ev if (self.shade and self.side is not None) else None
            out = _C.rasterize_gaussians(st.bg, sct["means3D"], feats_in.detach(), vfeats_in.detach(), empty,
                                         sct["opacities"], sct["scales"], sct["rotations"], st.scale_modifier, empty,
                                         st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx,
                                         st.tanfovy, st.image_height, st.image_width, sct["shs"], st.sh_degree,
                                         st.campos, False, False, st.config, features_ready=ready)
            (R, color, normal, depth, opac, feat, vfeat, weights, radii, gb, bb, ib) = out
            if not self.train:
                return R, color, weights
            g = _C.rasterize_gaussians_backward(st.bg, sct["means3D"], feats_in.detach(), vfeats_in.detach(), radii, empty,
                                                sct["scales"], sct["rotations"], st.scale_modifier, empty,
                                                st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx,
                                                st.tanfovy, gt["color"], gt["normal"], gt["depth"], gt["opacity"],
                                                gt["feature"], gt["vfeature"], sct["shs"], st.sh_degree, st.campos,
                                                gb, R, bb, ib, False, st.config)
            if self.shade and self.training:
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


def timed(wl, args, world, dev, dry=None):
    """warm-up, then `repeats` x (barrier, sync, K steps, barrier, sync); returns (per-region seconds [max over ranks], R,
    stage timings)."""
    import torch
    from svgir_harness import view_parallel as vp
    mvec = torch.zeros(3, dtype=torch.float32, device=dev)

    def metrics(R, color, gmean):
        # "loss"-like scalars gathered across ranks with ONE collective per step.  They are bench scaffolding, not the hot
        # path, so they are kept to one small device-side reduction per step (the middle row of the image, written straight
        # into the gather's source vector); the full checksums of the last step are taken after the timed region.
        torch.sum(color[0, color.shape[1] // 2] if color.dim() == 3 else color, dim=0, keepdim=True, out=mvec[0:1])
        return mvec

    def checksums(R, color, gmean):
        return torch.stack([color.sum(), gmean.sum(), torch.tensor(float(R), device=dev)])

    sync = (lambda: None) if dry else torch.cuda.synchronize
    step = dry or wl.step
    gatherer = vp.MetricsGatherer(3, dev)   # async: the collective of step i overlaps with step i+1
    R = 0
    for _ in range(args.warmup):
        R, color, gm = step()
        gatherer.submit(metrics(R, color, gm))
    gatherer.drain()
    if not dry:
        from gaussian_renderer import _native
    regions = []
    nrep = max(1, args.repeats)

    def region():
        nonlocal R, color, gm
        vp.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            R, color, gm = step()
            if world > 1:   # (one rank: nothing to gather -- the checksums of the last step are taken after the region)
                gatherer.submit(metrics(R, color, gm))
        if world > 1:
            gatherer.results()   # inside the timed region: the last collective has completed
        vp.barrier()
        sync()
        return time.perf_counter() - t0

    color = gm = None
    for rep in range(nrep):
        regions.append(region())
    # Per-stage / per-kernel durations: ONE more region of the same K steps with the library's HIP-event stage marks on
    # (an event record per stage boundary costs ~4 us on the stream -- ~50 us per cfg2 step -- so the regions that
    # produce `value` run without them; this region is not part of `value`).
    stage = {}
    if not dry:
        # (the stage marks sit on ONE stream: the shading goes back to the rasterizer's stream for this region, otherwise the
        # composite's mark would include its wait for the side stream)
        side = getattr(wl, "side", None)
        if side is not None:
            wl.side = None
        _native.set_profiling(True)
        region()
        stage = {n: (ms, cnt) for n, ms, cnt in _native.last_timings(with_counts=True)}
        _native.set_profiling(False)
        if side is not None:
            wl.side = side
    el = torch.tensor(regions, dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    if gatherer.results() is None:   # (no warm-up steps on one rank)
        gatherer.submit(metrics(R, color, gm))
    table = gatherer.results().clone()
    table[:, 1:] = vp.gather_rows(checksums(R, color, gm))[:, 1:]
    return el.cpu().numpy(), int(R), stage, table.cpu().numpy()


def roofline_of(wl, R, stage, workload):
    ab = algorithmic_bytes(wl.P, R, wl.W, wl.H, wl.S, wl.VS, wl.variant == "svgss")
    out = {}
    if "render_bwd" in stage:
        dom_ms = stage["render_bwd"][0]
        dom_name = ("render_bwd_plain_kernel" if wl.VS == 0 else "render_bwd_kernel") + " (backward composite)"
        if "grad_reduce" in stage:
            # svgss: the gradient read-modify-write part of B_bwd is carried out by the row stores of render_bwd plus the
            # per-Gaussian reduce kernel; the roofline is taken over both so that the byte model stays comparable
            dom_ms += stage["grad_reduce"][0]
            dom_name = "render_bwd_kernel + grad_reduce_kernel (backward composite incl. its gradient accumulation)"
        achieved = ab["bwd"] / (dom_ms * 1e-3) / 1e9
        out = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": ab["bwd"],
               "avg_launch_ms": dom_ms, "launches": stage["render_bwd"][1]}
    fwd_ms = stage["render"][0] + stage.get("cull", (0.0, 0))[0]
    fwd = {"kernel": "cull_kernel + render_fwd_kernel (forward composite)", "achieved": ab["fwd"] / (fwd_ms * 1e-3) / 1e9,
           "frac": ab["fwd"] / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": fwd_ms,
           "algorithmic_bytes_per_launch": ab["fwd"]}
    if not out:   # forward-only workload: the forward composite is the dominant kernel
        out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "launches": stage["render"][1], **fwd}
    else:
        out["fwd_composite"] = fwd
    # HBM bytes per launch of the dominant kernel from the PMC counters: a committed measurement of this workload
    # (scripts/pmc_traffic.sh -> profiles/traffic_<workload>.json: separate rocprofv3 --pmc passes, gfx950 correction),
    # valid only for the kernel sources it was taken with
    tpath = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        same_lib = tj.get("library_hash") == library_hash()
        if tj.get("kernel_source_hash") == kernel_source_hash() and (same_lib or library_is_current()):
            if not same_lib:   # another build of the SAME sources (e.g. rebuilt on the measuring machine)
                out["traffic_note"] = "library rebuilt from the kernel sources profiles/traffic_%s.json was measured with" % workload
            tr = 0
            for kname, kv in tj.get("kernels", {}).items():
                if kname.startswith("render_bwd") or kname.startswith("grad_reduce_kernel"):
                    tr += kv["read_bytes"] + kv["write_bytes"]
            out["traffic"] = tr or None
        else:
            out["traffic_note"] = "profiles/traffic_%s.json was measured with different kernel sources / another library build" % workload
    return out


def shading_record(wl, stage):
    sf_per_sample = 32 if not isinstance(wl.dirs, wl.shading.FibonacciLattice) else 16
    sf = wl.P * wl.Ns * sf_per_sample + wl.P * (31 + 70 + (0 if wl.training else wl.S + wl.VS) + (0 if sf_per_sample == 32 else 4)) * 4
    rec = {"config": f"rendering_equation4 + packing, Ns={wl.Ns} incident samples/surfel, env 32x64, incident directions "
                     f"{'streamed from HBM' if sf_per_sample == 32 else 'generated in the kernels (Fibonacci lattice)'}, "
                     f"{'forward+backward' if wl.training else 'forward only (eval)'}",
           "fwd": {"avg_launch_ms": stage["shade_fwd"][0], "algorithmic_bytes_per_launch": sf, "bytes_per_sample": sf_per_sample,
                   "achieved": sf / (stage["shade_fwd"][0] * 1e-3) / 1e9, "unit": "GB/s",
                   "frac": sf / (stage["shade_fwd"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if wl.training and "shade_bwd" in stage:
        sb = wl.P * wl.Ns * (sf_per_sample + 12) + wl.P * (31 + 70 + 28) * 4   # + dL_dradiance per sample, per-surfel gradients
        rec["bwd"] = {"avg_launch_ms": stage["shade_bwd"][0], "algorithmic_bytes_per_launch": sb,
                      "achieved": sb / (stage["shade_bwd"][0] * 1e-3) / 1e9, "unit": "GB/s",
                      "frac": sb / (stage["shade_bwd"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return rec


def two_streams(name, dev, args):
    """Supplementary record (never the headline `value`): TWO views of the workload in flight on one GPU, one HIP stream and one
    host thread each.  One view leaves the SIMDs under-occupied (cfg2: 2 930 forward waves for 1 024 SIMDs); a
    per-GPU driver that keeps two views in flight fills those issue slots."""
    import threading
    import time
    import torch
    wls = [Workload(name, dev, 0, 1, args), Workload(name, dev, 1, 2, args)]   # the same replicated scene from two cameras
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    wl = wls[0]
    rounds = max(10, args.steps)

    def work(w, s, k):
        with torch.cuda.stream(s):
            for _ in range(k):
                w.step()

    for w, s in zip(wls, streams):
        work(w, s, 3)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(5):
        th = [threading.Thread(target=work, args=(w, s, rounds)) for w, s in zip(wls, streams)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ms_view = best / rounds / 2 * 1e3
    return {"views_in_flight": 2, "ms_per_view": ms_view, "value": wl.P / (ms_view * 1e-3), "unit": "surfels/s",
            "note": "two views of the same workload on two HIP streams (two host threads), wall clock over %d rounds, best of 5; "
                    "supplementary, the headline value is one view per step" % rounds}


def cpu_baseline(wl, args):
    """The CPU oracle on the host cores: `-O3 -march=native` build (compiled here, on the machine that
    runs it) when the compiler is available, else the parity build."""
    import torch
    from oracle import oracle as orc
    fast = orc.build_fast()
    var_id = orc.SVGSS if wl.variant == "svgss" else orc.RGSS
    cores = orc.max_threads()
    o = orc.OracleRun(wl.sc, var_id, fast=fast)
    g = wl.grads

    def cpu_step(oo):
        oo.forward()
        oo.backward(g["color"], g["normal"], g["depth"], g["opacity"], g["feature"], g.get("vfeature"))

    tc = time.perf_counter()
    cpu_step(o)  # warm-up (thread pool, page faults); also sizes the sample
    one = time.perf_counter() - tc
    n_cpu = args.cpu_steps or max(2, min(200, int(12.0 / max(one, 1e-3))))
    tc = time.perf_counter()
    for _ in range(n_cpu):
        cpu_step(o)
    cpu_el = time.perf_counter() - tc
    per_step = cpu_el / n_cpu
    build = "-O3 -march=native" if fast else "-O2 (parity build)"
    sample = f"{n_cpu} fwd+bwd rasterizer steps of the same {wl.name} view ({cpu_el:.1f} s, OpenMP oracle/svgir_oracle.cpp " \
             f"{build}, {cores} threads)"
    res = {"value": wl.P / per_step, "unit": "surfels/s", "cores": cores, "kind": "port", "sample": sample}
    # the same rasterizer step on ONE host thread (SURVEY 8d asks for both); 2 steps, a few seconds
    o1 = orc.OracleRun(wl.sc, var_id, num_threads=1, fast=fast)
    tc = time.perf_counter()
    for _ in range(2):
        cpu_step(o1)
    res["single_thread"] = {"value": 2 * wl.P / (time.perf_counter() - tc), "unit": "surfels/s", "cores": 1,
                            "sample": "2 fwd+bwd rasterizer steps, 1 thread"}
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    from svgir_harness import view_parallel as vp
    rank, world, local = vp.init_from_env(backend="gloo" if args.dry_run else None)
    if world != max(1, args.gpus):
        print(f"bench.py: WORLD_SIZE={world} does not match --gpus {args.gpus}", file=sys.stderr)
        sys.exit(2)
    name = args.workload or ("cfg2" if world == 1 else "cfg4")

    if args.dry_run:
        dev = torch.device("cpu")
        tok = torch.arange(8, dtype=torch.float32) + rank

        def dry_step():
            return 100 + rank, tok * 2.0, tok + 1.0

        regions, R, stage, table = timed(None, args, world, dev, dry=dry_step)
        if rank == 0:
            med = float(np.median(regions))
            print(json.dumps({"metric": "dry-run (no GPU work)", "value": world * args.steps / med, "unit": "steps/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "repeats": len(regions), "ms_per_step": med / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": name + " (dry run)", "views_per_step": world},
                              "per_rank_R": [int(r[2]) for r in table]}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    if torch.cuda.device_count() <= local or not torch.cuda.is_available():
        print("bench.py needs a GPU per rank (the rasterizer has no CPU path)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    wl = Workload(name, dev, rank, world, args)
    regions, R, stage, table = timed(wl, args, world, dev)
    # which physical GPU every rank ran on (a rank that silently fell back to another device would show up here)
    pr = torch.cuda.get_device_properties(dev)
    ids = vp.gather_rows(torch.tensor([float(torch.cuda.current_device()), float(getattr(pr, "pci_domain_id", -1)),
                                       float(getattr(pr, "pci_bus_id", -1)), float(getattr(pr, "pci_device_id", -1))],
                                      dtype=torch.float64, device=dev)).cpu().numpy()
    res = None
    if rank == 0:
        med = float(np.median(regions))
        res = {
            "metric": "Gaussian-surfels/sec fwd+bwd @800x800 (1 view)",
            "value": world * wl.P * args.steps / med, "unit": "surfels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": med / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": len(regions), "ms_per_step_min": float(regions.min()) / args.steps * 1e3,
            "ms_per_step_max": float(regions.max()) / args.steps * 1e3,
            "config": {"workload": f"{name}: {wl.variant} path, P={wl.P} surfels, {wl.W}x{wl.H}, SH degree {wl.sc['sh_degree']}, "
                                   f"S={wl.S}, VS={wl.VS}, {'fwd+bwd' if wl.train else 'fwd'}, one view per step per GPU "
                                   f"(BASELINE.json configs[1] = cfg2 at N=1; configs[3] = cfg4, view r on rank r, at N>1)",
                       "num_rendered": int(R), "views_per_step": world, "parallelism": f"view-parallel x{world}",
                       "per_rank_num_rendered": [int(r[2]) for r in table],
                       "per_rank_device": [{"rank": i, "ordinal": int(r[0]), "pci": "%04x:%02x:%02x" % (int(r[1]) & 0xffff, int(r[2]) & 0xff, int(r[3]) & 0xff)}
                                           for i, r in enumerate(ids)]},
            "roofline": roofline_of(wl, R, stage, name),
            "stage_ms": {k: round(v[0], 4) for k, v in stage.items()},
        }
        if wl.shade:
            res["config"]["shading"] = shading_record(wl, stage)["config"]
            res["shading"] = {k: v for k, v in shading_record(wl, stage).items() if k != "config"}
    # the "shaded + blended" number of north_star: cfg3_train with the shading stage, same measurement, extra keys
    if world == 1 and args.workload is None and not args.no_shaded:
        wl.sct = wl.gt = None
        torch.cuda.empty_cache()
        sa = argparse.Namespace(**vars(args))
        sa.repeats = max(1, args.repeats // 5)
        w3 = Workload("cfg3_train", dev, rank, world, sa)
        reg3, R3, st3, _ = timed(w3, sa, world, dev)
        m3 = float(np.median(reg3))
        res["shaded"] = {"workload": f"cfg3_train: svgss path, P={w3.P}, {w3.W}x{w3.H}, S={w3.S}, VS={w3.VS}, SV-BRDF shading "
                                     f"(Ns={w3.Ns}) + rasterizer, fwd+bwd (BASELINE.json configs[2])",
                         "value": w3.P * sa.steps / m3, "unit": "surfels/s", "ms_per_step": m3 / sa.steps * 1e3, "repeats": len(reg3),
                         "num_rendered": int(R3), "roofline": roofline_of(w3, R3, st3, "cfg3_train"),
                         "shading": shading_record(w3, st3), "stage_ms": {k: round(v[0], 4) for k, v in st3.items()}}
        del w3
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(wl, args)
    if rank == 0 and world == 1 and not args.no_concurrent:   # (last: it creates streams of its own)
        torch.cuda.empty_cache()
        res["two_streams"] = two_streams(name, dev, args)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
