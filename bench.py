#!/usr/bin/env python3
"""Benchmark of the hot path: Gaussian-surfels/s, forward + backward, one 800x800 view per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3_train|cfg3_eval|cfg4|cfg5|cfg5_dense|train_step|tracers]

`--workload train_step` times one whole stage-2 training iteration (shade -> rasterize -> unpack -> L1 + SSIM -> backward ->
densification statistics -> fused Adam; reference: train.py:133-134, gaussian_renderer/svgss.py:15-262,
scene/gaussian_model.py:775-813, :1270-1276) on the cfg3_train scene; `--workload tracers` times the visibility / radiance cache
producers (scene/gaussian_model.py:435-522: BVH builds + trace_visibility + render_radiance_with_sampling_SH, P x 64 rays each, the
reference's chunk loop) on the cfg3 geometry.  Both print the same kind of JSON line (their own metric / unit).

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
# Instruction-issue roof of the chip, measured (profiles/r03_valu_rate_probe.txt, HISTORY.md 4): a SIMD issues one wave64 VALU
# instruction per 1.67 cycles at best (8 resident waves), a scalar instruction costs 1.3 more; 256 CUs x 4 SIMDs at 2.4 GHz
SIMDS, CLOCK_HZ, VALU_ISSUE_CYCLES, SALU_ISSUE_CYCLES = 1024, 2.4e9, 1.67, 1.3


def issue_record(workload, prefixes, launch_ms):
    """Issue-side roofline of the kernels whose names start with `prefixes`: wave-level VALU / SALU instruction counts per launch
    from the committed PMC pass (profiles/issue_<workload>.json, scripts/pmc_issue.sh; valid only for the kernel sources and
    the library it was measured with) priced at the measured best-case issue rate, over the SIMD cycles of the launch as timed NOW:
        issue_frac = (VALU * 1.67 + SALU * 1.3) cycles / (1024 SIMDs * launch seconds * 2.4 GHz)."""
    path = os.path.join(ROOT, "profiles", f"issue_{workload}.json")
    if not os.path.exists(path) or not launch_ms:
        return None
    with open(path) as f:
        tj = json.load(f)
    if tj.get("kernel_source_hash") != kernel_source_hash() or not (tj.get("library_hash") == library_hash() or library_is_current()):
        return {"issue_frac": None, "note": f"profiles/issue_{workload}.json was measured with different kernel sources / another library build"}
    valu = salu = 0.0
    for pre in (prefixes if isinstance(prefixes, tuple) else (prefixes,)):
        # (of a kernel's template variants -- the forward composite's two occupancy variants -- the one with the most launches: the steady state)
        ks = [(kv.get("launches", 0), kv["valu"], kv["salu"]) for kname, kv in tj.get("kernels", {}).items() if kname.startswith(pre)]
        if ks:
            _, v, sa = max(ks)
            valu += v; salu += sa
    if valu == 0.0:
        return None
    cyc = valu * VALU_ISSUE_CYCLES + salu * SALU_ISSUE_CYCLES
    return {"issue_frac": cyc / (SIMDS * launch_ms * 1e-3 * CLOCK_HZ), "valu_insts_per_launch": valu, "salu_insts_per_launch": salu,
            "method": "wave-level VALU / SALU instruction counts (rocprofv3 --pmc SQ_INSTS_*, profiles/issue_%s.json) x 1.67 / 1.3 issue cycles "
                      "(profiles/r03_valu_rate_probe.txt) over 1024 SIMDs x launch time x 2.4 GHz" % workload}


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
    ap.add_argument("--repeats", type=int, default=0, help="repetitions of the K-step timed region (median reported); 0 = as many as "
                    "keep the GPU busy for ~3 s (at least 25, at most 500)")
    ap.add_argument("--workload", default=None, help="default: cfg2 (N = 1), cfg4 (N > 1); also cfg3_train, cfg3_eval, cfg5, cfg5_dense, "
                    "train_step (whole stage-2 iteration), tracers (visibility / radiance cache producers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shaded", action="store_true", help="skip the extra records of the default line: cfg3_train (shaded + blended), the "
                                                              "compact cfg3_eval / cfg4 / cfg5 records and the rotating-camera records")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the supplementary two-views-on-two-streams record")
    ap.add_argument("--no-overlap", action="store_true", help="svgss workloads: run the shading forward on the rasterizer's stream "
                    "instead of a side stream that overlaps the binning (svgir_params.features_ready)")
    ap.add_argument("--unfused", action="store_true", help="svgss workloads: shade ALL surfels with svgir_shade_forward / _backward around "
                    "the rasterizer calls (rounds 1-4) instead of the view's working set inside them (svgir_params.shade)")
    ap.add_argument("--radiance-grad", action="store_true", help="svgss training workloads: differentiate the [P,Ns,3] radiance cache itself "
                    "(rounds 1-5a) instead of the reference's get_radiances = nan_to_num(_radiances.detach() * _radiance_ratio)")
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


def pin_rank_to_cores(local, world):
    """One process per GPU: rank r keeps to its own slice of the host cores (the host side of a view is a Python thread that launches
    ~25 kernels and polls pinned memory for the instance count for up to 50 ms; eight such ranks that share cores -- or migrate across
    NUMA nodes -- delay each other's launches).  Cores = the affinity mask this process inherited (a container's cpuset), split evenly
    in order; SVGIR_NO_AFFINITY=1 leaves the scheduler alone.  Returns the sorted list of cores, [] when nothing was pinned."""
    if world <= 1 or os.environ.get("SVGIR_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return []
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // world
    if per < 1:
        return []
    mine = cores[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return []
    return mine


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
            # the radiance cache enters as the reference's get_radiances (scene/gaussian_model.py:323-324): detached, times the learnable
            # scalar _radiance_ratio -- the product is formed inside the kernels and the backward returns the scalar's gradient
            self.rad_grad = bool(getattr(args, "radiance_grad", False))
            self.leaves = {k: sd[k].clone().requires_grad_(self.training and (k != "radiance" or self.rad_grad))
                           for k in ("base_color", "roughness", "normals", "radiance", "env")}
            self.ratio = None if self.rad_grad else torch.ones((), device=dev, requires_grad=self.training)
            self.light = shade_inputs.Light(self.leaves["env"])
            # the shading forward does not depend on the binning of the view (and vice versa): it runs on a side stream and only
            # the composite kernel waits for it (include/svgir_raster.h: svgir_params.features_ready)
            self.fused = not getattr(args, "unfused", False)
            self.side = None if (getattr(args, "no_overlap", False) or self.fused) else torch.cuda.Stream(dev)
            self.feat_ev = torch.cuda.Event() if self.side is not None else None
            if self.fused:   # outputs the library writes: the packed rows, and (training) the gradients of the shading's inputs
                self.f_buf = torch.empty((self.P, self.S), dtype=torch.float32, device=dev)
                self.vf_buf = torch.empty((self.P, self.VS), dtype=torch.float32, device=dev)
                self.sgrads = None
                if self.training:   # (the per-surfel gradient tensors are carved out of the rasterizer's gradient allocation per step)
                    self.senv = (torch.empty_like(sd["env"]), torch.empty(sd["env"].numel() + self.shading.RATIO_WORK, device=dev))
                    self.sshapes = dict(dL_dbase_color=sd["base_color"].shape, dL_droughness=sd["roughness"].shape,
                                        dL_dshade_normals=sd["normals"].shape)
                    if self.rad_grad:
                        self.sshapes["dL_dradiance"] = sd["radiance"].shape
                    self.sratio = torch.empty(1, device=dev)

    def shading_light(self):
        from svgir_harness import shade_inputs
        return shade_inputs.Light(self.leaves["env"].detach())

    def add_orbit_views(self, n, seed):
        """n seeded cameras on the radius-4 sphere around the scene (azimuth: n equal sectors of 0-360 degrees with a random offset inside
        each, elevation uniform in [-60, 80] degrees), looking at the origin; use_view(k) makes camera k the one step() renders.  The
        reference pops a random training camera every iteration (train.py:127): no two consecutive steps share a view."""
        import numpy as np
        import torch
        from svgir_harness import cameras, runner
        rng = np.random.default_rng(seed)
        self.views = []
        for k in range(n):
            az = (k + rng.random()) * 360.0 / n
            el = -60.0 + 140.0 * rng.random()
            cam = cameras.make_camera(self.W, self.H, cameras.orbit_eye(4.0, az, el))
            sct = dict(self.sct)
            sct.update(runner.to_torch(cam, self.dev))
            st = runner.settings(sct, self.variant)
            vd = torch.nn.functional.normalize(st.campos[None, :] - self.sct["means3D"], dim=-1) if self.shade else None
            self.views.append((st, vd, (az, el)))
        return self.views

    def use_view(self, k):
        st, vd, _ = self.views[k]
        self.st = st
        if vd is not None:
            self.sd["viewdirs"] = vd

    def step(self):
        """One forward (+ backward) through the binding layer; returns (R, colour image, a gradient tensor)."""
        import torch
        st, sct, gt, _C, empty = self.st, self.sct, self.gt, self._C, self.empty
        if self.variant == "svgss" and self.shade and self.fused:
            # the shading runs inside the two library calls, for the surfels this view reads (svgir_fused_shade)
            lv, sd = self.leaves, self.sd
            fs, keep = self.shading.fused_shade(lv["base_color"].detach(), lv["roughness"].detach(), lv["normals"].detach(), sd["viewdirs"],
                                                lv["radiance"].detach(), self.shading_light(), sd["visibility"], self.dirs, self.areas,
                                                st.viewmatrix, self.training, radiance_ratio=self.ratio)
            out = _C.rasterize_gaussians(st.bg, sct["means3D"], self.f_buf, self.vf_buf, empty, sct["opacities"], sct["scales"],
                                         sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint,
                                         st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height, st.image_width, sct["shs"],
                                         st.sh_degree, st.campos, False, False, st.config, shade=fs)
            (R, color, normal, depth, opac, feat, vfeat, weights, radii, gb, bb, ib) = out
            if not self.train:
                return R, color, weights
            kw = dict(out_weights=weights)
            if self.training:
                self.sgrads = dict(dL_denv=self.senv[0], env_grad_work=self.senv[1], dL_dreduced=None, out_weights=weights,
                                   _shapes=dict(self.sshapes))
                if self.ratio is not None:
                    self.sgrads["dL_dradiance_ratio"] = self.sratio
                kw = dict(shade=fs, shade_grads=self.sgrads, scratch_feature_grads=True)
            g = _C.rasterize_gaussians_backward(st.bg, sct["means3D"], self.f_buf, self.vf_buf, radii, empty, sct["scales"],
                                                sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint,
                                                st.patch_bbox, st.tanfovx, st.tanfovy, gt["color"], gt["normal"], gt["depth"],
                                                gt["opacity"], gt["feature"], gt["vfeature"], sct["shs"], st.sh_degree, st.campos,
                                                gb, R, bb, ib, False, st.config, **kw)
        elif self.variant == "svgss":
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
                        sd["visibility"], self.dirs, self.areas, st.viewmatrix, self.training, radiance_ratio=self.ratio)
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
                                                gb, R, bb, ib, False, st.config, out_weights=weights)
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
                                                st.campos, gb, R, bb, ib, True, False, out_weights=weights)
        return R, color, g[3]


def timed(wl, args, world, dev, dry=None, solo=False, step_fn=None, stages=True):
    """warm-up, then `repeats` x (barrier, sync, K steps, barrier, sync); returns (per-region seconds [max over ranks], R,
    stage timings).  solo: this rank alone (no barrier, no collective: the other ranks of a multi-GPU job wait outside);
    step_fn: what a step is (default wl.step); stages: run the extra region with the library's stage marks."""
    import torch
    from svgir_harness import view_parallel as vp
    mvec = torch.zeros(3, dtype=torch.float32, device=dev)
    barrier = (lambda: None) if solo else vp.barrier

    def metrics(R, color, gmean):
        # "loss"-like scalars gathered across ranks with ONE collective per step.  They are bench scaffolding, not the hot
        # path, so they are kept to one small device-side reduction per step (the middle row of the image, written straight
        # into the gather's source vector); the full checksums of the last step are taken after the timed region.
        torch.sum(color[0, color.shape[1] // 2] if color.dim() == 3 else color, dim=0, keepdim=True, out=mvec[0:1])
        return mvec

    def checksums(R, color, gmean):
        return torch.stack([color.sum(), gmean.sum(), torch.tensor(float(R), device=dev)])

    sync = (lambda: None) if dry else torch.cuda.synchronize
    coll = (world > 1 or vp.FORCE) and not solo   # (FORCE: a one-rank job that issues its collectives anyway, tests/test_gpu_view_parallel.py)
    step = dry or step_fn or wl.step
    gatherer = vp.MetricsGatherer(3, dev)   # async: the collective of step i overlaps with step i+1
    R = 0
    # (untimed, like the warm-up: the library sizes its speculative launches -- instance capacity, state slots, depth-sort passes --
    # from the last three views of a workload; with W < 4 the first timed steps would still be the unprimed ones)
    for _ in range(0 if dry else max(0, 4 - args.warmup)):
        step()
    for _ in range(args.warmup):
        R, color, gm = step()
        if not solo:
            gatherer.submit(metrics(R, color, gm))
    if not solo:
        gatherer.drain()
    if not dry:
        from gaussian_renderer import _native
    regions = []
    nrep = args.repeats

    def region():
        nonlocal R, color, gm
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            R, color, gm = step()
            if coll:   # (one rank: nothing to gather -- the checksums of the last step are taken after the region)
                gatherer.submit(metrics(R, color, gm))
        if coll:
            gatherer.results()   # inside the timed region: the last collective has completed
        barrier()
        sync()
        return time.perf_counter() - t0

    color = gm = None
    if nrep <= 0:   # size the measurement: enough regions for ~3 s of GPU work (an independent utilisation probe then sees the GPU busy)
        probe = region()
        nrep = 25 if dry else int(min(500, max(25, 3.0 / max(probe, 1e-4))))
        if coll:   # every rank must run the same number of regions (they contain barriers)
            t = torch.tensor([nrep], dtype=torch.int64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            nrep = int(t.item())
    for rep in range(nrep):
        regions.append(region())
    # Per-stage / per-kernel durations: ONE more region of the same K steps with the library's HIP-event stage marks on
    # (an event record per stage boundary costs ~4 us on the stream -- ~50 us per cfg2 step -- so the regions that
    # produce `value` run without them; this region is not part of `value`).
    stage = {}
    if not dry and stages:
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
    if solo:
        return el.cpu().numpy(), int(R), stage, None
    if coll:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    if gatherer.results() is None:   # (no warm-up steps on one rank)
        gatherer.submit(metrics(R, color, gm))
    table = gatherer.results().clone()
    table[:, 1:] = vp.gather_rows(checksums(R, color, gm))[:, 1:]
    return el.cpu().numpy(), int(R), stage, table.cpu().numpy()


def walked_instances(wl):
    """Instances any implementation of the composite MUST touch: per tile, the entries in front of (and including) the deepest
    contributor of its pixels, sum over tiles of max(n_contrib) -- what is left of R when every pixel of a tile saturates early
    (dense scenes: cfg5_dense walks 17 % of its 18.3 M instances).  The SURVEY 8(d) byte model charges all R instances (the
    reference's backward does fetch every tile's whole list, backward.cu:617-650); both are reported."""
    import numpy as np
    from svgir_harness import runner
    if getattr(wl, "sct", None) is None:
        return None
    raw = runner.forward_raw(wl.sct, wl.variant)
    nc = raw["n_contrib"]
    H, W = nc.shape
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), dtype=nc.dtype)
    pad[:H, :W] = nc
    tmax = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1).astype(np.int64)
    lens = (raw["ranges"][:, 1].astype(np.int64) - raw["ranges"][:, 0].astype(np.int64))
    return int(np.minimum(tmax, lens).sum())


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
            tr = trf = 0
            # (a kernel that exists in several template variants -- the forward composite's two occupancy variants, grad_reduce's batch
            # sizes -- is represented by the variant with the most launches: the steady state, not the workload's first views)
            def steady(prefix):
                ks = [(kv.get("launches", 0), kv["read_bytes"] + kv["write_bytes"]) for kn, kv in tj.get("kernels", {}).items() if kn.startswith(prefix)]
                return max(ks)[1] if ks else 0
            tr = steady("render_bwd") + steady("grad_reduce_kernel")
            trf = steady("cull_kernel") + steady("render_fwd")
            if "render_bwd" in stage:
                out["traffic"] = tr or None
                out["fwd_composite"]["traffic"] = trf or None
            else:
                out["traffic"] = trf or None
        else:
            out["traffic_note"] = "profiles/traffic_%s.json was measured with different kernel sources / another library build" % workload
    try:
        Rw = walked_instances(wl)
    except Exception:   # noqa: BLE001 -- supplementary
        Rw = None
    if Rw:
        abw = algorithmic_bytes(wl.P, Rw, wl.W, wl.H, wl.S, wl.VS, wl.variant == "svgss")
        key = "bwd" if "render_bwd" in stage else "fwd"
        out["walked"] = {"instances": Rw, "share_of_num_rendered": Rw / max(R, 1), "bytes_per_launch": abw[key],
                         "frac": abw[key] / (out["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "the same byte model over the instances in front of each tile's deepest contributor (what any "
                                 "implementation must touch); `frac` charges all num_rendered instances like SURVEY 8(d) and can exceed "
                                 "1 when most of a tile's list lies behind saturated pixels"}
    # which roof the kernel is nearer to: `frac` stays BASELINE's metric (algorithmic bytes / time / 8 TB/s); `issue_frac` prices the
    # instructions the kernel issues (committed PMC pass) at the chip's measured issue rate; `bound` names the larger of the two
    if "render_bwd" in stage:
        ir = issue_record(workload, ("render_bwd", "grad_reduce"), out.get("avg_launch_ms"))
    else:
        ir = issue_record(workload, ("render_fwd", "cull_kernel"), out.get("avg_launch_ms"))
    if ir:
        out["issue_frac"] = ir.get("issue_frac")
        out["issue"] = {k: v for k, v in ir.items() if k != "issue_frac"}
        if ir.get("issue_frac") is not None:
            out["bound"] = "issue" if ir["issue_frac"] > out["frac"] else "hbm"
    if "fwd_composite" in out:
        irf = issue_record(workload, ("render_fwd", "cull_kernel"), out["fwd_composite"]["avg_launch_ms"])
        if irf and irf.get("issue_frac") is not None:
            out["fwd_composite"]["issue_frac"] = irf["issue_frac"]
            out["fwd_composite"]["bound"] = "issue" if irf["issue_frac"] > out["fwd_composite"]["frac"] else "hbm"
    # where the step's time goes: the stage with the largest share (the composite backward on the training / rgss configurations; the
    # shading forward can lead at the evaluation sample counts) -- `kernel` above stays the composite, BASELINE's roofline metric
    core = {k: v[0] for k, v in stage.items() if k not in ("shade", "shade_env_table", "shade_bwd_prologue", "shade_env_grad")}
    if core:
        top = max(core, key=core.get)
        out["dominant_stage_by_time"] = {"stage": top, "ms": core[top], "share_of_stage_sum": core[top] / sum(core.values())}
    return out


def shading_counts(wl):
    """(surfels the shading forward shades, surfels its backward differentiates) for one view of a fused svgss Workload: the
    candidates of at least one 8x8 sub-tile, and the surfels with out_weights > 0."""
    import torch
    if not getattr(wl, "fused", False):
        return wl.P, wl.P
    _, _, third = wl.step()
    torch.cuda.synchronize()
    nf = int((wl.vf_buf.abs().sum(1) > 0).sum())
    w = wl.sgrads["out_weights"] if wl.sgrads is not None else third   # (forward-only workloads return the weights)
    return nf, int((w > 0).sum())


def shading_record(wl, stage, counts=None):
    sf_per_sample = 32 if not isinstance(wl.dirs, wl.shading.FibonacciLattice) else 16
    # surfels the kernels actually shade: all P (--unfused), or the view's working set (fused); the byte model charges those
    nf, nb = counts if counts is not None else shading_counts(wl)
    Pf = nf
    sf = Pf * wl.Ns * sf_per_sample + Pf * (31 + 70 + (0 if wl.training else wl.S + wl.VS) + (0 if sf_per_sample == 32 else 4)) * 4
    rec = {"surfels": wl.P, "surfels_shaded_fwd": nf, "surfels_shaded_bwd": nb if wl.training else None,
           "mode": "fused into svgir_forward / svgir_backward: the view's working set" if getattr(wl, "fused", False) else "all surfels (svgir_shade_forward / _backward around the rasterizer)",
           "radiance": ("the [P,Ns,3] cache is differentiated (--radiance-grad)" if getattr(wl, "rad_grad", False) else
                        "get_radiances = nan_to_num(_radiances.detach() * _radiance_ratio) formed in the kernels; the backward returns dL/d_radiance_ratio"),
           "config": f"rendering_equation4 + packing, Ns={wl.Ns} incident samples/surfel, env 32x64, incident directions "
                     f"{'streamed from HBM' if sf_per_sample == 32 else 'generated in the kernels (Fibonacci lattice)'}, "
                     f"{'forward+backward' if wl.training else 'forward only (eval)'}",
           "fwd": {"avg_launch_ms": stage["shade_fwd"][0], "algorithmic_bytes_per_launch": sf, "bytes_per_sample": sf_per_sample,
                   "achieved": sf / (stage["shade_fwd"][0] * 1e-3) / 1e9, "unit": "GB/s",
                   "frac": sf / (stage["shade_fwd"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if wl.training and "shade_bwd" in stage:
        rg = 12 if getattr(wl, "rad_grad", False) else 0   # dL_dradiance per sample (only when the cache itself is differentiated)
        sb = nb * wl.Ns * (sf_per_sample + rg) + nb * (31 + 70 + 28) * 4 + (wl.P - nb) * (wl.Ns * rg + 28 * 4)   # + per-surfel gradients (zero rows for the rest)
        rec["bwd"] = {"avg_launch_ms": stage["shade_bwd"][0], "algorithmic_bytes_per_launch": sb,
                      "achieved": sb / (stage["shade_bwd"][0] * 1e-3) / 1e9, "unit": "GB/s",
                      "frac": sb / (stage["shade_bwd"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    wname = getattr(wl, "name", None) or ("cfg3_train" if wl.training else "cfg3_eval")
    for part, pre in (("fwd", ("shade_fwd",)), ("bwd", ("shade_bwd",))):
        if part in rec:
            ir = issue_record("cfg3_train" if wname == "train_step" else wname, pre, rec[part]["avg_launch_ms"])
            if ir and ir.get("issue_frac") is not None:
                rec[part]["issue_frac"] = ir["issue_frac"]
                rec[part]["bound"] = "issue" if ir["issue_frac"] > rec[part]["frac"] else "hbm"
    return rec


def compact_record(name, dev, args, repeats=9):
    """One more BASELINE configuration in the default line: the same region protocol as the headline (K steps between barrier +
    synchronize), a fixed number of regions, and the two composite roofline fractions with the measured instance count."""
    import numpy as np
    import torch
    sa = argparse.Namespace(**vars(args))
    sa.repeats = repeats
    wl = Workload(name, dev, 0, 1, sa)
    reg, R, st, _ = timed(wl, sa, 1, dev)
    med = float(np.median(reg))
    rf = roofline_of(wl, R, st, name)
    rec = {"workload": f"{name}: {wl.variant} path, P={wl.P}, {wl.W}x{wl.H}, S={wl.S}, VS={wl.VS}" + (f", shading Ns={wl.Ns}" if wl.shade else "") + ", fwd+bwd",
           "value": wl.P * sa.steps / med, "unit": "surfels/s", "ms_per_step": med / sa.steps * 1e3, "repeats": len(reg), "num_rendered": int(R),
           "bwd_composite": {k: rf.get(k) for k in ("frac", "achieved", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "issue_frac", "bound")},
           "fwd_composite": {k: rf.get("fwd_composite", {}).get(k) for k in ("frac", "achieved", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "issue_frac")},
           "walked_share_of_num_rendered": (rf.get("walked") or {}).get("share_of_num_rendered"),
           "dominant_stage_by_time": rf.get("dominant_stage_by_time"),
           "stage_ms": {k: round(v[0], 4) for k, v in st.items()}}
    if wl.shade and "shade_fwd" in st:
        sr = shading_record(wl, st)
        rec["shade_fwd"] = {k: sr["fwd"].get(k) for k in ("avg_launch_ms", "frac", "issue_frac")}
        rec["surfels_shaded_fwd"] = sr["surfels_shaded_fwd"]
    del wl
    torch.cuda.empty_cache()
    return rec


def rotating_views(name, dev, args, fixed_ms, n_views=24, seed=606, repeats=9):
    """The same workload with a DIFFERENT camera every step: n_views seeded orbit cameras (Workload.add_orbit_views) visited in a seeded
    random order, fwd+bwd, same region protocol -- the reference never renders one view twice in a row (train.py:127), and a repeated view
    is the best case of every speculation in the library (instance capacity, state slots, three-pass depth sort) and of the 256 MB
    Infinity Cache.  Reports the step time next to the fixed-view one and what the speculative launches did during the regions."""
    import numpy as np
    import torch
    from gaussian_renderer import _native
    sa = argparse.Namespace(**vars(args))
    sa.repeats = repeats
    wl = Workload(name, dev, 0, 1, sa)
    views = wl.add_orbit_views(n_views, seed)
    order = np.random.default_rng(seed + 1).permutation(n_views * 64) % n_views
    for i in range(1, len(order)):   # (never the same camera twice in a row)
        if order[i] == order[i - 1]:
            order[i] = (order[i] + 1) % n_views
    pos = [0]

    def step():
        wl.use_view(int(order[pos[0] % len(order)]))
        pos[0] += 1
        return wl.step()

    for k in range(n_views):   # every camera once, untimed: per-view instance counts for the record (and the allocator's pools)
        wl.use_view(k)
        wl.step()
    torch.cuda.synchronize()
    Rs, per_view_ms = [], []
    for k in range(n_views):   # every camera as a FIXED view (3 untimed + 10 timed steps): what the rotation itself costs is the
        wl.use_view(k)         # rotating loop against the mean of these, not against the one camera of the headline
        for _ in range(3):
            Rs_k = int(wl.step()[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            wl.step()
        torch.cuda.synchronize()
        per_view_ms.append((time.perf_counter() - t0) / 10 * 1e3)
        Rs.append(Rs_k)
    before = _native.speculation_stats()
    reg, R, _, _ = timed(wl, sa, 1, dev, step_fn=step, stages=False)
    after = _native.speculation_stats()
    med = float(np.median(reg))
    ms = med / sa.steps * 1e3
    rec = {"workload": name, "views": n_views, "order": "seeded random, no camera twice in a row", "azimuth_deg": "0-360", "elevation_deg": "-60..80",
           "value": wl.P * sa.steps / med, "unit": "surfels/s", "ms_per_step": ms, "repeats": len(reg), "steps_timed": len(reg) * sa.steps,
           "fixed_view_ms_per_step": fixed_ms, "ratio_to_fixed_view": ms / fixed_ms if fixed_ms else None,
           "same_cameras_fixed_ms_per_step": {"mean": float(np.mean(per_view_ms)), "min": float(np.min(per_view_ms)), "max": float(np.max(per_view_ms))},
           "ratio_to_same_cameras_fixed": ms / float(np.mean(per_view_ms)),
           "num_rendered_min": min(Rs), "num_rendered_max": max(Rs), "num_rendered_mean": float(np.mean(Rs)),
           "speculation": {k: after[k] - before[k] for k in after}}
    del wl
    torch.cuda.empty_cache()
    return rec


def two_streams(name, dev, args):
    """Supplementary record (never the headline `value`): TWO views of the workload in flight on one GPU, one HIP stream and one
    host thread each.  One view leaves the SIMDs under-occupied (cfg2: 2 930 forward waves for 1 024 SIMDs, HISTORY.md 4); a
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


def one_thread_batch(name, dev, args, V=4):
    """Supplementary record: V views of the workload in flight from ONE host thread -- `svgir_forward_batch` launches the forwards of
    all V views (one stream each) before it waits for the first view's instance count, the V backwards follow on the views' streams.
    No second thread, no Python GIL hand-over between views (the `two_streams` record needs two threads for two views)."""
    import time
    import torch
    wls = [Workload(name, dev, r, max(2, V), args) for r in range(V)]   # the same replicated scene from V cameras
    streams = [torch.cuda.Stream(dev) for _ in range(V)]
    w0 = wls[0]
    if w0.variant != "rgss":
        return None
    _C, empty = w0._C, w0.empty

    def fwd_args(w):
        st, sct = w.st, w.sct
        return ((st.bg, sct["means3D"], sct["features"], empty, sct["opacities"], sct["scales"], sct["rotations"], st.scale_modifier, empty,
                 st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.cx, st.cy, st.image_height, st.image_width, sct["shs"],
                 st.sh_degree, st.campos, False, False, False), {})

    def bwd(w, out):
        st, sct, gt = w.st, w.sct, w.gt
        (R, ncontrib, color, normal, opac, depth, feat, pn, sx, weights, radii, gb, bb, ib) = out
        return _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], radii, empty, sct["scales"], sct["rotations"],
                                               st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, gt["color"],
                                               gt["normal"], gt["opacity"], gt["depth"], gt["feature"], sct["shs"], st.sh_degree, st.campos,
                                               gb, R, bb, ib, True, False)

    def round_():
        outs = _C.rasterize_gaussians_batch([fwd_args(w) for w in wls], dev, streams)
        for w, s, out in zip(wls, streams, outs):
            with torch.cuda.stream(s):
                bwd(w, out)

    rounds = max(10, args.steps)
    for _ in range(4):
        round_()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(rounds):
            round_()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ms_view = best / rounds / V * 1e3
    return {"views_in_flight": V, "host_threads": 1, "ms_per_view": ms_view, "value": w0.P / (ms_view * 1e-3), "unit": "surfels/s",
            "note": "svgir_forward_batch: %d views of the same workload launched from ONE host thread on %d HIP streams, forward + backward, "
                    "wall clock over %d rounds, best of 5; supplementary, the headline value is one view per step" % (V, V, rounds)}


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


def _time_regions(fn, args, dev):
    """W warm-up calls, then `repeats` regions of K calls (synchronize on both sides); returns seconds per region."""
    import torch
    for _ in range(args.warmup):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(max(1, args.repeats or 25)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return out


def bench_train_step(args, dev):
    """One stage-2 training iteration per step (svgir_harness.workloads.TrainStep)."""
    import numpy as np
    import torch
    from gaussian_renderer import _native
    from svgir_harness import workloads
    ts = workloads.TrainStep(dev, radiance_grad=bool(getattr(args, "radiance_grad", False)))
    regions = _time_regions(ts.step, args, dev)
    med = float(np.median(regions))
    # kernel-level stage marks of the library (rasterizer + shading stages) over one more region, and the phase table
    _native.set_profiling(True)
    for _ in range(args.steps):
        R, _, _ = ts.step()
    torch.cuda.synchronize()
    stage = {n: (ms, cnt) for n, ms, cnt in _native.last_timings(with_counts=True)}
    _native.set_profiling(False)
    phases = ts.stage_table(min(10, args.steps))

    class _W:   # what roofline_of / shading_record read
        pass
    w = _W()
    w.P, w.W, w.H, w.S, w.VS, w.variant, w.Ns, w.training = ts.P, ts.W, ts.H, ts.S, ts.VS, "svgss", ts.Ns, True
    w.shading, w.dirs = ts.shading, ts.shading.FibonacciLattice(ts.geo_n, ts.Ns, None)
    w.fused = ts.fused
    counts = (ts.P, ts.P)
    if ts.fused:   # (the same scene and camera as the cfg3_train workload: its working-set sizes)
        w3 = Workload("cfg3_train", dev, 0, 1, args)
        counts = shading_counts(w3)
        del w3
    n_adam = ts.n_param_elems
    res = {
        "metric": "Gaussian-surfels/sec, whole stage-2 training iteration @800x800 (shade + rasterize + unpack + L1/SSIM + backward + Adam)",
        "value": ts.P * args.steps / med, "unit": "surfels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": med / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "repeats": len(regions), "ms_per_step_min": min(regions) / args.steps * 1e3,
        "ms_per_step_max": max(regions) / args.steps * 1e3,
        "config": {"workload": f"train_step: cfg3_train scene (svgss, P={ts.P}, {ts.W}x{ts.H}, S=4, VS=52, Ns={ts.Ns}), one optimisation step per "
                               f"step: FibonacciLattice (random azimuths) -> shade_and_pack -> GaussianRasterizer -> unpack -> l1_ssim -> "
                               f"loss.backward() -> add_densification_stats -> FusedAdam.step(nan scrub, zero_grad) over {n_adam} parameter "
                               f"elements (geometry, SH, SV-BRDF, env map and " +
                               ("the radiance cache as a differentiated leaf: --radiance-grad)" if ts.radiance_grad else
                                "the scalar _radiance_ratio: the radiance cache enters detached as get_radiances = nan_to_num(_radiances.detach() * "
                                "_radiance_ratio), scene/gaussian_model.py:323-324, 526-528 -- it sits in the optimizer but never has a gradient)"),
                   "num_rendered": int(R), "reference": "train.py:133-134, gaussian_renderer/svgss.py:15-262, scene/gaussian_model.py:775-813, 1270-1276"},
        "phase_ms": {k: round(v, 4) for k, v in phases.items()},
        "stage_ms": {k: round(v[0], 4) for k, v in stage.items()},
        "roofline": roofline_of(w, R, stage, "cfg3_train"),
        "shading": {k: v for k, v in shading_record(w, stage, counts).items() if k != "config"},
        "optimizer": {"kernel": "adam_kernel (multi-tensor, NaN scrub)", "algorithmic_bytes_per_launch": 28 * n_adam,
                      "note": "28 B per parameter element (param, grad, two moments read; param, two moments written)"},
    }
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_train_step(ts, args)
    return res


def cpu_baseline_train_step(ts, args):
    """CPU restatement of the same iteration on the host cores: the C++ rasterizer oracle (fwd + bwd, all threads) on the full
    view; the torch-fp64 shading oracle with autograd, the numpy / torch epilogue + L1 / SSIM oracle and torch.optim.Adam on bounded
    samples, scaled to the full sizes (the sample sizes are stated)."""
    import numpy as np
    import torch
    from oracle import epilogue_oracle as eo
    from oracle import oracle as orc
    from oracle import shading_oracle as so
    from svgir_harness import scenes, shade_inputs
    fast = orc.build_fast()
    o = orc.OracleRun(ts.sc, orc.SVGSS, fast=fast)
    g = scenes.upstream_grads(ts.sc, "svgss")
    t0 = time.perf_counter()
    n_r = 2
    for _ in range(n_r):
        o.forward()
        o.backward(g["color"], g["normal"], g["depth"], g["opacity"], g["feature"], g.get("vfeature"))
    t_raster = (time.perf_counter() - t0) / n_r
    # shading forward + backward (torch fp64, autograd) on a sample of the surfels
    n_s = 4000
    d = shade_inputs.make(n_s, ts.Ns, seed=3)
    names = ("base_color", "roughness", "normals", "radiance", "env")
    lo = {k: d[k].double().requires_grad_(True) for k in names}
    t0 = time.perf_counter()
    ref = so.shade(lo["base_color"], lo["roughness"], lo["normals"], d["viewdirs"].double(), lo["radiance"], d["visibility"].double(),
                   d["dirs"].double(), d["areas"].double(), lo["env"])
    (ref["pbr"].sum() + ref["diffuse_light"].sum()).backward()
    t_shade = (time.perf_counter() - t0) * (ts.P / n_s)
    # image-space epilogue + L1 / SSIM with autograd at full resolution
    im = torch.rand(3, ts.H, ts.W, dtype=torch.float64, requires_grad=True)
    gt = torch.rand(3, ts.H, ts.W, dtype=torch.float64)
    t0 = time.perf_counter()
    l1, ss = eo.l1_ssim_torch(im, gt)
    (0.8 * l1 + 0.2 * (1 - ss)).backward()
    t_loss = time.perf_counter() - t0
    # Adam on a sample of the parameter block
    n_a = 4_000_000
    par = torch.nn.Parameter(torch.zeros(n_a))
    opt = torch.optim.Adam([par], lr=1e-3, eps=1e-15)
    par.grad = torch.ones(n_a)
    opt.step()
    t0 = time.perf_counter()
    opt.step()
    t_adam = (time.perf_counter() - t0) * (ts.n_param_elems / n_a)
    total = t_raster + t_shade + t_loss + t_adam
    return {"value": ts.P / total, "unit": "surfels/s", "cores": orc.max_threads(), "kind": "port",
            "seconds": {"rasterizer_fwd_bwd": t_raster, "shading_fwd_bwd_scaled": t_shade, "l1_ssim_fwd_bwd": t_loss, "adam_scaled": t_adam},
            "sample": f"{n_r} fwd+bwd rasterizer steps of the cfg3_train view (OpenMP oracle/svgir_oracle.cpp, {orc.max_threads()} threads); "
                      f"shading oracle (torch fp64 + autograd) on {n_s} of {ts.P} surfels, scaled; L1 + SSIM oracle with autograd at "
                      f"{ts.W}x{ts.H}; torch.optim.Adam on {n_a} of {ts.n_param_elems} elements, scaled (torch CPU threads: {torch.get_num_threads()})"}


def bench_tracers(args, dev):
    """update_visibility + update_radiace of the reference per step (svgir_harness.workloads.TracerCache, cfg3 geometry)."""
    import numpy as np
    import torch
    from svgir_harness import workloads
    tc = workloads.TracerCache(dev)
    args = argparse.Namespace(**vars(args))
    args.steps = min(args.steps, 3); args.warmup = min(args.warmup, 1); args.repeats = min(args.repeats or 3, 3)   # a step is ~1.5 s
    regions = _time_regions(tc.step, args, dev)
    med = float(np.median(regions))
    ms_vis, vis = tc.timed(tc.update_visibility, n=2)
    ms_rad, (rad, vis2, idx) = tc.timed(tc.update_radiance, n=1)
    rays = 2 * tc.rays
    res = {
        "metric": "rays/sec, visibility + radiance cache update (BVH builds + trace_visibility + render_radiance_with_sampling_SH)",
        "value": rays * args.steps / med, "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": med / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "repeats": len(regions),
        "config": {"workload": f"tracers: cfg3 geometry (P={tc.P} flat surfels), {tc.sample_num} incident directions per surfel "
                               f"(sample_incident_rays), the reference's chunk loop (P // 3 surfels per call); one step = update_visibility "
                               f"+ update_radiace = {rays} rays",
                   "reference": "scene/gaussian_model.py:435-522, submodules/bvh/src/{construct,trace}.cu, pbgi/bvhworkers/intersect_test.slang:1879-1990"},
        "producers": {"update_visibility": {"ms": ms_vis, "rays_per_s": tc.rays / (ms_vis * 1e-3), "mean_visibility": float(vis.mean())},
                      "update_radiance": {"ms": ms_rad, "rays_per_s": tc.rays / (ms_rad * 1e-3), "hit_fraction": float((idx >= 0).float().mean()),
                                          "mean_radiance": float(rad.mean())}},
        # data-dependent traversals: the byte model counts what every ray must move (origin / direction in, results out) -- the tree
        # traffic (64 B per visited internal node, 96 B per visited leaf; ~2 400 visits per radiance ray on this scene,
        # profiles/r04_tracer_stats.txt) is served by L2
        "roofline": {"bound": "issue", "kernel": "pbgi_trace_kernel (radiance tracer)", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": tc.rays * (12 + 12 + 4 + 4 + 8), "avg_launch_ms": ms_rad,
                     "achieved": tc.rays * 40 / (ms_rad * 1e-3) / 1e9, "frac": tc.rays * 40 / (ms_rad * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None,
                     "note": "divergent tree traversal: instruction-issue / latency bound (profiles/r04_tracer_pmc.txt), not an HBM stream"},
    }
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_tracers(tc, args)
    return res


def cpu_baseline_tracers(tc, args):
    """The CPU oracles of both tracers on a bounded number of surfel rows (64 rays each), all host threads."""
    import numpy as np
    from oracle import bvh_oracle as bo
    from oracle import pbgi_oracle as po
    from gaussian_renderer import shading
    f = lambda t: t.detach().cpu().numpy()   # noqa: E731
    xyz, scales, rot, nrm, op, cov, shs = (f(t) for t in (tc.xyz, tc.scales, tc.rot, tc.normals, tc.opacity, tc.cov_inv, tc.shs))
    rows = np.sort(np.random.default_rng(1).choice(tc.P, 2000, replace=False))
    dirs, _ = shading.sample_incident_rays(tc.normals[rows], False, tc.sample_num)
    dirs = f(dirs)
    t0 = time.perf_counter()
    info, aabb, _ = po.build(xyz, scales)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    po.trace(info, aabb, xyz[rows], dirs, xyz, scales, rot, nrm, op, cov, shs)
    t_rad = time.perf_counter() - t0
    nv = 40   # (the visibility oracle is a tree-free loop over ALL surfels per ray: O(P) per ray)
    boxes = bo.leaf_boxes(xyz, scales, rot)
    t0 = time.perf_counter()
    bo.trace_visibility(boxes, np.repeat(xyz[rows[:nv], None], tc.sample_num, 1), dirs[:nv], xyz, cov, op.reshape(-1), nrm)
    t_vis = time.perf_counter() - t0
    rays_rad, rays_vis = len(rows) * tc.sample_num, nv * tc.sample_num
    return {"value": rays_rad / t_rad, "unit": "rays/s (radiance tracer)", "cores": os.cpu_count(), "kind": "port",
            "visibility_rays_per_s": rays_vis / t_vis, "pbgi_bvh_build_s": t_build,
            "sample": f"oracle/pbgi_oracle.cpp (OpenMP) on {len(rows)} random rows x {tc.sample_num} rays of the same scene ({t_rad:.1f} s) after a "
                      f"{t_build:.2f} s tree build; oracle/bvh_oracle.cpp (tree-free, O(P) per ray) on {nv} rows x {tc.sample_num} rays ({t_vis:.1f} s)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    from svgir_harness import view_parallel as vp
    cores = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
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

        n1 = None
        if world > 1:   # the same-workload single-rank reference of the N > 1 line (see below): rank 0 alone, the others wait
            vp.barrier()
            if rank == 0:
                sreg, _, _, _ = timed(None, args, world, dev, dry=dry_step, solo=True)
                n1 = float(np.median(sreg))
            vp.barrier()
        regions, R, stage, table = timed(None, args, world, dev, dry=dry_step)
        core_tab = vp.gather_rows(torch.tensor([float(len(cores)), float(cores[0]) if cores else -1.0, float(cores[-1]) if cores else -1.0]))
        if rank == 0:
            med = float(np.median(regions))
            print(json.dumps({"metric": "dry-run (no GPU work)", "value": world * args.steps / med, "unit": "steps/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "repeats": len(regions), "ms_per_step": med / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": name + " (dry run)", "views_per_step": world},
                              "per_rank_R": [int(r[2]) for r in table],
                              "per_rank_cpus": [list(range(int(c[1]), int(c[2]) + 1)) if c[0] > 0 else [] for c in core_tab],
                              **({"n1_same_workload": {"value": args.steps / n1, "unit": "steps/s", "ms_per_step": n1 / args.steps * 1e3},
                                  "scaling_efficiency": (world * args.steps / med) / (world * args.steps / n1)} if n1 else {})}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    if torch.cuda.device_count() <= local or not torch.cuda.is_available():
        print("bench.py needs a GPU per rank (the rasterizer has no CPU path)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if name in ("train_step", "tracers"):
        if world != 1:
            print(f"bench.py: --workload {name} is a single-GPU measurement", file=sys.stderr)
            sys.exit(2)
        print(json.dumps(bench_train_step(args, dev) if name == "train_step" else bench_tracers(args, dev)))
        return
    wl = Workload(name, dev, rank, world, args)
    # N > 1: the line must carry its own single-GPU reference -- the SAME workload (rank 0's view of it) on ONE GPU of the same node,
    # measured by the same process right before the concurrent regions while the other ranks wait at a barrier.  value / (N x that) is the
    # view-parallel scaling efficiency; the N = 1 default line times another workload (cfg2, BASELINE's headline) and is not comparable.
    n1 = None
    if world > 1:
        vp.barrier()
        if rank == 0:
            sa = argparse.Namespace(**vars(args))
            sa.repeats = args.repeats if args.repeats > 0 else 15
            sreg, sR, _, _ = timed(wl, sa, world, dev, solo=True, stages=False)
            n1 = (float(np.median(sreg)), int(sR))
        vp.barrier()
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
                       "per_rank_cpus": "pinned: cores split evenly over the ranks (bench.py pin_rank_to_cores)" if cores else "not pinned",
                       "per_rank_device": [{"rank": i, "ordinal": int(r[0]), "pci": "%04x:%02x:%02x" % (int(r[1]) & 0xffff, int(r[2]) & 0xff, int(r[3]) & 0xff)}
                                           for i, r in enumerate(ids)]},
            "roofline": roofline_of(wl, R, stage, name),
            "stage_ms": {k: round(v[0], 4) for k, v in stage.items()},
        }
        if wl.shade:
            srec = shading_record(wl, stage)
            res["config"]["shading"] = srec["config"]
            res["shading"] = {k: v for k, v in srec.items() if k != "config"}
        if n1:
            v1 = wl.P * args.steps / n1[0]
            res["n1_same_workload"] = {"value": v1, "unit": "surfels/s", "ms_per_step": n1[0] / args.steps * 1e3, "num_rendered": n1[1],
                                       "what": f"{name}, rank 0's view alone on one GPU of this node (the other {world - 1} ranks idle at a barrier), "
                                               "same process, same region protocol, right before the concurrent regions"}
            res["scaling_efficiency"] = res["value"] / (world * v1)
            res["scaling_note"] = ("the scaling curve of this job = value (N views in flight on N GPUs) over N x n1_same_workload.value; the "
                                   "N = 1 default line (cfg2) is BASELINE's headline and a different workload -- its cfg4 record "
                                   "(`configs.cfg4`) is the N = 1 point on another box")
    # the "shaded + blended" number of north_star: cfg3_train with the shading stage, same measurement, extra keys
    if world == 1 and args.workload is None and not args.no_shaded:
        wl.sct = wl.gt = None
        torch.cuda.empty_cache()
        sa = argparse.Namespace(**vars(args))
        w3 = Workload("cfg3_train", dev, rank, world, sa)
        reg3, R3, st3, _ = timed(w3, sa, world, dev)
        m3 = float(np.median(reg3))
        res["shaded"] = {"workload": f"cfg3_train: svgss path, P={w3.P}, {w3.W}x{w3.H}, S={w3.S}, VS={w3.VS}, SV-BRDF shading "
                                     f"(Ns={w3.Ns}) + rasterizer, fwd+bwd (BASELINE.json configs[2])",
                         "value": w3.P * sa.steps / m3, "unit": "surfels/s", "ms_per_step": m3 / sa.steps * 1e3, "repeats": len(reg3),
                         "num_rendered": int(R3), "roofline": roofline_of(w3, R3, st3, "cfg3_train"),
                         "shading": shading_record(w3, st3), "stage_ms": {k: round(v[0], 4) for k, v in st3.items()}}
        res["value_shaded"] = res["shaded"]["value"]   # north_star's "shaded + blended" number (cfg3_train with the SV-BRDF shading)
        del w3
        torch.cuda.empty_cache()
        # the other BASELINE configurations, compactly: cfg3_eval (configs[2], evaluation widths), cfg4 (configs[3]: the workload of the
        # N > 1 line -- this record is its N = 1 point), cfg5 (configs[4]: the one where bandwidth can bind)
        res["configs"] = {}
        for cname in ("cfg3_eval", "cfg4", "cfg5"):
            try:
                res["configs"][cname] = compact_record(cname, dev, args)
            except Exception as e:   # noqa: BLE001 -- supplementary records never cost the headline
                res["configs"][cname] = {"error": repr(e)[:300]}
        # a different camera every step (the reference's training loop), cfg2 and cfg3_train
        res["rotating_views"] = {}
        for cname, fixed in (("cfg2", res["ms_per_step"]), ("cfg3_train", res["shaded"]["ms_per_step"])):
            try:
                res["rotating_views"][cname] = rotating_views(cname, dev, args, fixed)
            except Exception as e:   # noqa: BLE001
                res["rotating_views"][cname] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_concurrent:   # (last: it creates streams of its own)
        torch.cuda.empty_cache()
        # (HIP streams share a few hardware queues, dealt in creation order: each record creates its streams right before it runs)
        ob = one_thread_batch(name, dev, args)
        if ob:
            res["one_thread_batch"] = ob
        torch.cuda.empty_cache()
        res["two_streams"] = two_streams(name, dev, args)
    # (last: the OpenMP oracle leaves its worker threads spinning for a while, which would slow the host-side launch loops above)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(wl, args)
    if rank == 0:
        print(json.dumps(res))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
