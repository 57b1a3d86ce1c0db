"""world_size-2 gloo test of the view-parallel driver (CPU).  The renderer injected here is the CPU oracle (test
infrastructure); the driver logic -- sharding, single fused all_gather per step, replicated scene -- is what runs
on the GPUs with RCCL."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import oracle as orc
    from svgir_harness import cameras, scenes, view_parallel as vp
    r, w, _ = vp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    base = scenes.surface_scene(P=800, W=64, H=48, seed=3, sh_degree=1, variant="svgss", S=1, VS=4, scale_lo=0.03,
                                scale_hi=0.1)
    # rank 0 owns the "real" Gaussians; everybody else starts from garbage and receives the broadcast
    t = {k: torch.from_numpy(v.copy()) for k, v in base.items() if isinstance(v, np.ndarray) and k in
         ("means3D", "scales", "rotations", "opacities", "shs", "features", "vfeatures")}
    if rank != 0:
        for k in t:
            t[k].zero_()
    vp.broadcast_scene(t)
    for k in t:
        assert np.array_equal(t[k].numpy(), base[k]), k

    def render_view(v):
        sc = dict(base)
        sc.update({k: x.numpy() for k, x in t.items()})
        sc.update(cameras.make_camera(64, 48, cameras.orbit_eye(4.0, 45.0 * v, 30.0)))
        o = orc.OracleRun(sc, orc.SVGSS, num_threads=1)
        R = o.forward()
        return torch.tensor([float(R), float(o.images()["color"].sum())])

    table = vp.run_views(render_view, 5, rank, world, dev, 2)
    assert vp.shard_views(5, rank, world) == [v for v in range(5) if v % world == rank]
    # asynchronous per-step metrics gather (what bench.py uses): results lag the submits by at most one step
    g = vp.MetricsGatherer(3, dev)
    for step in range(4):
        g.submit(torch.tensor([float(rank), float(step), 10.0 * rank + step]))
    got = g.results()
    assert got.shape == (world, 3)
    for r_ in range(world):
        assert got[r_].tolist() == [float(r_), 3.0, 10.0 * r_ + 3.0], got
    g.drain()
    vp.barrier()
    q.put((rank, table.numpy()))


def test_view_parallel_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank ends with the same full table, every view rendered exactly once
    assert np.array_equal(res[0], res[1])
    assert not np.isnan(res[0]).any()
    # and it equals the single-process result
    for p_ in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    from oracle import oracle as orc
    from svgir_harness import cameras, scenes
    base = scenes.surface_scene(P=800, W=64, H=48, seed=3, sh_degree=1, variant="svgss", S=1, VS=4, scale_lo=0.03,
                                scale_hi=0.1)
    for v in range(5):
        sc = dict(base)
        sc.update(cameras.make_camera(64, 48, cameras.orbit_eye(4.0, 45.0 * v, 30.0)))
        o = orc.OracleRun(sc, orc.SVGSS, num_threads=1)
        R = o.forward()
        assert res[0][v, 0] == R
        np.testing.assert_allclose(res[0][v, 1], o.images()["color"].sum(), rtol=1e-6)


def test_bench_self_launches_ranks_dry_run():
    """`python bench.py --gpus 2` without torchrun starts the two ranks itself (fresh processes, before anything touches
    the GPU), runs the timed regions with the per-step metrics all_gather (gloo here, RCCL on the GPUs) and prints ONE
    JSON line with n_gpus = 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1",
                          "--repeats", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["repeats"] == 3 and d["scaling"] == "weak"
    assert d["per_rank_R"] == [100, 101] and d["config"]["workload"].startswith("cfg4")
    # the N > 1 line carries its own single-rank reference of the same workload (rank 0 alone, the others wait at a barrier): the two
    # numbers that form the scaling curve are `value` and N x n1_same_workload.value
    assert d["n1_same_workload"]["value"] > 0 and d["n1_same_workload"]["ms_per_step"] > 0
    assert abs(d["scaling_efficiency"] - d["value"] / (2 * d["n1_same_workload"]["value"])) < 1e-9
    # a request for more GPUs than the machine has fails loudly instead of silently running one rank
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s) are visible" in out.stderr


def _worker8(rank, world, port, q):
    for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from svgir_harness import view_parallel as vp
    vp.init_from_env(backend="gloo")
    dev = torch.device("cpu")
    calls = []

    def metrics(v):
        return torch.tensor([float(v), float(v * v % 97), float(rank)])

    def render_view(v):
        calls.append(v)
        return metrics(v)

    def render_batch(vs):
        calls.append(tuple(vs))
        return [metrics(v) for v in vs]

    t1 = vp.run_views(render_view, 200, rank, world, dev, 3)
    assert calls == vp.shard_views(200, rank, world) and len(calls) == 25       # a 200-frame TensoIR test set: 25 views per rank
    calls.clear()
    t2 = vp.run_views(None, 200, rank, world, dev, 3, render_batch=render_batch, batch=4)
    assert [v for c in calls for v in c] == vp.shard_views(200, rank, world) and max(len(c) for c in calls) == 4 and len(calls) == 7
    t3 = vp.run_views(render_view, 13, rank, world, dev, 3)                      # fewer views than 2 x world: ragged shares
    vp.barrier()
    q.put((rank, t1.numpy(), t2.numpy(), t3.numpy()))


def test_view_parallel_world8_gloo_200_views():
    """8-GPU readiness that does not need the node: eight ranks (gloo), the 200 views of a TensoIR test set, 25 per rank, one
    all_gather at the end (no per-round collective, no host synchronisation inside the loop), per-view and batched drivers."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29820 + (os.getpid() % 150)
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    v = np.arange(200, dtype=np.float32)
    exp = np.stack([v, (v * v) % 97, v % 8], axis=1)
    for rank, t1, t2, t3 in res:
        assert np.array_equal(t1, exp) and np.array_equal(t2, exp), rank
        assert np.array_equal(t3, exp[:13]), rank


def test_bench_dry_run_world8():
    """`python bench.py --gpus 8 --dry-run`: the launcher starts eight ranks, every region carries the barrier + metrics
    all_gather, rank 0 prints ONE line with n_gpus = 8 and the aggregate over the eight ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1",
                          "--repeats", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["per_rank_R"] == [100 + r for r in range(8)]
    assert d["config"]["workload"].startswith("cfg4") and len(d["per_rank_cpus"]) == 8
    # disjoint core sets (or all unpinned where the machine has fewer cores than ranks x 1)
    cpus = [set(c) for c in d["per_rank_cpus"] if c]
    assert all(a.isdisjoint(b) for i, a in enumerate(cpus) for b in cpus[i + 1:])
