"""View-parallel execution: independent camera views are sharded across ranks (one process per GPU).

The reference is single-GPU; views are its natural independent unit (one camera per training
iteration, train.py:127-135; independent frames in eval, eval_relighting_tensoIR.py:303-378).  All per-Gaussian
arrays are replicated on every rank (one broadcast), rank r renders views {v : v mod world == r}, and there is
NO collective inside forward/backward.  The only data-path-adjacent collective is one fused all_gather of a small
fp32 metrics vector per step (loss / checksum / timings) -- RCCL over xGMI on GPUs, gloo in the CPU tests;
latency-bound (tens of bytes), so it is a single call, never one call per metric.
"""
import os

import torch
import torch.distributed as dist


# With one rank nothing needs exchanging and every helper below returns at once.  SVGIR_VP_FORCE_COLLECTIVES=1 makes a one-rank
# job create its process group and issue the collectives anyway: tests/test_gpu_view_parallel.py runs the whole driver on ONE
# real GPU with backend "nccl" (RCCL init, device binding, broadcast / all_gather_into_tensor on HIP tensors, async work handles),
# so that the first multi-GPU run is not also the first RCCL run.
FORCE = os.environ.get("SVGIR_VP_FORCE_COLLECTIVES", "") not in ("", "0")


def _collective():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":   # one process per GPU: bind the device BEFORE the communicator is created
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(num_views, rank, world):
    """Round-robin ownership: view v belongs to rank v % world."""
    return [v for v in range(num_views) if v % world == rank]


def broadcast_scene(tensors, src=0):
    """One-time replication of the per-Gaussian arrays (dict name -> tensor), in place."""
    if not _collective():
        return tensors
    for k in sorted(tensors):
        if torch.is_tensor(tensors[k]):
            dist.broadcast(tensors[k], src=src)
    return tensors


def gather_metrics(vec):
    """vec: 1-D fp32 tensor of k per-rank scalars -> [world, k] on every rank with ONE all_gather."""
    if not _collective():
        return vec.reshape(1, -1).clone()
    world = dist.get_world_size()
    out = torch.empty(world * vec.numel(), dtype=vec.dtype, device=vec.device)
    dist.all_gather_into_tensor(out, vec.contiguous().reshape(-1))
    return out.reshape(world, vec.numel())


class MetricsGatherer:
    """One fused all_gather of a k-float metrics vector per step, issued asynchronously: the collective of step i
    overlaps with the rendering of step i+1 and is waited for one step later (`results()` drains the last one).
    Nothing in the data path depends on it, so no rank ever stalls on the collective."""

    def __init__(self, k, device, dtype=torch.float32):
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.collective = _collective()
        self.k = k
        self.bufs = [torch.zeros(self.world * k, dtype=dtype, device=device) for _ in range(2)]
        self.src = [torch.zeros(k, dtype=dtype, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.step = 0
        self.last = None

    def submit(self, vec):
        i = self.step & 1
        if self.pending[i] is not None:          # the gather issued two steps ago used this slot
            self.pending[i].wait()
            self.pending[i] = None
        if not self.collective:       # nothing to exchange: the table is the vector itself
            self.bufs[i] = vec.reshape(-1)
        else:
            self.src[i].copy_(vec.reshape(-1))
            self.pending[i] = dist.all_gather_into_tensor(self.bufs[i], self.src[i], async_op=True)
        self.last = i
        self.step += 1

    def results(self):
        """[world, k] table of the most recent submit (waits for it)."""
        if self.last is None:
            return None
        if self.pending[self.last] is not None:
            self.pending[self.last].wait()
            self.pending[self.last] = None
        return self.bufs[self.last].reshape(self.world, self.k)

    def drain(self):
        for i in (0, 1):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None


def gather_rows(vec):
    """[world, k] table of every rank's 1-D `vec` (blocking; one all_gather)."""
    vec = vec.reshape(-1)
    if not _collective():
        return vec[None, :].clone()
    out = torch.empty(dist.get_world_size() * vec.numel(), dtype=vec.dtype, device=vec.device)
    dist.all_gather_into_tensor(out, vec.contiguous())
    return out.reshape(dist.get_world_size(), vec.numel())


def barrier():
    if _collective():
        dist.barrier()


def run_views(render_view, num_views, rank, world, device, metrics_dim, render_batch=None, batch=4):
    """Renders this rank's share of `num_views` and returns the [num_views, metrics_dim] table assembled from all ranks (rows of
    views no rank owns stay NaN).  render_view(v) -> 1-D fp32 tensor[metrics_dim]; or render_batch([v, ...]) -> list of such
    tensors for up to `batch` views at a time (one host thread keeping several views in flight: render_views_in_flight /
    svgir_forward_batch).  No collective and no host synchronisation while the views render: every rank keeps its rows on the
    device and ONE all_gather at the end assembles the table (the reference's evaluation loop is sequential on one GPU,
    eval_relighting_tensoIR.py:303-378)."""
    mine = shard_views(num_views, rank, world)
    n_max = (num_views + world - 1) // world           # rows every rank contributes (padded: view id -1)
    local = torch.zeros((n_max, 1 + metrics_dim), dtype=torch.float32, device=device)
    local[:, 0] = -1.0
    if render_batch is not None:
        done = 0
        for c0 in range(0, len(mine), max(1, batch)):
            chunk = mine[c0:c0 + max(1, batch)]
            ms = render_batch(chunk)
            assert len(ms) == len(chunk)
            for v, m in zip(chunk, ms):
                local[done, 0] = float(v)
                local[done, 1:] = m.to(device=device, dtype=torch.float32).reshape(-1)
                done += 1
    else:
        for i, v in enumerate(mine):
            local[i, 0] = float(v)
            local[i, 1:] = render_view(v).to(device=device, dtype=torch.float32).reshape(-1)
    allr = gather_rows(local.reshape(-1)).reshape(-1, 1 + metrics_dim)     # the one collective
    table = torch.full((num_views + 1, metrics_dim), float("nan"), dtype=torch.float32, device=device)
    ids = allr[:, 0].to(torch.int64)
    ids = torch.where(ids >= 0, ids, torch.full_like(ids, num_views))         # padding rows land in the spare last row
    table[ids] = allr[:, 1:]
    return table[:num_views]


def render_views_in_flight(make_call, finish, views, device, rasterize_batch, in_flight=4, streams=None):
    """Renders `views` on ONE GPU from ONE host thread with up to `in_flight` of them in flight (svgir_forward_batch): for each
    group, make_call(v) -> (args, kwargs) of the binding's rasterize_gaussians is evaluated with the view's stream current,
    `rasterize_batch` (= `_C.rasterize_gaussians_batch` of either binding) launches the whole group before it waits for the first
    view's instance count, and finish(v, result_tuple) -> anything runs on the view's stream again.  Returns finish's results in
    the order of `views`.  The worker streams first wait for the caller's current stream and are joined into it at the end."""
    views = list(views)
    cur = torch.cuda.current_stream(device)
    if streams is None:
        streams = [torch.cuda.Stream(device) for _ in range(max(1, min(in_flight, len(views))))]
    for s in streams:
        s.wait_stream(cur)
    out = []
    for c0 in range(0, len(views), len(streams)):
        chunk = views[c0:c0 + len(streams)]
        calls = []
        for i, v in enumerate(chunk):
            with torch.cuda.stream(streams[i]):
                calls.append(make_call(v))
        res = rasterize_batch(calls, device, streams[:len(chunk)])
        for i, v in enumerate(chunk):
            with torch.cuda.stream(streams[i]):
                out.append(finish(v, res[i]))
    for s in streams:
        cur.wait_stream(s)
    return out


def render_in_flight(render_view, views, device, in_flight=2):
    """Renders `views` on ONE GPU keeping `in_flight` of them in flight: one HIP stream and one host thread each.

    A single view leaves the SIMDs under-occupied (the composite kernels run fewer than three waves per SIMD on the BASELINE
    scenes, HISTORY.md 4); with two views in flight the same GPU renders 1.3-1.4x as many views per second (bench.py
    `two_streams`; 4+ gain nothing: the host threads serialise on the per-view read-back).  The library's shared state
    (capacity cache, side stream, allocator callbacks) is thread-safe: tests/test_gpu_concurrency.py.

    render_view(v) runs inside `torch.cuda.stream(...)` of its worker and returns anything; results come back as a list in
    the order of `views`.  Exceptions of a worker are re-raised here.  Every worker stream first waits for the caller's
    current stream (parameters just written by the optimizer launch, densify appends, uploaded cameras are complete before a
    view reads them), and the caller's stream waits for the workers at the end."""
    import threading
    views = list(views)
    out, err = [None] * len(views), []
    nxt = iter(range(len(views)))
    lock = threading.Lock()
    cur = torch.cuda.current_stream(device)   # (captured here: the workers' "current stream" is their own)
    streams = []

    def worker():
        s = torch.cuda.Stream(device)
        s.wait_stream(cur)
        with lock:
            streams.append(s)
        with torch.cuda.stream(s):
            while True:
                with lock:
                    i = next(nxt, None)
                if i is None or err:
                    break
                try:
                    out[i] = render_view(views[i])
                except BaseException as e:   # noqa: BLE001 -- handed to the caller
                    err.append(e)
                    break
            s.synchronize()

    th = [threading.Thread(target=worker) for _ in range(max(1, min(in_flight, len(views))))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for s in streams:
        cur.wait_stream(s)
    if err:
        raise err[0]
    return out
