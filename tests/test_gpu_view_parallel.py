"""The view-parallel driver on ONE real GPU with backend "nccl" (= RCCL): process-group creation with the device bound first,
`broadcast_scene`, `gather_metrics`, the asynchronous `MetricsGatherer` and `run_views` all issue their collectives on HIP tensors
(SVGIR_VP_FORCE_COLLECTIVES=1 -- with one rank they would otherwise return early), and `bench.py` joins a world of one under
torchrun-style environment variables.  The multi-rank logic itself is covered on the CPU (tests/test_view_parallel.py, gloo, world 2);
this test makes sure the first multi-GPU run is not also the first RCCL run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys
ROOT = sys.argv[1]
for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
from svgir_harness import cameras, runner, scenes, view_parallel as vp
assert vp.FORCE
rank, world, local = vp.init_from_env()              # backend None -> "nccl" on a GPU box
assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
assert torch.cuda.current_device() == 0
base = scenes.surface_scene(P=3000, W=128, H=96, seed=3, sh_degree=1, variant="svgss", S=1, VS=4, scale_lo=0.03, scale_hi=0.1)
sct = runner.to_torch(base, dev)
per_gaussian = {k: sct[k] for k in ("means3D", "scales", "rotations", "opacities", "shs", "features", "vfeatures")}
before = {k: v.clone() for k, v in per_gaussian.items()}
vp.broadcast_scene(per_gaussian)                      # RCCL broadcast on HIP tensors (root = this rank: values unchanged)
for k in before:
    assert torch.equal(per_gaussian[k], before[k]), k

def render_view(v):
    sc = dict(base)
    sc.update(cameras.make_camera(128, 96, cameras.orbit_eye(4.0, 45.0 * v, 30.0)))
    out, _ = runner.render(runner.to_torch(sc, dev), "svgss")
    return torch.stack([torch.tensor(float(out["num_rendered"]), device=dev), out["color"].sum()])

table = vp.run_views(render_view, 3, rank, world, dev, 2)          # one all_gather_into_tensor per round
assert table.shape == (3, 2) and not torch.isnan(table).any()
for v in range(3):
    assert torch.allclose(table[v], render_view(v), rtol=1e-6)
g = vp.MetricsGatherer(3, dev)                                     # async all_gather, waited for one step later
assert g.collective
for step in range(5):
    g.submit(torch.tensor([1.0, float(step), 10.0 + step], device=dev))
assert g.results().tolist() == [[1.0, 4.0, 14.0]]
g.drain()
rows = vp.gather_rows(torch.arange(4, dtype=torch.float64, device=dev))
assert rows.tolist() == [[0.0, 1.0, 2.0, 3.0]]
vp.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def _env(port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SVGIR_VP_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_view_parallel_driver_on_rccl_world_of_one(built):
    out = subprocess.run([sys.executable, "-c", _SCRIPT, ROOT], env=_env(29700 + os.getpid() % 200), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_joins_a_torchrun_world_of_one_with_rccl(built):
    """bench.py under torchrun-style variables (WORLD_SIZE set: it joins the given world instead of launching ranks), collectives
    forced: the timed regions run with the per-step metrics all_gather and the barriers on RCCL."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "2",
                          "--workload", "cfg2", "--no-cpu-baseline", "--no-shaded", "--no-concurrent"],
                         env=_env(29900 + os.getpid() % 100), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["per_rank_num_rendered"] == [d["config"]["num_rendered"]]
