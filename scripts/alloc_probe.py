"""GPU probe: wall time spent inside the blob-allocation callbacks vs the 'emit' stage time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from gaussian_renderer import _native as N
from svgir_harness import runner, scenes

times = []
orig = N.BlobAllocator.fn
def fn(self, name):
    def alloc(nbytes, _ctx):
        t0 = time.perf_counter()
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self.tensors[name] = t
        times.append((name, int(nbytes), time.perf_counter() - t0))
        return t.data_ptr()
    f = N.ALLOC_FN(alloc)
    self._fns[name] = f
    return f
N.BlobAllocator.fn = fn

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
variant = scenes.CONFIGS[name][1]["variant"]
sc = scenes.make(name)
sct = runner.to_torch(sc, dev)
N.set_profiling(True)
for it in range(6):
    times.clear()
    t0 = time.perf_counter()
    out, leaves = runner.render(sct, variant, requires_grad=False)
    torch.cuda.synchronize()
    print(it, "forward wall %.3f ms" % ((time.perf_counter() - t0) * 1e3), [(n, b >> 20, round(t * 1e6)) for n, b, t in times])
    del out
print({k: round(v, 4) for k, v in N.last_timings()})
