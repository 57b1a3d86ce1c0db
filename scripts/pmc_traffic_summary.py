"""Per-kernel HBM bytes per launch from rocprofv3 --pmc TCC_EA0_* passes (csv).
read bytes  = 2 * 64 * (RDREQ - RDREQ_32B) + 32 * RDREQ_32B   (gfx950: 128-byte requests are tallied as 64 B --
              MI355X_MICROARCH.md, HBM section -- hence the factor 2 on the non-32B requests)
write bytes = 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B)         (uncalibrated on gfx950, reported as counted)
usage: pmc_traffic_summary.py <rd_dir> <wr_dir> <workload>"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # its hashes: the measurement is only valid for the kernel sources AND the library build it was taken with


def kernel_source_hash():
    return bench.kernel_source_hash()


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, launches=max(len(v) for v in cs.values())) for k, cs in acc.items()}


rd, wr = load(sys.argv[1]), load(sys.argv[2])
out = {"workload": sys.argv[3], "kernel_source_hash": kernel_source_hash(), "library_hash": bench.library_hash(), "method": __doc__.split("usage")[0].strip(), "kernels": {}}
for k in rd:
    if not any(p in k for p in ("render_", "cull_kernel", "contrib_prepass", "shade_fwd", "shade_bwd", "grad_reduce")):
        continue
    r, w = rd[k], wr.get(k, {})
    rq, r32 = r.get("TCC_EA0_RDREQ_sum", 0.0), r.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    wq, w64 = w.get("TCC_EA0_WRREQ_sum", 0.0), w.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    short = k.replace("svgir::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    out["kernels"][short] = {"read_bytes": 2 * 64 * (rq - r32) + 32 * r32, "write_bytes": 64 * w64 + 32 * (wq - w64),
                             "launches": int(r.get("launches", 0)),   # (the composite forward exists in two variants: the steady state is the one with most launches)
                             "raw": {"RDREQ": rq, "RDREQ_32B": r32, "WRREQ": wq, "WRREQ_64B": w64}}
print(json.dumps(out))
