"""Per-kernel instruction counts per launch from a rocprofv3 --pmc SQ_INSTS_* pass (csv), stamped with the hashes of the kernel
sources and of the library they were measured with (bench.py reports roofline.issue_frac from it only while both still match).
usage: pmc_issue_summary.py <dir> <workload>"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"workload": sys.argv[2], "kernel_source_hash": bench.kernel_source_hash(), "library_hash": bench.library_hash(),
       "method": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_{VALU,SALU,LDS,VMEM,SMEM} (+ MFMA MOPS), wave-level instruction counts per launch "
                 "(averaged over the launches of the run); durations of this profiled run in us", "kernels": {}}
for k, cs in acc.items():
    if not any(p in k for p in ("render_", "cull_kernel", "contrib_prepass", "shade_fwd", "shade_bwd", "grad_reduce", "geom_bwd", "preprocess")):
        continue
    short = k.replace("svgir::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    out["kernels"][short] = {"waves": c.get("SQ_WAVES", 0.0), "valu": c.get("SQ_INSTS_VALU", 0.0), "salu": c.get("SQ_INSTS_SALU", 0.0),
                             "lds": c.get("SQ_INSTS_LDS", 0.0), "vmem": c.get("SQ_INSTS_VMEM", 0.0), "smem": c.get("SQ_INSTS_SMEM", 0.0),
                             "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), "profiled_us": sum(dur[k]) / len(dur[k]),
                             "launches": len(dur[k])}
print(json.dumps(out))
