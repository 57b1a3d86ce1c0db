"""Per-kernel averages of the PMC counters collected by scripts/pmc_probe.sh (rocprofv3 --pmc, csv output).
usage: python scripts/pmc_summary.py gpurun_out/<tag>_a gpurun_out/<tag>_b [name-substring ...]"""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def main():
    dirs = [a for a in sys.argv[1:] if "/" in a]
    pats = [a for a in sys.argv[1:] if "/" not in a] or ["render_", "shade_", "radix", "column_scan", "emit", "geom_bwd"]
    acc, dur = defaultdict(dict), {}
    for d in dirs:
        a, du = load(d)
        for k, v in a.items():
            for c, vals in v.items():
                acc[k][c] = sum(vals) / len(vals)
            dur[k] = sum(du[k]) / len(du[k])
    for k in sorted(acc, key=lambda k: -dur.get(k, 0)):
        if not any(p in k for p in pats):
            continue
        c = acc[k]
        short = k.replace("svgir::(anonymous namespace)::", "").split("(")[0][:48]
        line = f"{short:48s} us={dur[k]:8.1f}"
        w = c.get("SQ_WAVES")
        if w:
            line += f" waves={w:9.0f}"
        busy = c.get("SQ_BUSY_CYCLES")       # summed over SEs/XCDs: per-XCD busy cycles
        gui = c.get("GRBM_GUI_ACTIVE")
        if "SQ_WAVE_CYCLES" in c and busy:
            line += f" wavecyc/busy={c['SQ_WAVE_CYCLES'] / busy:7.2f}"
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
            if n in c and w:
                line += f" {n[9:]}/w={c[n] / w:8.1f}"
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
                line += f" {n[3:]}/wc={c[n] / c['SQ_WAVE_CYCLES']:.3f}"
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            line += f" ldsconf={c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}"
        if gui:
            line += f" gui={gui:.0f}"
        print(line)
        print("      raw:", {n: round(v) for n, v in sorted(c.items())})


if __name__ == "__main__":
    main()
