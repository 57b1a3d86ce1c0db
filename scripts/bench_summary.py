"""Prints one line per bench JSON (gpurun_out/<dir>/*.json): value, ms/step, shaded surfels, stage times."""
import glob
import json
import sys

for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        print(f, "ERR", e)
        continue
    sh = r.get("shading") or (r.get("shaded") or {}).get("shading") or {}
    print(f"{f.split('/')[-1]:32s} {r['value'] / 1e6:8.1f} M  {r['ms_per_step']:8.4f} ms  "
          + (f"shaded {r['value_shaded'] / 1e6:.1f} M {r['shaded']['ms_per_step']:.4f} ms  " if "value_shaded" in r else "")
          + f"fwd {sh.get('surfels_shaded_fwd')} bwd {sh.get('surfels_shaded_bwd')}")
    st = r["shaded"]["stage_ms"] if "shaded" in r else r.get("stage_ms", {})
    print("      " + " ".join(f"{k}={v * 1e3:.0f}" for k, v in st.items()))
    if "phase_ms" in r:
        print("      phases " + " ".join(f"{k}={v * 1e3:.0f}" for k, v in r["phase_ms"].items()))
