"""CPU analysis of the composite workload of a config (uses the oracle): per-tile list lengths, per-wave (8x8)
candidate counts after the conservative cull, exact any-pixel-pass counts and blended pairs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as orc
from svgir_harness import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
variant = scenes.CONFIGS[name][1]["variant"]
sc = scenes.make(name)
o = orc.OracleRun(sc, orc.SVGSS if variant == "svgss" else orc.RGSS)
R = o.forward()
P = sc["means3D"].shape[0]
W, H = sc["W"], sc["H"]
gx = (W + 15) // 16
rg = o.get("ranges").reshape(-1, 2).astype(np.int64)
pl = o.get("point_list")
m2 = o.get("means2D").reshape(P, 2).astype(np.float64)
co = o.get("conic_opacity").reshape(P, 4).astype(np.float64)
ncon = o.get("n_contrib").reshape(H, W)
lens = rg[:, 1] - rg[:, 0]
ne = np.nonzero(lens)[0]
print(name, "R", R, "tiles", len(lens), "nonempty", len(ne), "len mean", lens[ne].mean(), "max", lens.max(), "p90", np.quantile(lens[ne], 0.9))
order = ne[np.argsort(-lens[ne])]
rng = np.random.default_rng(0)
sample = list(order[:10]) + list(rng.choice(ne, size=60, replace=False))
tot = dict(len=0, wave_slots=0, cand=0, anypass=0, pairs_pass=0, pairs_eval=0, proc=0)
rows = []
for t in sample:
    ids = pl[rg[t, 0]:rg[t, 1]]
    tx, ty = t % gx, t // gx
    st = dict(len=len(ids), cand=0, anypass=0, pairs_pass=0, proc=0)
    for wv in range(4):
        x0, y0 = tx * 16 + (wv & 1) * 8, ty * 16 + (wv >> 1) * 8
        xs, ys = np.meshgrid(np.arange(x0, x0 + 8), np.arange(y0, y0 + 8))
        xs, ys = xs.reshape(-1), ys.reshape(-1)
        dx = m2[ids, 0][:, None] - xs[None]; dy = m2[ids, 1][:, None] - ys[None]
        a, b, c, op = co[ids, 0][:, None], co[ids, 1][:, None], co[ids, 2][:, None], co[ids, 3][:, None]
        q = a * dx * dx + 2 * b * dx * dy + c * dy * dy
        passm = (op * np.exp(-0.5 * q) >= 1 / 255.0)
        anyp = passm.any(1)
        # conservative: fine-sample the rectangle (approximates the analytic min)
        # walked prefix: up to the wave's deepest contributor
        nc = ncon[y0:y0 + 8, x0:x0 + 8]
        wmax = nc.max() if nc.size else 0
        st["anypass"] += int(anyp[:wmax].sum()); st["pairs_pass"] += int(passm[:wmax].sum()); st["proc"] += int(wmax)
        uu = np.linspace(0, 7, 29)
        gxs, gys = np.meshgrid(x0 + uu, y0 + uu)
        dxf = m2[ids, 0][:, None] - gxs.reshape(-1)[None]; dyf = m2[ids, 1][:, None] - gys.reshape(-1)[None]
        qf = (a * dxf * dxf + 2 * b * dxf * dyf + c * dyf * dyf).min(1)
        cand = qf <= 2 * np.log(np.maximum(255 * op[:, 0], 1e-30)) + 0.02
        st["cand"] += int(cand[:wmax].sum())
    rows.append((t, st))
    for k in st: tot[k] += st[k]
print("heaviest tiles:")
for t, st in rows[:10]:
    print("  tile", t, st, "cand/len/4 %.2f anypass/cand %.2f pairs/anypass %.1f" % (st["cand"] / max(1, 4 * st["len"]), st["anypass"] / max(1, st["cand"]), st["pairs_pass"] / max(1, st["anypass"])))
print("sample totals", tot, "cand per wave-slot %.3f ; walked fraction %.3f; anypass/cand %.3f ; pairs per anypass %.1f" % (
    tot["cand"] / (4 * tot["len"]), tot["proc"] / (4 * tot["len"]), tot["anypass"] / tot["cand"], tot["pairs_pass"] / tot["anypass"]))
