"""Error statistics shared by the GPU parity tests and scripts/parity_report.py.

north_star: "match the reference within 1e-4 relative fp32".  Two views of the same comparison:
  * normalised: |a-b| <= tol * (max|b| + |b|) -- what a consumer of the image / gradient tensor sees;
  * element-wise relative on the entries that carry signal (|b| > 1e-3 max|b|): |a-b| <= rel_tol * |b|.
Entries outside either band are "flips": an alpha >= 1/255 or T < 1e-4 decision that fell the other way for an isolated
(pixel, splat) pair because exp differs in its last bit.  Their share and their size are both bounded."""
import numpy as np


def stats(a, b, tol=1e-4, rel_tol=1e-3):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return dict(n=0, n_signal=0, scale=0.0, max_abs=0.0, max_norm=0.0, p9999_norm=0.0, flip_frac=0.0, rel_frac=0.0, p9999_rel=0.0, tail_q=0.99,
                    tail_rel=0.0, tail_norm=0.0, finite=True)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    norm = err / (scale + np.abs(b))
    big = np.abs(b) > 1e-3 * scale
    rel = err[big] / np.abs(b[big]) if big.any() else np.zeros(0)
    # `tail`: the extreme quantile a sample of this size supports (the p99.99 of a few thousand entries is its maximum: one element)
    q = tail_quantile(int(big.sum()))
    return dict(n=int(b.size), n_signal=int(big.sum()), scale=scale, max_abs=float(err.max()), max_norm=float(norm.max()),
                p9999_norm=float(np.quantile(norm, 0.9999)), flip_frac=float((norm > tol).mean()),
                rel_frac=float((rel > rel_tol).mean()) if rel.size else 0.0,
                p9999_rel=float(np.quantile(rel, 0.9999)) if rel.size else 0.0,
                tail_q=q, tail_rel=float(np.quantile(rel, q)) if rel.size else 0.0, tail_norm=float(np.quantile(norm, tail_quantile(int(b.size)))),
                finite=bool(np.isfinite(a).all()))


def tail_quantile(n):
    """p99.99 from 100 000 entries on, p99.9 from 10 000, else p99: at least ~10 entries beyond the quantile."""
    return 0.9999 if n >= 100000 else (0.999 if n >= 10000 else 0.99)
