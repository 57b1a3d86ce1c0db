"""Synthetic scenes for the pbgi radiance tracer: small hollow shells of surfels whose normals point inward, so that rays
leaving a surfel along its normal hemisphere find other surfels within the tracer's 0.2 range."""
import numpy as np


def quat_from_z(n):
    """Unit quaternions (r, x, y, z) whose rotation maps (0,0,1) to n (n: [P,3] unit vectors), with a random twist."""
    z = np.array([0.0, 0.0, 1.0])
    h = n + z
    bad = np.linalg.norm(h, axis=1) < 1e-6
    h[bad] = np.array([1.0, 0.0, 0.0])
    h /= np.linalg.norm(h, axis=1, keepdims=True)
    # rotation by pi about h maps z to n: q = (0, h)
    return np.concatenate([np.zeros((n.shape[0], 1)), h], axis=1)


def rotmat(q):
    q = q / np.sqrt((q * q).sum(1, keepdims=True) + 1e-8)
    r, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                     np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                     np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)


def make(P=2000, shells=20, S=16, seed=0, radius=0.08, duplicates=0):
    rng = np.random.default_rng(seed)
    per = P // shells
    P = per * shells
    centres = rng.uniform(-1, 1, size=(shells, 3))
    u = rng.normal(size=(P, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    xyz = np.repeat(centres, per, axis=0) + u * radius * rng.uniform(0.85, 1.0, size=(P, 1))
    normals = -u                                             # inward
    q = quat_from_z(normals.copy()) * rng.uniform(0.5, 2.0, size=(P, 1))   # un-normalised on purpose
    scales = np.exp(rng.uniform(np.log(0.008), np.log(0.03), size=(P, 3))); scales[:, 2] = 1e-3
    if duplicates:                                           # identical centres: equal Morton codes
        xyz[:duplicates] = xyz[0]
    R = rotmat(q)
    inv = R @ (np.eye(3)[None] / (scales ** 2)[:, None, :]) @ np.transpose(R, (0, 2, 1))
    cov_inv = np.stack([inv[:, 0, 0], inv[:, 0, 1], inv[:, 0, 2], inv[:, 1, 1], inv[:, 1, 2], inv[:, 2, 2]], -1)
    opacity = rng.uniform(0.2, 0.95, size=(P, 1))
    shs = rng.normal(size=(P, 16, 3)) * 0.3; shs[:, 0] += 1.0
    # rays: cosine-ish hemisphere about the normal of every surfel
    d = rng.normal(size=(P, S, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    flip = (d * normals[:, None]).sum(-1, keepdims=True) < 0
    d = np.where(flip, -d, d) * rng.uniform(0.5, 2.0, size=(P, S, 1))      # un-normalised on purpose
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(P=P, S=S, xyz=f(xyz), scales=f(scales), rot=f(q), normals=f(normals * rng.uniform(0.5, 2.0, size=(P, 1))), opacity=f(opacity),
                cov_inv=f(cov_inv), shs=f(shs), ray_d=f(d))
