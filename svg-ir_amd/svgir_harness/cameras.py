"""Camera matrices in the reference's conventions (scene/cameras.py:69-84, utils/graphics_utils.py:136-168).

viewmatrix  = W2C.T   (row-vector convention; memory == column-major W2C, which is what the kernels index)
projmatrix  = W2C.T @ P.T  ("full_proj_transform")
focal       = W / (2 tan(fov/2))
"""
import math

import numpy as np

TENSOIR_FOVX = 0.6911112070083618  # relighting.py:150


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:149-168 restated."""
    t = math.tan(fovy / 2) * znear
    r = math.tan(fovx / 2) * znear
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (2 * r)
    P[1, 1] = 2.0 * znear / (2 * t)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_w2c(eye, target=(0, 0, 0), up=(0, 0, 1)):
    """World->camera with x right, y down, z forward (points in front have view z > 0)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, dtype=np.float64)
    right = np.cross(f, upv)
    if np.linalg.norm(right) < 1e-8:
        right = np.cross(f, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(f, right)
    Rw2c = np.stack([right, down, f], axis=0)
    M = np.eye(4)
    M[:3, :3] = Rw2c
    M[:3, 3] = -Rw2c @ eye
    return M.astype(np.float32)


def make_camera(W, H, eye, target=(0, 0, 0), fovx=TENSOIR_FOVX, znear=0.01, zfar=100.0):
    """Returns the dict of camera quantities the rasterizer settings need."""
    fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    w2c = look_at_w2c(eye, target)
    view = np.ascontiguousarray(w2c.T)
    P = projection_matrix(znear, zfar, fovx, fovy)
    proj = np.ascontiguousarray((view.astype(np.float32) @ P.T.astype(np.float32)).astype(np.float32))
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return {
        "W": int(W), "H": int(H),
        "tanfovx": math.tan(fovx * 0.5), "tanfovy": math.tan(fovy * 0.5),
        "viewmatrix": view, "projmatrix": proj, "campos": campos,
        "cx": W / 2.0, "cy": H / 2.0,
        "patch_bbox": np.array([0, 0, H, W], dtype=np.float32),
        "prcppoint": np.array([0.5, 0.5], dtype=np.float32),
    }


def orbit_eye(radius, azimuth_deg, elevation_deg):
    a, e = math.radians(azimuth_deg), math.radians(elevation_deg)
    return np.array([radius * math.cos(e) * math.cos(a), radius * math.cos(e) * math.sin(a), radius * math.sin(e)])
