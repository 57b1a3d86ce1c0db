"""CPU-only checks of the C-ABI shared library and the drop-in Python surface (no GPU compute is launched)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_symbol_of_the_header(built):
    from gaussian_renderer import _native
    hdr = open(os.path.join(ROOT, "include", "svgir_raster.h")).read()
    declared = set(re.findall(r"\b(svgir_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"svgir_alloc_fn"}
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    lib = C.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.svgir_abi_version() == _native.ABI_VERSION == 14


def test_struct_layouts_match_header_field_order(built):
    """The ctypes mirrors must list the same fields, in the same order, as the C structs."""
    from gaussian_renderer import _native
    hdr = open(os.path.join(ROOT, "include", "svgir_raster.h")).read()

    def fields(struct):
        m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S) or re.search(r"\nstruct %s \{(.*?)\};" % struct, hdr, re.S)
        body = m.group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.replace("*", " ").split(None, 1)[1] if not decl.startswith("const") else decl.replace("*", " ").split(None, 2)[2]
            out += [n.strip() for n in names.split(",")]
        return out

    assert fields("svgir_params") == [f[0] for f in _native.Params._fields_]
    assert fields("svgir_outputs") == [f[0] for f in _native.Outputs._fields_]
    assert fields("svgir_grads") == [f[0] for f in _native.Grads._fields_]
    from gaussian_renderer import shading
    assert fields("svgir_shade_params") == [f[0] for f in shading.ShadeParams._fields_]
    assert fields("svgir_fused_shade") == [f[0] for f in _native.FusedShade._fields_]
    from svgir_harness import optim
    assert fields("svgir_adam_tensor") == [f[0] for f in optim._AdamTensor._fields_]
    assert fields("svgir_row_tensor") == [f[0] for f in optim._RowTensor._fields_]
    assert re.search(r"#define SVGIR_ADAM_MAX_TENSORS (\d+)", hdr).group(1) == str(optim.MAX_TENSORS)


def test_blob_sizes_and_error_reporting(built):
    from gaussian_renderer import _native as N
    lib = N.lib
    assert lib.svgir_geom_bytes(1000) > 1000 * 96
    assert lib.svgir_geom_bytes(2000) > lib.svgir_geom_bytes(1000)
    assert lib.svgir_image_bytes(800, 800) >= 800 * 800 * 12
    off = lib.svgir_image_ncontrib_offset(800, 800)
    assert off % 256 == 0 and off + 800 * 800 * 4 <= lib.svgir_image_bytes(800, 800)
    assert lib.svgir_binning_bytes(0, 64, 64, 0, 0) > 0 and lib.svgir_binning_bytes(10 ** 6, 800, 800, 5, 0) >= 16 * 10 ** 6
    # invalid parameter blocks are rejected before any HIP call, with a message
    p = N.Params()
    p.variant, p.P, p.W, p.H = 7, 10, 16, 16
    rc = lib.svgir_forward(p, N.Outputs(), N.ALLOC_FN(lambda n, c: 0), None, N.ALLOC_FN(lambda n, c: 0), None,
                           N.ALLOC_FN(lambda n, c: 0), None, None)
    assert rc == -1 and "variant" in N.last_error()
    p.variant = N.SVGSS
    rc = lib.svgir_forward(p, N.Outputs(), N.ALLOC_FN(lambda n, c: 0), None, N.ALLOC_FN(lambda n, c: 0), None,
                           N.ALLOC_FN(lambda n, c: 0), None, None)
    assert rc == -1 and "must be provided" in N.last_error()
    with pytest.raises(RuntimeError):
        N.check(rc, "forward")


@pytest.mark.parametrize("mod", ["svgss_rasterization", "rgss_rasterization"])
def test_binding_surface_matches_reference_module(built, mod):
    """Field names / argument orders captured from the reference's own modules (scripts/make_golden.py)."""
    import importlib
    import inspect
    g = np.load(os.path.join(GOLD, "binding_api.npz"))
    m = importlib.import_module("gaussian_renderer." + mod)
    assert list(m.GaussianRasterizationSettings._fields) == list(g[mod + ".settings_fields"])
    assert list(inspect.signature(m.GaussianRasterizer.forward).parameters) == list(g[mod + ".forward_args"])
    assert list(inspect.signature(m._RasterizeGaussians.forward).parameters) == list(g[mod + ".function_forward_args"])
    assert list(inspect.signature(m._RasterizeGaussians.backward).parameters) == list(g[mod + ".function_backward_args"])
    for name in g[mod + ".public"]:
        assert hasattr(m, str(name)), name
    for fn in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        assert hasattr(m._C, fn)


@pytest.mark.parametrize("mod", ["svgss_rasterization", "rgss_rasterization"])
def test_argument_validation_like_reference(built, mod):
    import importlib
    m = importlib.import_module("gaussian_renderer." + mod)
    rast = m.GaussianRasterizer(raster_settings=None)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=x, means2D=x, opacities=x[:, :1])
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=x, means2D=x, opacities=x[:, :1], shs=x, colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=x, means2D=x, opacities=x[:, :1], shs=x, scales=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=x, means2D=x, opacities=x[:, :1], shs=x, scales=x, rotations=x, cov3D_precomp=x)


def test_cpu_tensors_fail_loudly_no_fallback(built):
    """There is no CPU / eager fallback behind the bindings: CPU tensors are rejected."""
    from svgir_harness import runner, scenes
    sc = scenes.random_cloud(P=16, W=32, H=32, variant="svgss")
    sct = runner.to_torch(sc, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        runner.render(sct, "svgss")
    sc = scenes.random_cloud(P=16, W=32, H=32, variant="rgss")
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        runner.render(runner.to_torch(sc, torch.device("cpu")), "rgss")


def test_image_size_limits_are_validated(built):
    """The state blobs pack tile coordinates into 10 bits and sub-tile ids into 18: larger images are rejected up
    front (before any HIP call) instead of corrupting silently."""
    from gaussian_renderer import _native as N
    p = N.Params()
    p.variant, p.P, p.W, p.H = N.SVGSS, 10, 16 * 1024, 64
    cb = N.ALLOC_FN(lambda n, c: 0)
    rc = N.lib.svgir_forward(p, N.Outputs(), cb, None, cb, None, cb, None, None)
    assert rc == -1 and "exceeds the supported size" in N.last_error()
    p.W, p.H = 16 * 300, 16 * 300      # 90 000 tiles: too many sub-tile ids
    rc = N.lib.svgir_forward(p, N.Outputs(), cb, None, cb, None, cb, None, None)
    assert rc == -1 and "exceeds the supported size" in N.last_error()


def test_backward_requires_scratch(built):
    from gaussian_renderer import _native as N
    assert N.lib.svgir_backward_scratch_bytes(N.RGSS, 1000, 0, 64, 64, 5, 0) >= 1000 * 20 * 4
    assert N.lib.svgir_backward_scratch_bytes(N.SVGSS, 1000, N.lib.svgir_binning_bytes(5000, 64, 64, 4, 52), 64, 64, 4, 52) > \
        N.lib.svgir_backward_scratch_bytes(N.RGSS, 1000, 0, 64, 64, 5, 0)


def test_xcd_tile_maps_are_bijections(built):
    """common.hpp xcd_tile_of_work (the per-tile cull's work-id -> tile map) and order_entries: every tile exactly once, on the XCD that
    owns its 4 x 4-tile block, for awkward grid shapes (checked on the host through a tiny program compiled against the header)."""
    src = r'''
#include <cstdio>
#include <vector>
#include "%s/svg-ir_amd/csrc/common.hpp"
using namespace svgir;
int main() {
    int dims[][2] = {{50,50},{100,100},{13,9},{1,1},{4,4},{5,3},{7,64},{1023,3},{17,17},{8,8},{9,1},{3,1023}};
    for (auto& d : dims) {
        int gx = d[0], gy = d[1]; size_t n = order_entries(gx, gy) / 4;
        std::vector<int> seen(gx * gy, 0);
        for (size_t q = 0; q < n + 64; q++) {
            uint32_t t = xcd_tile_of_work((uint32_t)q, gx, gy);
            if (t == ORDER_NONE) continue;
            if (q >= n || t >= (uint32_t)(gx * gy) || xcd_of_tile(t %% gx, t / gx) != (q & 7)) { printf("BAD %%d %%d\n", gx, gy); return 1; }
            seen[t]++;
        }
        for (int v : seen) if (v != 1) { printf("BAD %%d %%d\n", gx, gy); return 1; }
    }
    printf("OK\n");
    return 0;
}
''' % ROOT
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "t.cpp"), "w") as f:
            f.write(src)
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O1", os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t")],
                       check=True, capture_output=True, timeout=600)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout
