// oracle/bvh_oracle.cpp -- CPU restatement of the reference's surfel visibility tracer (submodules/bvh).
//
// TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke()); the product never links or loads this file.
//
// Parity status.  The reference tracer is CUDA + thrust (src/*.cu): not buildable here, and the reference holds no tests or
// vectors for it => the per-ray arithmetic below is "parity unpinned" by the reference's own outputs.  What IS pinned by the
// reference run in the authoring container (scripts/make_golden_bvh.py -> tests/golden/bvh.npz): the leaf boxes, the
// argument order and the 0.05 origin offset of `RayTracer` (__init__.py:28-71, imported with a recording stub for `_C`).
//
// What it restates:
//   orc_bvh_leaf_boxes  __init__.py:30-58 (build_rotation utils/general_utils.py:82-103): box of the eight corners
//                       mean +- 3 s_a a +- 3 s_b b +- 3 s_c c.
//   orc_bvh_trace       trace_bvh_opacity_cuda (src/trace.cu:186-262) with ray_intersects / gaussian_fn of
//                       include/utility.cuh:35-121.  The reference walks a BVH, but a leaf is reached exactly when its own
//                       box passes the slab test (every ancestor's box contains it; only the root is entered untested), so
//                       the walk is restated as a loop over ALL surfels gated by the slab test of their leaf box -- no tree,
//                       nothing shared with the product's builder.  Surfels are taken in index order; the reference's order
//                       (tree order) only changes the rounding of the running product and the moment the 0.9 cut-off is met.
#include <cmath>
#include <cstdint>
#include <omp.h>

namespace {

template <typename F>
inline void slab(const F* lo, const F* hi, const F* o, const F* d, F& tmin_out, F& tmax_out) {
    // utility.cuh:35-90 (divisions, thrust::swap, the two early returns {-1,-1})
    F tmin = (lo[0] - o[0]) / d[0], tmax = (hi[0] - o[0]) / d[0];
    if (tmin > tmax) { F t = tmin; tmin = tmax; tmax = t; }
    F tymin = (lo[1] - o[1]) / d[1], tymax = (hi[1] - o[1]) / d[1];
    if (tymin > tymax) { F t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) { tmin_out = -1; tmax_out = -1; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    F tzmin = (lo[2] - o[2]) / d[2], tzmax = (hi[2] - o[2]) / d[2];
    if (tzmin > tzmax) { F t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) { tmin_out = -1; tmax_out = -1; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    tmin_out = tmin; tmax_out = tmax;
}

template <typename F>
void trace(int P, const float* boxes, long long num_rays, const float* rays_o, const float* rays_d, float t_offset,
           const float* means, const float* cov, const float* opacity, const float* normals, int32_t* contribute,
           float* visibility) {
#pragma omp parallel for schedule(dynamic, 16)
    for (long long r = 0; r < num_rays; r++) {
        F o[3], d[3];
        for (int c = 0; c < 3; c++) {
            d[c] = (F)rays_d[3 * r + c];
            // RayTracer.trace_visibility: rays_o + rays_d * 0.05 (fp32 tensor ops in the reference)
            o[c] = sizeof(F) == 4 ? (F)(rays_o[3 * r + c] + (float)(rays_d[3 * r + c] * t_offset))
                                  : (F)rays_o[3 * r + c] + d[c] * (F)t_offset;
        }
        int count = 0;
        F ray_opacity = 1;
        bool blocked = false;
        for (int g = 0; g < P && !blocked; g++) {
            if (P > 1) {   // (the root is entered without a test: a single surfel is always examined)
                F lo[3], hi[3], tmin, tmax;
                for (int c = 0; c < 3; c++) { lo[c] = (F)boxes[6 * g + c]; hi[c] = (F)boxes[6 * g + 3 + c]; }
                slab<F>(lo, hi, o, d, tmin, tmax);
                if (!(tmax > 0)) continue;   // trace.cu:250-262: a child is pushed when intersection.y > 0
            }
            if (opacity[g] < 1.f / 255.f) continue;                                              // trace.cu:222
            const F nx = (F)normals[3 * g], ny = (F)normals[3 * g + 1], nz = (F)normals[3 * g + 2];
            if (nx * d[0] + ny * d[1] + nz * d[2] > 0) continue;                                 // :224
            const F* dummy = nullptr; (void)dummy;
            F c[6];
            for (int k = 0; k < 6; k++) c[k] = (F)cov[6 * g + k];
            const F m[3] = {(F)means[3 * g], (F)means[3 * g + 1], (F)means[3 * g + 2]};
            const F mx = m[0] - o[0], my = m[1] - o[1], mz = m[2] - o[2];
            // utility.cuh:99-110
            const F t1 = c[0] * mx * d[0] + c[1] * mx * d[1] + c[2] * mx * d[2] + c[1] * my * d[0] + c[3] * my * d[1] + c[4] * my * d[2] +
                         c[2] * mz * d[0] + c[4] * mz * d[1] + c[5] * mz * d[2];
            const F t2 = c[0] * d[0] * d[0] + c[1] * d[0] * d[1] + c[2] * d[0] * d[2] + c[1] * d[1] * d[0] + c[3] * d[1] * d[1] +
                         c[4] * d[1] * d[2] + c[2] * d[2] * d[0] + c[4] * d[2] * d[1] + c[5] * d[2] * d[2];
            const F t = t1 / t2;
            if ((double)t < 0.01) continue;                                                      // :227 (0.01 is a double literal: the float is promoted)
            const F pos[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
            const F dx = m[0] - pos[0], dy = m[1] - pos[1], dz = m[2] - pos[2];
            // utility.cuh:113-120: -0.5 (double) * float sum, returned as float
            const F sum = dx * dx * c[0] + dy * dy * c[3] + dz * dz * c[5] + 2 * dx * dy * c[1] + 2 * dx * dz * c[2] + 2 * dy * dz * c[4];
            const F power = (F)(-0.5 * (double)sum);
            if (power > 0) continue;                                                             // :236
            count += 1;
            const F alpha = (F)opacity[g] * (sizeof(F) == 4 ? (F)expf((float)power) : (F)exp((double)power));
            ray_opacity *= 1 - alpha;
            if ((double)ray_opacity < 0.9) blocked = true;                                        // :240-243 (0.9: double literal)
        }
        contribute[r] = blocked ? 0 : count;          // the early return leaves the zero-initialised count (bvh.cu:97)
        visibility[r] = blocked ? 0.0f : (float)ray_opacity;
    }
}

}  // namespace

extern "C" {

void orc_bvh_leaf_boxes(int P, const float* means, const float* scales, const float* rots, float* boxes) {
    for (int i = 0; i < P; i++) {
        float q[4] = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
        const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; k++) q[k] = q[k] / nrm;
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        const float R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                               {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                               {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
        const float s[3] = {3 * scales[3 * i], 3 * scales[3 * i + 1], 3 * scales[3 * i + 2]};
        for (int c = 0; c < 3; c++) {
            float mn = 0, mx = 0;
            for (int k = 0; k < 8; k++) {
                // x111 = means3D + a*sa + b*sb + c*sc, ... (left to right)
                const float v = ((means[3 * i + c] + ((k & 4) ? 1.f : -1.f) * (R[c][0] * s[0])) + ((k & 2) ? 1.f : -1.f) * (R[c][1] * s[1])) +
                                ((k & 1) ? 1.f : -1.f) * (R[c][2] * s[2]);
                mn = k == 0 ? v : fminf(mn, v);
                mx = k == 0 ? v : fmaxf(mx, v);
            }
            boxes[6 * i + c] = mn; boxes[6 * i + 3 + c] = mx;
        }
    }
}

void orc_bvh_trace(int P, const float* boxes, long long num_rays, const float* rays_o, const float* rays_d, float t_offset,
                   const float* means, const float* cov_inv, const float* opacity, const float* normals, int32_t* contribute,
                   float* visibility, int fp64) {
    if (fp64) trace<double>(P, boxes, num_rays, rays_o, rays_d, t_offset, means, cov_inv, opacity, normals, contribute, visibility);
    else trace<float>(P, boxes, num_rays, rays_o, rays_d, t_offset, means, cov_inv, opacity, normals, contribute, visibility);
}

}  // extern "C"
