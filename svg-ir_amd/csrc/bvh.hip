// svg-ir_amd/csrc/bvh.hip -- linear BVH over the surfels and the visibility tracer.
//
// Replaces submodules/bvh: `RayTracer.__init__` + `create_bvh` (__init__.py:28-58, src/construct.cu:151-265: leaf boxes from
// the eight corners mean +- 3 s_a a +- 3 s_b b +- 3 s_c c, 30-bit Morton codes of the box centroids, stable sort, Karras
// hierarchy on the unique keys morton << 31 | index, bottom-up box refit) and `trace_bvh_opacity` (src/trace.cu:186-262:
// per ray the product of (1 - opacity exp(power)) over the surfels whose leaf box the ray enters, 0 as soon as it drops
// below 0.9) -- the producer of the `[P,Ns,1]` visibility the shading kernels stream.
//
// A leaf is reached by the reference's traversal exactly when its own box passes the slab test (every ancestor's box
// contains it), so the result does not depend on the shape of the tree: only the leaf boxes, the slab test and the per-leaf
// arithmetic are the reference's; the tree layout is ours:
//   * internal node = one 64-byte record {box of child 0, box of child 1, child ids, parent}: one fetch per visited node,
//     both slab tests from it (the reference reads node[5] and two separate boxes);
//   * the per-surfel data the leaf test needs {mean, opacity, inverse covariance, normal, id} is gathered once per trace
//     call into 64-byte records in Morton order, so neighbouring leaves are neighbouring memory;
//   * Morton sort = the rasterizer's radix sort (binning.hip); the refit hands boxes upward through the parents' records
//     with one device-scope acquire-release counter per node (no second pass to lay the child boxes out);
//   * one lane per ray, traversal stack of child ids in private memory; rays are taken in memory order (the reference's
//     callers pass [surfels, samples] blocks: 64 consecutive rays share their origin).
#include <algorithm>

#include "common.hpp"
#include "lbvh.hpp"

namespace svgir {

namespace {

struct BvhLayout {
    float* leaf_box;        // [P][6] lower.xyz, upper.xyz (surfel order)
    uint32_t* key[2];       // [P] Morton codes ping/pong
    uint32_t* val[2];       // [P] surfel ids ping/pong (val[sorted] = Morton order)
    uint32_t* radix_tbl;    // radix scratch
    float4* nodes;          // [P-1][4]: {lo0.xyz, hi0.x} {hi0.yz, lo1.xy} {lo1.z, hi1.xyz} {child0, child1, parent, -} (bits)
    uint32_t* leaf_parent;  // [P] internal node above leaf i (Morton position) | slot << 31
    uint32_t* arrive;       // [P-1] refit arrival counters
    uint32_t* whole;        // [8] whole box as order-preserving integers: min xyz, max xyz
    float4* leaf_rec;       // [P][4]: {mean.xyz, opacity} {c0 c1 c2 c3} {c4 c5 n.x n.y} {n.z, id, -, -}
    size_t bytes;
};
BvhLayout bvh_layout(char* base, int P) {
    BvhLayout b;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t p = (size_t)(P > 0 ? P : 1);
    b.leaf_box = (float*)take(p * 6 * 4);
    b.key[0] = (uint32_t*)take(p * 4); b.key[1] = (uint32_t*)take(p * 4);
    b.val[0] = (uint32_t*)take(p * 4); b.val[1] = (uint32_t*)take(p * 4);
    b.radix_tbl = (uint32_t*)take(radix_table_words(P) * 4);
    b.nodes = (float4*)take(p * 64);
    b.leaf_parent = (uint32_t*)take(p * 4);
    b.arrive = (uint32_t*)take(p * 4);
    b.whole = (uint32_t*)take(32);
    b.leaf_rec = (float4*)take(p * 64);
    b.bytes = off;
    return b;
}
constexpr int BVH_SORT_BITS = 30, BVH_SORT_PASSES = 4;   // 10 bits per axis; 8 + 8 + 8 + 6

// ---- leaf boxes (__init__.py:32-58) + whole box (construct.cu:164-173) -------------------------------------------------
__global__ void __launch_bounds__(BLOCK) bvh_leaf_kernel(int P, const float* __restrict__ means, const float* __restrict__ scales,
                                                         const float* __restrict__ rots, float* __restrict__ leaf_box,
                                                         uint32_t* __restrict__ whole) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    float lo[3] = {100000.f, 100000.f, 100000.f}, hi[3] = {-100000.f, -100000.f, -100000.f};   // (the reduce's identity box)
    if (i < P) {
        // build_rotation (utils/general_utils.py:82-103): normalised quaternion (r, x, y, z) -> R; a, b, c = columns of R
        float q0 = rots[4 * i], q1 = rots[4 * i + 1], q2 = rots[4 * i + 2], q3 = rots[4 * i + 3];
        const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        q0 /= nrm; q1 /= nrm; q2 /= nrm; q3 /= nrm;
        const float r = q0, x = q1, y = q2, z = q3;
        const float R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                               {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                               {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
        const float sa = 3 * scales[3 * i], sb = 3 * scales[3 * i + 1], sc = 3 * scales[3 * i + 2];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float m = means[3 * i + c], ea = R[c][0] * sa, eb = R[c][1] * sb, ec = R[c][2] * sc;
            // the eight corners, evaluated left to right like the reference's tensor expressions
            float mn = 0.f, mx = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float v = ((m + ((k & 4) ? ea : -ea)) + ((k & 2) ? eb : -eb)) + ((k & 1) ? ec : -ec);
                mn = k == 0 ? v : fminf(mn, v);
                mx = k == 0 ? v : fmaxf(mx, v);
            }
            lo[c] = mn; hi[c] = mx;
            leaf_box[6 * i + c] = mn; leaf_box[6 * i + 3 + c] = mx;
        }
    }
    // whole box: wave reduce, one atomic per wave and component
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float mn = lo[c], mx = hi[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn = fminf(mn, __shfl_xor(mn, d)); mx = fmaxf(mx, __shfl_xor(mx, d)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&whole[c], f2ord(mn)); atomicMax(&whole[3 + c], f2ord(mx)); }
    }
}

// ---- Morton codes of the box centroids (construct.cu:6-52) -------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) bvh_morton_kernel(int P, const float* __restrict__ leaf_box, const uint32_t* __restrict__ whole,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float wl = ord2f(whole[c]), wu = ord2f(whole[3 + c]);
        float p = (leaf_box[6 * i + 3 + c] + leaf_box[6 * i + c]) * 0.5f;
        p -= wl;
        p /= (wu - wl);
        p = fminf(fmaxf(p * 1024.0f, 0.0f), 1024.0f - 1.0f);
        code |= expand_bits((uint32_t)p) << (2 - c);   // xx * 4 + yy * 2 + zz
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// ---- Karras hierarchy on the unique keys morton << 31 | id (construct.cu:54-149, 204-229) --------------------------------
__device__ __forceinline__ int delta_of(const uint32_t* __restrict__ mk, const uint32_t* __restrict__ id, int n, unsigned long long self, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long o = ((unsigned long long)mk[j] << 31) | id[j];
    return __clzll((long long)(self ^ o));
}
__global__ void __launch_bounds__(BLOCK) bvh_hierarchy_kernel(int P, const uint32_t* __restrict__ mk, const uint32_t* __restrict__ id,
                                                              float4* __restrict__ nodes, uint32_t* __restrict__ leaf_parent) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= P - 1) return;
    const unsigned long long self = ((unsigned long long)mk[idx] << 31) | id[idx];
    int first = 0, last = P - 1;
    if (idx != 0) {
        const int Ld = delta_of(mk, id, P, self, idx - 1), Rd = delta_of(mk, id, P, self, idx + 1);
        const int d = Rd > Ld ? 1 : -1;
        const int dmin = min(Ld, Rd);
        int lmax = 2;
        while (delta_of(mk, id, P, self, idx + d * lmax) > dmin) lmax <<= 1;
        int l = 0;
        for (int t = lmax >> 1; t > 0; t >>= 1)
            if (delta_of(mk, id, P, self, idx + (l + t) * d) > dmin) l += t;
        const int j = idx + l * d;
        first = min(idx, j); last = max(idx, j);
    }
    // split: highest key bit that differs inside [first, last] (the keys are unique)
    const unsigned long long fk = ((unsigned long long)mk[first] << 31) | id[first];
    const int dnode = delta_of(mk, id, P, fk, last);
    int split = first, stride = last - first;
    do {
        stride = (stride + 1) >> 1;
        const int mid = split + stride;
        if (mid < last && delta_of(mk, id, P, fk, mid) > dnode) split = mid;
    } while (stride > 1);
    // children: leaf (encoded ~position) when the range ends there, internal node otherwise
    const bool lleaf = first == split, rleaf = last == split + 1;
    const uint32_t c0 = lleaf ? ~(uint32_t)split : (uint32_t)split, c1 = rleaf ? ~(uint32_t)(split + 1) : (uint32_t)(split + 1);
    // (component stores: .z of this record is written by the parent's thread)
    nodes[4 * idx + 3].x = __builtin_bit_cast(float, c0); nodes[4 * idx + 3].y = __builtin_bit_cast(float, c1);
    if (idx == 0) nodes[3].z = __builtin_bit_cast(float, 0xffffffffu);
    // parent links (slot in bit 31)
    if (lleaf) leaf_parent[split] = (uint32_t)idx; else nodes[4 * split + 3].z = __builtin_bit_cast(float, (uint32_t)idx);
    if (rleaf) leaf_parent[split + 1] = (uint32_t)idx | 0x80000000u;
    else nodes[4 * (split + 1) + 3].z = __builtin_bit_cast(float, (uint32_t)idx | 0x80000000u);
}

// ---- bottom-up refit (construct.cu:234-264): every leaf walks up; the second child to arrive at a node carries the union on ----
__device__ __forceinline__ void store_child_box(float4* node, int slot, const float lo[3], const float hi[3]) {
    float* f = reinterpret_cast<float*>(node);
    float* d = f + 6 * slot;
    d[0] = lo[0]; d[1] = lo[1]; d[2] = lo[2]; d[3] = hi[0]; d[4] = hi[1]; d[5] = hi[2];
}
__global__ void __launch_bounds__(BLOCK) bvh_refit_kernel(int P, const uint32_t* __restrict__ id, const float* __restrict__ leaf_box,
                                                          float4* nodes, const uint32_t* __restrict__ leaf_parent, uint32_t* arrive) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = id[i];
    float lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = leaf_box[6 * g + c]; hi[c] = leaf_box[6 * g + 3 + c]; }
    uint32_t link = leaf_parent[i];
    for (int guard = 0; guard < 128; guard++) {
        const uint32_t parent = link & 0x7fffffffu;
        const int slot = (int)(link >> 31);
        float4* node = nodes + 4 * (size_t)parent;
        store_child_box(node, slot, lo, hi);
        // release our box / acquire the sibling's: device-scope acquire-release on the arrival counter
        const uint32_t old = __hip_atomic_fetch_add(&arrive[parent], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;   // first to arrive: the sibling's thread finishes this node
        const float* s = reinterpret_cast<const float*>(node) + 6 * (1 - slot);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            lo[c] = fminf(lo[c], __hip_atomic_load(s + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            hi[c] = fmaxf(hi[c], __hip_atomic_load(s + 3 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        link = __builtin_bit_cast(uint32_t, __hip_atomic_load(reinterpret_cast<const float*>(node) + 14, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (link == 0xffffffffu) return;   // the root
    }
}

// ---- per-trace leaf records in Morton order ----------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) bvh_leaf_rec_kernel(int P, const uint32_t* __restrict__ id, const float* __restrict__ means,
                                                             const float* __restrict__ cov_inv, const float* __restrict__ opacity,
                                                             const float* __restrict__ normals, float4* __restrict__ rec) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = id[i];
    const float* c = cov_inv + 6 * (size_t)g;
    rec[4 * i + 0] = make_float4(means[3 * g], means[3 * g + 1], means[3 * g + 2], opacity[g]);
    rec[4 * i + 1] = make_float4(c[0], c[1], c[2], c[3]);
    rec[4 * i + 2] = make_float4(c[4], c[5], normals[3 * g], normals[3 * g + 1]);
    rec[4 * i + 3] = make_float4(normals[3 * g + 2], __builtin_bit_cast(float, g), 0.f, 0.f);
}

// ---- visibility tracing (trace.cu:186-262) -------------------------------------------------------------------------------
// slab test of utility.cuh:35-90 with the reference's operations (true divisions by the ray direction, its comparisons; NaNs
// from 0 / 0 behave as there): returns tmax, or -1 for a miss -- the traversal only looks at tmax
__device__ __forceinline__ float slab_tmax(const float lo[3], const float hi[3], const float o[3], const float d[3]) {
    float tmin = (lo[0] - o[0]) / d[0], tmax = (hi[0] - o[0]) / d[0];
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (lo[1] - o[1]) / d[1], tymax = (hi[1] - o[1]) / d[1];
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) return -1.f;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo[2] - o[2]) / d[2], tzmax = (hi[2] - o[2]) / d[2];
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) return -1.f;
    if (tzmax < tmax) tmax = tzmax;
    return tmax;
}

constexpr int BVH_STACK = 64;
constexpr int BVH_LDS_DEPTH = 32;   // stack levels kept in LDS, [level][lane] (lanes at different depths never share a bank; a private array
                                    // indexed by a lane-varying depth makes every push / pop 64 separate cache lines); deeper levels in scratch
constexpr int BVH_WAVE = 64;        // one wave per workgroup

__global__ void __launch_bounds__(BLOCK) bvh_fill_kernel(long long n, float v, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ void __launch_bounds__(BVH_WAVE) bvh_trace_kernel(int P, long long num_rays, const float4* __restrict__ nodes,
                                                          const float4* __restrict__ rec, const float* __restrict__ rays_o,
                                                          const float* __restrict__ rays_d, float t_offset,
                                                          int32_t* __restrict__ contribute, float* __restrict__ visibility) {
    __shared__ uint32_t s_stack[BVH_LDS_DEPTH * BVH_WAVE];
    const long long r = (long long)blockIdx.x * BVH_WAVE + threadIdx.x;
    if (r >= num_rays) return;
    uint32_t* st = s_stack + threadIdx.x;
    float o[3], d[3];
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            d[c] = rays_d[3 * r + c];
            o[c] = rays_o[3 * r + c] + d[c] * t_offset;   // RayTracer.trace_visibility: rays_o + rays_d * 0.05
        }
    }
    uint32_t deep[BVH_STACK - BVH_LDS_DEPTH];
    int sp = 0;
    auto push = [&](uint32_t id) {
        if (sp < BVH_LDS_DEPTH) st[sp * BVH_WAVE] = id; else deep[sp - BVH_LDS_DEPTH] = id;
        sp++;
    };
    push(P > 1 ? 0u : ~0u);   // (a single surfel: the root is its leaf, entered unconditionally like the reference's)
    int count = 0;
    float ray_opacity = 1.0f;
    bool blocked = false;
    while (sp > 0 && !blocked) {
        --sp;
        const uint32_t nid = sp < BVH_LDS_DEPTH ? st[sp * BVH_WAVE] : deep[sp - BVH_LDS_DEPTH];
        // leaf record and node record are both 64 bytes: requested BEFORE the lanes split by node type -- one memory round trip per
        // step for all lanes (inside the two branches they were two, one after the other)
        const bool is_leaf = (nid & 0x80000000u) != 0u;
        const float4* q = is_leaf ? rec + 4 * (size_t)(~nid) : nodes + 4 * (size_t)nid;
        const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        if (is_leaf) {
            // ---- leaf: one surfel (trace.cu:218-247) ----
            const float4 A = q0, C0 = q1, C1 = q2, N = q3;
            if (A.w < 1.f / 255.f) continue;
            if (C1.z * d[0] + C1.w * d[1] + N.x * d[2] > 0) continue;   // back-facing
            const float c0 = C0.x, c1 = C0.y, c2 = C0.z, c3 = C0.w, c4 = C1.x, c5 = C1.y;
            const float mx = A.x - o[0], my = A.y - o[1], mz = A.z - o[2];
            // t of the point of the ray closest to the mean in the Mahalanobis sense (utility.cuh:99-110)
            const float t1 = c0 * mx * d[0] + c1 * mx * d[1] + c2 * mx * d[2] + c1 * my * d[0] + c3 * my * d[1] + c4 * my * d[2] +
                             c2 * mz * d[0] + c4 * mz * d[1] + c5 * mz * d[2];
            const float t2 = c0 * d[0] * d[0] + c1 * d[0] * d[1] + c2 * d[0] * d[2] + c1 * d[1] * d[0] + c3 * d[1] * d[1] +
                             c4 * d[1] * d[2] + c2 * d[2] * d[0] + c4 * d[2] * d[1] + c5 * d[2] * d[2];
            const float t = t1 / t2;
            if (t <= 0.01f) continue;   // trace.cu:227 compares with the DOUBLE literal: (double)t < 0.01  <=>  t <= 0.01f (0.01f < 0.01 < its successor)
            const float dx = A.x - (o[0] + t * d[0]), dy = A.y - (o[1] + t * d[1]), dz = A.z - (o[2] + t * d[2]);
            const float power = -0.5f * (dx * dx * c0 + dy * dy * c3 + dz * dz * c5 + 2 * dx * dy * c1 + 2 * dx * dz * c2 + 2 * dy * dz * c4);
            if (power > 0) continue;   // (a NaN power -- degenerate direction or covariance -- goes on and poisons the ray, as in the reference)
            count += 1;
            const float alpha = A.w * __expf(power);
            ray_opacity *= 1 - alpha;
            if (ray_opacity <= 0.9f) blocked = true;   // trace.cu:240 (double)ray_opacity < 0.9  <=>  <= 0.9f (0.9f < 0.9 < its successor)
        } else {
            // ---- internal node: both child boxes come with it; the child with the larger exit distance is pushed first ----
            const float4 a = q0, b = q1, c = q2, w = q3;
            const float lo0[3] = {a.x, a.y, a.z}, hi0[3] = {a.w, b.x, b.y}, lo1[3] = {b.z, b.w, c.x}, hi1[3] = {c.y, c.z, c.w};
            const float tl = slab_tmax(lo0, hi0, o, d), tr = slab_tmax(lo1, hi1, o, d);
            const uint32_t lid = __builtin_bit_cast(uint32_t, w.x), rid = __builtin_bit_cast(uint32_t, w.y);
            if (tl > tr) {
                if (tl > 0 && sp < BVH_STACK) push(lid);
                if (tr > 0 && sp < BVH_STACK) push(rid);
            } else {
                if (tr > 0 && sp < BVH_STACK) push(rid);
                if (tl > 0 && sp < BVH_STACK) push(lid);
            }
        }
    }
    // a blocked ray returns early in the reference: opacity 0, and its count is never written (stays 0)
    contribute[r] = blocked ? 0 : count;
    visibility[r] = blocked ? 0.0f : ray_opacity;
}

}  // namespace

}  // namespace svgir

extern "C" {

size_t svgir_bvh_bytes(int32_t P) { return svgir::bvh_layout(nullptr, P).bytes; }

int svgir_bvh_build(int32_t P, const float* means3D, const float* scales, const float* rotations, char* bvh, void* stream) {
    using namespace svgir;
    if (P < 0 || (P > 0 && (!means3D || !scales || !rotations || !bvh))) return SVGIR_ERR_INVALID;
    if (P == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const BvhLayout B = bvh_layout(bvh, P);
    const int nb = (P + BLOCK - 1) / BLOCK;
    if (hipMemsetAsync(B.whole, 0xff, 12, s) != hipSuccess) return SVGIR_ERR_HIP;
    if (hipMemsetAsync(B.whole + 3, 0, 12, s) != hipSuccess) return SVGIR_ERR_HIP;
    if (hipMemsetAsync(radix_gtot(B.radix_tbl, P), 0, radix_gtot_words(P) * 4, s) != hipSuccess) return SVGIR_ERR_HIP;
    if (hipMemsetAsync(B.arrive, 0, (size_t)P * 4, s) != hipSuccess) return SVGIR_ERR_HIP;
    hipLaunchKernelGGL(bvh_leaf_kernel, dim3(nb), dim3(BLOCK), 0, s, P, means3D, scales, rotations, B.leaf_box, B.whole);
    hipLaunchKernelGGL(bvh_morton_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.leaf_box, B.whole, B.key[0], B.val[0]);
    launch_radix_sort(B.key, B.val, P, nullptr, BVH_SORT_BITS, 8, B.radix_tbl, s);
    const int fin = BVH_SORT_PASSES & 1;
    if (P > 1) {
        hipLaunchKernelGGL(bvh_hierarchy_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.key[fin], B.val[fin], B.nodes, B.leaf_parent);
        hipLaunchKernelGGL(bvh_refit_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.val[fin], B.leaf_box, B.nodes, B.leaf_parent, B.arrive);
    }
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_bvh_trace_visibility(int32_t P, char* bvh, int64_t num_rays, const float* rays_o, const float* rays_d,
                               float t_offset, const float* means3D, const float* cov_inv, const float* opacity,
                               const float* normals, int32_t* contribute, float* visibility, void* stream) {
    using namespace svgir;
    if (P < 0 || num_rays < 0) return SVGIR_ERR_INVALID;
    if (num_rays == 0) return 0;
    if (!rays_o || !rays_d || !contribute || !visibility) return SVGIR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {   // nothing to hit: every ray is unoccluded (trace_bvh_opacity's initial values, bvh.cu:97-98)
        if (hipMemsetAsync(contribute, 0, (size_t)num_rays * 4, s) != hipSuccess) return SVGIR_ERR_HIP;
        hipLaunchKernelGGL(bvh_fill_kernel, dim3((unsigned)((num_rays + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, (long long)num_rays, 1.0f, visibility);
        return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
    }
    if (!bvh || !means3D || !cov_inv || !opacity || !normals) return SVGIR_ERR_INVALID;
    const BvhLayout B = bvh_layout(bvh, P);
    const int fin = BVH_SORT_PASSES & 1;
    hipLaunchKernelGGL(bvh_leaf_rec_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, P, B.val[fin], means3D, cov_inv, opacity,
                       normals, B.leaf_rec);
    hipLaunchKernelGGL(bvh_trace_kernel, dim3((unsigned)((num_rays + BVH_WAVE - 1) / BVH_WAVE)), dim3(BVH_WAVE), 0, s, P, (long long)num_rays,
                       B.nodes, B.leaf_rec, rays_o, rays_d, t_offset, contribute, visibility);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

}  // extern "C"
