// svg-ir_amd/csrc/pbgi.hip -- the radiance-cache producer of SURVEY 8f row f3: linear BVH over the surfels and the
// closest-hit radiance tracer of the reference's point-based-GI renderer.
//
// Replaces (reference = slang kernels compiled by slangtorch at run time):
//   * pbgi/bvhhelpers.py:96-156 get_gs_bvh + bvhworkers/{get_elements,lbvh_morton_codes,lbvh_single_radixsort,
//     lbvh_hierarchy,lbvh_bounding_boxes}.slang: boxes centre +- 3 max|scale|, 30-bit Morton codes of the box centres in
//     the scene extent, a stable sort by code, the Karras hierarchy with duplicate codes resolved by sorted position,
//     bottom-up box unions.  The reference sorts with ONE 256-thread workgroup and refits with one launch per tree level
//     (each with a host round trip for the height); here: the library's multi-block radix sort and ONE refit launch
//     (arrival counters, device-scope acquire/release);
//   * pbgi/renderer.py:596-615 render_radiance_with_sampling_SH + bvhworkers/intersect_test.slang:1879-1990 (the ray loop),
//     :251-437 (gs_bvh_hit), :94-148 (ellipse_hit), :21-42 (aabb_hit), sh_utils.slang (eval_sh).
//
// Unlike the visibility tracer (bvh.hip) the result of this one DEPENDS on the tree and on the traversal order: the
// transmittance factor a traversal returns is that of the LAST accepted leaf, not of the closest one (oracle/pbgi_oracle.cpp
// lists this and the other reproduced quirks, Q-a .. Q-e).  So the tree is the reference's tree -- same node numbering
// (internal nodes 0 .. P-2, leaf of sorted position j at P-1+j), children pushed left then right, boxes tested when popped
// against the closest hit so far -- and the kernels keep fp contraction off.  What is ours is the data layout: 32-byte node
// records {box, left, right} (one fetch per visited node instead of 3 + 6 scalar loads), and 96-byte leaf records in sorted
// order prepared once per trace call (centre, scales, plane normal, the two rows of the inverse rotation the ellipse test
// needs, unit normal, inverse covariance) instead of a quaternion -> matrix -> inverse evaluation at every visited leaf.
// One lane per ray, rays in memory order ([rows, samples]: the 64 rays of a wave share their origin).
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "lbvh.hpp"

namespace svgir {

namespace {

struct PbgiNode { float lo[3], hi[3]; int32_t left, right; };   // 32 bytes
static_assert(sizeof(PbgiNode) == 32, "node record");

struct PbgiLayout {
    float* box;             // [P][6] element boxes (primitive order)
    uint32_t* whole;        // [8] scene extent as order-preserving integers: min xyz, max xyz
    uint32_t* key[2];       // [P] Morton codes ping/pong
    uint32_t* val[2];       // [P] primitive ids ping/pong
    uint32_t* radix_tbl;
    PbgiNode* node;         // [2P-1]
    float4* pair;           // [P-1][4] traversal records of the internal nodes: the boxes of BOTH children + their ids (64 bytes)
    uint32_t* parent;       // [2P-1]
    uint32_t* arrive;       // [P-1]
    float4* rec;            // [P][6] leaf records (sorted order), filled per trace call
    unsigned long long* queue;   // [8] per XCD: next ray of its part of the launch that no wave has claimed yet (reset per trace call)
    uint32_t* rkey[2];      // [P] Morton codes of the ray origins of a trace call (rows are traced in that order), ping/pong
    uint32_t* rval[2];      // [P] row ids, ping/pong
    size_t bytes;
};
PbgiLayout pbgi_layout(char* base, int P) {
    PbgiLayout b;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t p = (size_t)(P > 0 ? P : 1);
    b.box = (float*)take(p * 24);
    b.whole = (uint32_t*)take(32);
    b.key[0] = (uint32_t*)take(p * 4); b.key[1] = (uint32_t*)take(p * 4);
    b.val[0] = (uint32_t*)take(p * 4); b.val[1] = (uint32_t*)take(p * 4);
    b.radix_tbl = (uint32_t*)take(radix_table_words(P) * 4);
    b.node = (PbgiNode*)take(2 * p * 32);
    b.pair = (float4*)take(p * 64);
    b.parent = (uint32_t*)take(2 * p * 4);
    b.arrive = (uint32_t*)take(p * 4);
    b.rec = (float4*)take(p * 96);
    b.queue = (unsigned long long*)take(64);
    b.rkey[0] = (uint32_t*)take(p * 4); b.rkey[1] = (uint32_t*)take(p * 4);
    b.rval[0] = (uint32_t*)take(p * 4); b.rval[1] = (uint32_t*)take(p * 4);
    b.bytes = off;
    return b;
}
constexpr int PBGI_SORT_BITS = 30, PBGI_SORT_PASSES = 4;

// ---- element boxes + scene extent (get_elements.slang:74-107, bvhhelpers.py:105-111) ---------------------------------------
__global__ void __launch_bounds__(BLOCK) pbgi_box_kernel(int P, const float* __restrict__ centers, const float* __restrict__ scales,
                                                         float* __restrict__ box, uint32_t* __restrict__ whole) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < P) {
        const float len = 3.0f * fmaxf(fabsf(scales[3 * i]), fmaxf(fabsf(scales[3 * i + 1]), fabsf(scales[3 * i + 2])));
#pragma unroll
        for (int c = 0; c < 3; c++) {
            lo[c] = centers[3 * i + c] - len; hi[c] = centers[3 * i + c] + len;
            box[6 * i + c] = lo[c]; box[6 * i + 3 + c] = hi[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float mn = lo[c], mx = hi[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mn = fminf(mn, __shfl_xor(mn, d)); mx = fmaxf(mx, __shfl_xor(mx, d)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&whole[c], f2ord(mn)); atomicMax(&whole[3 + c], f2ord(mx)); }
    }
}

// ---- Morton codes (lbvh_morton_codes.slang:22-80) ------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) pbgi_morton_kernel(int P, const float* __restrict__ box, const uint32_t* __restrict__ whole,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float gl = ord2f(whole[c]), gu = ord2f(whole[3 + c]);
        const float lo = box[6 * i + c], hi = box[6 * i + 3 + c];
        const float centre = lo + 0.5f * (hi - lo);
        const float m = (centre - gl) / (gu - gl);
        const float cell = fminf(fmaxf(m * 1024.0f, 0.0f), 1023.0f);
        code += expand_bits((uint32_t)cell) << (2 - c);   // xx * 4 + yy * 2 + zz (disjoint bits)
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// ---- hierarchy (lbvh_hierarchy.slang:40-244): Karras 2012 on the sorted codes, equal codes told apart by position ----------
__device__ __forceinline__ int pbgi_lcp(const uint32_t* __restrict__ code, int n, int i, uint32_t ci, int j) {
    if (j < 0 || j > n - 1) return -1;
    const uint32_t cj = code[j];
    if (ci == cj) return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
    return __clz((int)(ci ^ cj));
}
__global__ void __launch_bounds__(BLOCK) pbgi_hierarchy_kernel(int P, const uint32_t* __restrict__ code, const uint32_t* __restrict__ prim,
                                                               const float* __restrict__ box, PbgiNode* __restrict__ node,
                                                               uint32_t* __restrict__ parent) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    const int L = P - 1;
    {   // leaf of sorted position i
        PbgiNode nd;
        const uint32_t g = prim[i];
#pragma unroll
        for (int c = 0; c < 3; c++) { nd.lo[c] = box[6 * g + c]; nd.hi[c] = box[6 * g + 3 + c]; }
        nd.left = 0; nd.right = 0;
        node[L + i] = nd;
    }
    if (i >= P - 1) return;
    const uint32_t ci = code[i];
    const int dl = pbgi_lcp(code, P, i, ci, i - 1), dr = pbgi_lcp(code, P, i, ci, i + 1);
    const int d = dr >= dl ? 1 : -1;
    const int dmin = min(dl, dr);
    int lmax = 2;
    while (pbgi_lcp(code, P, i, ci, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t > 0; t >>= 1)
        if (pbgi_lcp(code, P, i, ci, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int first = min(i, j), last = max(i, j);
    const uint32_t cf = code[first];
    const int common = pbgi_lcp(code, P, first, cf, last);
    int split = first, stride = last - first;
    do {
        stride = (stride + 1) >> 1;
        const int cand = split + stride;
        if (cand < last && pbgi_lcp(code, P, first, cf, cand) > common) split = cand;
    } while (stride > 1);
    const int left = split == first ? L + split : split, right = split + 1 == last ? L + split + 1 : split + 1;
    node[i].left = left; node[i].right = right;
    parent[left] = (uint32_t)i; parent[right] = (uint32_t)i;
    if (i == 0) parent[0] = 0xffffffffu;
}

// ---- boxes of the internal nodes, bottom-up in one launch (lbvh_bounding_boxes.slang does one launch per tree level) --------
__global__ void __launch_bounds__(BLOCK) pbgi_refit_kernel(int P, PbgiNode* node, const uint32_t* __restrict__ parent, uint32_t* arrive) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    uint32_t p = parent[P - 1 + i];
    for (int guard = 0; guard < 128 && p != 0xffffffffu; guard++) {
        // the second child to arrive owns the node: release / acquire of the children's boxes on the arrival counter
        const uint32_t old = __hip_atomic_fetch_add(&arrive[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        const int a = node[p].left, b = node[p].right;
        const float* fa = reinterpret_cast<const float*>(node + a);
        const float* fb = reinterpret_cast<const float*>(node + b);
        float* fo = reinterpret_cast<float*>(node + p);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float la = __hip_atomic_load(fa + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), lb = __hip_atomic_load(fb + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float ha = __hip_atomic_load(fa + 3 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hb = __hip_atomic_load(fb + 3 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fo + c, fminf(la, lb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fo + 3 + c, fmaxf(ha, hb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        p = parent[p];
    }
}

// ---- traversal records: {box of the left child, box of the right child, left, right} per internal node ------------------------
__global__ void __launch_bounds__(BLOCK) pbgi_pair_kernel(int P, const PbgiNode* __restrict__ node, float4* __restrict__ pair) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P - 1) return;
    const int l = node[i].left, r = node[i].right;
    const PbgiNode a = node[l], b = node[r];
    float4* o = pair + 4 * (size_t)i;
    o[0] = make_float4(a.lo[0], a.lo[1], a.lo[2], a.hi[0]);
    o[1] = make_float4(a.hi[1], a.hi[2], b.lo[0], b.lo[1]);
    o[2] = make_float4(b.lo[2], b.hi[0], b.hi[1], b.hi[2]);
    o[3] = make_float4(__builtin_bit_cast(float, l), __builtin_bit_cast(float, r), 0.f, 0.f);
}

// ---- the tree in the reference's tensors: LBVHNode_info [2P-1][3] = {left, right, primitive}, LBVHNode_aabb [2P-1][6] ------
__global__ void __launch_bounds__(BLOCK) pbgi_export_kernel(int P, const PbgiNode* __restrict__ node, const uint32_t* __restrict__ code,
                                                            const uint32_t* __restrict__ prim, int32_t* __restrict__ info,
                                                            float* __restrict__ aabb, int32_t* __restrict__ sorted) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= 2 * P - 1) return;
    const PbgiNode nd = node[i];
    info[3 * i] = nd.left; info[3 * i + 1] = nd.right; info[3 * i + 2] = i >= P - 1 ? (int32_t)prim[i - (P - 1)] : 0;
#pragma unroll
    for (int c = 0; c < 3; c++) { aabb[6 * i + c] = nd.lo[c]; aabb[6 * i + 3 + c] = nd.hi[c]; }
    if (sorted && i < P) { sorted[2 * i] = (int32_t)code[i]; sorted[2 * i + 1] = (int32_t)prim[i]; }
}

// Morton codes of the ray origins of a trace call, in the tree's own grid: rows are traced in this order, so that the waves that run
// at the same time walk neighbouring parts of the tree (the 38 MB of node / leaf records of a 200 k-surfel tree do not fit a 4 MB L2;
// with rows in memory order -- random in space -- nearly every visit was a trip to HBM)
__global__ void __launch_bounds__(BLOCK) pbgi_row_code_kernel(int n, const float* __restrict__ ray_o, const uint32_t* __restrict__ whole,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float gl = ord2f(whole[c]), gu = ord2f(whole[3 + c]);
        const float m = (ray_o[3 * (size_t)i + c] - gl) / fmaxf(gu - gl, 1e-30f);
        const float cell = fminf(fmaxf(m * 1024.0f, 0.0f), 1023.0f);   // (NaN -> 0)
        code += expand_bits((uint32_t)cell) << (2 - c);
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// The rays of a launch (rows in Morton order of their origins) are dealt to the XCDs in eight contiguous parts: the waves of XCD c
// (workgroups go to the XCDs round-robin: blockIdx.x & 7) start in part c and take its chunks from queue[c], so that an XCD's L2
// serves one eighth of the scene's neighbourhoods instead of all of them; a wave whose part is used up helps the next one.
__global__ void pbgi_queue_init_kernel(unsigned long long* queue, unsigned long long part, unsigned long long chunk, unsigned nw) {
    const unsigned c = threadIdx.x;
    if (c < 8u) queue[c] = (unsigned long long)c * part + (unsigned long long)((nw + 7u - c) / 8u) * chunk;   // the first chunks belong to the waves from the start
}

// ---- leaf records (per trace call: the reference reads these tensors at every visited leaf) --------------------------------
// {c.xyz, sx} {sy, opacity, nw.xy} {nw.z, i00 i01 i02} {i10 i11 i12, n.x} {n.yz, ci0 ci1} {ci2 .. ci5}
__global__ void __launch_bounds__(BLOCK) pbgi_leaf_rec_kernel(int P, const uint32_t* __restrict__ prim, const float* __restrict__ centers,
                                                              const float* __restrict__ scales, const float* __restrict__ rot,
                                                              const float* __restrict__ normals, const float* __restrict__ opacity,
                                                              const float* __restrict__ cov_inv, float4* __restrict__ rec) {
#pragma clang fp contract(off)
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= P) return;
    const size_t g = prim[j];
    // matrixFromRotationQuaternions, intersect_test.slang:224-248
    const float q0 = rot[4 * g], q1 = rot[4 * g + 1], q2 = rot[4 * g + 2], q3 = rot[4 * g + 3];
    const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3 + 0.00000001f);
    const float r = q0 / qn, x = q1 / qn, y = q2 / qn, z = q3 / qn;
    const float m00 = 1 - 2 * (y * y + z * z), m01 = 2 * (x * y - r * z), m02 = 2 * (x * z + r * y);
    const float m10 = 2 * (x * y + r * z), m11 = 1 - 2 * (x * x + z * z), m12 = 2 * (y * z - r * x);
    const float m20 = 2 * (x * z - r * y), m21 = 2 * (y * z + r * x), m22 = 1 - 2 * (x * x + y * y);
    // inverse(rotateMat) = adjugate * (1 / det): rows 0 and 1 (ellipse_hit only uses posM.x, posM.y)
    const float det = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
    const float id = 1.0f / det;
    const float i00 = (m11 * m22 - m12 * m21) * id, i01 = (m02 * m21 - m01 * m22) * id, i02 = (m01 * m12 - m02 * m11) * id;
    const float i10 = (m12 * m20 - m10 * m22) * id, i11 = (m00 * m22 - m02 * m20) * id, i12 = (m02 * m10 - m00 * m12) * id;
    const float nx = normals[3 * g], ny = normals[3 * g + 1], nz = normals[3 * g + 2];
    const float nl = sqrtf(nx * nx + ny * ny + nz * nz);
    const float* ci = cov_inv + 6 * g;
    float4* o = rec + 6 * (size_t)j;
    o[0] = make_float4(centers[3 * g], centers[3 * g + 1], centers[3 * g + 2], scales[3 * g]);
    o[1] = make_float4(scales[3 * g + 1], opacity[g], m02, m12);
    o[2] = make_float4(m22, i00, i01, i02);
    o[3] = make_float4(i10, i11, i12, nx / nl);
    o[4] = make_float4(ny / nl, nz / nl, ci[0], ci[1]);
    o[5] = make_float4(ci[2], ci[3], ci[4], ci[5]);
}

// ---- tracing ---------------------------------------------------------------------------------------------------------------
struct F3 { float x, y, z; };
__device__ __forceinline__ float dot3(F3 a, F3 b) {
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ F3 unit3(F3 v) {   // normalize(v) = v / sqrt(dot(v, v))
#pragma clang fp contract(off)
    const float l = sqrtf(dot3(v, v));
    return {v.x / l, v.y / l, v.z / l};
}
struct SlabDir { float inv[3]; };
__device__ __forceinline__ SlabDir slab_dir(F3 d) {   // aabb_hit's per-axis reciprocal (zero components become 1e-6), intersect_test.slang:25-27
#pragma clang fp contract(off)
    SlabDir s;
    const float dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; i++) s.inv[i] = 1.0f / (dd[i] == 0.f ? 0.000001f : dd[i]);
    return s;
}
__device__ __forceinline__ bool box_hit(const float4 q0, const float4 q1, F3 o, const SlabDir& sd, float t_min, float t_max) {   // :21-42
#pragma clang fp contract(off)
    const float lo[3] = {q0.x, q0.y, q0.z}, hi[3] = {q0.w, q1.x, q1.y}, oo[3] = {o.x, o.y, o.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float inv = sd.inv[i];
        float t0 = (lo[i] - oo[i]) * inv, t1 = (hi[i] - oo[i]) * inv;
        if (inv < 0.0f) { const float t = t1; t1 = t0; t0 = t; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

constexpr int PBGI_STACK = 64;        // the reference's MAX_STACK_SIZE; the tree is at most 30 + 32 - clz(P) + 1 < 63 levels deep (30-bit
                                      // codes, equal codes split by position), and the stack holds at most one pending sibling per level
#ifndef PBGI_LDS_DEPTH_V
#define PBGI_LDS_DEPTH_V 20
#endif
constexpr int PBGI_LDS_DEPTH = PBGI_LDS_DEPTH_V;   // (round 4, 129 VGPRs = 12 waves per CU whatever the LDS: 32 / 24 / 16 levels: 1 017 / 963 / 1 023 ms on the cfg3 geometry, 1 735 / 1 392 / 1 238 ms on the shell scene.  Round 5, 124 VGPRs: 24 levels = 13 waves per CU 873 / 1 255 ms; 20 levels = 16 waves per CU 866 / 1 060 ms; 16 levels at 5 waves per SIMD spills: 1 508 / 2 120 ms)    // stack levels kept in LDS ([level][lane]: conflict-free whatever the lanes' depths); deeper ones in scratch
constexpr int PBGI_WAVE = 64;         // one wave per workgroup
constexpr int PBGI_MAX_HITS = 4096;   // guard of the ray loop (every accepted hit removes >= 1/255 of the transmittance: < 1800 hits)


#if defined(SVGIR_DEV)
// development builds: traversal statistics of the last trace (queries, tested boxes, boxes passed, leaves visited, leaves accepted, rays)
__device__ unsigned long long g_pbgi_stats[8];
__device__ unsigned long long g_pbgi_qhist[16];   // rays by floor(log2(queries)); [12..15]: box tests spent by rays with >= 1, 16, 128, 1024 queries
#define PBGI_STAT(i) st_[i]++
#else
#define PBGI_STAT(i)
#endif

// The per-axis part of aabb_hit (intersect_test.slang:21-42) split from its interval: aabb_hit narrows [t_min, t_max] axis by
// axis and fails as soon as it is empty; because the lower end only grows and the upper end only shrinks this is exactly
//     min(t_max, exit) > max(t_min, entry)   with   entry = max_i t0_i,  exit = min_i t1_i   (NaNs skipped by the same selects),
// i.e.  exit > entry'  and  t_max > entry'  with entry' = max(t_min, entry).  entry' and exit do not depend on t_max (the closest hit so
// far): a box tested when it is PUSHED keeps its entry', and the test the reference makes when it POPS the node -- against the
// closest hit found meanwhile -- is the single comparison closest > entry'.
__device__ __forceinline__ bool box_entry(const float lo[3], const float hi[3], F3 o, const SlabDir& sd, float t_min, float& entry) {
#pragma clang fp contract(off)
    const float oo[3] = {o.x, o.y, o.z};
    float ex = INFINITY;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float inv = sd.inv[i];
        float t0 = (lo[i] - oo[i]) * inv, t1 = (hi[i] - oo[i]) * inv;
        if (inv < 0.0f) { const float t = t1; t1 = t0; t0 = t; }
        t_min = t0 > t_min ? t0 : t_min;
        ex = t1 < ex ? t1 : ex;
    }
    entry = t_min;
    return ex > t_min;
}

__device__ __forceinline__ void eval_sh3(const float* __restrict__ sh, F3 dir, float out[3]) {   // sh_utils.slang:3-67
#pragma clang fp contract(off)
    dir = unit3(dir);
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f, C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
    const float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f, C33 = 0.3731763325901154f,
                C34 = -0.4570457994644658f, C35 = 1.445305721320277f, C36 = -0.5900435899266435f;
    const float x = dir.x, y = dir.y, z = dir.z;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float r = C0 * sh[c];
        r = r - C1 * y * sh[3 + c] + C1 * z * sh[6 + c] - C1 * x * sh[9 + c];
        r = r + C20 * x * y * sh[12 + c] + C21 * y * z * sh[15 + c] + C22 * (2.0f * z * z - x * x - y * y) * sh[18 + c] + C23 * x * z * sh[21 + c] +
            C24 * (x * x - y * y) * sh[24 + c];
        r = r + C30 * y * (3.0f * x * x - y * y) * sh[27 + c] + C31 * x * y * z * sh[30 + c] + C32 * y * (4.0f * z * z - x * x - y * y) * sh[33 + c] +
            C33 * z * (2.0f * z * z - 3.0f * x * x - 3.0f * y * y) * sh[36 + c] + C34 * x * (4.0f * z * z - x * x - y * y) * sh[39 + c] +
            C35 * z * (x * x - y * y) * sh[42 + c] + C36 * x * (x * x - 3.0f * y * y) * sh[45 + c];
        out[c] = r + 0.5f;
    }
}

// One visited leaf of gs_bvh_hit: ellipse_hit (:94-148) and the acceptance tests (:367-404) for the surfel of sorted position j.
// The reference evaluates the ellipse first; its results (hit, uv) are only read for ACCEPTED leaves (8 % of the visited ones:
// t >= t_min, power <= 0, alpha >= 1/255), so the plane intersection and the acceptance come first and the ellipse -- four IEEE
// divisions -- last.  `d` is the direction in effect at this visit (re-normalised by the caller, Q-c).
struct LeafRes { bool acc, hit; float t, alpha, u, v; };
__device__ __forceinline__ LeafRes leaf_eval_regs(const float4 A, const float4 B, const float4 C, const float4 D, const float4 E, const float4 G, F3 o,
                                                  F3 d, float t_min) {
#pragma clang fp contract(off)
    LeafRes r = {false, false, 0.f, 0.f, 0.5f, 0.5f};
    const F3 c = {A.x, A.y, A.z};
    const float sx = A.w, sy = B.x;
    const F3 nw = {B.z, B.w, C.x};
    const float denom = dot3(nw, d);
    if (!(fabsf(denom) < 1e-6f)) {
        const F3 co = {c.x - o.x, c.y - o.y, c.z - o.z};
        const float t_now = dot3(co, nw) / denom;
        if (!(t_now < t_min)) {   // (:367-371; a parallel ray has t_now = 0 < t_min)
            const F3 pos = {o.x + d.x * t_now, o.y + d.y * t_now, o.z + d.z * t_now};
            const F3 dd = {c.x - pos.x, c.y - pos.y, c.z - pos.z};
            const float power = -0.5f * (dd.x * dd.x * E.z + dd.y * dd.y * G.y + dd.z * dd.z * G.w + 2 * dd.x * dd.y * E.w + 2 * dd.x * dd.z * G.x +
                                         2 * dd.y * dd.z * G.z);
            if (!(power > 0.0f)) {
                const float alpha = fminf(0.99f, B.y * expf(power));
                if (!(alpha < 1.0f / 255.0f)) {
                    const F3 w = {pos.x - c.x, pos.y - c.y, pos.z - c.z};
                    const float px = C.y * w.x + C.z * w.y + C.w * w.z, py = D.x * w.x + D.y * w.y + D.z * w.z;
                    float a = px / sx, b = py / sy;
                    if (a < b) { const float t = a; a = b; b = t; }
                    a = a * 0.5f + 0.5f; b = b * 0.5f + 0.5f;
                    r.u = fminf(fmaxf(a, 0.001f), 0.999f);
                    r.v = fminf(fmaxf(b, 0.001f), 0.999f);
                    const float dis = (px * px) / (sx * sx) + (py * py) / (sy * sy);
                    bool hit = dis <= 9.0f;
                    const F3 nrm = {D.w, E.x, E.y};
                    if (!(dot3(d, nrm) < -0.0f)) hit = false;   // :399-404
                    r.acc = true; r.hit = hit; r.t = t_now; r.alpha = alpha;
                }
            }
        }
    }
    return r;
}

__device__ __forceinline__ LeafRes leaf_eval(const float4* __restrict__ rec, int j, F3 o, F3 d, float t_min) {
    const float4* lr = rec + 6 * (size_t)j;
    return leaf_eval_regs(lr[0], lr[1], lr[2], lr[3], lr[4], lr[5], o, d, t_min);
}

// gs_bvh_hit (intersect_test.slang:251-437) inside render_radiance_with_sampling_SH (:1879-1990).
//
// Per ray the reference runs closest-hit queries in a loop (one per accepted surfel, until the transmittance is used up or nothing is
// hit); a query pops a node, tests its box against the closest hit so far, pushes left then right.  The results depend on that visit
// order (Q-a .. Q-e), so every ray here makes exactly the reference's visits and decisions -- restated so that
//   * a visit costs ONE fetch: an internal node's record holds the boxes of both children, tested when the parent is visited; the
//     right child is visited next (it would be popped next), the left child is pushed with its entry distance and re-checked against
//     the closest hit when popped (box_entry above): boxes that fail never reach the stack, popped entries that fail cost no memory
//     access.  That needs the direction to be final: the reference re-normalises it at every visited leaf (Q-c) and later box tests
//     use the new one.  While |d| does not round to exactly 1 the direction may still change; until then the left child is pushed
//     untested (entry = -inf) and tested with its own 32-byte box when popped, exactly like the reference does;
//   * the stack lives in LDS, [level][lane]: lanes at different depths never collide on a bank;
//   * LANES NEVER WAIT FOR EACH OTHER'S RAYS: 19 % of the rays of a cache update hit something and then need 7 queries on average
//     (some 30+), the others one -- with one ray per lane for the life of the wave, 64 lanes waited for the longest ray of the wave
//     (measured: 3.5 s for 12.8 M rays, SIMD lanes ~10 % busy).  Here a wave owns a POOL of consecutive rays; a lane whose traversal
//     ends finishes its query (shading update, next query or outputs) and pulls the next ray of the pool at once, while the other
//     lanes keep walking: one loop, one traversal step per iteration, per-lane state.
#ifndef PBGI_WPE
#define PBGI_WPE 4
#endif
__global__ void __launch_bounds__(PBGI_WAVE) __attribute__((amdgpu_waves_per_eu(PBGI_WPE, 8))) pbgi_trace_kernel(int P, const PbgiNode* __restrict__ node, const float4* __restrict__ pair, const float4* __restrict__ rec,
                                                               const uint32_t* __restrict__ prim, int N, int S, const float* __restrict__ ray_o,
                                                               const float* __restrict__ ray_d, const float* __restrict__ centers,
                                                               const float* __restrict__ shs, float* __restrict__ radiance,
                                                               float* __restrict__ visibility, int32_t* __restrict__ hit_indices, float* __restrict__ uvs,
                                                               int chunk, unsigned long long* __restrict__ queue, long long part,
                                                               const uint32_t* __restrict__ row_order, int row_base) {
#pragma clang fp contract(off)
    __shared__ int s_ids[PBGI_LDS_DEPTH * PBGI_WAVE];
    __shared__ float s_ens[PBGI_LDS_DEPTH * PBGI_WAVE];
    const int lane = threadIdx.x;
    int* s_id = s_ids + lane;
    float* s_en = s_ens + lane;
    const long long total = (long long)N * S;
    // wave-uniform: the wave's current chunk of consecutive rays [next_ray, pool_end); further chunks come from the launch-wide queue
    // (the wave's XCD part first: [q_cur * part, (q_cur + 1) * part) of the launch's rays in tracing order, see pbgi_queue_init_kernel)
    int q_cur = (int)(blockIdx.x & 7u), q_tried = 0;
    long long next_ray = (long long)q_cur * part + (long long)(blockIdx.x >> 3) * chunk;
    long long pool_end = min(min(total, (long long)(q_cur + 1) * part), next_ray + (long long)chunk);
    bool drained = false;   // every part's queue is empty
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const float t_max = 0.2f;
    const int L = P - 1;
#if defined(SVGIR_DEV)
    unsigned st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // ---- per-lane ray state (the loop of :1879-1990) ----
    bool have_ray = false;
    long long ri = 0;
    int row = 0, first_hit = -1, it = 0;
    F3 dir = {0.f, 0.f, 1.f}, o = {0.f, 0.f, 0.f};
    float fu = 0.f, fv = 0.f, T = 1.0f, t_min = 0.042f;
    bool visible = true;
    float sh[3] = {0.f, 0.f, 0.f};
    // ---- per-lane traversal state (gs_bvh_hit) ----
    bool walking = false;   // a query is in progress
    int deep_id[PBGI_STACK - PBGI_LDS_DEPTH];
    float deep_en[PBGI_STACK - PBGI_LDS_DEPTH];
    int count = 0, cur = 0;
    bool own_test = true;   // `cur` still has to pass its own box test (root; nodes pushed while the direction could still change)
    float closest = t_max, cu = 0.f, cv = 0.f, hit_t = 0.f, keep_l = 0.f, hu = 0.f, hv = 0.f;
    uint32_t closest_index = 0;
    bool any_hit = false, fixed = false;
    F3 d = dir;
    SlabDir sd = slab_dir(d);
    auto push = [&](int id, float en) {
        if (count < PBGI_LDS_DEPTH) { s_id[count * PBGI_WAVE] = id; s_en[count * PBGI_WAVE] = en; }
        else if (count < PBGI_STACK) { deep_id[count - PBGI_LDS_DEPTH] = id; deep_en[count - PBGI_LDS_DEPTH] = en; }
        else return;   // a full stack drops the entry, as the reference does (it cannot happen: see PBGI_STACK) -- never an out-of-bounds pop
        count++;
    };
    for (;;) {
        if (__any(!walking)) {
            // ---- a query ended (or the lane has no ray yet): ray bookkeeping, :1925-1975 ----
            if (!walking && have_ray) {
                const int h_index = any_hit ? (int)closest_index : -1;
                const bool hit = h_index == row ? false : any_hit;   // (Q-d)
                bool more = false;
                if (hit) {
                    if (first_hit == -1) { first_hit = h_index; fu = hu; fv = hv; t_min = 0.01f; }
                    const F3 sdir = {centers[3 * (size_t)h_index] - o.x, centers[3 * (size_t)h_index + 1] - o.y, centers[3 * (size_t)h_index + 2] - o.z};
                    o = {o.x + dir.x * hit_t, o.y + dir.y * hit_t, o.z + dir.z * hit_t};
                    float e[3];
                    eval_sh3(shs + 48 * (size_t)h_index, sdir, e);
#pragma unroll
                    for (int c = 0; c < 3; c++) sh[c] += e[c] * (1 - keep_l) * T;
                    T = T * keep_l;
                    if (T < 0.2f) visible = false;
                    it++;
                    more = it < PBGI_MAX_HITS && T > 0.001f;
                }
                if (!more) {
#if defined(SVGIR_DEV)
                    {
                        const unsigned nq = (unsigned)it + (hit ? 0u : 1u);
                        atomicAdd(&g_pbgi_qhist[min(11, 31 - __clz((int)max(nq, 1u)))], 1ull);
                        atomicMax(&g_pbgi_stats[6], (unsigned long long)nq);
                        const unsigned long long bt = st_[1] - st_[7];
                        atomicAdd(&g_pbgi_qhist[12], bt);
                        if (nq >= 16) atomicAdd(&g_pbgi_qhist[13], bt);
                        if (nq >= 128) atomicAdd(&g_pbgi_qhist[14], bt);
                        if (nq >= 1024) atomicAdd(&g_pbgi_qhist[15], bt);
                        st_[7] = st_[1];
                    }
#endif
#pragma unroll
                    for (int c = 0; c < 3; c++) radiance[3 * ri + c] = fminf(fmaxf(sh[c], 0.0f), 10.0f);
                    visibility[ri] = visible ? T : 0.0f;
                    hit_indices[ri] = first_hit;
                    uvs[2 * ri] = fu; uvs[2 * ri + 1] = fv;
                    have_ray = false;
                }
            }
            {   // lanes without a ray take the next rays of the wave's chunk, in lane order; an empty chunk is replaced from the queue
                const bool want = !walking && !have_ray;
                if (next_ray >= pool_end && !drained && __any(want)) {
                    for (;;) {
                        unsigned long long got = 0;
                        if (lane == 0) got = atomicAdd(queue + q_cur, (unsigned long long)chunk);
                        got = (unsigned long long)__shfl((long long)got, 0);
                        const long long part_end = min(total, (long long)(q_cur + 1) * part);
                        next_ray = (long long)got;
                        pool_end = min(part_end, next_ray + (long long)chunk);
                        if (next_ray < part_end) break;
                        // this part is used up: the next XCD's part (its own waves are still on it, its tree neighbourhoods are the nearest)
                        q_cur = (q_cur + 1) & 7;
                        if (++q_tried == 8) { drained = true; next_ray = pool_end = total; break; }
                    }
                }
                const unsigned long long m = __ballot(want);
                const long long mine = next_ray + (long long)__popcll(m & lt_mask);
                next_ray += (long long)__popcll(m);
                if (want && mine < pool_end) {
                    // `mine` counts the rays of this launch in TRACING order (rows sorted by the Morton code of their origin)
                    const long long pr = mine / S;
                    row = row_base + (int)row_order[pr];
                    ri = (long long)row * S + (mine - pr * S);
                    dir = unit3(F3{ray_d[3 * ri], ray_d[3 * ri + 1], ray_d[3 * ri + 2]});
                    o = {ray_o[3 * (size_t)row], ray_o[3 * (size_t)row + 1], ray_o[3 * (size_t)row + 2]};
                    first_hit = -1; fu = 0.f; fv = 0.f; T = 1.0f; t_min = 0.042f; visible = true; it = 0;
                    sh[0] = 0.f; sh[1] = 0.f; sh[2] = 0.f;
                    have_ray = true;
                    PBGI_STAT(5);
                }
            }
            if (!walking && have_ray) {   // next query of the lane's ray
                PBGI_STAT(0);
                count = 0; cur = 0; own_test = true;
                closest = t_max; cu = 0.f; cv = 0.f; hit_t = 0.f; keep_l = 0.f; hu = 0.f; hv = 0.f;
                closest_index = 0; any_hit = false;
                d = dir; sd = slab_dir(d);
                fixed = sqrtf(dot3(d, d)) == 1.0f;   // the re-normalisation at a visited leaf leaves d as it is
                walking = true;
            }
            if (!__any(walking)) break;   // the pool is empty and every lane is done
        }
        if (walking) {
            // ---- one traversal step ----
            // The node's record -- the pair record of an internal node (64 bytes) or the leaf record (96 bytes) -- is requested BEFORE the
            // lanes split by node type: one memory round trip per step for all lanes (requested inside the two branches they were two,
            // one after the other).  A node that still has to pass its own box (rare: the root, nodes pushed while the direction could
            // change) has it requested speculatively.
            const bool is_int = cur < L;
            const float4* nrec = is_int ? pair + 4 * (size_t)cur : rec + 6 * (size_t)(cur - L);
            const float4 n0 = nrec[0], n1 = nrec[1], n2 = nrec[2], n3 = nrec[3];
            float4 n4 = n0, n5 = n0;
            if (!is_int) { n4 = nrec[4]; n5 = nrec[5]; }
            bool alive = true;
            if (own_test) {
                const float4* q = reinterpret_cast<const float4*>(node + cur);
                const float4 q0 = q[0], q1 = q[1];
                PBGI_STAT(1);
                alive = box_hit(q0, q1, o, sd, t_min, closest);
            }
            bool descend = false;
            if (alive) {
                PBGI_STAT(2);
                if (cur < L) {
                    // ---- internal node: both children's boxes ----
                    const float4 r0 = n0, r1 = n1, r2 = n2, r3 = n3;
                    const int left = __builtin_bit_cast(int, r3.x), right = __builtin_bit_cast(int, r3.y);
                    const float lo0[3] = {r0.x, r0.y, r0.z}, hi0[3] = {r0.w, r1.x, r1.y}, lo1[3] = {r1.z, r1.w, r2.x}, hi1[3] = {r2.y, r2.z, r2.w};
                    if (fixed) {
                        float eL;
                        PBGI_STAT(1);
                        if (box_entry(lo0, hi0, o, sd, t_min, eL) && closest > eL) push(left, eL);
                    } else {
                        push(left, -INFINITY);
                    }
                    float eR;
                    PBGI_STAT(1);
                    if (box_entry(lo1, hi1, o, sd, t_min, eR) && closest > eR) { cur = right; own_test = false; descend = true; }
                } else {
                    // ---- leaf ----
                    const int j = cur - L;
                    PBGI_STAT(3);
                    if (!fixed) {   // :342 -- the re-normalised direction also serves the box tests that follow (Q-c).  A direction whose
                        // length already rounds to 1 is left bit-identical by the division
                        const float l = sqrtf(dot3(d, d));
                        if (l != 1.0f) { d = {d.x / l, d.y / l, d.z / l}; sd = slab_dir(d); fixed = sqrtf(dot3(d, d)) == 1.0f; }
                        else fixed = true;
                    }
                    const LeafRes lf = leaf_eval_regs(n0, n1, n2, n3, n4, n5, o, d, t_min);
                    if (lf.acc) {
                        PBGI_STAT(4);
                        const bool update = lf.hit && lf.t < closest;
                        closest = lf.hit ? fminf(lf.t, closest) : closest;
                        closest_index = update ? prim[j] : closest_index;
                        cu = update ? lf.u : cu; cv = update ? lf.v : cv;
                        if (lf.hit) { any_hit = true; hit_t = closest; keep_l = 1 - lf.alpha; hu = cu; hv = cv; }   // (Q-a, Q-b)
                    }
                }
            }
            if (!descend) {
                // ---- next node: the stack's top, unless the closest hit found since it was pushed already excludes its box ----
                bool got = false;
                while (count > 0) {
                    count--;
                    float en;
                    if (count < PBGI_LDS_DEPTH) { cur = s_id[count * PBGI_WAVE]; en = s_en[count * PBGI_WAVE]; }
                    else { cur = deep_id[count - PBGI_LDS_DEPTH]; en = deep_en[count - PBGI_LDS_DEPTH]; }
                    if (en == -INFINITY) { own_test = true; got = true; break; }
                    if (closest > en) { own_test = false; got = true; break; }
                }
                walking = got;
            }
        }
    }
#if defined(SVGIR_DEV)
#pragma unroll
    for (int i = 0; i < 6; i++) {
        unsigned v = st_[i];
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) v += (unsigned)__shfl_xor((int)v, dd);
        if (lane == 0) atomicAdd(&g_pbgi_stats[i], (unsigned long long)v);
    }
#endif
}


}  // namespace

}  // namespace svgir

extern "C" {

#if defined(SVGIR_DEV)
// development builds only: reads and clears the traversal statistics
int svgir_dev_pbgi_stats(unsigned long long* out) {   // out[24]: 8 counters + 16 histogram words
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(svgir::g_pbgi_stats), 64) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(svgir::g_pbgi_qhist), 128) != hipSuccess) return -1;
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(svgir::g_pbgi_stats), z, 64);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(svgir::g_pbgi_qhist), z, 128);
    return 0;
}
#endif

size_t svgir_pbgi_bvh_bytes(int32_t P) { return svgir::pbgi_layout(nullptr, P).bytes; }

int svgir_pbgi_bvh_build(int32_t P, const float* centers, const float* scales, char* bvh, void* stream) {
    using namespace svgir;
    if (P <= 0 || !centers || !scales || !bvh) return SVGIR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const PbgiLayout B = pbgi_layout(bvh, P);
    const int nb = (P + BLOCK - 1) / BLOCK;
    if (hipMemsetAsync(B.whole, 0xff, 12, s) != hipSuccess) return SVGIR_ERR_HIP;
    if (hipMemsetAsync(B.whole + 3, 0, 12, s) != hipSuccess) return SVGIR_ERR_HIP;
    if (hipMemsetAsync(radix_gtot(B.radix_tbl, P), 0, radix_gtot_words(P) * 4, s) != hipSuccess) return SVGIR_ERR_HIP;
    if (hipMemsetAsync(B.arrive, 0, (size_t)P * 4, s) != hipSuccess) return SVGIR_ERR_HIP;
    hipLaunchKernelGGL(pbgi_box_kernel, dim3(nb), dim3(BLOCK), 0, s, P, centers, scales, B.box, B.whole);
    hipLaunchKernelGGL(pbgi_morton_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.box, B.whole, B.key[0], B.val[0]);
    launch_radix_sort(B.key, B.val, P, nullptr, PBGI_SORT_BITS, 8, B.radix_tbl, s);
    const int fin = PBGI_SORT_PASSES & 1;
    hipLaunchKernelGGL(pbgi_hierarchy_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.key[fin], B.val[fin], B.box, B.node, B.parent);
    if (P > 1) hipLaunchKernelGGL(pbgi_refit_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.node, B.parent, B.arrive);
    if (P > 1) hipLaunchKernelGGL(pbgi_pair_kernel, dim3(nb), dim3(BLOCK), 0, s, P, B.node, B.pair);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_pbgi_bvh_export(int32_t P, char* bvh, int32_t* info, float* aabb, int32_t* sorted, void* stream) {
    using namespace svgir;
    if (P <= 0 || !bvh || !info || !aabb) return SVGIR_ERR_INVALID;
    const PbgiLayout B = pbgi_layout(bvh, P);
    const int fin = PBGI_SORT_PASSES & 1;
    hipLaunchKernelGGL(pbgi_export_kernel, dim3((2 * P - 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, P, B.node, B.key[fin],
                       B.val[fin], info, aabb, sorted);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_pbgi_trace_radiance(int32_t P, char* bvh, int32_t N, int32_t S, const float* ray_o, const float* ray_d, const float* centers,
                              const float* scales, const float* rotations, const float* normals, const float* opacity,
                              const float* cov3D_inverse, const float* shs, float* radiance, float* visibility, int32_t* hit_indices,
                              float* uvs, void* stream) {
    using namespace svgir;
    if (P <= 0 || N < 0 || S <= 0 || !bvh) return SVGIR_ERR_INVALID;
    if (N == 0) return 0;
    if (!ray_o || !ray_d || !centers || !scales || !rotations || !normals || !opacity || !cov3D_inverse || !shs || !radiance || !visibility ||
        !hit_indices || !uvs)
        return SVGIR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const PbgiLayout B = pbgi_layout(bvh, P);
    const int fin = PBGI_SORT_PASSES & 1;
    hipLaunchKernelGGL(pbgi_leaf_rec_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, P, B.val[fin], centers, scales, rotations, normals,
                       opacity, cov3D_inverse, B.rec);
    // Persistent waves (one per resident slot: 10 KB of LDS each, 16 per CU) take chunks of consecutive rays (whole rows: the rays of a row
    // share their origin) from a launch-wide queue, so no wave slot idles while rays are left and the end of the launch is one ray deep.
    // (Static pools, measured on the cfg3 geometry, 4.27 M rays per launch: 256 / 1 024 / 4 096 / 16 384 rays per wave = 2 129 / 1 329 /
    // 1 534 / 4 249 ms -- small pools end with 63 lanes waiting for one long ray, large ones leave wave slots empty.)
    long long chunk = std::max<long long>(S, (64 + S - 1) / S * S);   // (64 / 256 / 1 024 rays per chunk: 1 005 / 1 046 / 1 178 ms on the cfg3 geometry)
    if (const char* e = getenv("SVGIR_PBGI_POOL")) { const long long v = atoll(e); if (v > 0) chunk = (v + S - 1) / S * S; }   // (tuning experiments)
    const int slots = 256 * std::min(8 * 4, (160 * 1024) / (PBGI_LDS_DEPTH * PBGI_WAVE * 8));   // resident waves
    for (int row0 = 0; row0 < N; row0 += P) {   // (the row-order buffers hold P rows: more rows than surfels go in blocks)
        const int n = std::min(P, N - row0);
        // rows in the Morton order of their origins
        if (hipMemsetAsync(radix_gtot(B.radix_tbl, n), 0, radix_gtot_words(n) * 4, s) != hipSuccess) return SVGIR_ERR_HIP;   // (n <= P: within the table)
        hipLaunchKernelGGL(pbgi_row_code_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, n, ray_o + 3 * (size_t)row0, B.whole, B.rkey[0], B.rval[0]);
        launch_radix_sort(B.rkey, B.rval, n, nullptr, PBGI_SORT_BITS, 8, B.radix_tbl, s);
        const long long rays = (long long)n * S;
        const long long nw = std::min<long long>((rays + chunk - 1) / chunk, slots);
        const long long nchunks = (rays + chunk - 1) / chunk;
        const long long part = (nchunks + 7) / 8 * chunk;   // rays per XCD part (whole chunks)
        hipLaunchKernelGGL(pbgi_queue_init_kernel, dim3(1), dim3(64), 0, s, B.queue, (unsigned long long)part, (unsigned long long)chunk, (unsigned)nw);
        hipLaunchKernelGGL(pbgi_trace_kernel, dim3((unsigned)nw), dim3(PBGI_WAVE), 0, s, P, B.node, B.pair, B.rec, B.val[fin], n, S, ray_o,
                           ray_d, centers, shs, radiance, visibility, hit_indices, uvs, (int)chunk, B.queue, part, B.rval[PBGI_SORT_PASSES & 1], row0);
    }
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

}  // extern "C"
