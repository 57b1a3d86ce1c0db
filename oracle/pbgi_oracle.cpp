// oracle/pbgi_oracle.cpp -- CPU restatement of the reference's point-based-GI radiance tracer (TEST INFRASTRUCTURE ONLY:
// used by tests/ and nothing else; never linked into or called by the product).
//
// What it restates (reference = /root/reference, slang sources compiled by slangtorch at run time):
//   * pbgi/bvhhelpers.py:96-156  get_gs_bvh: element boxes -> scene extent -> Morton codes -> radix sort -> hierarchy ->
//     bounding boxes;
//   * pbgi/bvhworkers/get_elements.slang:74-107 (generateGaussianElements), lbvh_morton_codes.slang:22-80,
//     lbvh_single_radixsort.slang (a stable LSD sort by the 32-bit code), lbvh_hierarchy.slang:40-244 (Karras 2012 with
//     duplicate codes resolved by the SORTED POSITION), lbvh_bounding_boxes.slang (bottom-up unions; min / max are exact,
//     so the order of the reference's height-by-height sweeps does not matter);
//   * pbgi/bvhworkers/intersect_test.slang:21-42 (aabb_hit), :94-148 (ellipse_hit), :189-195 (gaussian_fn), :224-248
//     (matrixFromRotationQuaternions), :251-437 (gs_bvh_hit), :1879-1990 (render_radiance_with_sampling_SH);
//     sh_utils.slang:1-67 (eval_sh, degree 3).
//
// PARITY UNPINNED: slangtorch / the slang compiler are not available here, so nothing of the above was executed; the
// reference holds no tests or golden vectors for it.  This file follows the slang text operation by operation in fp32
// (-ffp-contract=off).  Two things the text does not determine are fixed here and in the HIP kernel alike:
// normalize(v) = v / sqrt(dot(v, v)), and inverse(M) = adjugate(M) * (1 / det(M)).
//
// Behaviour of the reference that is reproduced on purpose (it shapes the outputs):
//   Q-a  gs_bvh_hit returns the index / t / uv of the CLOSEST accepted leaf but the transmittance factor (1 - alpha) of the
//        LAST accepted leaf in traversal order (debug_res is overwritten on every accepted leaf, :414-418), so the result
//        depends on the tree and on the traversal order (children pushed left then right, popped right first);
//   Q-b  a leaf is "accepted" without any upper bound on its t: one beyond t_max sets any_hit with the initial
//        closest index 0 and t_hit = t_max (:408-416);
//   Q-c  the ray direction is re-normalised at every visited leaf and the re-normalised direction is used by the following
//        box tests of the same traversal (:342);
//   Q-d  the self-hit test compares the hit primitive with the ROW of the ray in the chunk (index_hit == gs_index, :1932),
//        and a rejected self hit ends the ray;
//   Q-e  box tests replace a zero direction component by 1e-6 (:26).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize(V3 v) { const float l = std::sqrt(dot(v, v)); return {v.x / l, v.y / l, v.z / l}; }

inline uint32_t expand_bits(uint32_t v) {   // lbvh_morton_codes.slang:24-30
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
inline float clamp_cell(float v) { return std::fmin(std::fmax(v * 1024.0f, 0.0f), 1023.0f); }   // fmax(NaN, 0) = 0 like the GPU's max
inline uint32_t morton3d(float x, float y, float z) {   // :34-42
    return expand_bits((uint32_t)clamp_cell(x)) * 4 + expand_bits((uint32_t)clamp_cell(y)) * 2 + expand_bits((uint32_t)clamp_cell(z));
}

// number of leading bits two sorted keys share; equal codes are told apart by their position (lbvh_hierarchy.slang:40-53)
struct Keys {
    const uint32_t* code; int n;
    int lcp(int i, int j) const {
        if (j < 0 || j > n - 1) return -1;
        const uint32_t a = code[i], b = code[j];
        if (a == b) return 32 + __builtin_clz((uint32_t)i ^ (uint32_t)j);   // (i != j whenever this is reached)
        return __builtin_clz(a ^ b);
    }
};

}  // namespace

extern "C" {

// Tree in the reference's own format: info[2P-1][3] = {left, right, primitive}, aabb[2P-1][6]; internal nodes first, leaf of
// sorted position j at P - 1 + j.  sorted[P][2] = {code, primitive} after the sort.
int orc_pbgi_build(int P, const float* centers, const float* scales, int32_t* info, float* aabb, int32_t* sorted) {
    if (P <= 0) return -1;
    std::vector<float> box((size_t)P * 6);
    float gmin[3] = {INFINITY, INFINITY, INFINITY}, gmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < P; i++) {   // get_elements.slang:86-105
        const float a0 = std::fabs(scales[3 * i]), a1 = std::fabs(scales[3 * i + 1]), a2 = std::fabs(scales[3 * i + 2]);
        const float len = 3.0f * std::fmax(a0, std::fmax(a1, a2));
        for (int c = 0; c < 3; c++) {
            box[6 * i + c] = centers[3 * i + c] - len;
            box[6 * i + 3 + c] = centers[3 * i + c] + len;
            gmin[c] = std::fmin(gmin[c], box[6 * i + c]);           // bvhhelpers.py:105-111
            gmax[c] = std::fmax(gmax[c], box[6 * i + 3 + c]);
        }
    }
    std::vector<uint32_t> code(P);
    for (int i = 0; i < P; i++) {   // lbvh_morton_codes.slang:66-76
        float m[3];
        for (int c = 0; c < 3; c++) {
            const float lo = box[6 * i + c], hi = box[6 * i + 3 + c];
            const float centre = lo + 0.5f * (hi - lo);
            m[c] = (centre - gmin[c]) / (gmax[c] - gmin[c]);
        }
        code[i] = morton3d(m[0], m[1], m[2]);
    }
    std::vector<int> order(P);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return code[a] < code[b]; });
    std::vector<uint32_t> scode(P);
    for (int j = 0; j < P; j++) { scode[j] = code[order[j]]; sorted[2 * j] = (int32_t)scode[j]; sorted[2 * j + 1] = order[j]; }
    const int L = P - 1;   // leaf offset
    for (int j = 0; j < P; j++) {
        info[3 * (L + j)] = 0; info[3 * (L + j) + 1] = 0; info[3 * (L + j) + 2] = order[j];
        std::memcpy(aabb + 6 * (size_t)(L + j), &box[6 * (size_t)order[j]], 24);
    }
    const Keys K{scode.data(), P};
    for (int i = 0; i < P - 1; i++) {   // lbvh_hierarchy.slang:55-113, :150-175 (Karras 2012, sections 3-4)
        const int dl = K.lcp(i, i - 1), dr = K.lcp(i, i + 1);
        const int d = dr >= dl ? 1 : -1;
        const int dmin = std::min(dl, dr);
        int lmax = 2;
        while (K.lcp(i, i + lmax * d) > dmin) lmax <<= 1;
        int l = 0;
        for (int t = lmax >> 1; t > 0; t >>= 1)
            if (K.lcp(i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int first = std::min(i, j), last = std::max(i, j);
        const int common = K.lcp(first, last);
        int split = first, stride = last - first;
        do {
            stride = (stride + 1) >> 1;
            const int cand = split + stride;
            if (cand < last && K.lcp(first, cand) > common) split = cand;
        } while (stride > 1);
        info[3 * i] = split == first ? L + split : split;
        info[3 * i + 1] = split + 1 == last ? L + split + 1 : split + 1;
        info[3 * i + 2] = 0;
    }
    // boxes of the internal nodes: unions of the children's, bottom-up (explicit stack, post-order)
    if (P > 1) {
        std::vector<int> st{0};
        std::vector<char> seen(P - 1, 0);
        while (!st.empty()) {
            const int n = st.back();
            if (n >= L) { st.pop_back(); continue; }
            if (!seen[n]) { seen[n] = 1; st.push_back(info[3 * n]); st.push_back(info[3 * n + 1]); continue; }
            st.pop_back();
            const float *a = aabb + 6 * (size_t)info[3 * n], *b = aabb + 6 * (size_t)info[3 * n + 1];
            for (int c = 0; c < 3; c++) {
                aabb[6 * (size_t)n + c] = std::fmin(a[c], b[c]);
                aabb[6 * (size_t)n + 3 + c] = std::fmax(a[3 + c], b[3 + c]);
            }
        }
    }
    return 0;
}

}  // extern "C"

namespace {

struct Scene {
    int P;
    const int32_t* info; const float* aabb;
    const float *centers, *scales, *rot, *normals, *opacity, *cov_inv, *shs;
};

inline bool box_hit(V3 o, V3 d, float t_min, float t_max, const float* b) {   // intersect_test.slang:21-42
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int i = 0; i < 3; i++) {
        float di = dd[i];
        if (di == 0.f) di = 0.000001f;
        const float inv = 1.0f / di;
        float t0 = (b[i] - oo[i]) * inv, t1 = (b[3 + i] - oo[i]) * inv;
        if (inv < 0.0f) std::swap(t0, t1);
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

struct Hit { bool any; float t, keep; int index; float u, v; };   // keep = debug_res.x = 1 - alpha of the last accepted leaf

// gs_bvh_hit, intersect_test.slang:251-437.  t / keep / index / uv are left untouched when nothing is accepted except that
// the index becomes -1, exactly like the reference's inout parameters.
void closest_hit(const Scene& S, V3 o, V3 d, float t_min, float t_max, float& t_hit, float& keep, int& index_hit, float& u, float& v, bool& any) {
    int stack[64];
    int count = 0;
    stack[count++] = 0;
    float closest = t_max, cu = 0.f, cv = 0.f, hit_t = 0.f, keep_l = 0.f, hu = 0.f, hv = 0.f;
    uint32_t closest_index = 0;
    bool any_hit = false;
    while (count > 0) {
        const int n = stack[--count];
        if (!box_hit(o, d, t_min, closest, S.aabb + 6 * (size_t)n)) continue;
        const int left = S.info[3 * n], right = S.info[3 * n + 1];
        if (left != 0 && right != 0) {
            if (count < 63) { stack[count++] = left; stack[count++] = right; }   // (the reference's 64-entry stack would overflow)
        } else if (left == 0 && right == 0) {
            const int g = S.info[3 * n + 2];
            const V3 c = {S.centers[3 * g], S.centers[3 * g + 1], S.centers[3 * g + 2]};
            const float sx = S.scales[3 * g], sy = S.scales[3 * g + 1];
            // :224-248
            const float q0 = S.rot[4 * g], q1 = S.rot[4 * g + 1], q2 = S.rot[4 * g + 2], q3 = S.rot[4 * g + 3];
            const float qn = std::sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3 + 0.00000001f);
            const float r = q0 / qn, x = q1 / qn, y = q2 / qn, z = q3 / qn;
            const float m00 = 1 - 2 * (y * y + z * z), m01 = 2 * (x * y - r * z), m02 = 2 * (x * z + r * y);
            const float m10 = 2 * (x * y + r * z), m11 = 1 - 2 * (x * x + z * z), m12 = 2 * (y * z - r * x);
            const float m20 = 2 * (x * z - r * y), m21 = 2 * (y * z + r * x), m22 = 1 - 2 * (x * x + y * y);
            d = normalize(d);   // :342 (Q-c)
            const V3 nrm = normalize(V3{S.normals[3 * g], S.normals[3 * g + 1], S.normals[3 * g + 2]});
            // ---- ellipse_hit, :94-148 ----
            bool hit = false;
            float t_now = 0.f, uh = 0.5f, vh = 0.5f;
            V3 pos = {0.f, 0.f, 0.f};
            {
                const V3 nw = {m02, m12, m22};   // L (0,0,1), L = R diag(sx, sy, 1)
                const float denom = dot(nw, d);
                if (!(std::fabs(denom) < 1e-6f)) {
                    t_now = dot(c - o, nw) / denom;
                    if (!(t_now < t_min)) {
                        pos = o + d * t_now;   // (origin + t_hit * dir)
                        const float det = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
                        const float id = 1.0f / det;
                        const V3 w = pos - c;
                        const float i00 = (m11 * m22 - m12 * m21) * id, i01 = (m02 * m21 - m01 * m22) * id, i02 = (m01 * m12 - m02 * m11) * id;
                        const float i10 = (m12 * m20 - m10 * m22) * id, i11 = (m00 * m22 - m02 * m20) * id, i12 = (m02 * m10 - m00 * m12) * id;
                        const float px = i00 * w.x + i01 * w.y + i02 * w.z, py = i10 * w.x + i11 * w.y + i12 * w.z;
                        float a = px / sx, b = py / sy;
                        if (a < b) std::swap(a, b);
                        a = a * 0.5f + 0.5f; b = b * 0.5f + 0.5f;
                        uh = std::fmin(std::fmax(a, 0.001f), 0.999f);
                        vh = std::fmin(std::fmax(b, 0.001f), 0.999f);
                        const float dis = (px * px) / (sx * sx) + (py * py) / (sy * sy);
                        hit = dis <= 9.0f;
                    }
                }
            }
            if (t_now < t_min) continue;   // :367-371
            const float* ci = S.cov_inv + 6 * (size_t)g;
            const V3 dd = c - pos;
            const float power = -0.5f * (dd.x * dd.x * ci[0] + dd.y * dd.y * ci[3] + dd.z * dd.z * ci[5] + 2 * dd.x * dd.y * ci[1] +
                                         2 * dd.x * dd.z * ci[2] + 2 * dd.y * dd.z * ci[4]);
            if (power > 0.0f) continue;
            const float alpha = std::fmin(0.99f, S.opacity[g] * std::exp(power));
            if (alpha < 1.0f / 255.0f) continue;
            if (!(dot(d, nrm) < -0.0f)) hit = false;   // :399-404
            const bool update = hit && t_now < closest;
            closest = hit ? std::fmin(t_now, closest) : closest;
            closest_index = update ? (uint32_t)g : closest_index;
            cu = update ? uh : cu; cv = update ? vh : cv;
            if (hit) { any_hit = true; hit_t = closest; keep_l = 1 - alpha; hu = cu; hv = cv; }   // (Q-a, Q-b)
        }
    }
    any = any_hit;
    if (any_hit) { t_hit = hit_t; keep = keep_l; index_hit = (int)closest_index; u = hu; v = hv; }
    else index_hit = -1;
}

void eval_sh(const float* sh, V3 dir, float out[3]) {   // sh_utils.slang:3-67; sh = [16][3] of one primitive
    dir = normalize(dir);
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f,
                         1.445305721320277f, -0.5900435899266435f};
    const float x = dir.x, y = dir.y, z = dir.z;
    for (int c = 0; c < 3; c++) {
        auto s = [&](int k) { return sh[3 * k + c]; };
        float r = C0 * s(0);
        r = r - C1 * y * s(1) + C1 * z * s(2) - C1 * x * s(3);
        r = r + C2[0] * x * y * s(4) + C2[1] * y * z * s(5) + C2[2] * (2.0f * z * z - x * x - y * y) * s(6) + C2[3] * x * z * s(7) +
            C2[4] * (x * x - y * y) * s(8);
        r = r + C3[0] * y * (3.0f * x * x - y * y) * s(9) + C3[1] * x * y * z * s(10) + C3[2] * y * (4.0f * z * z - x * x - y * y) * s(11) +
            C3[3] * z * (2.0f * z * z - 3.0f * x * x - 3.0f * y * y) * s(12) + C3[4] * x * (4.0f * z * z - x * x - y * y) * s(13) +
            C3[5] * z * (x * x - y * y) * s(14) + C3[6] * x * (x * x - 3.0f * y * y) * s(15);
        out[c] = r + 0.5f;
    }
}

}  // namespace

extern "C" {

// render_radiance_with_sampling_SH, intersect_test.slang:1879-1990: N rows x S rays; ray_o [N,3], ray_d [N,S,3].
// Outputs: radiance [N,S,3], visibility [N,S], hit_indices [N,S] (first accepted primitive or -1), uvs [N,S,2].
int orc_pbgi_trace(int P, const int32_t* info, const float* aabb, int N, int S, const float* ray_o, const float* ray_d, const float* centers,
                   const float* scales, const float* rotations, const float* normals, const float* opacity, const float* cov_inv,
                   const float* shs, float* radiance, float* visibility, int32_t* hit_indices, float* uvs) {
    if (P <= 0 || N < 0 || S <= 0) return -1;
    const Scene Sc{P, info, aabb, centers, scales, rotations, normals, opacity, cov_inv, shs};
#pragma omp parallel for schedule(dynamic, 4)
    for (long long ri = 0; ri < (long long)N * S; ri++) {
        const int row = (int)(ri / S);
        const V3 dir = normalize(V3{ray_d[3 * ri], ray_d[3 * ri + 1], ray_d[3 * ri + 2]});
        V3 o = {ray_o[3 * row], ray_o[3 * row + 1], ray_o[3 * row + 2]};
        int index_hit = -1, first_hit = -1;
        float u = 0.f, v = 0.f, fu = 0.f, fv = 0.f;
        float T = 1.0f, t_min = 0.042f, t_hit = 0.f, keep = 0.f;
        const float t_max = 0.2f;
        bool done = false, visible = true;
        float sh[3] = {0.f, 0.f, 0.f};
        while (T > 0.001f && !done) {
            bool hit;
            closest_hit(Sc, o, dir, t_min, t_max, t_hit, keep, index_hit, u, v, hit);
            hit = index_hit == row ? false : hit;   // (Q-d)
            if (hit) {
                if (first_hit == -1) { first_hit = index_hit; fu = u; fv = v; t_min = 0.01f; }
                const V3 hc = {centers[3 * index_hit], centers[3 * index_hit + 1], centers[3 * index_hit + 2]};
                const V3 sd = hc - o;
                o = o + dir * t_hit;
                float e[3];
                eval_sh(shs + 48 * (size_t)index_hit, sd, e);
                for (int c = 0; c < 3; c++) sh[c] += e[c] * (1 - keep) * T;
                T = T * keep;
                if (T < 0.2f) visible = false;
            } else {
                done = true;
            }
        }
        for (int c = 0; c < 3; c++) radiance[3 * ri + c] = std::fmin(std::fmax(sh[c], 0.0f), 10.0f);
        visibility[ri] = visible ? T : 0.0f;
        hit_indices[ri] = first_hit;
        uvs[2 * ri] = fu; uvs[2 * ri + 1] = fv;
    }
    return 0;
}

}  // extern "C"
