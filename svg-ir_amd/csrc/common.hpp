// svg-ir_amd/csrc/common.hpp -- shared declarations of the gfx950 surfel rasterizer (product code).
//
// HBM layout (all blobs are opaque to the callers; only the sizes are part of the C ABI):
//   geometry blob  : per-Gaussian "splat record" (24 floats = 96 B, one gather unit for the composite kernels),
//                    cov3D, SH clamp mask, tiles_touched, depth-sort ping/pong (key, id), instance offsets,
//                    scan / radix scratch, device instance counter.
//   image blob     : final_T, final_D, n_contrib (int32) planes + per-tile ranges (uint2).
//   binning blob   : tile-key and Gaussian-id ping/pong buffers of R entries + radix scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "../../include/svgir_raster.h"

namespace svgir {

constexpr int TILE = 16;
constexpr int BLOCK = 256;
constexpr int REC = 24;  // floats per splat record

// Record field offsets (floats).  First 6 floats = the "header" the composite kernels stage in LDS.
enum RecField {
    R_X = 0, R_Y = 1, R_CX = 2, R_CY = 3,      // float4 #0: mean2D, conic.x, conic.y
    R_CZ = 4, R_OP = 5, R_DEPTH = 6, R_DA = 7, // float4 #1: conic.z, opacity, view depth, d(depth)/d(dx)
    R_J0 = 8, R_J1 = 9, R_J2 = 10, R_J3 = 11,  // float4 #2: screen->tangent 2x2
    R_DB = 12, R_R = 13, R_G = 14, R_B = 15,   // float4 #3: d(depth)/d(dy), rgb
                                               //   depth differencing (auxiliary.h:390-397, .z): (dx J0 + dy J1) ax0.z +
                                               //   (dx J2 + dy J3) ax1.z = dx DA + dy DB, DA = J0 ax0.z + J2 ax1.z,
                                               //   DB = J1 ax0.z + J3 ax1.z -- the only way ax0.z / ax1.z are ever used
    R_NX = 16, R_NY = 17, R_NZ = 18, R_IU = 19,// float4 #4: view normal, 1/(0.5*scale.x+0.1)
    R_IV = 20, R_IBASE = 21, R_RECT = 22, R_PAD2 = 23  // float4 #5: 1/(0.5*scale.y+0.1); [written by emit:] index of the
                                                       // Gaussian's first instance in emit order (u32 bits); tile rect
                                                       // x0 | y0 << 10 | (x1 - x0) << 20 (u32 bits)
};

// Without vfeatures (every rgss call; svgss with VS = 0) the records carry the Gaussian's feature row: nothing reads the tangent-plane
// terms (J, 1/umax) then, and with S <= 5
// the row fits their floats -- the specialised composite kernels then make ONE gather per candidate (the 96-byte record) instead of
// record + feature row (a second, 20-byte-strided gather: its own cache lines, its own scalar loads in the backward).
// Channel ch sits at rec_feature_slot(ch); preprocess writes it, render_fwd_kernel<S, 0, .> / render_bwd_plain_kernel<S, .> read it.
#if defined(__HIPCC__)
#define SVGIR_HD __host__ __device__
#else
#define SVGIR_HD
#endif
SVGIR_HD constexpr bool rec_embeds_features(int S, int VS) { return VS == 0 && S >= 1 && S <= 5; }
SVGIR_HD constexpr int rec_feature_slot(int ch) { return ch < 4 ? 8 + ch : 19; }   // R_J0 .. R_J3, R_IU

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- radix sort geometry -----------------------------------------------------------------------------------
constexpr int SORT_ITEMS = 16;                      // elements per thread per block
constexpr int SORT_BLOCK_ELEMS = BLOCK * SORT_ITEMS;  // 4096
// upper bound of the number of radix blocks for n keys (1024 keys per block for small inputs)
inline int sort_blocks(int n) { return n <= 0 ? 1 : (n + 1023) / 1024; }
// radix scratch: [sort_blocks][256] block histograms + [4 passes][sort_blocks/32 + 1][256] group totals
inline size_t radix_gtot_words(int n) { return (size_t)4 * 256 * (sort_blocks(n) / 32 + 1); }
inline size_t radix_table_words(int n) { return (size_t)256 * sort_blocks(n) + radix_gtot_words(n); }
// group totals of a sort over (up to) n elements; they must be ZERO when launch_radix_sort runs (the stage in front of each
// sort clears them in passing: preprocess for the depth sort, emit for the tile sort)
inline uint32_t* radix_gtot(uint32_t* table, int n) { return table + (size_t)256 * sort_blocks(n); }
constexpr int SCAN_BLOCK_ELEMS = 2048;
inline int scan_blocks(int n) { return n <= 0 ? 1 : (n + SCAN_BLOCK_ELEMS - 1) / SCAN_BLOCK_ELEMS; }

struct GeomLayout {
    float* rec;            // [P*24]
    float* cov3D;          // [P*6]
    uint32_t* clamped;     // [P] bit c set => SH colour channel c was clamped
    uint32_t* tiles;       // [2P] per Gaussian {tiles touched (0 => culled), tile rectangle x0 | y0 << 10 | width << 20}: ONE 8-byte gather tells
                           // the emit kernel everything it needs (it used to gather the count, the radius and the record's mean2D)
    uint32_t* key[2];      // [P] depth keys ping/pong
    uint32_t* idx[2];      // [P] Gaussian ids ping/pong (idx[final] = depth-sorted order)
    uint32_t* offsets;     // [P] exclusive scan of tiles in depth-sorted order
    uint32_t* scan_tmp;    // [2 * (scan_blocks(P)+1)]: block sums | per-block depth-key summaries
    uint32_t* radix_tbl;   // [256 * sort_blocks(P)]
    uint32_t* counters;    // [4]: [0] = R, [1] = prefilter violation, [2] = top bytes of the visible depth keys {AND << 8 | OR},
                           // [3] = visible span of the depth order (1 + position of the last Gaussian with tiles > 0)
    uint32_t* key_top;     // [ceil(P / 64)] the same per preprocess wave (identity 0xff00 where a wave has no visible Gaussian)
    // working set of the view for the fused shading (svgir_params.shade; subset.hip):
    uint8_t* needed;       // [P] 1 <=> receives a blend weight (cleared by preprocess, set by the contribution pre-pass of the composite)
    uint32_t* shade_list;  // [P] partition of 0..P-1: selected surfels in front (forward with pre-pass: `needed`; backward: out_weights > 0)
    uint32_t* shade_work;  // [partition_work_words(P)] scan scratch; its LAST word = number of selected surfels
    size_t bytes;
};
inline GeomLayout geom_layout(char* base, int P) {
    GeomLayout g;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t p = (size_t)(P > 0 ? P : 1);
    g.rec = (float*)take(p * REC * 4);
    g.cov3D = (float*)take(p * 6 * 4);
    g.clamped = (uint32_t*)take(p * 4);
    g.tiles = (uint32_t*)take(p * 8);
    g.key[0] = (uint32_t*)take(p * 4);
    g.key[1] = (uint32_t*)take(p * 4);
    g.idx[0] = (uint32_t*)take(p * 4);
    g.idx[1] = (uint32_t*)take(p * 4);
    g.offsets = (uint32_t*)take(p * 4);
    g.scan_tmp = (uint32_t*)take(((size_t)scan_blocks(P) + 1) * 8);
    g.radix_tbl = (uint32_t*)take(radix_table_words(P) * 4);
    g.counters = (uint32_t*)take(16);
    g.key_top = (uint32_t*)take((p + 63) / 64 * 4);
    g.needed = (uint8_t*)take(p);
    g.shade_list = (uint32_t*)take(p * 4);
    g.shade_work = (uint32_t*)take(((p + BLOCK * 8 - 1) / (BLOCK * 8) + 2) * 4);
    g.bytes = off;
    return g;
}

// Dispatch order of the composite forward's waves (ImageLayout::sub_order, written by order_desc_kernel): workgroup b takes sub-tile
// sub_order[b]; ORDER_NONE = padding (the workgroup exits).  Two orders exist (binning.hip): one global longest-first list, or one
// longest-first list per XCD, interleaved (entry 8 j + c = the j-th item of XCD c), where an XCD owns the image blocks of 4 x 4 tiles
// with (bx + 3 by) & 7 == c.  order_entries() = entries of the padded per-XCD form (>= 4 T): 8 x the longest XCD list.
constexpr uint32_t ORDER_NONE = 0xffffffffu;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t xcd_of_tile(int tx, int ty) { return (uint32_t)((tx >> 2) + 3 * (ty >> 2)) & 7u; }
inline size_t order_entries(int gx, int gy) {
    size_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int by = 0; 4 * by < gy; by++)
        for (int bx = 0; 4 * bx < gx; bx++) cnt[xcd_of_tile(4 * bx, 4 * by)] += (size_t)std::min(4, gx - 4 * bx) * (size_t)std::min(4, gy - 4 * by);
    size_t m = 0;
    for (size_t c : cnt) m = std::max(m, c);
    return std::max((size_t)8 * 4 * m, (size_t)4 * gx * gy);
}

// The same ownership for the per-tile cull (one workgroup per tile): work id q runs on XCD q & 7 and takes the (q >> 3)-th tile of that
// XCD in (block row, block, row-major inside the block) order -- the ~6 tiles a splat touches are then culled on ONE XCD and its
// 32-byte record header is fetched into one L2 instead of six (ORDER_NONE: past the end of the XCD's list; order_entries() / 4 ids).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t xcd_tile_of_work(uint32_t q, int gx, int gy) {
    const int c = (int)(q & 7u), nbx = (gx + 3) >> 2, nby = (gy + 3) >> 2;
    const int wl = gx - 4 * (nbx - 1);   // width of the last block of a row (1..4)
    uint32_t rem = q >> 3;
    for (int by = 0; by < nby; by++) {
        const int bh = gy - 4 * by < 4 ? gy - 4 * by : 4;
        const int r = (c - 3 * by) & 7;                               // this XCD's blocks of the row: r, r + 8, ...
        const int nblk = nbx > r ? (nbx - r + 7) / 8 : 0;
        const bool has_last = nblk > 0 && ((nbx - 1 - r) & 7) == 0;   // ... the row's last (possibly narrower) block among them
        const uint32_t cnt = (uint32_t)(bh * (4 * nblk - (has_last ? 4 - wl : 0)));
        if (rem >= cnt) { rem -= cnt; continue; }
        const int k = (int)(rem / (uint32_t)(bh * 4));                // (only the last block can be narrower, and it is the last of the list)
        const int bx = r + 8 * k;
        const uint32_t within = rem - (uint32_t)(k * bh * 4);
        const int bw = bx == nbx - 1 ? wl : 4;
        return (uint32_t)((4 * by + (int)(within / (uint32_t)bw)) * gx + 4 * bx + (int)(within % (uint32_t)bw));
    }
    return ORDER_NONE;
}

constexpr int SEG_CLASSES = 5, SEG_BLOCK_STRIDE = 8;   // backward-segment length classes (see SEG); per-256-tile-block counters padded to 8

struct ImageLayout {
    float* final_T;      // [N]
    float* final_D;      // [N]
    int32_t* n_contrib;  // [N]
    uint32_t* ranges;    // [2*T]
    uint32_t* seg_block; // [ceil(T / 256)][SEG_BLOCK_STRIDE] live backward segments per block of 256 tiles and length class (summed by the
                         // forward; directly behind ranges)
    uint32_t* sub_total; // [4*T] #entries of each 8x8 sub-tile's compact candidate list (written by the cull kernel)
    uint32_t* sub_order; // [order_entries] sub-tiles sorted by descending candidate count (heaviest work is dispatched first), see ORDER_NONE
    uint32_t* sub_count; // [4*T] #candidates the forward composite consumed before every pixel was done (<= sub_total)
    uint32_t* sub_ndump; // [4*T] #segment-boundary states the forward dumped for the sub-tile (see SEG)
    uint32_t* sub_pair_base; // [4*T] exclusive prefix of sub_total over the sub-tiles: first gradient row of a sub-tile's candidates
    uint32_t* sub_slot_base; // [4*T] exclusive prefix of seg_slots(sub_total): first dumped-state slot of a sub-tile (BinLayout::seg_state)
    uint32_t* counters;  // [8]: [0] = number of live backward segments (entries of BinLayout::seg_list); [1] = sum of sub_total (pairs);
                         // [2] = sum of seg_slots(sub_total) (state slots); [3] = magic, [4] = instance capacity, [5..6] = state-slot
                         // capacity (int64; -1: worst case) the forward laid the binning blob out for (api.hip view_from_blob)
    size_t ncontrib_off;
    size_t bytes;
};
inline ImageLayout image_layout(char* base, int W, int H) {
    ImageLayout im;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t N = (size_t)W * H;
    const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    im.final_T = (float*)take(N * 4);
    im.final_D = (float*)take(N * 4);
    im.ncontrib_off = off;
    im.n_contrib = (int32_t*)take(N * 4);
    im.ranges = (uint32_t*)take((T * 2 + (T + 255) / 256 * SEG_BLOCK_STRIDE) * 4);
    im.seg_block = im.ranges ? im.ranges + T * 2 : nullptr;
    im.sub_total = (uint32_t*)take(T * 4 * 4);
    im.sub_order = (uint32_t*)take(order_entries((W + TILE - 1) / TILE, (H + TILE - 1) / TILE) * 4);
    im.sub_count = (uint32_t*)take(T * 4 * 4);
    im.sub_ndump = (uint32_t*)take(T * 4 * 4);
    im.sub_pair_base = (uint32_t*)take(T * 4 * 4);
    im.sub_slot_base = (uint32_t*)take(T * 4 * 4);
    im.counters = (uint32_t*)take(32);
    im.bytes = off;
    return im;
}

// Gradient rows.  Instead of one float atomic per (wave, splat, output), a backward wave writes the complete gradient
// row of its (instance, sub-tile) pair and a second kernel sums each Gaussian's rows in a fixed order: no atomics, deterministic
// gradients.  Rows are COMPACT (round 4): row = sub_pair_base[sub-tile] + position of the candidate in the sub-tile's list, i.e. one
// row per pair that survived the cull (1.22 per instance at cfg2) instead of four per instance; the wave also records the row in the
// reverse map row_of[4 * (emit-order instance index) + sub-tile] = row + 1 (0: no row), where the reduce kernel -- which walks a
// Gaussian's instances -- finds it.  Scratch layout: row_of[4 * capacity] | rows[row capacity][RS].
// Row layout (floats): [0, NC0) colour3, normal3 (x10), depth, features S | pad to 4 | [P4, P4+VS) vfeatures |
// [P4+VS, P4+VS+6) mean2D.xy, conic.xyz, opacity | pad to 4.
struct GradRowGeom { int NC0, P4, VS, GEO, RS; };
inline GradRowGeom grad_row_geom(int S, int VS) {
    GradRowGeom g;
    g.NC0 = 7 + S; g.P4 = (g.NC0 + 3) / 4 * 4; g.VS = VS; g.GEO = g.P4 + VS; g.RS = (g.GEO + 6 + 3) / 4 * 4;
    return g;
}
inline size_t grad_rowof_bytes(int cap) { return align_up((size_t)4 * cap * 4); }
inline size_t grad_scratch_bytes(int cap, size_t rows, int S, int VS) {   // reverse map + `rows` gradient rows
    return grad_rowof_bytes(cap) + align_up(rows * grad_row_geom(S, VS).RS * 4);
}

// Backward segments.  The backward composite is parallelised over depth: one wave per SEG consecutive candidates of a
// sub-tile's compact list.  The forward dumps its per-pixel blend state (T and every accumulator) after each SEG-th
// candidate and once at the end; a backward segment starts its back-to-front replay from the state at its far end
// (transmittance there, and "everything behind" = (final - prefix) / T) instead of from the end of the list.
// State slots are COMPACT (round 4): a sub-tile whose list has `total` candidates dumps at most total / SEG boundary states + the final
// one, and none at all below SEG candidates -- seg_slots(total) -- and its first slot is the exclusive prefix of that over the
// sub-tiles (ImageLayout::sub_slot_base, a by-product of order_desc_kernel); slot of (sub-tile, segment k) = sub_slot_base + k.
// After the forward composite seg_build_kernel lists the live segments -- seg_list (compact ids) and seg_desc (everything a
// backward wave needs to start: SegDesc) -- LONGEST FIRST: class 0 = full segments (SEG candidates), classes 1..4 = the
// partial last segments of the sub-tiles by length quarter; tile order inside a class.  The backward's waves take the list
// round-robin, so the last, partly filled round consists of the shortest items (a kernel lasts ceil(items / resident waves)
// rounds: with ~2.1 rounds of equal items a third of the time was the tail).  Work ids map to list positions in blocks of
// SEG_XCD_BLOCK consecutive items per XCD (seg_item_of): neighbouring sub-tiles share splat records, gradient planes and
// dumped states, which then stay in ONE 4 MiB L2.
constexpr int SEG = 64;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t seg_slots(uint32_t total) { return total >= (uint32_t)SEG ? total / (uint32_t)SEG + 1u : 0u; }
// upper bound of the number of live segments (seg_list / seg_desc) and -- the worst case over the cull, 4 full lists per tile -- of
// the state slots
inline size_t seg_capacity(int R, int T) { return (size_t)4 * (size_t)(R > 0 ? R : 0) / SEG + (size_t)4 * T + 1; }
constexpr int SEG_K_BITS = 14;   // seg_list entry = (sub-tile id << SEG_K_BITS) | k
struct SegDesc { uint32_t sm, r0, len, count, ndump, pair_base, slot_base, pad2; };   // 32 B: seg_list entry, tile range start / length, sub_count, sub_ndump, first gradient row / first state slot of the sub-tile
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int seg_class(uint32_t nent) { return nent >= (uint32_t)SEG ? 0 : 4 - (int)(nent * 4u / (uint32_t)SEG); }   // 64 | 48-63 | 32-47 | 16-31 | 1-15
// Work id w (workgroups are dealt to the XCDs round-robin: w & 7 = XCD) -> list position: XCD c takes the blocks c, c + 8,
// c + 16, ... of SEG_XCD_BLOCK consecutive items.  Independent of the item count (a wave can fetch its descriptor and the
// count with two independent loads); ids >= 8 * ceil(n / 8 / B) * B ... are simply past the end (position >= n).
#ifndef SEG_XCD_BLOCK_V
#define SEG_XCD_BLOCK_V 32
#endif
constexpr uint32_t SEG_XCD_BLOCK = SEG_XCD_BLOCK_V;   // (experiment builds: -DSEG_XCD_BLOCK_V=...)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t seg_item_of(uint32_t w) {
    const uint32_t c = w & 7u, q = w >> 3;
    return ((q / SEG_XCD_BLOCK) * 8u + c) * SEG_XCD_BLOCK + (q % SEG_XCD_BLOCK);
}
// number of work ids that cover n items
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t seg_work_ids(uint32_t n) { return (n + 8u * SEG_XCD_BLOCK - 1u) / (8u * SEG_XCD_BLOCK) * (8u * SEG_XCD_BLOCK); }

// The binning blob is laid out for an instance CAPACITY (a multiple of 4096 >= R): the forward may size it from a
// guess before the host knows R, and the backward recovers the capacity from the blob's byte size.
inline int binning_capacity(long long R) {
    const long long c = ((R > 0 ? R : 1) + 4095) / 4096 * 4096;
    return (int)(c > 0x7ffff000LL ? 0x7ffff000LL : c);
}

// Tile-sort plan: #bits of the tile id split into equal passes of <= 8 bits (both fwd and bwd derive the final
// ping/pong slot from it) -- or, up to 4096 tiles (800 x 800 and below), ONE counting pass over the whole tile id (binning.hip
// launch_tile_sort12: histogram, column prefix, stable scatter; the scatter also knows every tile's range, so no ranges kernel).
struct TileSortPlan { int bits, passes, bits_per_pass; bool single; };
constexpr int TS12_BINS = 4096, TS12_KEYS = 2048;   // bins of the single-pass tile sort; keys per workgroup
inline bool tile_sort_single_enabled() {
    static const bool on = [] { const char* e = getenv("SVGIR_TILE_SORT12"); return !(e && e[0] == '0'); }();
    return on;
}
inline TileSortPlan tile_sort_plan(int T) {
    int bits = 1;
    while ((1 << bits) < T) bits++;
    TileSortPlan p;
    p.bits = bits;
    p.single = T <= TS12_BINS && tile_sort_single_enabled();
    p.passes = p.single ? 1 : (bits + 7) / 8;
    p.bits_per_pass = p.single ? bits : (bits + p.passes - 1) / p.passes;
    return p;
}
// scratch of the single-pass sort for up to n keys: per-block histogram rows + the column totals
inline size_t tile12_table_words(int n) { return ((size_t)((n > 0 ? n : 1) + TS12_KEYS - 1) / TS12_KEYS + 1) * TS12_BINS; }

struct BinLayout {
    uint32_t* key[2];  // [R] tile ids ping/pong
    uint32_t* val[2];  // [R] Gaussian ids ping/pong
    uint32_t* radix_tbl;
    uint2* sub_list;   // [4*R] compact per-sub-tile candidate lists {Gaussian id, slot in the tile list}; sub-tile w
                       // of a tile with range [r0,r1) owns entries [4*r0 + w*(r1-r0), 4*r0 + (w+1)*(r1-r0))
    uint32_t* seg_list; // [seg_capacity] live backward segments, longest first (seg_build_kernel)
    SegDesc* seg_desc;  // [seg_capacity + 256] their descriptors (same order)
    float* seg_state;  // [slot_cap][nstate][64] dumped forward states: T, colour3, normal3, depth, feature S, vfeature VC
    size_t seg_cap;    // entries of seg_list / seg_desc
    size_t slot_cap;   // state slots of seg_state
    size_t bytes;
};
inline int seg_nstate(int S, int VS) { return 8 + S + VS / 4; }
// `slots` < 0: the worst case (seg_capacity); else a state-slot capacity chosen by the forward (from the pair statistics of recent
// views); such blobs are marked by an odd multiple of 128 bytes, so a size alone tells which kind it is
inline BinLayout bin_layout(char* base, int R, int T, int nstate, long long slots = -1) {
    BinLayout b;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t r = (size_t)(R > 0 ? R : 1);
    b.key[0] = (uint32_t*)take(r * 4);
    b.key[1] = (uint32_t*)take(r * 4);
    b.val[0] = (uint32_t*)take(r * 4);
    b.val[1] = (uint32_t*)take(r * 4);
    b.radix_tbl = (uint32_t*)take(std::max(radix_table_words(R), tile_sort_plan(T).single ? tile12_table_words(R) : (size_t)0) * 4);
    b.sub_list = (uint2*)take(r * 4 * 8);
    b.seg_cap = seg_capacity(R, T);
    b.seg_list = (uint32_t*)take(b.seg_cap * 4);
    b.seg_desc = (SegDesc*)take((b.seg_cap + 8 * SEG_XCD_BLOCK) * sizeof(SegDesc));
    b.slot_cap = slots < 0 ? b.seg_cap : (size_t)slots;
    b.seg_state = (float*)take((b.slot_cap > 0 ? b.slot_cap : 1) * (size_t)nstate * 64 * 4);
    b.bytes = off + (slots < 0 ? 0 : 128);
    return b;
}
inline bool bin_bytes_compact(size_t bytes) { return bytes % 256 == 128; }

// inverse of bin_layout(...).bytes (worst-case state slots) over the capacities binning_capacity() produces (bytes is strictly increasing in it)
inline int binning_capacity_from_bytes(size_t bytes, int T, int nstate) {
    long long lo = 1, hi = 0x7ffff000LL / 4096;
    while (lo < hi) {
        const long long mid = (lo + hi) / 2;
        if (bin_layout(nullptr, (int)(mid * 4096), T, nstate).bytes < bytes) lo = mid + 1; else hi = mid;
    }
    return (int)(lo * 4096);
}

// The svgss `config` tensor ([surface, normalize_depth, per_pixel_depth, (lrn_cam)]) stays on the device, exactly as
// in the reference (kernels read config[i] > 0); entries >= len read as false (quirk Q7).  rgss uses the
// compile-time constant {1,1,1} of its auxiliary.h:41-46, encoded as len < 0.
struct CfgRef { const float* ptr; int len; };
#if defined(__HIPCC__)
__device__ __forceinline__ bool cfg_flag(const CfgRef& c, int i) {
    if (c.len < 0) return i < 3;
    return i < c.len && c.ptr[i] > 0.f;
}
#endif

// the two small tables of the shading kernels (shade_tables.hpp), built in passing by kernels of the fused path
struct ShadeTables {
    const float* env = nullptr; float4* env_tab = nullptr; int ntexel = 0, softplus = 0;   // env == nullptr: nothing to do
    int Ns = 0; float4* lat_tab = nullptr;                                                  // lat_tab == nullptr: directions are streamed
    float* zero = nullptr; int nzero = 0;                                                   // backward: the env-gradient accumulator
#if defined(__HIPCC__)
    __host__ __device__
#endif
    int entries() const { return env ? (ntexel > (lat_tab ? Ns : 0) ? ntexel : (lat_tab ? Ns : 0)) : 0; }
};

// ---- kernel argument blocks --------------------------------------------------------------------------------
struct PreArgs {
    int P, D, M, W, H, gx, gy;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *campos, *patchbbox;
    float scale_modifier, tanx, tany, focal_x, focal_y;
    CfgRef cfg;
    float* rec; float* cov3D; uint32_t* clamped; uint32_t* tiles; uint32_t* key; uint32_t* idx; int32_t* radii;
    float* out_weights;              // [P] zeroed here (accumulated with atomics by the composite)
    uint8_t* needed;                 // [P] or null: zeroed here (set by the contribution pre-pass of the fused shading)
    uint32_t* span;                  // GeomLayout::counters + 3, zeroed here
    ShadeTables tabs; int pblocks;   // fused shading: workgroups >= pblocks build the shading tables (shade_tables.hpp)
    uint32_t* zero_words; int n_zero_words;   // small table cleared in passing (the depth sort's group totals)
    int spec_top; uint32_t* key_top;   // speculated common top byte of the visible depth keys (-1: none) -> key of a culled Gaussian; per-wave summary out
    uint32_t* prefilter_violation;   // non-null <=> `prefiltered`: set to 1 when a frustum / back-face cull fires (auxiliary.h:163-167)
    const float* features; int embed_S;   // rec_embeds_features(): the [P][embed_S] feature rows copied into the records (0: none)
};

struct RenderArgs {
    int W, H, gx, gy, S, VS;
    const uint32_t* ranges; const uint32_t* point_list;
    const float* rec; const float* features; const float* vfeatures;
    const float* bg;
    CfgRef cfg;
    uint2* sub_list; uint32_t* sub_total; uint32_t* sub_order; uint32_t* sub_count;
    uint32_t* sub_ndump; uint32_t* seg_list; SegDesc* seg_desc; uint32_t* seg_count; uint32_t* seg_block; float* seg_state;
    uint32_t* sub_pair_base; uint32_t* sub_slot_base; uint32_t slot_cap;
    int bg_in_render;   // 1: the pixels of EMPTY tiles and the all-zero planes (zero_a / zero_b) are written by the composite kernel's waves -- the
                        // empty sub-tiles' waves run last, in the kernel's idle tail -- instead of by the cull kernel (specialised kernels)
    int order_n;     // entries of sub_order = workgroups of the composite forward (4 T, or order_entries() with one list per XCD)
    int hi_fill;     // != 0: the launch has many rounds of waves (api.hip guess_fill): the composite's high-occupancy variant (render_fwd.hip)
    int dump_only;   // composite forward: 1 = replay for the state dumps alone (svgir_backward, a view that exceeded its slot capacity): no output
                     // is written
    float *final_T, *final_D; int32_t* n_contrib;
    float *out_color, *out_normal, *out_depth, *out_opacity, *out_feature, *out_vfeature, *out_weights;
    float *zero_a, *zero_b;   // [3,H,W] planes the cull pass clears (rgss pseudo normal / surface xyz when not computed), or null
    uint8_t* needed;          // [P] or null: contrib_prepass_kernel marks every Gaussian that receives a blend weight
};

struct RenderBwdArgs {
    int W, H, gx, gy, S, VS;
    const uint32_t* ranges; const uint32_t* point_list;
    const float* rec; const float* features; const float* vfeatures;
    const float* bg;
    CfgRef cfg; int backward_geometry;
    const uint2* sub_list; const uint32_t* sub_count;
    const uint32_t* sub_ndump; const uint32_t* seg_list; const SegDesc* seg_desc; const uint32_t* seg_count; const float* seg_state; int seg_cap;
    const float *final_T, *final_D; const int32_t* n_contrib;
    const float *g_color, *g_normal, *g_depth, *g_opacity, *g_feature, *g_vfeature;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dfeature, *dL_dvfeature, *dL_dnormal, *dL_ddepth;
    float* grad_rows; uint32_t* row_of; uint32_t rows_cap;   // svgss (VS > 0): compact gradient rows + reverse map; else: packed rows [P][RS]
    uint4* clear; size_t clear_n16;         // the caller's gradient allocation, zeroed in passing by the composite backward's waves (or null)
};

struct GradReduceArgs {
    const uint32_t* list; const uint32_t* list_count;   // optional: walk only the Gaussians list[0 .. *list_count) (those that were blended)
    int P, S, VS;
    const int32_t* radii; const uint32_t* tiles; const float* rec;
    const float* grad_rows; const uint32_t* row_of;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dfeature, *dL_dvfeature, *dL_dnormal, *dL_ddepth;
};

struct GeomBwdArgs {
    const uint32_t* list; const uint32_t* list_count;   // optional: as GradReduceArgs (all other Gaussians keep their cleared, zero gradients)
    int P, D, M;
    const float *means3D, *shs, *scales, *rotations, *cov3D, *view, *proj, *campos;
    const int32_t* radii; const uint32_t* clamped;
    float scale_modifier, tanx, tany, focal_x, focal_y;
    CfgRef cfg; int svgss;
    // composite gradients: the caller's tensors; when `packed` is set (one GradRowGeom(S, 0) row per Gaussian, written by
    // the backward composite) the kernel first unpacks the Gaussian's row into them
    float *dL_dmean2D, *dL_dconic, *dL_dcolor, *dL_dnormal, *dL_ddepth, *dL_dopacity, *dL_dfeature;
    const float* packed; int S;
    float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot, *dL_dviewmat, *dL_dprojmat, *dL_dcampos;
};

// ---- stage timing shared by the entry points (api.hip owns the state; no-ops unless svgir_set_profiling(1)) ----
struct StageMarks { hipStream_t s; bool on; hipEvent_t prev; };
StageMarks stage_begin(hipStream_t s);
void stage_mark(StageMarks& t, const char* name);   // `name` = the stage that ENDS here (static string)

// ---- host-side launchers (one per .hip file) ---------------------------------------------------------------
void launch_preprocess(const PreArgs& a, bool svgss, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
// stable LSD radix sort of (u32 key, u32 value) pairs on bits [0, total_bits) in passes of bits_per_pass (<= 8);
// input in slot 0 of the ping/pong buffers, result in slot (passes & 1); table: radix_table_words(n) counters
// element count = n, or min(n, *n_dev) read on the device when n_dev != nullptr (n then only sizes launch + scratch)
void launch_radix_sort(uint32_t* const key[2], uint32_t* const val[2], int n, const uint32_t* n_dev, int total_bits,
                       int bits_per_pass, uint32_t* table, hipStream_t s);
// offsets[i] = exclusive prefix sum of tiles[order[i]]; total -> *total_out
// total_out[2] = the per-wave depth-key summaries `key_top[n_key_top]` folded into one word; host_out (pinned host memory, or null):
// {host_tag << 32 | total}, {host_tag << 32 | (*violation != 0) << 16 | that word} stored by the last block
void launch_offsets_scan(const uint32_t* tiles, const uint32_t* order, uint32_t* offsets, uint32_t* scan_tmp, int n,
                         uint32_t* total_out, const uint32_t* key_top, int n_key_top, const uint32_t* violation,
                         unsigned long long* host_out, uint32_t host_tag, hipStream_t s);
// also clears ranges[2*gx*gy] + the per-tile-block segment counts behind it, the live-segment counter and the group totals of the tile sort's table (`sort_table`,
// sized for `cap` elements)
void launch_emit(int P, const uint32_t* order, const uint32_t* tiles, const uint32_t* offsets, float* rec,
                 const int32_t* radii, int gx, int gy, uint32_t* tile_keys, uint32_t* vals, int cap, uint32_t* ranges,
                 uint32_t* seg_count, uint32_t* sort_table, const uint32_t* span, hipStream_t s);   // span: GeomLayout::counters + 3
void launch_ranges(int R, const uint32_t* R_dev, const uint32_t* tile_keys, uint32_t* ranges, int T, hipStream_t s);
// single-pass stable counting sort of (tile id, value) pairs for T <= TS12_BINS tiles: slot 0 -> slot 1; writes ranges[2 T] (zero for empty
// tiles, like identifyTileRanges); table: tile12_table_words(n)
void launch_tile_sort12(uint32_t* const key[2], uint32_t* const val[2], int n, const uint32_t* n_dev, uint32_t* table, uint32_t* ranges, int T,
                        hipStream_t s);
// order[] = item ids sorted by descending counts[] (longest-processing-time-first dispatch of the composite waves); also
// prefix[i] = exclusive prefix sum of counts[], slot_prefix[i] = the same of seg_slots(counts[]), totals[1] / totals[2] = the two sums,
// host_totals[0] / [1] = host_tag << 32 | sum in pinned host memory (any of them may be null)
// totals[3..6] = {magic, cap_R, cap_slots lo, hi}: the capacities of the launch sequence, kept in the image blob
// gx: tiles per image row; order_n: entries of order[] (>= n; the rest is padded with ORDER_NONE); per_xcd: one longest-first list per XCD
void launch_order_desc(const uint32_t* counts, int n, uint32_t* order, uint32_t* prefix, uint32_t* slot_prefix, uint32_t* totals,
                       unsigned long long* host_totals, uint32_t host_tag, uint32_t cap_R, long long cap_slots, uint32_t magic, int gx, int order_n,
                       bool per_xcd, hipStream_t s);
// per-tile cull of the depth-ordered splat lists against the four 8x8 sub-tiles -> sub_list, sub_total
void launch_cull(const RenderArgs& a, hipStream_t s);
// fused shading: needed[id] = 1 for every surfel that receives a blend weight in this view (the composite's transmittance walk alone)
void launch_contrib_prepass(const RenderArgs& a, hipStream_t s);
// tile-ordered list of the live backward segments (seg_list, seg_desc, seg_count) from the forward's sub_count / sub_ndump; also zeroes
// `clear_bytes` bytes at `clear` (a multiple of 16; the backward's scratch clear rides on this launch)
void launch_seg_build(const RenderArgs& a, void* clear, size_t clear_bytes, const ShadeTables& tabs, const float* weights, int P, uint32_t* part_sums,
                      hipStream_t s);   // (+ the shading tables of the fused path; + per-chunk counts of weights > 0 for launch_partition_scatter)
int launch_render_fwd(const RenderArgs& a, bool svgss, hipStream_t s);      // <0 (nothing launched) if (S,VS) has no specialised kernel
int launch_render_bwd(const RenderBwdArgs& a, bool svgss, hipStream_t s);  // <0 (nothing launched) if (S,VS) has no specialised kernel
int launch_render_bwd_plain(const RenderBwdArgs& a, bool svgss, hipStream_t s);   // VS = 0 widths (render_bwd_plain.hip); <0 if not specialised
bool render_specialised(int S, int VS, bool svgss);
// run-time-width composite kernels for every other (S, VS) the reference accepts (render_generic.hip)
void launch_render_fwd_generic(const RenderArgs& a, bool svgss, hipStream_t s);
void launch_render_bwd_generic(const RenderBwdArgs& a, bool svgss, hipStream_t s);
void launch_grad_reduce(const GradReduceArgs& a, hipStream_t s);
void launch_geom_bwd(const GeomBwdArgs& a, hipStream_t s);
void launch_image_ops(int W, int H, const float* view, float focal_x, float focal_y, float cx, float cy,
                      const float* opacity, const float* depth, float* pseudo_normal, float* surface_xyz,
                      hipStream_t s);

// ---- working set of a view (subset.hip): list[0 .. *count_dev) = the surfels with flags[i] != 0 (or, flags == nullptr, positive[i] > 0)
// in index order, list[P-1-j] = the j-th other one; work: partition_work_words(P) uint32 of scratch
constexpr int PART_ELEMS = BLOCK * 8;   // surfels per partition workgroup
// the second half alone: `work` already holds the per-chunk counts of selected surfels (chunk = PART_ELEMS consecutive surfels), e.g.
// counted in passing by the live-segment kernel of the backward
void launch_partition_scatter(int P, const float* positive, uint32_t* list, const uint32_t* work, uint32_t* count_dev, hipStream_t s);
size_t partition_work_words(int P);
void launch_partition(int P, const uint8_t* flags, const float* positive, uint32_t* list, uint32_t* work, uint32_t* count_dev, hipStream_t s);
// zeroes row list[P-1-j], j < P - *count_dev, of up to 6 row-major fp32 tensors (null tensors are skipped)
void launch_zero_rows(int P, const uint32_t* list, const uint32_t* count_dev, float* const* tensors, const int* row_floats, int n, hipStream_t s);

// svgir_shade_backward with one more switch: rows_precleared = the four per-surfel gradient tensors are already zero (svgir_backward
// lets the composite backward's waves clear them in passing), so the rows outside the subset need no zero-fill launch
// tables_ready = the f(env) / lattice tables (and, backward, the zeroed env-gradient accumulator) were already produced by a kernel in
// front (shade_tables.hpp): no prologue launch
ShadeTables shade_tables(const svgir_shade_params* p, float* zero, int nzero);
int shade_forward_impl(const svgir_shade_params* p, float* reduced, float* features, float* vfeatures, bool tables_ready, void* stream);
int shade_backward_impl(const svgir_shade_params* p, const float* dL_dreduced, const float* dL_dfeatures, const float* dL_dvfeatures,
                        float* dL_dbase_color, float* dL_droughness, float* dL_dnormals, float* dL_dradiance, float* dL_denv,
                        float* env_grad_work, float* dL_dradiance_ratio, bool rows_precleared, bool tables_ready, void* stream);

#if defined(__HIPCC__)
// ---- device helpers ----------------------------------------------------------------------------------------
// Wave64 sum with DPP row shifts + row broadcasts (gfx9 family); the total lands in lane 63.  All 64 lanes must
// be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {   // v of the lane DPP control CTRL names; 0 where it names none
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_scan_last(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {  // uniform result
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_scan_last(v)), 63));
}
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
// Ordering point for LDS traffic that is private to ONE wave (single-wave workgroups, or per-wave LDS regions): the DS
// operations of a wave execute in issue order, so only the compiler has to be kept from moving LDS accesses across it.
// Unlike __syncthreads() it does not drain vmcnt, i.e. it never waits for outstanding global atomics / stores.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

}  // namespace svgir
