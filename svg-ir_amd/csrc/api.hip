// svg-ir_amd/csrc/api.hip -- C ABI (include/svgir_raster.h) and host-side orchestration of the kernels.
//
// Host counterpart of CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (svgss rasterizer_impl.cu:141-153, 209-382, 386-523; rgss :141-153, 209-407, 411-535).
// Every kernel is launched on the caller's stream (the reference uses the legacy default stream); the only
// host synchronisation is the 4-byte read of the instance count R that sizes the binning blob -- the same one
// the reference has at rasterizer_impl.cu:311 -- plus, when svgir_set_profiling(1), one sync per call to read
// the HIP event timings.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

using namespace svgir;

namespace {

thread_local std::string g_err;
// Profiling state is process-wide (the autograd engine runs backward on its own thread).  Stage boundaries are
// HIP events recorded on the launch stream; they are resolved lazily (svgir_last_timings), so enabling profiling
// adds no synchronisation to forward/backward.
std::atomic<bool> g_prof{false};
// Speculative-capacity history: the instance counts of the last eight forwards PER WORKLOAD KEY (device, image size,
// Gaussian count, channel widths, variant), so that scenes / resolutions that alternate in one process (a 256x256
// preview next to a 1600x1600 render, several scenes, several devices) neither re-run each other's dependent stages nor
// over-allocate each other's blobs.  A small fixed table, least-recently-used replacement.
struct CapKey { int dev, W, H, P, S, VS, variant, scope; };
// (P is NOT part of a workload's identity as long as it moves slowly: densification / pruning changes it every few hundred iterations,
// scene/gaussian_model.py:1229-1253, and the history must survive that -- every sample remembers the Gaussian count it was taken at and
// is scaled to the caller's: instances and state slots grow with the surfel count on a fixed view.  A caller whose P is more than a
// factor of two away from the entry's latest sample is another model: it gets its own entry, and a scaled sample never exceeds four
// times the largest unscaled one.  `scope` = svgir_params.workload_scope: models that share (device, image size, widths, variant) keep
// separate histories by giving each its own id.)
struct CapEntry { CapKey key; int hist[8]; int hist_P[8]; long long hist_slots[8]; int hist_slots_P[8]; unsigned next, next_slots; unsigned long long stamp; bool used;
                  long long fill; int fill_P;   // non-empty 8x8 sub-tiles of the workload's latest view (-1 / 0: none seen), and its Gaussian count
                  int top_byte, top_streak;   // common top byte of the visible depth keys of the last `top_streak` views (0: none / not common)
                  const void* last_view; int last_view_P; };   // image blob of the workload's latest forward: its slot total is read when the next one starts
std::mutex g_cap_mu;
CapEntry g_cap[16];
unsigned long long g_cap_clock = 0;
bool same_key(const CapKey& a, const CapKey& b) {   // a: the entry's key (P = the Gaussian count of its latest sample), b: the caller's
    if (!(a.dev == b.dev && a.W == b.W && a.H == b.H && a.S == b.S && a.VS == b.VS && a.variant == b.variant && a.scope == b.scope)) return false;
    return a.P <= 0 || b.P <= 0 || ((long long)a.P <= 2ll * b.P && (long long)b.P <= 2ll * a.P);
}
CapEntry* cap_entry(const CapKey& k, bool create) {
    CapEntry* lru = &g_cap[0];
    for (auto& e : g_cap) {
        if (e.used && same_key(e.key, k)) { e.stamp = ++g_cap_clock; return &e; }
        if (!e.used) { if (lru->used) lru = &e; }
        else if (lru->used && e.stamp < lru->stamp) lru = &e;
    }
    if (!create) return nullptr;
    *lru = CapEntry{};
    lru->key = k; lru->used = true; lru->stamp = ++g_cap_clock;
    return lru;
}
inline long long scale_to(long long v, int from_P, int to_P) {   // a count measured at from_P Gaussians, expected at to_P
    if (from_P <= 0 || from_P == to_P) return v;
    return (long long)((double)v * (double)to_P / (double)from_P) + 1;
}
int guess_R(const CapKey& k) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    const CapEntry* e = cap_entry(k, false);
    long long m = 0, raw = 0;
    if (e) for (unsigned i = 0; i < std::min(e->next, 8u); i++) { m = std::max(m, scale_to(e->hist[i], e->hist_P[i], k.P)); raw = std::max<long long>(raw, e->hist[i]); }
    return (int)std::min<long long>(std::min(m, 4 * raw), 0x7ffff000LL);
}
void record_R(const CapKey& k, int R) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    CapEntry* e = cap_entry(k, true);
    e->hist[e->next % 8] = R; e->hist_P[e->next % 8] = k.P;
    e->next++;
    e->key.P = k.P;   // (the entry follows its model's Gaussian count)
}
// state slots (common.hpp seg_slots summed over the sub-tiles) of recent views of the workload: -1 = none seen yet
long long guess_slots(const CapKey& k) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    const CapEntry* e = cap_entry(k, false);
    long long m = -1, raw = 0;
    if (e && e->next_slots) for (unsigned i = 0; i < std::min(e->next_slots, 8u); i++) { m = std::max(m, scale_to(e->hist_slots[i], e->hist_slots_P[i], k.P)); raw = std::max(raw, e->hist_slots[i]); }
    return m < 0 ? m : std::min(m, 4 * raw + 64);
}
void record_slots(const CapKey& k, long long slots, long long nonempty) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    CapEntry* e = cap_entry(k, true);
    e->hist_slots[e->next_slots % 8] = slots; e->hist_slots_P[e->next_slots % 8] = k.P;
    e->next_slots++;
    if (nonempty >= 0) { e->fill = nonempty + 1; e->fill_P = k.P; }   // (stored + 1: a zero-initialised entry has seen none)
}
// The FILL of the composite launch: non-empty sub-tiles (= waves with work) of the workload's latest view.  The forward composite exists in
// two occupancy variants (render_fwd.hip): below ~4 rounds of waves the machine is under-filled and the variant with more registers per
// wave wins; above, the one with more resident waves.  -1: no view seen yet.
long long guess_fill(const CapKey& k) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    const CapEntry* e = cap_entry(k, false);
    return (e && e->fill > 0) ? e->fill - 1 : -1;
}
// Depth-key speculation.  The depth keys are positive floats; in a bounded scene they share their top byte (sign + 7 exponent bits: all
// depths in [2, 8), or [8, 32) ...), and then the fourth 8-bit pass of the depth sort orders nothing.  The preprocess reports AND / OR of
// the visible keys' top bytes (read back with the instance count); once kTopStreak consecutive views of a workload had one common byte, the
// next view is launched with three passes, its culled keys carrying that byte -- and re-run from scratch, with four, if a visible key
// turns out to differ (the streak then starts over, so at most one view in kTopStreak + 1 can ever be re-run).
constexpr int kTopStreak = 3;
// {forwards, re-runs for the instance capacity, for the state-slot capacity, for the depth-key byte, views sorted in three passes}
std::atomic<long long> g_spec_stats[5];
int guess_top(const CapKey& k) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    const CapEntry* e = cap_entry(k, false);
    return (e && e->top_streak >= kTopStreak) ? e->top_byte : -1;
}
void record_top(const CapKey& k, uint32_t summary) {   // {AND << 8 | OR}; AND = 0xff, OR = 0: nothing visible (no information)
    const int av = (int)((summary >> 8) & 0xffu), ov = (int)(summary & 0xffu);
    if (av == 0xff && ov == 0) return;
    std::lock_guard<std::mutex> lk(g_cap_mu);
    CapEntry* e = cap_entry(k, true);
    if (av != ov) { e->top_streak = 0; return; }
    if (e->top_streak > 0 && e->top_byte == av) e->top_streak = std::min(e->top_streak + 1, 1 << 20);
    else { e->top_byte = av; e->top_streak = 1; }
}
// Pinned landing slots for the instance count (+ prefilter violation + depth-key summary): the scan's last block stores them there,
// tagged, with system-scope stores -- no copy operation and no event on the stream -- and the host spins on the tag.  A slot belongs to
// ONE forward from its launch until that forward has read the count (a free list, not a ring: any number of forwards may be in flight
// across threads / streams; the ones that find no free slot read the count with a blocking copy like the reference does,
// rasterizer_impl.cu:307-312).
unsigned long long* g_pinned = nullptr;
std::atomic<uint32_t> g_pinned_tag{0};
constexpr unsigned kPinnedSlots = 256;
std::once_flag g_pinned_once;
std::mutex g_pinned_mu;
uint64_t g_pinned_busy[kPinnedSlots / 64];   // bit set: the slot is owned by a forward in flight
struct PinnedSlot {
    unsigned long long* at = nullptr; uint32_t tag = 0; int index = -1;
    PinnedSlot() {
        std::call_once(g_pinned_once, [] {
            void* ptr = nullptr;
            // (explicitly host-coherent: the device's system-scope stores must become visible to the polling host thread)
            if (hipHostMalloc(&ptr, kPinnedSlots * 2 * sizeof(unsigned long long), hipHostMallocCoherent) == hipSuccess) {
                g_pinned = (unsigned long long*)ptr;
                for (unsigned i = 0; i < 2 * kPinnedSlots; i++) g_pinned[i] = 0ull;
            } else {
                (void)hipGetLastError();
            }
        });
        uint32_t t = ++g_pinned_tag;
        if (t == 0) t = ++g_pinned_tag;   // (never 0: the slots start as 0)
        tag = t;
        if (!g_pinned) return;
        // (SVGIR_PINNED_SLOTS: fewer slots, for tests of the no-free-slot path)
        static const unsigned usable = [] { const char* e = getenv("SVGIR_PINNED_SLOTS"); return e ? (unsigned)std::min<long>(std::max<long>(atol(e), 0), kPinnedSlots) : kPinnedSlots; }();
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        for (unsigned i = 0; i < usable; i++)
            if (!(g_pinned_busy[i / 64] >> (i % 64) & 1ull)) {
                g_pinned_busy[i / 64] |= 1ull << (i % 64);
                index = (int)i;
                at = g_pinned + 2 * index;
                return;
            }
    }
    ~PinnedSlot() {
        if (index < 0) return;
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned_busy[index / 64] &= ~(1ull << (index % 64));
    }
    PinnedSlot(const PinnedSlot&) = delete;
    PinnedSlot& operator=(const PinnedSlot&) = delete;
};
// How long a host thread polls pinned memory for a tagged word before it stops polling and blocks on the stream instead (the value is
// typically tens of microseconds away; a longer wait means the stream holds a backlog -- tracer updates, a shared GPU, a profiler that
// serialises kernels -- and then blocking is the right way to wait).  SVGIR_SPIN_MS overrides it (tests use 0: always the blocking path).
double spin_budget_s() {
    static const double v = [] {
        const char* e = getenv("SVGIR_SPIN_MS");
        return e ? std::max(0.0, atof(e)) * 1e-3 : 0.05;
    }();
    return v;
}
inline bool tagged_pair(volatile unsigned long long* at, uint32_t tag, uint32_t* w0, uint32_t* w1) {
    const unsigned long long v0 = at[0], v1 = at[1];
    if ((uint32_t)(v0 >> 32) != tag || (uint32_t)(v1 >> 32) != tag) return false;
    *w0 = (uint32_t)v0; *w1 = (uint32_t)v1;
    return true;
}
// polls for the two tagged words; false when the spin budget ran out (NOT an error: the caller then blocks on the stream and looks again)
bool pinned_spin(volatile unsigned long long* at, uint32_t tag, uint32_t* w0, uint32_t* w1) {
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const double budget = spin_budget_s();
    for (long long spin = 0;; spin++) {
        if (tagged_pair(at, tag, w0, w1)) return true;
        if ((spin & 255) == 255 || budget == 0.0) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double el = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            if (el >= budget) return false;
            if (el > 5e-3) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
        }
    }
}
// Per-view counts that exist only behind the cull -- the surviving (sub-tile, instance) pairs (= gradient rows the svgss backward needs)
// and the state slots the composite forward may dump into -- reach the host as tagged 8-byte stores of order_desc_kernel into pinned
// memory (no copy operation, no event on the stream) and are kept per IMAGE BLOB together with the capacities the forward laid the
// binning blob out for: the backward and svgir_backward_scratch_bytes_for() find them there.  A host wait right behind the cull costs
// nothing: the composite is still queued (measured with a full event synchronisation there: 0.4353 vs 0.4361 ms per cfg2 step).
struct ViewEntry { const void* key = nullptr; uint32_t tag = 0; int cap_R = 0; long long cap_slots = -1; unsigned long long stamp = 0;
                   bool recorded = false;   // its slot total has entered the workload's history
                   hipStream_t stream = nullptr; bool has_stream = false; };   // the stream the forward ran on: what a waiter without a stream of its own blocks on   // its slot total has entered the workload's history
constexpr int kViewEntries = 1024;   // forwards whose backward may still come (least recently used entry replaced)
std::mutex g_view_mu;
ViewEntry g_view[kViewEntries];
constexpr int kViewWords = 4;
unsigned long long* g_view_pinned = nullptr;   // [kViewEntries][kViewWords] {tag << 32 | pairs, tag << 32 | slots, tag << 32 | non-empty sub-tiles, -}
unsigned long long g_view_clock = 0;
uint32_t g_view_tag = 0;
// registers the launch sequence of the forward that owns `image_blob`: returns where order_desc_kernel writes its totals and the tag
unsigned long long* view_note(const void* image_blob, int cap_R, long long cap_slots, uint32_t* tag, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_view_mu);
    if (!g_view_pinned) {
        void* ptr = nullptr;
        if (hipHostMalloc(&ptr, kViewEntries * kViewWords * sizeof(unsigned long long), hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        g_view_pinned = (unsigned long long*)ptr;
        for (int i = 0; i < kViewWords * kViewEntries; i++) g_view_pinned[i] = 0ull;
    }
    int slot = 0;
    for (int i = 0; i < kViewEntries; i++) {
        if (g_view[i].key == image_blob) { slot = i; break; }
        if (g_view[i].stamp < g_view[slot].stamp) slot = i;
    }
    ViewEntry& e = g_view[slot];
    e.key = image_blob; e.stamp = ++g_view_clock; e.cap_R = cap_R; e.cap_slots = cap_slots; e.recorded = false;
    e.stream = stream; e.has_stream = true;
    e.tag = ++g_view_tag ? g_view_tag : ++g_view_tag;   // (never 0: the slots start as 0)
    *tag = e.tag;
    return g_view_pinned + kViewWords * slot;
}
// The same four numbers live in the image blob itself (ImageLayout::counters, written by order_desc_kernel): a blob the host table no
// longer knows -- more than kViewEntries forwards ago, or a binder that moved / cloned the saved buffer -- is still self-describing, like
// the reference's blobs; the table is the fast path (no device read, no synchronisation).
constexpr uint32_t kBlobMagic = 0x53564931u;   // "SVI1" in counters[3]: order_desc_kernel of this library version wrote the words behind it
struct ViewCounts { int cap_R = 0; long long cap_slots = -1, pairs = -1, slots = -1; };
// blocking read of the blob's own copy; the forward that wrote it must be complete on the device (the callers synchronise first)
bool view_from_blob(const uint32_t* counters_dev, ViewCounts* out) {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!counters_dev || hipMemcpy(w, counters_dev, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (w[3] != kBlobMagic) return false;
    out->pairs = (long long)w[1]; out->slots = (long long)w[2]; out->cap_R = (int)w[4];
    out->cap_slots = (long long)((unsigned long long)w[5] | ((unsigned long long)w[6] << 32));
    return true;
}
// the entry of the forward that owns `image_blob` (capacities; counts when `wait`): false = unknown blob.  The counts are in host memory
// as soon as the forward's order kernel has run.  wait = 1: poll for spin_budget_s(), then BLOCK -- on `*sync_stream` when the caller has
// the stream the forward ran on (or one ordered behind it), else on the stream the forward itself was launched on (remembered in the
// entry; the whole device only if that stream no longer exists) -- and look again: a backlog in front of the forward is not an error,
// and backlogs on OTHER streams (tracer updates of another view, a second model) are not waited for.  pairs / slots stay -1 only if the forward never wrote them (it failed on the device) or the entry was recycled.
// wait = 2: one look.
bool view_lookup(const void* image_blob, int wait, int* cap_R, long long* cap_slots, long long* pairs, long long* slots,
                 const hipStream_t* sync_stream = nullptr, long long* nonempty = nullptr) {
    volatile unsigned long long* at = nullptr;
    uint32_t tag = 0;
    hipStream_t fwd_stream = nullptr; bool have_fwd_stream = false;
    {
        std::lock_guard<std::mutex> lk(g_view_mu);
        for (int i = 0; i < kViewEntries; i++)
            if (g_view[i].key == image_blob && g_view_pinned) {
                at = g_view_pinned + kViewWords * i; tag = g_view[i].tag;
                fwd_stream = g_view[i].stream; have_fwd_stream = g_view[i].has_stream;
                if (cap_R) *cap_R = g_view[i].cap_R;
                if (cap_slots) *cap_slots = g_view[i].cap_slots;
                break;
            }
    }
    if (pairs) *pairs = -1;
    if (slots) *slots = -1;
    if (nonempty) *nonempty = -1;
    if (!at) return false;
    if (!wait) return true;
    uint32_t w0 = 0, w1 = 0;
    bool got = wait == 2 ? tagged_pair(at, tag, &w0, &w1) : pinned_spin(at, tag, &w0, &w1);
    if (!got && wait == 1) {
        hipError_t e = hipErrorInvalidHandle;
        if (sync_stream) e = hipStreamSynchronize(*sync_stream);
        else if (have_fwd_stream) {   // (the handle may be stale: a query tells; the null stream is always valid)
            const hipError_t q = fwd_stream ? hipStreamQuery(fwd_stream) : hipSuccess;
            if (q == hipSuccess || q == hipErrorNotReady) e = hipStreamSynchronize(fwd_stream);
            else (void)hipGetLastError();
        }
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipDeviceSynchronize(); }
        if (e != hipSuccess) (void)hipGetLastError();
        got = tagged_pair(at, tag, &w0, &w1);
    }
    if (got) {
        if (pairs) *pairs = (long long)w0;
        if (slots) *slots = (long long)w1;
        const unsigned long long v2 = at[2];
        if (nonempty && (uint32_t)(v2 >> 32) == tag) *nonempty = (long long)(uint32_t)v2;
    }
    return true;
}
// A view's state-slot total enters its workload's history once, through whoever sees it first: the workload's next forward (mode 2:
// a look, no wait -- the forward itself never waits for the cull) or the view's own backward (mode 1: the value is there by then).
void note_view_slots(const CapKey& k, const void* image_blob, int mode, const hipStream_t* sync_stream = nullptr) {
    long long slots = -1, nonempty = -1;
    if (!image_blob || !view_lookup(image_blob, mode, nullptr, nullptr, nullptr, &slots, sync_stream, &nonempty) || slots < 0) return;
    {
        std::lock_guard<std::mutex> lk(g_view_mu);
        bool found = false;
        for (int i = 0; i < kViewEntries; i++)
            if (g_view[i].key == image_blob) {
                if (g_view[i].recorded) return;
                g_view[i].recorded = true; found = true;
                break;
            }
        if (!found) return;
    }
    record_slots(k, slots, nonempty);
}
// Side stream of the backward: the gradient tensors are cleared there while the composite backward (which only writes the
// scratch) runs on the caller's stream.  One per device, created on first use; fork / join through events.
struct SideStream { hipStream_t s = nullptr; bool ok = false; };
std::mutex g_side_mu;
SideStream g_side[16];
hipStream_t side_stream(hipStream_t of) {
    // the device that owns the caller's stream (the current device need not be it); the null stream belongs to the current one
    int dev = 0;
    hipDevice_t sd;
    if (of != nullptr && hipStreamGetDevice(of, &sd) == hipSuccess) dev = (int)sd;
    else if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream& ss = g_side[dev];
    if (!ss.ok) {
        // lowest priority: the clear only has to be done by the time the composite backward ends
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
        ss.ok = true;
    }
    return ss.s;
}
std::mutex g_times_mu;
struct Pending { hipEvent_t a, b; const char* name; };
std::vector<Pending> g_pending;
struct Accum { const char* name; double sum_ms; int count; };
std::vector<Accum> g_accum;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(SVGIR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Resolve queued event pairs into per-stage sums.  Each event appears as `b` of one pair and possibly `a` of the
// next; destroy an event after its last use.
void resolve_pending() {
    std::vector<Pending> todo;
    {
        std::lock_guard<std::mutex> lk(g_times_mu);
        todo.swap(g_pending);
    }
    std::vector<hipEvent_t> seen;
    for (auto& p : todo) {
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_times_mu);
            bool found = false;
            for (auto& acc : g_accum)
                if (acc.name == p.name) { acc.sum_ms += ms; acc.count++; found = true; break; }
            if (!found) g_accum.push_back({p.name, (double)ms, 1});
        }
        for (hipEvent_t e : {p.a, p.b}) {
            bool dup = false;
            for (auto x : seen) if (x == e) dup = true;
            if (!dup) seen.push_back(e);
        }
    }
    for (auto e : seen) (void)hipEventDestroy(e);
}

}  // namespace

namespace svgir {
StageMarks stage_begin(hipStream_t s) {
    StageMarks t{s, g_prof.load(), nullptr};
    if (t.on) {
        if (hipEventCreate(&t.prev) != hipSuccess) t.on = false;
        else (void)hipEventRecord(t.prev, s);
    }
    return t;
}
void stage_mark(StageMarks& t, const char* name) {
    if (!t.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { t.on = false; return; }
    (void)hipEventRecord(e, t.s);
    {
        std::lock_guard<std::mutex> lk(g_times_mu);
        g_pending.push_back({t.prev, e, name});
    }
    t.prev = e;
}
}  // namespace svgir

namespace {

// Stage timer (stage_begin / stage_mark above): one event per stage boundary; consecutive pairs are queued for
// lazy resolution.
struct StageTimer {
    StageMarks t;
    explicit StageTimer(hipStream_t s) : t(stage_begin(s)) {}
    void mark(const char* name) { stage_mark(t, name); }
};


CfgRef cfg_ref(const svgir_params* p) {
    CfgRef c;
    if (p->variant == SVGIR_SVGSS) { c.ptr = p->config; c.len = p->config ? p->config_len : 0; }
    else { c.ptr = nullptr; c.len = -1; }
    return c;
}

// svgir_grads.out_weights given: below this many Gaussians the two launches of the partition cost more than the per-Gaussian kernels save
// by walking the blended Gaussians only (the fused shading needs the partition anyway)
// (measured: cfg3_train, P = 200 k, svgss rows: grad_reduce 85 -> 64 us, geom_bwd 28 -> 24 us against ~10 us for the partition; cfg2,
// P = 200 k, rgss packed rows: only geom_bwd gains, 31 -> ~25 us: not worth it; cfg5, P = 2 M: 498 -> 345 us and 171 -> 88 us)
constexpr int kListMinP = 400000, kListMinPRows = 50000;   // (round 5, rgss at P = 200 k with the count folded into seg_build: geom_bwd -7 us, scatter launch +5, seg_build +2: no gain)

// Does a composite launch with `nonempty` waves of work fill the machine several times over?  (256 CUs x 8-11 resident waves: from ~4
// rounds on; SVGIR_FWD_FILL = 0 / 1 forces the low- / high-occupancy variant)
bool high_fill(long long nonempty) {
    static const int forced = [] { const char* e = getenv("SVGIR_FWD_FILL"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    return forced >= 0 ? forced != 0 : nonempty >= 8192;
}

// fused shading: run the contribution pre-pass?  (SVGIR_PREPASS = 0 / 1 forces it off / on)
bool shade_prepass(int Ns) {
    static const int forced = [] { const char* e = getenv("SVGIR_PREPASS"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    return forced >= 0 ? forced != 0 : Ns >= 128;
}

int validate(const svgir_params* p, bool fwd) {
    if (!p) return fail(SVGIR_ERR_INVALID, "params is NULL");
    if (p->variant != SVGIR_RGSS && p->variant != SVGIR_SVGSS) return fail(SVGIR_ERR_INVALID, "unknown variant %d", p->variant);
    if (p->P < 0 || p->W <= 0 || p->H <= 0) return fail(SVGIR_ERR_INVALID, "bad sizes P=%d W=%d H=%d", p->P, p->W, p->H);
    if (p->P > 40000000) return fail(SVGIR_ERR_INVALID, "P=%d exceeds the supported 40 000 000 Gaussians (32-bit byte offsets into the splat records)", p->P);
    {   // packing limits of the state blobs: tile rectangle x0 | y0 << 10 | width << 20 (common.hpp R_RECT) and
        // (sub-tile id << SEG_K_BITS) | segment (seg_list)
        const long long gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
        if (gx > 1023 || gy > 1023 || 4 * gx * gy >= (1ll << (32 - SEG_K_BITS)))
            return fail(SVGIR_ERR_INVALID, "image %dx%d exceeds the supported size (at most 1023 tiles per side, %lld tiles in total)",
                        p->W, p->H, (1ll << (32 - SEG_K_BITS)) / 4 - 1);
    }
    if (p->P == 0) return 0;
    if (!p->means3D || !p->viewmatrix || !p->projmatrix || !p->background)
        return fail(SVGIR_ERR_INVALID, "means3D/viewmatrix/projmatrix/background must be provided");
    if (fwd && !p->opacities) return fail(SVGIR_ERR_INVALID, "opacities must be provided");
    if (!p->colors_precomp && !p->shs)
        return fail(SVGIR_ERR_INVALID, "For non-RGB, provide precomputed Gaussian colors!");  // rasterizer_impl.cu:264-267
    if (p->shs && (p->D < 0 || p->D > 3 || p->M < (p->D + 1) * (p->D + 1)))
        return fail(SVGIR_ERR_INVALID, "SH degree %d needs M >= %d coefficients (M=%d)", p->D, (p->D + 1) * (p->D + 1), p->M);
    if (p->shs && !p->cam_pos) return fail(SVGIR_ERR_INVALID, "cam_pos is required with SHs");
    if (!p->cov3D_precomp && !(p->scales && p->rotations))
        return fail(SVGIR_ERR_INVALID, "provide scales+rotations or cov3D_precomp");
    if (p->S < 0 || (p->S > 0 && !p->features)) return fail(SVGIR_ERR_INVALID, "features missing for S=%d", p->S);
    if (p->variant == SVGIR_SVGSS) {
        if (p->VS < 0 || p->VS % 4 != 0) return fail(SVGIR_ERR_INVALID, "VS=%d must be a non-negative multiple of 4", p->VS);
        if (p->VS > 0 && !p->vfeatures) return fail(SVGIR_ERR_INVALID, "vfeatures missing for VS=%d", p->VS);
        if (!p->patchbbox) return fail(SVGIR_ERR_INVALID, "patchbbox is required for svgss");
        if (p->S > 50 || p->VS / 4 > 20) return fail(SVGIR_ERR_INVALID, "svgss supports S<=50, VS/4<=20 (Q9)");
    } else {
        if (p->VS != 0) return fail(SVGIR_ERR_INVALID, "rgss has no vfeatures");
        if (p->S > 33) return fail(SVGIR_ERR_INVALID, "rgss supports S<=33 (Q9)");
    }
    if (p->shade) {   // fused shading: the packed rows are produced inside the call
        const svgir_shade_params& sp = p->shade->sp;
        if (p->variant != SVGIR_SVGSS) return fail(SVGIR_ERR_INVALID, "fused shading belongs to the svgss variant");
        if (sp.P != p->P) return fail(SVGIR_ERR_INVALID, "fused shading: sp.P = %d but P = %d", sp.P, p->P);
        if (p->S != (sp.training ? 4 : 7) || p->VS != (sp.training ? 52 : 64))
            return fail(SVGIR_ERR_INVALID, "fused shading packs S = %d, VS = %d (training = %d), not S = %d, VS = %d", sp.training ? 4 : 7,
                        sp.training ? 52 : 64, sp.training, p->S, p->VS);
        if (!sp.viewmatrix) return fail(SVGIR_ERR_INVALID, "fused shading needs sp.viewmatrix (packed view-space normals)");
    }
    return 0;
}

}  // namespace

extern "C" {

int svgir_abi_version(void) { return SVGIR_ABI_VERSION; }
size_t svgir_geom_bytes(int32_t P) { return geom_layout(nullptr, P).bytes; }
size_t svgir_image_bytes(int32_t W, int32_t H) { return image_layout(nullptr, W, H).bytes; }
size_t svgir_binning_bytes(int32_t R, int32_t W, int32_t H, int32_t S, int32_t VS) {
    const int T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    return bin_layout(nullptr, binning_capacity(R), T, seg_nstate(S, VS)).bytes;
}
size_t svgir_image_ncontrib_offset(int32_t W, int32_t H) { return image_layout(nullptr, W, H).ncontrib_off; }
size_t svgir_image_ranges_offset(int32_t W, int32_t H) {
    const ImageLayout I = image_layout((char*)256, W, H);   // (non-null dummy base: the layout returns pointers)
    return (size_t)((char*)I.ranges - (char*)256);
}
size_t svgir_binning_point_list_offset(size_t binning_bytes, const char* image_blob, int32_t W, int32_t H, int32_t S, int32_t VS) {
    const int T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const int nstate = seg_nstate(S, VS);
    int cap = 0;
    if (bin_bytes_compact(binning_bytes)) {   // laid out for a state-slot capacity of the forward's choosing: the view's entry knows
        if (!image_blob) return (size_t)-1;
        if (!view_lookup(image_blob, false, &cap, nullptr, nullptr, nullptr)) {   // not in the host table: the blob's own copy
            ViewCounts vc;
            if (hipDeviceSynchronize() != hipSuccess || !view_from_blob(image_layout(const_cast<char*>(image_blob), W, H).counters, &vc)) return (size_t)-1;
            cap = vc.cap_R;
        }
    } else {
        cap = binning_capacity_from_bytes(binning_bytes, T, nstate);
    }
    const BinLayout B = bin_layout((char*)256, cap, T, nstate);   // (the instance arrays come first: independent of the slot capacity)
    return (size_t)((char*)B.val[tile_sort_plan(T).passes & 1] - (char*)256);
}
const char* svgir_last_error(void) { return g_err.c_str(); }
void svgir_set_profiling(int enabled) {
    resolve_pending();
    if (enabled) {
        std::lock_guard<std::mutex> lk(g_times_mu);
        g_accum.clear();
    }
    g_prof.store(enabled != 0);
}
int svgir_last_timings(const char** names, float* avg_ms, int* counts, int cap) {
    resolve_pending();
    std::lock_guard<std::mutex> lk(g_times_mu);
    int n = 0;
    for (auto& acc : g_accum) {
        if (n >= cap) break;
        names[n] = acc.name;
        avg_ms[n] = (float)(acc.sum_ms / (acc.count > 0 ? acc.count : 1));
        if (counts) counts[n] = acc.count;
        n++;
    }
    return n;
}

// One svgir_forward, in two halves: begin() validates, allocates and launches EVERYTHING -- the count-dependent stages speculatively, for
// capacities guessed from the workload's recent views -- without waiting for the GPU; finish() waits for the instance count (it only
// confirms the guess, or re-runs the dependent stages) and returns it.  svgir_forward is begin() + finish(); svgir_forward_batch begins all
// its views before it finishes the first, so that ONE host thread keeps several views in flight (on as many streams).
struct ForwardCall {
    const svgir_params* p; const svgir_outputs* o;
    svgir_alloc_fn geom, binning, image; void *geom_ctx, *binning_ctx, *image_ctx;
    hipStream_t s; bool key_spec;
    // set by begin()
    int P = 0, W = 0, H = 0, gx = 0, gy = 0, T = 0, nstate = 0, fin = 0, spec_top = -1, cap = 0;
    size_t N = 0;
    bool svgss = false, shade_subset = false, prepass = false, done = false;
    CfgRef cfg{}; float focal_x = 0.f, focal_y = 0.f;
    char *gblob = nullptr, *iblob = nullptr, *bblob = nullptr;
    GeomLayout G{}; ImageLayout I{}; TileSortPlan plan{};
    CapKey ckey{}; const uint32_t* depth_order = nullptr; long long cap_slots = -1;
    hipEvent_t features_ready = nullptr;
    PinnedSlot R_pin; unsigned long long* R_slot = nullptr; uint32_t R_tag = 0;
    const uint32_t* prefilter_violation = nullptr;
    StageTimer tm;

    ForwardCall(const svgir_params* p_, const svgir_outputs* o_, svgir_alloc_fn geom_, void* geom_ctx_, svgir_alloc_fn binning_,
                void* binning_ctx_, svgir_alloc_fn image_, void* image_ctx_, void* stream, bool key_spec_)
        : p(p_), o(o_), geom(geom_), binning(binning_), image(image_), geom_ctx(geom_ctx_), binning_ctx(binning_ctx_), image_ctx(image_ctx_),
          s((hipStream_t)stream), key_spec(key_spec_), tm((hipStream_t)stream) {}

    int check(const char* what) {
        if (!p->debug) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(SVGIR_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
            return 0;
        }
        hipError_t e = hipStreamSynchronize(s);  // reference CHECK_CUDA(debug), auxiliary.h:425-432
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return fail(SVGIR_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return 0;
    }

    // Everything behind the offsets scan depends on the instance count R that the GPU is still computing.  The stages are launched for an
    // instance CAPACITY `cap` and read R on the device (min(cap, R)); the binning blob is laid out for `cap`.  `timed`: stage marks are
    // only recorded for the launch sequence that counts.
    int run_binning_and_render(char* bblob, int cap, long long cap_slots, bool timed, bool cull_only = false) {
        const BinLayout B = bin_layout(bblob, cap, T, nstate, cap_slots);
        launch_emit(P, depth_order, G.tiles, G.offsets, G.rec, o->radii, gx, gy, B.key[0], B.val[0], cap, I.ranges,
                    I.counters, B.radix_tbl, G.counters + 3, s);
        if (int rc = check("emit")) return rc;
        if (timed) tm.mark("emit");
        if (plan.single) {   // up to 4096 tiles: one counting pass over the whole tile id, which also yields the tile ranges
            launch_tile_sort12(B.key, B.val, cap, G.counters, B.radix_tbl, I.ranges, T, s);
            if (int rc = check("tile sort")) return rc;
            if (timed) tm.mark("sort_tile");
        } else {
            launch_radix_sort(B.key, B.val, cap, G.counters, plan.bits, plan.bits_per_pass, B.radix_tbl, s);
            if (int rc = check("tile sort")) return rc;
            if (timed) tm.mark("sort_tile");
            launch_ranges(cap, G.counters, B.key[fin], I.ranges, T, s);
            if (int rc = check("ranges")) return rc;
            if (timed) tm.mark("ranges");
        }

        RenderArgs ra;
        ra.W = W; ra.H = H; ra.gx = gx; ra.gy = gy; ra.S = p->S; ra.VS = svgss ? p->VS : 0;
        ra.ranges = I.ranges; ra.point_list = B.val[fin]; ra.rec = G.rec; ra.features = p->features; ra.vfeatures = p->vfeatures;
        ra.bg = p->background;
        ra.cfg = cfg; ra.sub_list = B.sub_list; ra.sub_total = I.sub_total; ra.sub_order = I.sub_order;
        ra.sub_pair_base = I.sub_pair_base; ra.sub_slot_base = I.sub_slot_base; ra.slot_cap = (uint32_t)std::min<size_t>(B.slot_cap, 0xffffffffu);
        ra.dump_only = 0;
        ra.hi_fill = high_fill(guess_fill(ckey)) ? 1 : 0;
        ra.order_n = (int)order_entries(gx, gy);
        ra.bg_in_render = render_specialised(p->S, svgss ? p->VS : 0, svgss) ? 1 : 0;
        ra.sub_count = I.sub_count;
        ra.sub_ndump = I.sub_ndump; ra.seg_list = B.seg_list; ra.seg_desc = B.seg_desc; ra.seg_count = I.counters; ra.seg_block = I.seg_block; ra.seg_state = B.seg_state;
        ra.final_T = I.final_T; ra.final_D = I.final_D; ra.n_contrib = I.n_contrib;
        ra.out_color = o->out_color; ra.out_normal = o->out_normal; ra.out_depth = o->out_depth; ra.out_opacity = o->out_opacity;
        ra.out_feature = o->out_feature; ra.out_vfeature = o->out_vfeature; ra.out_weights = o->out_weights;
        // rgss without computer_pseudo_normal: the two stencil outputs are all zero (rasterize_points.cu:85-86)
        const bool clear_stencil = !svgss && !p->computer_pseudo_normal;
        ra.zero_a = clear_stencil ? o->out_pseudo_normal : nullptr;
        ra.zero_b = clear_stencil ? o->out_surface_xyz : nullptr;
        ra.needed = nullptr;
        launch_cull(ra, s);
        // dispatch order of the sub-tiles, first gradient row / first state slot of each, and the two totals (device + tagged host copy)
        uint32_t vtag = 0;
        unsigned long long* vslot = view_note(iblob, cap, cap_slots, &vtag, s);
        const bool row_path = svgss && p->VS > 0 && render_specialised(p->S, p->VS, true);   // (only the svgss backward writes gradient rows)
        // (a launch of many rounds of waves gets one longest-first list per XCD instead of one global list: common.hpp ORDER_NONE)
        static const int xcd_forced = [] { const char* e = getenv("SVGIR_FWD_XCD"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
        const bool per_xcd = ra.bg_in_render && (xcd_forced >= 0 ? xcd_forced != 0 : ra.hi_fill != 0);   // (specialised composite kernels only)
        launch_order_desc(I.sub_total, 4 * T, I.sub_order, row_path ? I.sub_pair_base : nullptr, I.sub_slot_base, I.counters, vslot, vtag,
                          (uint32_t)cap, cap_slots, kBlobMagic, gx, ra.order_n, per_xcd, s);
        if (int rc = check("cull")) return rc;
        if (timed) tm.mark("cull");
        if (cull_only) return 0;   // (the sizing phase of a workload's first view: see finish())
        if (prepass) {
            RenderArgs rp = ra;
            rp.S = 0; rp.VS = 0; rp.features = nullptr; rp.vfeatures = nullptr; rp.needed = G.needed;
            launch_contrib_prepass(rp, s);
            if (int rc = check("prepass")) return rc;
            if (timed) tm.mark("prepass");
        }
        if (p->shade) {
            // The per-splat shading of this view, for the surfels its composite is about to read (the reference shades all P before
            // it knows the view, svgss.py:116-141).  Rows of the others: zero.
            svgir_shade_params sp = p->shade->sp;
            sp.subset = nullptr; sp.subset_count = nullptr;
            if (prepass) {
                uint32_t* cnt = G.shade_work + partition_work_words(P) - 1;
                launch_partition(P, G.needed, nullptr, G.shade_list, G.shade_work, cnt, s);
                sp.subset = G.shade_list; sp.subset_count = cnt;
            } else if (shade_subset) {   // (a permutation of 0..P-1 whose first counters[3] entries hold every surfel that touches a tile)
                sp.subset = depth_order; sp.subset_count = G.counters + 3;
            }
            if (shade_forward_impl(&sp, p->shade->reduced, const_cast<float*>(p->features), const_cast<float*>(p->vfeatures), true, s) != 0)
                return fail(SVGIR_ERR_INVALID, "fused shading: svgir_shade_forward rejected its parameters");
            if (int rc = check("shade")) return rc;
            if (timed) tm.mark("shade");
        }
        if (features_ready && hipStreamWaitEvent(s, features_ready, 0) != hipSuccess) return fail(SVGIR_ERR_HIP, "waiting for the features event");
        if (launch_render_fwd(ra, svgss, s) < 0) launch_render_fwd_generic(ra, svgss, s);   // run-time-width kernels
        if (int rc = check("render")) return rc;
        if (timed) tm.mark("render");
        // (the list of live backward segments is built by svgir_backward, next to its clears: a forward-only call never pays for it)
        return 0;
    }

    // returns a negative status, or 0 (launched; finish() must follow), or 1 (nothing to finish: P == 0)
    int begin() {
        if (int rc = validate(p, true)) return rc;
        // features / vfeatures may still be in production on another stream (the shading kernels do not depend on the binning and
        // the binning does not read them): only the composite kernel waits for the caller's event
        features_ready = (hipEvent_t)p->features_ready;
        if (!o || !geom || !binning || !image) return fail(SVGIR_ERR_INVALID, "outputs / allocators must be provided");
        P = p->P; W = p->W; H = p->H;
        N = (size_t)W * H;
        gx = (W + TILE - 1) / TILE; gy = (H + TILE - 1) / TILE; T = gx * gy;
        svgss = p->variant == SVGIR_SVGSS;
        cfg = cfg_ref(p);
        if (P == 0) {  // rasterize_points.cu:100: nothing runs, outputs stay zero
            HIP_OK(hipMemsetAsync(o->out_color, 0, 3 * N * 4, s));
            HIP_OK(hipMemsetAsync(o->out_normal, 0, 3 * N * 4, s));
            HIP_OK(hipMemsetAsync(o->out_depth, 0, N * 4, s));
            HIP_OK(hipMemsetAsync(o->out_opacity, 0, N * 4, s));
            if (p->S) HIP_OK(hipMemsetAsync(o->out_feature, 0, (size_t)p->S * N * 4, s));
            if (svgss && p->VS) HIP_OK(hipMemsetAsync(o->out_vfeature, 0, (size_t)(p->VS / 4) * N * 4, s));
            if (!svgss && o->out_pseudo_normal) HIP_OK(hipMemsetAsync(o->out_pseudo_normal, 0, 3 * N * 4, s));
            if (!svgss && o->out_surface_xyz) HIP_OK(hipMemsetAsync(o->out_surface_xyz, 0, 3 * N * 4, s));
            done = true;
            return 1;
        }
        focal_y = H / (2.0f * p->tan_fovy); focal_x = W / (2.0f * p->tan_fovx);
        gblob = geom(geom_layout(nullptr, P).bytes, geom_ctx);
        iblob = image(image_layout(nullptr, W, H).bytes, image_ctx);
        if (!gblob || !iblob) return fail(SVGIR_ERR_ALLOC, "geometry/image blob allocation failed");
        G = geom_layout(gblob, P);
        I = image_layout(iblob, W, H);

        PreArgs pa;
        pa.P = P; pa.D = p->D; pa.M = p->M; pa.W = W; pa.H = H; pa.gx = gx; pa.gy = gy;
        pa.means3D = p->means3D; pa.shs = p->colors_precomp ? nullptr : p->shs; pa.colors_precomp = p->colors_precomp;
        pa.opacities = p->opacities; pa.scales = p->scales; pa.rotations = p->rotations; pa.cov3D_precomp = p->cov3D_precomp;
        pa.view = p->viewmatrix; pa.proj = p->projmatrix; pa.campos = p->cam_pos; pa.patchbbox = p->patchbbox;
        pa.scale_modifier = p->scale_modifier; pa.tanx = p->tan_fovx; pa.tany = p->tan_fovy;
        pa.focal_x = focal_x; pa.focal_y = focal_y; pa.cfg = cfg;
        pa.rec = G.rec; pa.cov3D = G.cov3D; pa.clamped = G.clamped; pa.tiles = G.tiles; pa.key = G.key[0]; pa.idx = G.idx[0];
        pa.radii = o->radii;
        pa.out_weights = o->out_weights;
        pa.features = nullptr; pa.embed_S = 0;
        if (rec_embeds_features(p->S, svgss ? p->VS : 0)) {
            // (the preprocess reads the rows: an event that says when they are complete is waited for here, not in front of the composite)
            if (features_ready && hipStreamWaitEvent(s, features_ready, 0) != hipSuccess) return fail(SVGIR_ERR_HIP, "waiting for the features event");
            features_ready = nullptr; pa.features = p->features; pa.embed_S = p->S; }   // feature rows ride in the records
        shade_subset = p->shade && !p->shade->all_surfels;   // shade the view's working set only (subset.hip)
        // With many incident samples per surfel (evaluation: 384) shading a surfel costs far more than compositing it, and a geometry-only
        // pass of the composite (the alpha / transmittance chain of the very same arithmetic: no channels, no outputs) first finds the surfels
        // that actually receive a blend weight -- 29 % at cfg3, 13 % at cfg5 -- for ~40 % of the full composite's time.  Otherwise the
        // working set is every surfel that touches a tile (44 % on the BASELINE scenes: the preprocess culls), which costs nothing to find:
        // the depth order holds them in front, and the offsets scan reports where they end.
        prepass = shade_subset && shade_prepass(p->shade->sp.Ns);
        pa.needed = prepass ? G.needed : nullptr;
        pa.span = G.counters + 3;
        if (p->shade) pa.tabs = shade_tables(&p->shade->sp, nullptr, 0);   // (the preprocess launch carries the shading kernels' tables)
        pa.zero_words = radix_gtot(G.radix_tbl, P); pa.n_zero_words = (int)radix_gtot_words(P);
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        ckey = CapKey{dev_id, W, H, P, p->S, svgss ? p->VS : 0, p->variant, p->workload_scope};
        static const bool key_spec_env = getenv("SVGIR_NO_KEY_SPEC") == nullptr;
        spec_top = (key_spec && key_spec_env) ? guess_top(ckey) : -1;
        pa.spec_top = spec_top; pa.key_top = G.key_top;
        pa.prefilter_violation = nullptr;
        if (p->prefiltered) {   // the violation flag sits next to the instance counter and is read back with it
            HIP_OK(hipMemsetAsync(G.counters, 0, 16, s));
            pa.prefilter_violation = G.counters + 1;
        }
        launch_preprocess(pa, svgss, s);
        if (int rc = check("preprocess")) return rc;
        tm.mark("preprocess");

        // depth sort of the P Gaussians: 4 x 8-bit stable passes (ends in slot 0), or 3 when the top byte is speculated to be common (slot 1)
        const int depth_bits = spec_top >= 0 ? 24 : 32;
        depth_order = G.idx[(depth_bits / 8) & 1];
        launch_radix_sort(G.key, G.idx, P, nullptr, depth_bits, 8, G.radix_tbl, s);
        if (int rc = check("depth sort")) return rc;
        tm.mark("sort_depth");

        R_slot = R_pin.at; R_tag = R_pin.tag;   // (the slot is released when this call object dies)
        launch_offsets_scan(G.tiles, depth_order, G.offsets, G.scan_tmp, P, G.counters, G.key_top, (P + 63) / 64, pa.prefilter_violation,
                            R_slot, R_tag, s);
        if (int rc = check("offsets scan")) return rc;
        tm.mark("scan");

        prefilter_violation = pa.prefilter_violation;
        nstate = seg_nstate(p->S, svgss ? p->VS : 0);
        plan = tile_sort_plan(T);
        fin = plan.passes & 1;

        // Speculative launch: capacities from this workload's recent views (+12.5 %) -- instances (binning arrays) and state slots
        // (seg_state) -- no host round trip in between.  The first view of a workload gets the exact instance capacity and the worst-case
        // slot count (4 full lists per tile); later ones typically a third of that.
        cap = 0; cap_slots = -1; bblob = nullptr;
        {   // the previous view of this workload: its slot total, if the backward has not recorded it already (forward-only loops)
            const void* prev = nullptr;
            CapKey pkey = ckey;
            { std::lock_guard<std::mutex> lk(g_cap_mu); if (const CapEntry* e = cap_entry(ckey, false)) { prev = e->last_view; pkey.P = e->last_view_P; } }
            note_view_slots(pkey, prev, 2);
        }
        if (const int guess = guess_R(ckey)) {
            cap = binning_capacity((long long)guess + guess / 8 + 1024);
            const long long gs = guess_slots(ckey);
            cap_slots = p->forward_only ? 0 : (gs < 0 ? -1 : std::min<long long>(gs + gs / 8 + 64, (long long)seg_capacity(cap, T)));
            bblob = binning(bin_layout(nullptr, cap, T, nstate, cap_slots).bytes, binning_ctx);
            // (a failed speculative allocation is not an error: the guess may be far larger than this view needs; fall
            // through to the exact-size path below)
            if (bblob) {
                if (int rc = run_binning_and_render(bblob, cap, cap_slots, true)) return rc;
            } else {
                cap = 0;
            }
        }
        return 0;
    }

    // waits for the instance count, confirms (or repairs) the speculation; returns R or a negative status
    int finish() {
        if (done) return 0;
        done = true;
        // the instance count (only: the speculative stages keep running)
        uint32_t R_host = 0, R_aux = 0;
        bool have_R = R_slot && pinned_spin(R_slot, R_tag, &R_host, &R_aux);
        if (!have_R) {
            // The count is further away than the spin budget -- a backlog in front of this forward on the stream (the reference's call order
            // puts update_visibility / update_radiace, seconds of work, right before a render), a shared GPU, a serialising profiler -- or no
            // landing slot was free.  Like the reference (rasterizer_impl.cu:307-312: a cudaMemcpy without a deadline): block, then look again;
            // a slow stream is not an error.
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) return fail(SVGIR_ERR_HIP, "the forward failed on the device: %s", hipGetErrorString(e));
            have_R = R_slot && tagged_pair(R_slot, R_tag, &R_host, &R_aux);
            if (!have_R) {   // the counters' device copy (same three words)
                uint32_t w[3] = {0, 0, 0};
                HIP_OK(hipMemcpy(w, G.counters, 12, hipMemcpyDeviceToHost));
                R_host = w[0]; R_aux = ((p->prefiltered && w[1]) ? 1u << 16 : 0u) | (w[2] & 0xffffu);
            }
        }
        if (p->prefiltered && (R_aux >> 16) != 0u) {
            (void)hipStreamSynchronize(s);
            return fail(SVGIR_ERR_INVALID, "Point is filtered although prefiltered is set. This shouldn't happen!");   // auxiliary.h:163-167
        }
        if (R_host > 0x7ffff000u) return fail(SVGIR_ERR_INVALID, "instance count %u overflows int32", R_host);
        const int R = (int)R_host;
        {   // the visible depth keys' top bytes: history for the next view; and did this view's speculation hold?
            const uint32_t summary = R_aux & 0xffffu;
            if (key_spec) { record_top(ckey, summary); g_spec_stats[0]++; }
            if (spec_top >= 0) g_spec_stats[4]++;
            const int av = (int)((summary >> 8) & 0xffu), ov = (int)(summary & 0xffu);
            if (spec_top >= 0 && !(av == 0xff && ov == 0) && (av != spec_top || ov != spec_top)) {
                // a visible key outside the speculated byte: the three-pass order is wrong -- run the whole view again, four passes
                HIP_OK(hipStreamSynchronize(s));
                g_spec_stats[3]++;
                ForwardCall again(p, o, geom, geom_ctx, binning, binning_ctx, image, image_ctx, (void*)s, false);
                if (int rc = again.begin()) return rc < 0 ? rc : 0;
                return again.finish();
            }
        }
        record_R(ckey, R);
        // (whether the state-slot guess held is the backward's business -- svgir_backward re-dumps the states of a view that exceeded it; the
        // forward does not wait for the cull.  Measured on the host-bound training step, bench.py --workload train_step: 2.11 ms with
        // the wait and a re-run here, see HISTORY.md 4)
        if (!bblob || R > cap) {
            // first view, or the scene grew past a guess: (re)do the dependent stages -- exact instance capacity, worst-case state slots
            const bool redo = bblob != nullptr;
            if (redo) { HIP_OK(hipStreamSynchronize(s)); g_spec_stats[1]++; }
            cap = binning_capacity(R);
            cap_slots = p->forward_only ? 0 : -1;
            if (cap_slots < 0 && R > 0) {
                // No history to size the state slots from (a workload's first view, or one that outgrew its guess): instead of the worst
                // case over the cull -- four full lists per tile, 3-4x what a view needs, 35 GB at cfg5_dense -- the binning and the
                // cull run once into a stream-ordered temporary WITHOUT state slots, the view's own slot total comes back (the same tagged
                // store the backward reads), and the blob the caller keeps is laid out for exactly that.  ~0.2 ms, once per workload.
                void* tmp = nullptr;
                if (hipMallocAsync(&tmp, bin_layout(nullptr, cap, T, nstate, 0).bytes, s) == hipSuccess) {
                    const int rc = run_binning_and_render((char*)tmp, cap, 0, false, true);
                    long long slots = -1;
                    if (rc == 0) (void)view_lookup(iblob, 1, nullptr, nullptr, nullptr, &slots, &s);
                    (void)hipFreeAsync(tmp, s);
                    if (rc) return rc;
                    if (slots >= 0) cap_slots = std::min<long long>(slots, (long long)seg_capacity(cap, T));
                } else {
                    (void)hipGetLastError();   // (no temporary: the worst-case layout, as before)
                }
            }
            bblob = binning(bin_layout(nullptr, cap, T, nstate, cap_slots).bytes, binning_ctx);
            if (!bblob) return fail(SVGIR_ERR_ALLOC, "binning blob allocation failed");
            if (redo && o->out_weights) HIP_OK(hipMemsetAsync(o->out_weights, 0, (size_t)P * 4, s));   // accumulated by atomics
            if (int rc = run_binning_and_render(bblob, cap, cap_slots, !redo)) return rc;
        }
        {
            std::lock_guard<std::mutex> lk(g_cap_mu);
            CapEntry* e = cap_entry(ckey, true);
            e->last_view = iblob; e->last_view_P = P;
        }

        if (!svgss && p->computer_pseudo_normal) {
            launch_image_ops(W, H, p->viewmatrix, focal_x, focal_y, p->cx, p->cy, o->out_opacity, o->out_depth,
                             o->out_pseudo_normal, o->out_surface_xyz, s);
            if (int rc = check("image ops")) return rc;
            tm.mark("image");
        }
        return R;
    }
};

}  // extern "C"  (the call object is C++)

extern "C" {

int svgir_forward(const svgir_params* p, const svgir_outputs* o, svgir_alloc_fn geom, void* geom_ctx,
                  svgir_alloc_fn binning, void* binning_ctx, svgir_alloc_fn image, void* image_ctx, void* stream) {
    ForwardCall c(p, o, geom, geom_ctx, binning, binning_ctx, image, image_ctx, stream, true);
    const int rc = c.begin();
    if (rc != 0) return rc < 0 ? rc : 0;
    return c.finish();
}

// Several views in flight from ONE host thread: every view is begun (validated, allocated, all of its kernels launched on ITS stream)
// before the first is finished (its instance count awaited).  On distinct streams the views overlap on the GPU -- one view leaves the
// SIMDs under-occupied -- and the host never idles between them.
int svgir_forward_batch(svgir_view_call* views, int32_t count) {
    if (count < 0 || (count > 0 && !views)) return fail(SVGIR_ERR_INVALID, "views is NULL");
    std::vector<std::unique_ptr<ForwardCall>> calls;
    calls.reserve((size_t)count);
    int first_err = 0;
    for (int v = 0; v < count; v++) {
        svgir_view_call& c = views[v];
        calls.emplace_back(new ForwardCall(c.params, c.outputs, c.geom, c.geom_ctx, c.binning, c.binning_ctx, c.image, c.image_ctx, c.stream, true));
        const int rc = calls.back()->begin();
        c.num_rendered = rc < 0 ? rc : 0;
        if (rc < 0 && !first_err) first_err = rc;
    }
    const std::string begin_err = first_err ? g_err : std::string();
    for (int v = 0; v < count; v++) {
        if (views[v].num_rendered < 0) continue;
        const int rc = calls[(size_t)v]->finish();
        views[v].num_rendered = rc;
        if (rc < 0 && !first_err) first_err = rc;
    }
    if (!begin_err.empty()) g_err = begin_err;
    return first_err;
}
void svgir_reset_workload_history(int32_t scope) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    for (auto& e : g_cap)
        if (e.used && (scope < 0 || e.key.scope == scope)) e = CapEntry{};
}
void svgir_speculation_stats(int64_t* out5) {
    if (out5) for (int i = 0; i < 5; i++) out5[i] = (int64_t)g_spec_stats[i].load();
}

size_t svgir_backward_scratch_bytes(int32_t variant, int32_t P, size_t binning_bytes, int32_t W, int32_t H, int32_t S,
                                    int32_t VS) {
    return svgir_backward_scratch_bytes_for(variant, P, binning_bytes, nullptr, W, H, S, VS);
}

size_t svgir_backward_scratch_bytes_for(int32_t variant, int32_t P, size_t binning_bytes, const char* image_blob, int32_t W, int32_t H,
                                        int32_t S, int32_t VS) {
    if (!render_specialised(S, variant == SVGIR_SVGSS ? VS : 0, variant == SVGIR_SVGSS))
        return 256;   // run-time-width kernels accumulate straight into the dL_d* tensors: no scratch (a token size, never touched)
    if (variant != SVGIR_SVGSS || VS == 0)   // one packed gradient row per Gaussian
        return align_up((size_t)(P > 0 ? P : 1) * grad_row_geom(S, 0).RS * 4);
    const int T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    // the binning blob's instance capacity: from its size (worst-case state slots) or, for a blob laid out for a slot capacity of the
    // forward's choosing, from the view's entry; and one gradient row per (sub-tile, instance) pair that survived the cull of THIS
    // view when its count is known, else four per instance
    int cap = 0;
    long long pairs = -1;
    bool known = image_blob && view_lookup(image_blob, true, &cap, nullptr, &pairs, nullptr);
    if (image_blob && (!known || pairs < 0)) {   // not in the host table (or its counts never arrived): the blob's own copy, blocking
        ViewCounts vc;
        if (hipDeviceSynchronize() == hipSuccess && view_from_blob(image_layout(const_cast<char*>(image_blob), W, H).counters, &vc)) {
            known = true; cap = vc.cap_R; pairs = vc.pairs;
        } else {
            (void)hipGetLastError();
        }
    }
    if (!bin_bytes_compact(binning_bytes)) cap = binning_capacity_from_bytes(binning_bytes, T, seg_nstate(S, VS));
    else if (!known) cap = binning_capacity((long long)(binning_bytes / 48));   // (no view given: an upper bound -- every instance owns 48 B of the blob)
    const size_t rows = pairs >= 0 ? (size_t)std::min<long long>(pairs, (long long)4 * cap) : (size_t)4 * cap;
    return grad_scratch_bytes(cap, rows > 0 ? rows : 1, S, VS);
}

int svgir_backward(const svgir_params* p, const svgir_grads* g, int32_t R, const int32_t* radii, char* geom_blob,
                   char* binning_blob, size_t binning_bytes, char* image_blob, char* scratch, size_t scratch_bytes,
                   void* stream) {
    if (int rc = validate(p, false)) return rc;
    if (p->P == 0) return 0;
    if (!g || !radii || !geom_blob || !binning_blob || !image_blob)
        return fail(SVGIR_ERR_INVALID, "grads / radii / blobs must be provided");
    if (p->shade) {   // fused shading: everything its backward needs, checked BEFORE anything is launched (no half-written gradients on a bad call)
        if (!g->dL_dbase_color || !g->dL_droughness || !g->dL_dshade_normals || (!g->dL_dradiance && !p->shade->sp.radiance_ratio) || !g->dL_denv ||
            !g->env_grad_work)
            return fail(SVGIR_ERR_INVALID, "fused shading: the gradient outputs of the shading inputs must be provided");
        if (!render_specialised(p->S, p->VS, true)) return fail(SVGIR_ERR_INVALID, "fused shading without a specialised composite");
        if (g->dL_dreduced && !p->shade->all_surfels) return fail(SVGIR_ERR_INVALID, "fused shading: dL_dreduced needs all_surfels");
        if (!p->shade->all_surfels && !g->out_weights) return fail(SVGIR_ERR_INVALID, "fused shading: out_weights (the forward's) must be provided");
    }
    hipStream_t s = (hipStream_t)stream;
    const int P = p->P, W = p->W, H = p->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    const bool svgss = p->variant == SVGIR_SVGSS;
    const CfgRef cfg = cfg_ref(p);
    const float focal_y = H / (2.0f * p->tan_fovy), focal_x = W / (2.0f * p->tan_fovx);
    const GeomLayout G = geom_layout(geom_blob, P);
    const ImageLayout I = image_layout(image_blob, W, H);
    const int nstate = seg_nstate(p->S, svgss ? p->VS : 0);
    int cap = 0;
    long long cap_slots = -1;
    if (bin_bytes_compact(binning_bytes)) {   // laid out by the forward for a state-slot capacity of its choosing: the view's entry knows
        if (!view_lookup(image_blob, false, &cap, &cap_slots, nullptr, nullptr) || cap_slots < 0) {
            // not in the host table (an old forward, or a binder that moved the saved buffer): the image blob carries its own copy
            ViewCounts vc;
            HIP_OK(hipStreamSynchronize(s));
            if (!view_from_blob(I.counters, &vc) || vc.cap_slots < 0)
                return fail(SVGIR_ERR_INVALID, "the binning blob (%zu bytes) has a compact layout, but the image blob does not describe it "
                                               "(not the image blob of the same svgir_forward?)", binning_bytes);
            cap = vc.cap_R; cap_slots = vc.cap_slots;
        }
    } else {
        cap = binning_capacity_from_bytes(binning_bytes, T, nstate);
    }
    if (cap < R || bin_layout(nullptr, cap, T, nstate, cap_slots).bytes != binning_bytes)
        return fail(SVGIR_ERR_INVALID, "binning blob of %zu bytes does not match any layout for R=%d", binning_bytes, R);
    const BinLayout B = bin_layout(binning_blob, cap, T, nstate, cap_slots);
    const int fin = tile_sort_plan(T).passes & 1;
    StageTimer tm(s);

    // The forward dumped its blend states into slots sized from the workload's previous views and never waited to learn whether this
    // view fits (that wait stalls a host-bound training loop).  By now the view's slot total is in host memory: it enters the
    // workload's history, and if it exceeds the capacity the states are dumped AGAIN, all of them, into a stream-ordered temporary -- a
    // replay of the composite forward that writes nothing else (one extra forward composite on the rare view that outgrows its guess).
    float* seg_state = B.seg_state;
    void* redump = nullptr;
    {
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        const CapKey ckey{dev_id, W, H, P, p->S, svgss ? p->VS : 0, p->variant, p->workload_scope};
        note_view_slots(ckey, image_blob, 1, &s);
        long long slots = -1;
        bool seen = cap_slots >= 0 && R > 0 && view_lookup(image_blob, 1, nullptr, nullptr, nullptr, &slots, &s) && slots >= 0;
        if (cap_slots >= 0 && R > 0 && !seen) {
            // The slot total is not in host memory even after blocking on the stream (the table entry was recycled, or the blob came
            // from elsewhere): read the blob's own copy.  The backward never runs on state slots it has not verified.
            ViewCounts vc;
            HIP_OK(hipStreamSynchronize(s));
            if (!view_from_blob(I.counters, &vc))
                return fail(SVGIR_ERR_INVALID, "the image blob does not carry this view's state-slot total (forward failed, or a foreign blob)");
            slots = vc.slots; seen = true;
        }
        static const bool trace = getenv("SVGIR_TRACE_SPEC") != nullptr;
        if (trace) fprintf(stderr, "[svgir] backward: R=%d capacity=%d state slots: capacity %lld, view %lld%s\n", R, cap, cap_slots, slots, (seen && slots > cap_slots) ? " -> re-dump" : "");
        if (seen && slots > cap_slots) {
            if (!render_specialised(p->S, svgss ? p->VS : 0, svgss)) return fail(SVGIR_ERR_INVALID, "state slots without a specialised composite");
            HIP_OK(hipMallocAsync(&redump, align_up((size_t)slots * nstate * 64 * 4), s));
            seg_state = (float*)redump;
            g_spec_stats[2]++;
            RenderArgs ra{};
            ra.W = W; ra.H = H; ra.gx = gx; ra.gy = gy; ra.S = p->S; ra.VS = svgss ? p->VS : 0;
            ra.ranges = I.ranges; ra.point_list = B.val[fin]; ra.rec = G.rec; ra.features = p->features; ra.vfeatures = p->vfeatures;
            ra.bg = p->background; ra.cfg = cfg; ra.sub_list = B.sub_list; ra.sub_total = I.sub_total; ra.sub_order = I.sub_order;
            ra.sub_pair_base = I.sub_pair_base; ra.sub_slot_base = I.sub_slot_base; ra.slot_cap = (uint32_t)std::min<long long>(slots, 0xffffffffll);
            ra.sub_count = I.sub_count; ra.sub_ndump = I.sub_ndump; ra.seg_block = I.seg_block; ra.seg_state = seg_state;
            ra.dump_only = 1;
            ra.hi_fill = 0;
            ra.order_n = (int)order_entries(gx, gy);
            if (launch_render_fwd(ra, svgss, s) < 0) { (void)hipFreeAsync(redump, s); return fail(SVGIR_ERR_INVALID, "state re-dump: no specialised composite"); }
            tm.mark("state_redump");
        }
    }
    struct FreeAsync { void* p; hipStream_t s; ~FreeAsync() { if (p) (void)hipFreeAsync(p, s); } } free_redump{redump, s};

    RenderBwdArgs ba;
    ba.W = W; ba.H = H; ba.gx = gx; ba.gy = gy; ba.S = p->S; ba.VS = svgss ? p->VS : 0;
    ba.ranges = I.ranges; ba.point_list = B.val[fin]; ba.rec = G.rec; ba.features = p->features; ba.vfeatures = p->vfeatures;
    ba.bg = p->background;
    ba.cfg = cfg; ba.sub_list = B.sub_list; ba.sub_count = I.sub_count;
    ba.sub_ndump = I.sub_ndump; ba.seg_list = B.seg_list; ba.seg_desc = B.seg_desc; ba.seg_count = I.counters; ba.seg_state = seg_state;
    ba.seg_cap = (int)B.seg_cap;
    ba.backward_geometry = p->backward_geometry;
    ba.final_T = I.final_T; ba.final_D = I.final_D; ba.n_contrib = I.n_contrib;
    ba.g_color = g->dL_dout_color; ba.g_normal = g->dL_dout_normal; ba.g_depth = g->dL_dout_depth;
    ba.g_opacity = g->dL_dout_opacity; ba.g_feature = g->dL_dout_feature; ba.g_vfeature = g->dL_dout_vfeature;
    ba.dL_dmean2D = g->dL_dmeans2D; ba.dL_dconic = g->dL_dconic; ba.dL_dopacity = g->dL_dopacity; ba.dL_dcolor = g->dL_dcolors;
    ba.dL_dfeature = g->dL_dfeatures; ba.dL_dvfeature = g->dL_dvfeatures; ba.dL_dnormal = g->dL_dnormal; ba.dL_ddepth = g->dL_ddepth;
    // Composite gradients go through the scratch: svgss (VS > 0) -> one row per (instance, sub-tile) pair, summed per
    // Gaussian by grad_reduce (no atomics, deterministic); otherwise one packed row per Gaussian accumulated with float
    // atomics and unpacked by geom_bwd.
    const bool generic = !render_specialised(p->S, ba.VS, svgss);   // run-time-width kernels: atomics on the dL_d* tensors
    // (svgss rows: sized for the pair count of this view when the forward's read-back of it is at hand, else for the worst case)
    const size_t need = svgir_backward_scratch_bytes_for(p->variant, P, binning_bytes, image_blob, W, H, p->S, ba.VS);
    if (!generic && (!scratch || scratch_bytes < need))
        return fail(SVGIR_ERR_INVALID, "backward scratch of %zu bytes is smaller than svgir_backward_scratch_bytes_for() = %zu",
                    scratch ? scratch_bytes : (size_t)0, need);
    const bool rows = ba.VS > 0 && !generic;
    // Clears:
    //   1. the backward scratch (gradient-row validity bytes / packed rows), needed by the composite backward: it rides on the launch
    //      that builds the list of live segments (one kernel in front of the composite instead of a memset + that kernel);
    //   2. the dL_d* outputs, which start from zero (the kernels write the visible Gaussians only): the specialised composite
    //      backward does not touch them (it accumulates in the scratch), so this clear runs on a side stream next to it and is joined
    //      before the per-Gaussian kernels; the run-time-width composite adds into them, so there the clear comes first, on the
    //      caller's stream.
    // (scope guard: every exit path -- including the error returns below -- joins the side stream with the caller's stream and
    // releases the event)
    struct ClearJoin {
        hipStream_t s; hipEvent_t ev = nullptr;
        void join() {
            if (ev) { (void)hipStreamWaitEvent(s, ev, 0); (void)hipEventDestroy(ev); ev = nullptr; }
        }
        ~ClearJoin() { join(); }
    } cleared{s};
    // (one allocation behind all gradient tensors, 16-byte granular, and a specialised composite about to run: its waves clear it)
    const bool clear_in_kernel = !generic && R > 0 && g->clear_base && g->clear_bytes && (((uintptr_t)g->clear_base | g->clear_bytes) & 15) == 0;
    ba.clear = clear_in_kernel ? (uint4*)g->clear_base : nullptr;
    ba.clear_n16 = clear_in_kernel ? g->clear_bytes / 16 : 0;
    if (!clear_in_kernel) {
        hipStream_t cs = generic ? s : side_stream(s);
        if (!cs) cs = s;
        hipEvent_t ev_fork = nullptr;
        if (cs != s) {   // the tensors may have been used on `s` before (stream-ordered allocators): order the clear after that
            HIP_OK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            HIP_OK(hipEventRecord(ev_fork, s));
            HIP_OK(hipStreamWaitEvent(cs, ev_fork, 0));
            (void)hipEventDestroy(ev_fork);
        }
        const size_t Pz = (size_t)P * 4;
        if (g->clear_base && g->clear_bytes) {
            HIP_OK(hipMemsetAsync(g->clear_base, 0, g->clear_bytes, cs));
        } else {
            struct { float* p; size_t n; } t[] = {
                {g->dL_dmeans2D, 3 * Pz}, {g->dL_dconic, 4 * Pz}, {g->dL_dopacity, Pz}, {g->dL_dcolors, 3 * Pz},
                {g->dL_dfeatures, (size_t)p->S * Pz}, {g->dL_dvfeatures, (size_t)ba.VS * Pz}, {g->dL_dnormal, 3 * Pz},
                {g->dL_ddepth, Pz}, {g->dL_dmeans3D, 3 * Pz}, {g->dL_dcov3D, 6 * Pz}, {g->dL_dsh, (size_t)p->M * 3 * Pz},
                {g->dL_dscales, 3 * Pz}, {g->dL_drotations, 4 * Pz}, {svgss ? g->dL_dviewmat : nullptr, 64},
                {svgss ? g->dL_dprojmat : nullptr, 64}, {svgss ? g->dL_dcampos : nullptr, 12}};
            for (auto& e : t)
                if (e.p && e.n) HIP_OK(hipMemsetAsync(e.p, 0, e.n, cs));
        }
        if (cs != s) {
            HIP_OK(hipEventCreateWithFlags(&cleared.ev, hipEventDisableTiming));
            HIP_OK(hipEventRecord(cleared.ev, cs));
        }
    }
    const GradRowGeom rg = grad_row_geom(p->S, ba.VS);
    ba.grad_rows = generic ? nullptr : (float*)scratch;
    ba.row_of = nullptr; ba.rows_cap = 0;
    void* sc_clear = nullptr;
    size_t sc_bytes = 0;
    if (!generic) {
        if (rows) {   // reverse map (cleared) | compact rows
            ba.row_of = (uint32_t*)scratch;
            ba.grad_rows = (float*)(scratch + grad_rowof_bytes(cap));
            ba.rows_cap = (uint32_t)std::min<size_t>((scratch_bytes - grad_rowof_bytes(cap)) / ((size_t)rg.RS * 4), 0xfffffff0u);
            sc_clear = ba.row_of; sc_bytes = grad_rowof_bytes(cap);
        } else {
            sc_clear = ba.grad_rows; sc_bytes = align_up((size_t)P * rg.RS * 4);
        }
    }
    ShadeTables shade_tabs;   // (env == nullptr: the shading backward launches its own prologue)
    bool use_list = false;
    if (R > 0) {
        // live backward segments, longest first (common.hpp SEG), from the forward's per-sub-tile counts: built here -- a forward-only
        // call never pays for it -- together with the scratch clear
        RenderArgs sa{};
        sa.W = W; sa.H = H; sa.gx = gx; sa.gy = gy; sa.S = p->S; sa.VS = ba.VS;
        sa.ranges = I.ranges; sa.sub_count = I.sub_count; sa.sub_ndump = I.sub_ndump; sa.seg_list = B.seg_list; sa.seg_desc = B.seg_desc;
        sa.seg_count = I.counters; sa.seg_block = I.seg_block; sa.sub_pair_base = I.sub_pair_base; sa.sub_slot_base = I.sub_slot_base;
        // (fused shading: the same launch builds the tables and zeroes the env-gradient accumulator of the shading backward below)
        if (p->shade && g->env_grad_work) shade_tabs = shade_tables(&p->shade->sp, g->env_grad_work, p->shade->sp.env_h * p->shade->sp.env_w * 3);
        // (and counts, per chunk, the surfels that received a blend weight: the first half of the partition the per-Gaussian kernels walk)
        use_list = g->out_weights && !generic && (p->shade || P >= (rows ? kListMinPRows : kListMinP));
        launch_seg_build(sa, sc_clear, sc_bytes, shade_tabs, use_list ? g->out_weights : nullptr, P, G.shade_work, s);
    } else if (sc_clear && !rows) {
        HIP_OK(hipMemsetAsync(sc_clear, 0, sc_bytes, s));   // (nothing rendered: geom_bwd still unpacks the -- zero -- packed rows)
    }
    tm.mark("seg_build");
    // The surfels that received a blend weight (the forward's out_weights > 0): only they own gradient rows, only their per-Gaussian
    // gradients are non-zero, only their shading is differentiated.  With the weights at hand the per-Gaussian kernels behind the
    // composite walk that list (13-29 % of the model on the BASELINE scenes) instead of all P.
    const uint32_t* blended = nullptr; const uint32_t* blended_n = nullptr;
    if (use_list && R > 0) {
        uint32_t* cnt = G.shade_work + partition_work_words(P) - 1;
        launch_partition_scatter(P, g->out_weights, G.shade_list, G.shade_work, cnt, s);
        blended = G.shade_list; blended_n = cnt;
    }
    if (R > 0) {
        if (generic) launch_render_bwd_generic(ba, svgss, s);
        else (void)launch_render_bwd(ba, svgss, s);
    }
    tm.mark("render_bwd");
    cleared.join();
    if (R > 0 && rows) {
        GradReduceArgs ra;
        ra.list = blended; ra.list_count = blended_n;
        ra.P = P; ra.S = p->S; ra.VS = ba.VS; ra.radii = radii; ra.tiles = G.tiles; ra.rec = G.rec;
        ra.grad_rows = ba.grad_rows; ra.row_of = ba.row_of;
        ra.dL_dmean2D = g->dL_dmeans2D; ra.dL_dconic = g->dL_dconic; ra.dL_dopacity = g->dL_dopacity; ra.dL_dcolor = g->dL_dcolors;
        ra.dL_dfeature = g->dL_dfeatures; ra.dL_dvfeature = g->dL_dvfeatures; ra.dL_dnormal = g->dL_dnormal; ra.dL_ddepth = g->dL_ddepth;
        launch_grad_reduce(ra, s);
        tm.mark("grad_reduce");
    }
    if (p->shade) {
        // dL_dfeatures / dL_dvfeatures are complete: the shading's backward, for the surfels that received a blend weight (the rows of all
        // others are exactly zero: no pixel blended them)
        const bool all = p->shade->all_surfels != 0;   // (the arguments were validated before the first launch)
        svgir_shade_params sp = p->shade->sp;
        sp.subset = nullptr; sp.subset_count = nullptr;
        if (!all) {
            if (!blended) {   // (R == 0: nothing was blended -- the partition of all-zero weights zero-fills every row)
                uint32_t* cnt = G.shade_work + partition_work_words(P) - 1;
                launch_partition(P, nullptr, g->out_weights, G.shade_list, G.shade_work, cnt, s);
                blended = G.shade_list; blended_n = cnt;
            }
            sp.subset = blended; sp.subset_count = blended_n;
        }
        // (a binder that lays the four per-surfel gradient tensors out inside clear_base gets their zero rows from the composite
        // backward's clearing sweep -- stores nobody waits for -- instead of a zero-fill launch in front of the shading backward)
        auto in_clear = [&](const float* t, size_t floats) {
            const char* b = (const char*)g->clear_base, *q = (const char*)t;
            return clear_in_kernel && q >= b && q + floats * 4 <= b + g->clear_bytes;
        };
        const size_t Pz = (size_t)P;
        const bool precleared = in_clear(g->dL_dbase_color, 12 * Pz) && in_clear(g->dL_droughness, 4 * Pz) &&
                                in_clear(g->dL_dshade_normals, 12 * Pz) &&
                                (!g->dL_dradiance || in_clear(g->dL_dradiance, 3 * Pz * (size_t)sp.Ns));
        if (shade_backward_impl(&sp, g->dL_dreduced, g->dL_dfeatures, g->dL_dvfeatures, g->dL_dbase_color, g->dL_droughness,
                                g->dL_dshade_normals, g->dL_dradiance, g->dL_denv, g->env_grad_work, g->dL_dradiance_ratio, precleared,
                                shade_tabs.env != nullptr, s) != 0)
            return fail(SVGIR_ERR_INVALID, "fused shading: svgir_shade_backward rejected its parameters");
        tm.mark("shade_bwd");
    }

    GeomBwdArgs ga;
    ga.list = R > 0 ? blended : nullptr; ga.list_count = blended_n;
    ga.P = P; ga.D = p->D; ga.M = p->M;
    ga.means3D = p->means3D; ga.shs = p->colors_precomp ? nullptr : p->shs; ga.scales = p->scales; ga.rotations = p->rotations;
    ga.cov3D = p->cov3D_precomp ? p->cov3D_precomp : G.cov3D; ga.view = p->viewmatrix; ga.proj = p->projmatrix; ga.campos = p->cam_pos;
    ga.radii = radii; ga.clamped = G.clamped;
    ga.scale_modifier = p->scale_modifier; ga.tanx = p->tan_fovx; ga.tany = p->tan_fovy; ga.focal_x = focal_x; ga.focal_y = focal_y;
    ga.cfg = cfg; ga.svgss = svgss;
    ga.dL_dmean2D = g->dL_dmeans2D; ga.dL_dconic = g->dL_dconic; ga.dL_dcolor = g->dL_dcolors; ga.dL_dnormal = g->dL_dnormal;
    ga.dL_ddepth = g->dL_ddepth;
    ga.packed = (rows || generic) ? nullptr : ba.grad_rows; ga.S = p->S;
    ga.dL_dopacity = g->dL_dopacity; ga.dL_dfeature = g->dL_dfeatures;
    ga.dL_dmean3D = g->dL_dmeans3D; ga.dL_dcov3D = g->dL_dcov3D; ga.dL_dsh = g->dL_dsh; ga.dL_dscale = g->dL_dscales;
    ga.dL_drot = g->dL_drotations; ga.dL_dviewmat = g->dL_dviewmat; ga.dL_dprojmat = g->dL_dprojmat; ga.dL_dcampos = g->dL_dcampos;
    if (ga.scales && !ga.rotations) return fail(SVGIR_ERR_INVALID, "rotations missing");

    launch_geom_bwd(ga, s);
    tm.mark("geom_bwd");
    hipError_t e = p->debug ? hipStreamSynchronize(s) : hipSuccess;
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(SVGIR_ERR_HIP, "backward failed: %s", hipGetErrorString(e));
    return 0;
}

int svgir_mark_visible(int32_t variant, int32_t P, const float* means3D, const float* viewmatrix,
                       const float* projmatrix, uint8_t* present, void* stream) {
    (void)projmatrix;
    if (P < 0) return fail(SVGIR_ERR_INVALID, "P < 0");
    if (P == 0 || variant == SVGIR_SVGSS) return 0;  // svgss: kernel body is a no-op in the reference (Q14)
    if (!means3D || !viewmatrix || !present) return fail(SVGIR_ERR_INVALID, "NULL pointer");
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SVGIR_ERR_HIP, "mark_visible failed: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
