// dsa_api.hip — extern "C" ABI of include/defuse_dsa.h on top of the gfx950 kernels.
//
// Replaces the per-candidate call SplitAlignmentTask::Align (tools/SplitAlignment.cpp:371-444 of the
// reference) for whole batches.  There is deliberately no CPU fallback in this file: without a HIP
// device dsa_create fails and every other entry point needs a ctx.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "dsa_kernels.hpp"
#include "dsa_long.hpp"
#include "dsa_diag_host.hpp"

using namespace dsa;

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct HostResult {            // pinned: filled by k_publish at the end of a slice's first phase
    Counters ctr;
    int64_t n_rec;
    PlanGlobals plan;          // of the slice's latest planning
};

struct PipeLane {
    hipStream_t stream = nullptr;
    hipEvent_t ev[7] = {};      // 0 pack start, 4 pack end, 1 fill start, 2 fill end, 3 phase-1 end, 5 emit start, 6 emit end
    hipStream_t aux = nullptr;  // the two emit kernels of a slice run side by side: fork to aux, join back
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    HostResult* host = nullptr;
    int slice = -1;             // slice whose phase 1 is in flight
    int64_t resident_upload = -1;   // the descriptors of (resident_upload, resident_slice) are on the device
    int resident_slice = -1;
    bool emit_pending = false;  // phase 2 launched, its time not yet accounted
    bool emit_early = false;    // phase 1 already wrote the slice's records, into room for emit_cap of them
    uint64_t emit_cap = 0;
    int emit_from = 5;          // event that marks the start of the emit in flight
    uint64_t last_gtasks = 0;   // entries of the generic replay's list the last time this lane finished a slice
    bool phase1_end_recorded = false;  // ev[3] marks the end of the slice's phase 1 (not recorded when the emit follows at once)
    unsigned tier_hint = 0xFu;  // fill kernels the lane's last slice needed (bit t: k_fill_fast<t>, bit 3: k_fill_generic): the ones launched
    unsigned tiers_launched = 0;    // ... for the slice in flight
    DevBuf<uint8_t> d_wg_tier;  // which fill kernel owns each workgroup (written by k_fill_fast<0>)
    DevBuf<uint32_t> d_rowcodes, d_rowbytes, d_bnd, d_cmax, d_rmax, d_tmask;
    DevBuf<PairState> d_state;
    DevBuf<KeptRow> d_kept;
    DevBuf<int64_t> d_rec_count, d_rec_offset;
    DevBuf<ReplayTask> d_tasks;
    DevBuf<uint64_t> d_masks;
    DevBuf<uint2> d_gtasks;
    DevBuf<Counters> d_ctr;
    DevBuf<uint8_t> d_scan_tmp;
    DevBuf<int32_t> d_tstop;
#ifdef DSA_PRUNE_STATS
    DevBuf<unsigned long long> d_stats;
#endif
    void release()
    {
        d_wg_tier.release(); d_rowcodes.release(); d_rowbytes.release();
        d_bnd.release(); d_cmax.release(); d_rmax.release(); d_tmask.release(); d_state.release(); d_kept.release();
        d_rec_count.release(); d_rec_offset.release(); d_tasks.release(); d_masks.release(); d_gtasks.release();
        d_ctr.release(); d_scan_tmp.release(); d_tstop.release();
    }
};

// The two pipeline lanes of a context: streams, events, pinned result words and all per-slice scratch planes.  Contexts that
// run one after the other on one device (a caller that keeps several uploads resident, bench.py) may share one set
// (dsa_share_scratch), so the scratch is paid once.
struct LaneSet {
    int device = -1;
    hipStream_t own[2] = {};     // the lanes' private streams (lane 0 may run on a caller's stream instead)
    PipeLane lane[2];
    ~LaneSet()
    {
        if (device >= 0) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        for (PipeLane& L : lane) {
            L.release();
            for (auto& e : L.ev)
                if (e) (void)hipEventDestroy(e);
            if (L.ev_fork) (void)hipEventDestroy(L.ev_fork);
            if (L.ev_join) (void)hipEventDestroy(L.ev_join);
            if (L.aux) (void)hipStreamDestroy(L.aux);
            if (L.host) (void)hipHostFree(L.host);
        }
        for (hipStream_t st : own)
            if (st) (void)hipStreamDestroy(st);
    }
};

std::atomic<int64_t> g_upload_serial{0};      // unique over all contexts: lanes may be shared

struct Slice {                       // a contiguous range of the caller's pair order that fits the scratch budget
    int64_t pair_begin = 0, pair_end = 0;
    Geom g{};
};

}  // namespace

struct dsa_ctx {
    int device = -1;
    // The stream a context queues its copies and its planning on is ALWAYS lane 0's (main_stream()), read where it is used:
    // lanes may be shared (dsa_share_scratch, dsa_stream) and dsa_set_stream on one sharer changes lane 0's stream for all —
    // a copy of the handle per context would go stale and let dsa_plan and dsa_run queue on two different streams.
    hipStream_t main_stream() const;
    hipStream_t user_stream = nullptr;
    std::string err;
    size_t scratch_budget = (size_t)16 << 30;
    unsigned plan_flags = 0;         // DSA_PLAN_NO_*: parts of the sweep planning switched off (dsa_set_plan_options; DEFUSE_DSA_NO_* at creation)

    // resident batch
    int64_t n_pairs = 0, ref_bytes_len = 0, read_bytes_len = 0;
    int32_t n_fusions = 0;
    DevBuf<uint8_t> d_ref, d_reads;
    DevBuf<dsa_fusion> d_fusions;
    int64_t upload_serial = 0;       // serial of the resident upload (unique over all contexts)
    int nch_all = 1;                 // tiles of the widest window of the upload: refcodes stride = nch_all * W
    DevBuf<uint32_t> d_refcodes;     // packed once per run for the whole upload
    hipEvent_t ev_pack[2] = {};      // around k_pack_refs
    DevBuf<int32_t> d_orig;          // sweep order -> caller's pair index (Geom::orig)
    DevBuf<dsa_pair> d_pairs_in;     // the pairs in the caller's order, as uploaded
    DevBuf<dsa_pair> d_pairs;        // the pairs in sweep order, with the per-pair score bound in the padding bytes (k_plan_permute)
    DevBuf<int32_t> d_min_score;
    // sweep planning (dsa_plan.hpp), all on the device: runs per fusion, sort keys and order of the fusions, starts, ranks, bounds
    DevBuf<PlanRun> plan_runs;
    DevBuf<uint32_t> plan_key, plan_key_sorted;
    DevBuf<int32_t> plan_fidx, plan_order, plan_bsum, plan_start, plan_rank;
    DevBuf<uint16_t> plan_bound;
    DevBuf<uint8_t> plan_flip, plan_sort_tmp;
    DevBuf<PlanGlobals> plan_glob;   // one per slice
    PlanParams plan_prm{};
    hipEvent_t ev_plan[2] = {};      // around the planning kernels
    bool plan_timed = false;         // ev_plan was recorded since the last run read it
    std::vector<Slice> slices;
    float last_plan_ms = 0.f;        // device time of the latest planning (upload's or dsa_plan's)
    std::vector<int32_t> h_min_score;
    // pairs beyond the 16-bit kernels (dsa_long.hpp): reads longer than FAST_MAX_READ or windows longer than FAST_MAX_REF
    std::vector<LongDesc> h_long;
    std::vector<int32_t> h_long_fusions;
    std::vector<LongState> h_long_state;
    DevBuf<LongDesc> d_long;
    DevBuf<int32_t> d_long_fusions, d_long_work, d_long_rows;
    DevBuf<LongState> d_long_state;
    DevBuf<uint32_t> d_long_bits;
    int64_t long_cells = 0;          // DP cells of the long pairs
    int64_t long_blank_cells = 0;    // what the planning counts for their blanked copies (an empty read on the blanked windows)

    // per-slice scratch lives in two pipeline lanes so that the latency-bound finish stage of one
    // slice overlaps the fill of the next (separate HIP streams)
    std::shared_ptr<LaneSet> lanes;
    DevBuf<dsa_record> d_records;
    int64_t n_records = 0;
    bool have_results = false;

    dsa_timing timing{};
};

hipStream_t dsa_ctx::main_stream() const { return lanes->lane[0].stream; }

namespace {

int fail(dsa_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPC(call)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(ctx, DSA_E_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                       \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// minScore exactly as the reference writes it (tools/SplitAlignment.cpp:379):
// (int)((float)len * (float)matchScore * 0.90)
int min_score_for(int lq)
{
    float f = (float)lq * (float)DSA_MATCH;
    return (int)((double)f * 0.90);
}

size_t slice_scratch_bytes(int64_t n_waves, int lq1, int nch)
{
    size_t rows = (size_t)n_waves * lq1 * WAVE * 4;
    return 3 * rows + rows / 4 + 2 * rows * (size_t)nch;      // rowcodes, rmax, tmask; the byte plane of the table sweeps; bnd, cmax
}

// Slices bound the scratch footprint: contiguous ranges of the caller's pair order that fit the budget (and
// DEFUSE_DSA_SLICE_PAIRS); inside a slice pair p of the SWEEP order lives in wave p/64, lane p%64.  Host work is
// O(slices): the geometry depends on the sizes of the upload only, everything that depends on the pairs themselves is
// planned on the device (enqueue_plan) or found by the fill kernels (rows and tiles per wave, fusions per workgroup).
void make_slices(dsa_ctx* ctx, int64_t n_pairs, int lqmax)
{
    ctx->slices.clear();
    const int lq1 = (lqmax + 1 + 3) & ~3;          // row planes are stored four rows per 16-byte word
    const size_t per_wg = slice_scratch_bytes(WG_WAVES, lq1, ctx->nch_all);
    int64_t chunk = (int64_t)std::min<size_t>((size_t)1 << 40, ctx->scratch_budget / std::max<size_t>(per_wg, 1)) * WG_LANES;
    // A batch that fits one slice runs as ONE fill launch.  Cutting it in two so that the latency-bound finish kernels of the
    // first half run beside the fill of the second (DEFUSE_DSA_SPLIT_LARGE=1) measured +1..2 % aligns/s on BASELINE configs[1]
    // with the two fills overlapping on the two lanes — which blurs the per-launch times the roofline is computed from — and
    // -10 % with the fills serialised; four or eight slices lose as well.  So it stays off.
    if (const char* e = getenv("DEFUSE_DSA_SPLIT_LARGE"))
        if (atoi(e) != 0 && chunk >= n_pairs && n_pairs >= ((int64_t)1 << 19)) chunk = ((n_pairs + 1) / 2 + WG_LANES - 1) / WG_LANES * WG_LANES;
    if (const char* e = getenv("DEFUSE_DSA_SLICE_PAIRS")) chunk = std::min<int64_t>(chunk, std::max<int64_t>(WG_LANES, atoll(e) / WG_LANES * WG_LANES));
    chunk = std::max<int64_t>(chunk, WG_LANES);
    for (int64_t b = 0; b < n_pairs; b += chunk) {
        Slice cur;
        cur.pair_begin = b;
        cur.pair_end = std::min(n_pairs, b + chunk);
        const int64_t n = cur.pair_end - cur.pair_begin;
        cur.g.n_waves = (int32_t)((n + WAVE - 1) / WAVE);
        cur.g.n_wgs = (int32_t)((n + WG_LANES - 1) / WG_LANES);
        cur.g.lq1 = lq1;
        cur.g.nch = ctx->nch_all;
        cur.g.lrp = ctx->nch_all * W;
        cur.g.n_fusions = ctx->n_fusions;
        cur.g.n_pairs = n;
        cur.g.orig = nullptr;                       // set when the buffers are known (enqueue_plan)
        ctx->slices.push_back(cur);
    }
}

// The sweep plan of the resident upload (dsa_plan.hpp), queued on the context's stream: no host round trip, nothing the host
// waits for.  Per slice: runs per fusion, one workgroup per fusion (diagonals, bounds, ranks, tiles, sort key), radix sort
// of the fusion keys, starts, permutation.  DEFUSE_DSA_NO_REORDER=1 keeps the caller's order (and drops the bounds),
// DEFUSE_DSA_NO_RANK / _NO_TIGHTEN / _NO_LPT switch the parts of the plan off one by one.
int enqueue_plan(dsa_ctx* ctx)
{
    const int nf = ctx->n_fusions;
    const int64_t n_pairs = ctx->n_pairs;
    if (n_pairs == 0 || nf == 0 || ctx->slices.empty()) return DSA_OK;
    hipStream_t st = ctx->main_stream();
    const bool no_reorder = ctx->plan_flags & DSA_PLAN_NO_REORDER, no_rank = ctx->plan_flags & DSA_PLAN_NO_RANK,
               no_tighten = ctx->plan_flags & DSA_PLAN_NO_TIGHTEN, no_lpt = ctx->plan_flags & DSA_PLAN_NO_LPT;
    int64_t max_chunk = 0;
    for (const Slice& sl : ctx->slices) max_chunk = std::max(max_chunk, sl.pair_end - sl.pair_begin);
    const int nb = (nf + PLACE_BLOCK - 1) / PLACE_BLOCK;
    HIPC(ctx->plan_runs.reserve((size_t)nf));
    HIPC(ctx->plan_key.reserve((size_t)nf));
    HIPC(ctx->plan_key_sorted.reserve((size_t)nf));
    HIPC(ctx->plan_fidx.reserve((size_t)nf));
    HIPC(ctx->plan_order.reserve((size_t)nf));
    HIPC(ctx->plan_bsum.reserve((size_t)nb));
    HIPC(ctx->plan_start.reserve((size_t)nf));
    HIPC(ctx->plan_flip.reserve((size_t)nf));
    HIPC(ctx->plan_rank.reserve((size_t)max_chunk));
    HIPC(ctx->plan_bound.reserve((size_t)max_chunk));
    HIPC(ctx->plan_glob.reserve(ctx->slices.size()));
    HIPC(ctx->d_orig.reserve((size_t)n_pairs));
    HIPC(ctx->d_pairs.reserve((size_t)n_pairs + 1));
    size_t sort_tmp = 0;
    HIPC(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, ctx->plan_key.p, ctx->plan_key_sorted.p, ctx->plan_fidx.p, ctx->plan_order.p, nf, 0, PLAN_KEY_BITS, st));
    HIPC(ctx->plan_sort_tmp.reserve(sort_tmp));
    const PlanParams prm = ctx->plan_prm;
    const size_t lds = plan_lds_bytes(prm.wc, prm.slots);
    HIPC(hipEventRecord(ctx->ev_plan[0], st));
    HIPC(hipMemsetAsync(ctx->plan_glob.p, 0, ctx->slices.size() * sizeof(PlanGlobals), st));
    for (size_t k = 0; k < ctx->slices.size(); ++k) {
        Slice& sl = ctx->slices[k];
        const int64_t n = sl.pair_end - sl.pair_begin;
        const dsa_pair* in = ctx->d_pairs_in.p + sl.pair_begin;
        sl.g.orig = ctx->d_orig.p + sl.pair_begin;          // slice-relative indices of the caller's order
        PlanGlobals* glob = ctx->plan_glob.p + k;
        if (!no_reorder) {
            HIPC(hipMemsetAsync(ctx->plan_runs.p, 0, (size_t)nf * sizeof(PlanRun), st));
            hipLaunchKernelGGL(k_plan_runs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, ctx->plan_runs.p);
            PlanParams pp = prm;
            pp.use_rank = no_rank ? 0 : 1;
            pp.use_bound = no_tighten ? 0 : 1;
            pp.use_lpt = no_lpt ? 0 : 1;
            hipLaunchKernelGGL(k_plan_fusion, dim3((unsigned)nf), dim3(PLAN_THREADS), lds, st, ctx->d_ref.p, ctx->d_fusions.p, ctx->d_reads.p, in,
                               ctx->plan_runs.p, ctx->plan_key.p, ctx->plan_fidx.p, ctx->plan_rank.p, ctx->plan_bound.p, glob, pp);
            size_t tmp = ctx->plan_sort_tmp.cap;
            HIPC(hipcub::DeviceRadixSort::SortPairs(ctx->plan_sort_tmp.p, tmp, ctx->plan_key.p, ctx->plan_key_sorted.p, ctx->plan_fidx.p,
                                                    ctx->plan_order.p, nf, 0, PLAN_KEY_BITS, st));
            hipLaunchKernelGGL(k_plan_place_a, dim3((unsigned)nb), dim3(PLACE_BLOCK), 0, st, ctx->plan_order.p, ctx->plan_runs.p, nf, ctx->plan_bsum.p);
            hipLaunchKernelGGL(k_plan_place_b, dim3((unsigned)nb), dim3(PLACE_BLOCK), 0, st, ctx->plan_order.p, ctx->plan_runs.p, nf, ctx->plan_bsum.p,
                               ctx->plan_start.p, ctx->plan_flip.p);
        }
        hipLaunchKernelGGL(k_plan_permute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, ctx->d_fusions.p, ctx->plan_runs.p, ctx->plan_start.p,
                           ctx->plan_flip.p, ctx->plan_rank.p, ctx->plan_bound.p, glob, no_reorder ? 1 : 0, ctx->d_pairs.p + sl.pair_begin,
                           ctx->d_orig.p + sl.pair_begin);
    }
    HIPC(hipEventRecord(ctx->ev_plan[1], st));
    HIPC(hipGetLastError());
    ctx->plan_timed = true;
    for (PipeLane& L : ctx->lanes->lane) L.resident_upload = -1;
    return DSA_OK;
}

int exclusive_scan(dsa_ctx* ctx, PipeLane& L, const int64_t* in, int64_t* out, int64_t n)
{
    size_t tmp = 0;
    HIPC(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, in, out, (int)n, L.stream));
    HIPC(L.d_scan_tmp.reserve(tmp));
    HIPC(hipcub::DeviceScan::ExclusiveSum(L.d_scan_tmp.p, tmp, in, out, (int)n, L.stream));
    return DSA_OK;
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int grow_records(dsa_ctx* ctx, size_t need)
{
    if (need <= ctx->d_records.cap) return DSA_OK;
    DevBuf<dsa_record> bigger;
    HIPC(bigger.reserve(need + need / 2 + 1024));
    HIPC(hipDeviceSynchronize());       // an emit of the other lane may still be writing the old buffer
    if (ctx->n_records)
        HIPC(hipMemcpy(bigger.p, ctx->d_records.p, ctx->n_records * sizeof(dsa_record), hipMemcpyDeviceToDevice));
    ctx->d_records.release();
    std::swap(ctx->d_records.p, bigger.p);
    std::swap(ctx->d_records.cap, bigger.cap);
    return DSA_OK;
}

FinishBufs finish_bufs(PipeLane& L)
{
    FinishBufs fb;
    fb.state = L.d_state.p;
    fb.kept = L.d_kept.p;
    fb.tasks = L.d_tasks.p;
    fb.masks = L.d_masks.p;
    fb.gtasks = L.d_gtasks.p;
    fb.ctr = L.d_ctr.p;
    fb.kept_cap = L.d_kept.cap;
    fb.task_cap = L.d_tasks.cap;
    fb.mask_cap = L.d_masks.cap / 2;
    fb.gtask_cap = L.d_gtasks.cap;
    fb.tstop = L.d_tstop.p;
    fb.rec_count = L.d_rec_count.p;
    return fb;
}

// the records of a slice, behind those of the earlier slices (ev[5]..ev[6] time it): the pairs counted in the fill
// kernel's tail by a per-pair launch, the others through the generic replay's task list
// The listed pairs' kernels map one list entry to a thread; blocks without entries and second rounds both cost (they are
// latency-bound kernels of a few hundred blocks), so the grid follows the length the list had the last time this lane ran
// (the same batch run again: exact; another upload of the job: close), or a tenth of the pairs before anything is known.
unsigned listed_grid(const PipeLane& L, int64_t n_pairs)
{
    const uint64_t expect = L.last_gtasks ? L.last_gtasks : (uint64_t)n_pairs / 10 + 1;
    const uint64_t blocks = (expect + expect / 64 + EMIT_BLOCK - 1) / EMIT_BLOCK + 1;
    return (unsigned)std::min<uint64_t>(std::max<uint64_t>(blocks, 16), 16384);
}
void launch_emit(dsa_ctx* ctx, PipeLane& L, const Slice& s, size_t cap_left, bool record_start = true)
{
    const int64_t np = s.g.n_pairs;
    Geom g = s.g;
    g.tiers_launched = L.tiers_launched;
    const dsa_pair* pairs = ctx->d_pairs.p + s.pair_begin;
    dsa_record* out = ctx->d_records.p + ctx->n_records;
    L.emit_from = record_start ? 5 : 2;
    if (record_start) (void)hipEventRecord(L.ev[5], L.stream);      // (every event between two kernels is a gap of a few microseconds)
    // the listed pairs' kernel is a few latency-bound waves, the counted pairs' one streams: side by side
    (void)hipEventRecord(L.ev_fork, L.stream);
    (void)hipStreamWaitEvent(L.aux, L.ev_fork, 0);
    hipLaunchKernelGGL(k_emit_listed<true>, dim3(listed_grid(L, np)), dim3(EMIT_BLOCK), 0, L.aux, L.d_gtasks.p, (uint64_t)L.d_gtasks.cap, L.d_ctr.p,
                       pairs, ctx->d_fusions.p, L.d_state.p, L.d_kept.p, L.d_tasks.p, (uint64_t)L.d_tasks.cap, L.d_masks.p,
                       (uint64_t)(L.d_masks.cap / 2), (uint64_t)L.d_kept.cap, L.d_rec_count.p, (const int64_t*)L.d_rec_offset.p, out,
                       (uint64_t)cap_left, (int64_t)s.pair_begin, g);
    (void)hipEventRecord(L.ev_join, L.aux);
    hipLaunchKernelGGL(k_emit_counted, dim3((unsigned)((np + EMIT_BLOCK - 1) / EMIT_BLOCK)), dim3(EMIT_BLOCK), 0, L.stream, pairs, ctx->d_fusions.p,
                       L.d_state.p, L.d_kept.p, L.d_tasks.p, L.d_masks.p, (const int64_t*)L.d_rec_offset.p, out, (uint64_t)cap_left,
                       (int64_t)s.pair_begin, L.d_ctr.p, (uint64_t)L.d_kept.cap, (uint64_t)L.d_tasks.cap, (uint64_t)(L.d_masks.cap / 2),
                       (uint64_t)L.d_gtasks.cap, g);
    if (!ctx->h_long.empty())
        hipLaunchKernelGGL(k_long_emit<true>, dim3((unsigned)((ctx->h_long.size() + 63) / 64)), dim3(64), 0, L.stream, ctx->d_long.p, ctx->d_long_state.p,
                           (int)ctx->h_long.size(), ctx->d_long_work.p, ctx->d_long_bits.p, (int64_t)s.pair_begin, (int64_t)s.pair_end, L.d_rec_count.p,
                           (const int64_t*)L.d_rec_offset.p, out, (uint64_t)cap_left);
    (void)hipStreamWaitEvent(L.stream, L.ev_join, 0);
    (void)hipEventRecord(L.ev[6], L.stream);
}

// fill (with the per-workgroup combine and table-driven replay in its tail) -> generic replay ->
// count -> scan, then async copies of the cursors and the record total into the lane's pinned result
int launch_compute(dsa_ctx* ctx, PipeLane& L, const Slice& s)
{
    Geom g = s.g;
#ifdef DSA_PRUNE_STATS
    HIPC(L.d_stats.reserve(16));
    HIPC(hipMemsetAsync(L.d_stats.p, 0, 16 * sizeof(unsigned long long), L.stream));
    g.stats = L.d_stats.p;
#endif
    hipStream_t st = L.stream;
    const int64_t np = g.n_pairs;
    const dsa_pair* pairs = ctx->d_pairs.p + s.pair_begin;
    const FinishBufs fb = finish_bufs(L);
    // (own kernels rather than hipMemsetAsync for the words that must be zero, and rather than copy commands for the
    // results: every command between two kernels costs a gap of its own on the stream)
    hipLaunchKernelGGL(k_reset_finish, dim3(1), dim3(64), 0, st, L.d_ctr.p, L.d_rec_count.p + np);
    HIPC(hipEventRecord(L.ev[1], st));
    // Every workgroup is run by exactly one of the fill kernels: k_fill_fast<0> finds the owner of each (its table tier,
    // or the generic kernel) and is always launched; the others only if the lane's last slice needed them (an empty launch
    // is a gap of its own on the stream).  The slice says what it needed (Counters::need_tiers): phase2 runs it again with
    // every kernel if one was missing.
    const unsigned mask = L.tier_hint | 1u;
    L.tiers_launched = mask;
    g.tiers_launched = mask;
    // (the instantiation with the in-register row-maximum reduction for 9-16 tiles only where a slice has that many:
    // windows beyond 512 bases, e.g. 2x150 bp reads)
#define DSA_LAUNCH_FILL(TIER, WIDE)                                                                                                              \
    hipLaunchKernelGGL((k_fill_fast<TIER, WIDE>), dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, L.d_wg_tier.p, ctx->d_refcodes.p, \
                       ctx->d_reads.p, L.d_rowcodes.p, L.d_rowbytes.p, ctx->d_min_score.p, ctx->d_fusions.p, L.d_bnd.p, L.d_cmax.p, L.d_rmax.p, L.d_tmask.p, fb, g)
    const bool wide = g.nch > 8 && g.nch <= 16;
    if (wide) DSA_LAUNCH_FILL(0, true); else DSA_LAUNCH_FILL(0, false);
    if (mask & 2u) { if (wide) DSA_LAUNCH_FILL(1, true); else DSA_LAUNCH_FILL(1, false); }
    if (mask & 4u) { if (wide) DSA_LAUNCH_FILL(2, true); else DSA_LAUNCH_FILL(2, false); }
#undef DSA_LAUNCH_FILL
    if (mask & 8u)
        hipLaunchKernelGGL(k_fill_generic, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, ctx->d_fusions.p, L.d_wg_tier.p, ctx->d_refcodes.p,
                           ctx->d_reads.p, L.d_rowcodes.p, ctx->d_min_score.p, L.d_bnd.p, L.d_cmax.p, L.d_rmax.p, L.d_tmask.p, fb, g);
    HIPC(hipEventRecord(L.ev[2], st));
    hipLaunchKernelGGL(k_replay, dim3(2048), dim3(REPLAY_BLOCK), 0, st, L.d_tasks.p, (uint64_t)L.d_tasks.cap, L.d_gtasks.p,
                       (uint64_t)L.d_gtasks.cap, L.d_ctr.p, L.d_state.p, L.d_kept.p, (uint64_t)L.d_kept.cap, pairs, ctx->d_fusions.p,
                       ctx->d_refcodes.p, L.d_rowcodes.p, L.d_bnd.p, L.d_tstop.p, L.d_masks.p, (uint64_t)(L.d_masks.cap / 2), g);
    hipLaunchKernelGGL(k_emit_listed<false>, dim3(listed_grid(L, np)), dim3(EMIT_BLOCK), 0, st, L.d_gtasks.p, (uint64_t)L.d_gtasks.cap, L.d_ctr.p,
                       pairs, ctx->d_fusions.p, L.d_state.p, L.d_kept.p, L.d_tasks.p, (uint64_t)L.d_tasks.cap, L.d_masks.p,
                       (uint64_t)(L.d_masks.cap / 2), (uint64_t)L.d_kept.cap, L.d_rec_count.p, (const int64_t*)nullptr, (dsa_record*)nullptr,
                       (uint64_t)0, (int64_t)s.pair_begin, g);
    if (!ctx->h_long.empty())           // the long pairs of this slice: their counts replace the zeros of the blanked copies
        hipLaunchKernelGGL(k_long_emit<false>, dim3((unsigned)((ctx->h_long.size() + 63) / 64)), dim3(64), 0, st, ctx->d_long.p, ctx->d_long_state.p,
                           (int)ctx->h_long.size(), ctx->d_long_work.p, ctx->d_long_bits.p, (int64_t)s.pair_begin, (int64_t)s.pair_end, L.d_rec_count.p,
                           (const int64_t*)nullptr, (dsa_record*)nullptr, (uint64_t)0);
    if (int rc = exclusive_scan(ctx, L, L.d_rec_count.p, L.d_rec_offset.p, np + 1)) return rc;
    // the cursors and the record total go to the lane's pinned result words
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, L.d_ctr.p, L.d_rec_offset.p + np, reinterpret_cast<const unsigned long long*>(ctx->plan_glob.p + (&s - ctx->slices.data())),
                       &L.host->ctr, &L.host->n_rec, reinterpret_cast<unsigned long long*>(&L.host->plan));
    // The first slice of a run knows where its records go: write them right away, into the room there is, without
    // waiting for the host to read the total (phase2 runs the emit again after growing the buffer if it was short).
    L.emit_cap = 0;
    L.emit_early = false;
    L.phase1_end_recorded = false;
    if (ctx->n_records == 0 && &s == &ctx->slices.front() && ctx->d_records.cap > 0) {
        L.emit_cap = ctx->d_records.cap;
        L.emit_early = true;
        launch_emit(ctx, L, s, L.emit_cap, false);        // timed from ev[2], the end of the fill, with the rest of the finish stage
    } else {
        HIPC(hipEventRecord(L.ev[3], st));
        L.phase1_end_recorded = true;
    }
    HIPC(hipGetLastError());
    return DSA_OK;
}

// phase 1 of a slice: pack, fill, finish up to the record counts — everything asynchronous
int phase1(dsa_ctx* ctx, PipeLane& L, int slice_idx)
{
    const Slice& s = ctx->slices[slice_idx];
    Geom g = s.g;
    hipStream_t st = L.stream;
    const int64_t np = g.n_pairs;
    if (g.lq1 > ((FAST_MAX_READ + 1 + 3) & ~3)) return fail(ctx, DSA_E_LIMIT, "internal: a long read reached the 16-bit kernels");
    const size_t n_rows = (size_t)g.n_waves * g.lq1 * WAVE;
    HIPC(L.d_wg_tier.reserve((size_t)g.n_wgs));
    HIPC(L.d_rowcodes.reserve(n_rows));
    HIPC(L.d_rowbytes.reserve(n_rows / 4));        // one byte per row for the table sweeps (lq1 is a multiple of 4)
    HIPC(L.d_bnd.reserve(n_rows * g.nch));
    HIPC(L.d_cmax.reserve(n_rows * g.nch));
    HIPC(L.d_rmax.reserve(n_rows));
    HIPC(L.d_tmask.reserve(n_rows));
    HIPC(L.d_tstop.reserve((size_t)g.n_waves * g.nch));
    HIPC(L.d_state.reserve(np));
    HIPC(L.d_rec_count.reserve(np + 1));
    HIPC(L.d_rec_offset.reserve(np + 1));
    HIPC(L.d_ctr.reserve(1));
    HIPC(L.d_kept.reserve((size_t)np * 2 + 1024));
    HIPC(L.d_tasks.reserve((size_t)np * 4 + 1024));
    HIPC(L.d_masks.reserve((size_t)np * 8 + 1024));
    HIPC(L.d_gtasks.reserve((size_t)np * 2 + 1024));
    if (int rc = launch_compute(ctx, L, s)) return rc;
    L.slice = slice_idx;
    return DSA_OK;
}

void account_emit(dsa_ctx* ctx, PipeLane& L)
{
    if (!L.emit_pending) return;
    (void)hipEventSynchronize(L.ev[6]);
    ctx->timing.finish_ms += elapsed(L.ev[L.emit_from], L.ev[6]);
    L.emit_pending = false;
}

// phase 2: wait for the slice's counts, grow whatever overflowed (rare; then the finish stage is
// re-run), and write its records behind those of the earlier slices
int phase2(dsa_ctx* ctx, PipeLane& L)
{
    const Slice& s = ctx->slices[L.slice];
    account_emit(ctx, L);
    for (int attempt = 0;; ++attempt) {
        HIPC(hipStreamSynchronize(L.stream));
        HIPC(hipGetLastError());
        const Counters c = L.host->ctr;
        const bool tiers_ok = (c.need_tiers & ~L.tiers_launched) == 0;
        L.tier_hint = tiers_ok ? c.need_tiers : 0xFu;          // what the next slice of this lane launches
        if (tiers_ok && c.n_kept <= L.d_kept.cap && c.n_tasks <= L.d_tasks.cap && c.n_masks <= L.d_masks.cap / 2 && c.n_gtasks <= L.d_gtasks.cap) break;
        if (attempt >= 3) return fail(ctx, DSA_E_DEVICE, "finish stage did not converge");
        HIPC(L.d_kept.reserve(c.n_kept + 1024));
        HIPC(L.d_tasks.reserve(c.n_tasks + 1024));
        HIPC(L.d_masks.reserve(2 * c.n_masks + 1024));
        HIPC(L.d_gtasks.reserve(c.n_gtasks + 1024));
        if (int rc = launch_compute(ctx, L, s)) return rc;
    }
#ifdef DSA_PRUNE_STATS
    diag_dump_plan();
    diag_dump_slice(L.d_stats.p, L.d_gtasks.p, (size_t)L.host->ctr.n_gtasks, L.d_tasks.p, (size_t)L.host->ctr.n_tasks);
#endif
    L.last_gtasks = L.host->ctr.n_gtasks;
    ctx->timing.cells += (int64_t)L.host->plan.cells;
    const int64_t n_rec = L.host->n_rec;
    if (!(L.emit_early && (uint64_t)n_rec <= L.emit_cap)) {
        if (L.emit_early) ctx->timing.finish_ms += elapsed(L.ev[L.emit_from], L.ev[6]);     // the short attempt was work too
        if (int rc = grow_records(ctx, (size_t)(ctx->n_records + n_rec))) return rc;
        launch_emit(ctx, L, s, ctx->d_records.cap - ctx->n_records);
    }
    HIPC(hipGetLastError());
    L.emit_early = false;
    L.emit_pending = true;
    ctx->n_records += n_rec;
    ctx->timing.fill_ms += elapsed(L.ev[1], L.ev[2]);
    if (L.phase1_end_recorded) ctx->timing.finish_ms += elapsed(L.ev[2], L.ev[3]);
    ctx->timing.fill_launches += 1;
    ctx->timing.n_replay_tasks += (int64_t)L.host->ctr.n_tasks;
    ctx->timing.n_generic_tasks += (int32_t)L.host->ctr.n_gtasks;
    L.slice = -1;
    return DSA_OK;
}

}  // namespace

extern "C" {

#ifndef DSA_BUILD_HASH
#define DSA_BUILD_HASH "unknown"
#endif
// the last word is the hash of the sources and flags this library was built from (defuse_amd/build.py): profiles carry the
// same hash, so a bench line can tell whether committed counters belong to the kernels that are running
const char* dsa_version(void) { return "defuse_amd dsa 0.3 (gfx950) src " DSA_BUILD_HASH; }
#ifndef DSA_BUILD_FLAGS
#define DSA_BUILD_FLAGS "sched=unknown"
#endif
// what the split-read kernels were compiled with: "sched=<iterative-ilp|default> <compiler flags>" — the scheduler is the one
// flag the fill kernel's speed depends on, and a compiler without it builds the library all the same (defuse_amd/build.py)
const char* dsa_build_flags(void) { return DSA_BUILD_FLAGS; }

int dsa_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// Among n devices: the first one, counting from pid mod n, whose lock file this process can take (held until the
// process ends, released by the kernel even after a crash); with all n taken, pid mod n.  So up to n tool processes
// started side by side (scripts/defuse_run.pl --parallel) land on n different GPUs instead of colliding at random.
int dsa_pick_device_among(int n)
{
    if (n <= 1) return 0;
    // The claim is one per process: a lock taken with flock belongs to its open file description, so a second call with a
    // fresh descriptor would find the process's own lock taken, move on to the next device and hold two.  The first
    // decision for a given n is remembered.
    static std::mutex pick_mutex;
    static int picked_n = -1, picked = 0;
    std::lock_guard<std::mutex> lk(pick_mutex);
    if (picked_n == n) return picked;
    const int start = (int)((unsigned long)getpid() % (unsigned long)n);
    const char* dir = getenv("DEFUSE_GPU_LOCK_DIR");
    if (!dir) dir = "/tmp";
    int found = start;
    static int held_fd = -1;                   // a claim for another n is given up first
    if (held_fd >= 0) { close(held_fd); held_fd = -1; }
    for (int k = 0; k < n; ++k) {
        const int d = (start + k) % n;
        char path[512];
        snprintf(path, sizeof path, "%s/defuse_gpu.%d.lock", dir, d);
        const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd < 0) continue;
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) {        // fd stays open on purpose: it is the claim
            held_fd = fd;
            found = d;
            break;
        }
        close(fd);
    }
    picked_n = n;
    picked = found;
    return found;
}

int dsa_pick_device(void)
{
    if (const char* e = getenv("DEFUSE_GPU")) return atoi(e);
    return dsa_pick_device_among(dsa_device_count());
}

int dsa_set_plan_options(dsa_ctx* ctx, unsigned flags)
{
    if (!ctx || (flags & ~(DSA_PLAN_NO_REORDER | DSA_PLAN_NO_RANK | DSA_PLAN_NO_TIGHTEN | DSA_PLAN_NO_LPT))) return DSA_E_ARG;
    ctx->plan_flags = flags;
    return DSA_OK;
}

int dsa_set_scratch_budget(dsa_ctx* ctx, int64_t bytes)
{
    if (!ctx || bytes <= 0) return DSA_E_ARG;
    ctx->scratch_budget = (size_t)bytes;
    return DSA_OK;
}

// A context on `device`; with `share` it uses that context's pipeline lanes from the start (the slots of a dsa_stream: no
// streams, events and pinned words of its own that dsa_share_scratch would throw away at once — creating a stream is
// creating a hardware queue, tens of milliseconds each for the first few of a process).
static int create_ctx(dsa_ctx** out, int device, dsa_ctx* share)
{
    if (!out) return DSA_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DSA_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DSA_E_DEVICE;
    dsa_ctx* ctx = new dsa_ctx();
    ctx->device = device;
    bool ok = true;
    for (const auto& sw : {std::pair<const char*, unsigned>{"DEFUSE_DSA_NO_REORDER", DSA_PLAN_NO_REORDER}, {"DEFUSE_DSA_NO_RANK", DSA_PLAN_NO_RANK},
                           {"DEFUSE_DSA_NO_TIGHTEN", DSA_PLAN_NO_TIGHTEN}, {"DEFUSE_DSA_NO_LPT", DSA_PLAN_NO_LPT}}) {
        const char* e = getenv(sw.first);
        if (e && atoi(e) != 0) ctx->plan_flags |= sw.second;
    }
    if (share) {
        ctx->lanes = share->lanes;
        ctx->scratch_budget = share->scratch_budget;
    } else {
        ctx->lanes = std::make_shared<LaneSet>();
        ctx->lanes->device = device;
        for (int l = 0; l < 2; ++l) {
            PipeLane& L = ctx->lanes->lane[l];
            ok = ok && hipStreamCreate(&ctx->lanes->own[l]) == hipSuccess;
            L.stream = ctx->lanes->own[l];
            for (auto& e : L.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
            ok = ok && hipStreamCreateWithFlags(&L.aux, hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipHostMalloc((void**)&L.host, sizeof(HostResult)) == hipSuccess;
        }
    }
    for (auto& e : ctx->ev_pack) ok = ok && hipEventCreate(&e) == hipSuccess;
    for (auto& e : ctx->ev_plan) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) {
        dsa_destroy(ctx);
        return DSA_E_DEVICE;
    }
    if (!share)
        if (const char* mb = getenv("DEFUSE_DSA_SCRATCH_MB")) {
            long v = atol(mb);
            if (v > 0) ctx->scratch_budget = (size_t)v << 20;
        }
    *out = ctx;
    return DSA_OK;
}

int dsa_create(dsa_ctx** out, int device) { return create_ctx(out, device, nullptr); }

void dsa_destroy(dsa_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->d_ref.release(); ctx->d_reads.release(); ctx->d_fusions.release(); ctx->d_pairs.release(); ctx->d_orig.release(); ctx->d_pairs_in.release();
    ctx->d_min_score.release(); ctx->d_records.release(); ctx->d_refcodes.release();
    ctx->plan_runs.release(); ctx->plan_key.release(); ctx->plan_key_sorted.release(); ctx->plan_fidx.release(); ctx->plan_order.release();
    ctx->plan_bsum.release(); ctx->plan_start.release(); ctx->plan_rank.release(); ctx->plan_bound.release(); ctx->plan_flip.release();
    ctx->plan_sort_tmp.release(); ctx->plan_glob.release();
    ctx->d_long.release(); ctx->d_long_fusions.release(); ctx->d_long_work.release(); ctx->d_long_rows.release(); ctx->d_long_state.release();
    ctx->d_long_bits.release();
    for (auto& e : ctx->ev_pack)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_plan)
        if (e) (void)hipEventDestroy(e);
    ctx->lanes.reset();          // the last context of a shared set frees the lanes (streams, events, scratch planes)
    delete ctx;
}

int dsa_share_scratch(dsa_ctx* ctx, dsa_ctx* donor)
{
    if (!ctx || !donor || ctx == donor) return DSA_E_ARG;
    if (ctx->device != donor->device) return fail(ctx, DSA_E_ARG, "dsa_share_scratch: contexts on different devices");
    HIPC(hipSetDevice(ctx->device));
    HIPC(hipDeviceSynchronize());
    ctx->lanes = donor->lanes;
    ctx->scratch_budget = donor->scratch_budget;
    return DSA_OK;
}

int dsa_get_limits(const dsa_ctx*, dsa_limits* out)
{
    if (!out) return DSA_E_ARG;
    // the 16-bit tile kernels take reads up to FAST_MAX_READ and windows up to FAST_MAX_REF; anything longer is swept in 32 bits
    // (dsa_long.hpp), about three orders of magnitude slower per cell
    out->max_read_len = LONG_MAX_READ;
    out->max_ref_len = LONG_MAX_REF;
    out->tile_cols = W;
    return DSA_OK;
}

const char* dsa_last_error(const dsa_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int dsa_set_stream(dsa_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DSA_E_ARG;
    ctx->user_stream = (hipStream_t)hip_stream;
    ctx->lanes->lane[0].stream = hip_stream ? (hipStream_t)hip_stream : ctx->lanes->own[0];      // lane 1 keeps its private stream
    return DSA_OK;
}

int dsa_synchronize(dsa_ctx* ctx)
{
    if (!ctx) return DSA_E_ARG;
    HIPC(hipStreamSynchronize(ctx->main_stream()));
    return DSA_OK;
}

}  // extern "C"

namespace {

// validation, buffers, copies (queued on st) and slice geometry of an upload; the planning is queued by the caller
int upload_enqueue(dsa_ctx* ctx, hipStream_t st, const uint8_t* ref_bytes, int64_t ref_bytes_len, const dsa_fusion* fusions,
               int32_t n_fusions, const uint8_t* read_bytes, int64_t read_bytes_len, const dsa_pair* pairs,
               int64_t n_pairs)
{
    if (!ctx) return DSA_E_ARG;
    if (n_fusions < 0 || n_pairs < 0 || ref_bytes_len < 0 || read_bytes_len < 0)
        return fail(ctx, DSA_E_ARG, "negative size");
    if (n_pairs >= ((int64_t)1 << 31)) return fail(ctx, DSA_E_LIMIT, "more than 2^31-1 pairs in one batch");
    if ((n_fusions && !fusions) || (n_pairs && !pairs) || (ref_bytes_len && !ref_bytes) || (read_bytes_len && !read_bytes))
        return fail(ctx, DSA_E_ARG, "null pointer with non-zero size");
    // A pair goes to the 16-bit tile kernels unless its read or one of its fusion's windows is too long for them; those few
    // are swept in 32 bits by kernels of their own (dsa_long.hpp), and the regular path sees them as empty reads / windows.
    ctx->h_long.clear();
    ctx->h_long_fusions.clear();
    ctx->long_cells = 0;
    ctx->long_blank_cells = 0;
    int nch_all = 1, maxwin = 0;
    for (int32_t f = 0; f < n_fusions; ++f) {
        const dsa_fusion& fu = fusions[f];
        if (fu.ref0_len < 0 || fu.ref1_len < 0 || fu.ref0_off < 0 || fu.ref1_off < 0 ||
            (int64_t)fu.ref0_off + fu.ref0_len > ref_bytes_len || (int64_t)fu.ref1_off + fu.ref1_len > ref_bytes_len)
            return fail(ctx, DSA_E_ARG, "fusion %d: reference window outside ref_bytes", f);
        if (fu.ref0_len > LONG_MAX_REF || fu.ref1_len > LONG_MAX_REF)
            return fail(ctx, DSA_E_LIMIT, "fusion %d: reference window longer than %d", f, LONG_MAX_REF);
        if (fu.ref0_len > FAST_MAX_REF || fu.ref1_len > FAST_MAX_REF) {
            ctx->h_long_fusions.push_back(f);
            continue;
        }
        nch_all = std::max(nch_all, std::max(cdiv(fu.ref0_len, W), cdiv(fu.ref1_len, W)));
        maxwin = std::max(maxwin, std::max(fu.ref0_len, fu.ref1_len));
    }
    const bool any_long_fusion = !ctx->h_long_fusions.empty();
    int lqmax = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        const dsa_pair& pr = pairs[p];
        if (pr.fusion_idx < 0 || pr.fusion_idx >= n_fusions) return fail(ctx, DSA_E_ARG, "pair %lld: bad fusion_idx", (long long)p);
        if (pr.read_len < 0 || pr.read_off < 0 || (int64_t)pr.read_off + pr.read_len > read_bytes_len)
            return fail(ctx, DSA_E_ARG, "pair %lld: read outside read_bytes", (long long)p);
        if (pr.read_len > LONG_MAX_READ) return fail(ctx, DSA_E_LIMIT, "pair %lld: read longer than %d", (long long)p, LONG_MAX_READ);
        if (pr.read_len > FAST_MAX_READ ||
            (any_long_fusion && (fusions[pr.fusion_idx].ref0_len > FAST_MAX_REF || fusions[pr.fusion_idx].ref1_len > FAST_MAX_REF))) {
            const dsa_fusion& fu = fusions[pr.fusion_idx];
            LongDesc d;
            d.pair_idx = (int32_t)p;
            d.read_off = pr.read_off;
            d.read_len = pr.read_len;
            d.ref0_off = fu.ref0_off;
            d.ref0_len = fu.ref0_len;
            d.ref1_off = fu.ref1_off;
            d.ref1_len = fu.ref1_len;
            d.fusion_id = fu.fusion_id;
            d.frag = pr.frag;
            d.read_end = pr.read_end;
            d.revcomp = pr.revcomp;
            d.min_score = min_score_for(pr.read_len);
            ctx->h_long.push_back(d);
            ctx->long_cells += (int64_t)(fu.ref0_len + 1 + fu.ref1_len + 1) * (pr.read_len + 1);
            ctx->long_blank_cells += (fu.ref0_len > FAST_MAX_REF || fu.ref1_len > FAST_MAX_REF) ? 2 : fu.ref0_len + 1 + fu.ref1_len + 1;
            continue;
        }
        lqmax = std::max(lqmax, (int)pr.read_len);
    }
    HIPC(hipSetDevice(ctx->device));
    ctx->have_results = false;
    ctx->n_records = 0;
    ctx->upload_serial = ++g_upload_serial;
    ctx->n_pairs = n_pairs;
    ctx->n_fusions = n_fusions;
    ctx->ref_bytes_len = ref_bytes_len;
    ctx->read_bytes_len = read_bytes_len;
    HIPC(ctx->d_ref.reserve((size_t)ref_bytes_len + 64));        // (whole dwords, as for the reads)
    HIPC(ctx->d_reads.reserve((size_t)read_bytes_len + 64));      // the planning kernels read whole dwords up to 36 bytes past a read's end
    HIPC(ctx->d_fusions.reserve((size_t)n_fusions + 1));
    HIPC(ctx->d_pairs_in.reserve((size_t)n_pairs + 1));
    ctx->nch_all = nch_all;
    HIPC(ctx->d_refcodes.reserve((size_t)n_fusions * nch_all * W + 1));
    if (ref_bytes_len) HIPC(hipMemcpyAsync(ctx->d_ref.p, ref_bytes, ref_bytes_len, hipMemcpyHostToDevice, st));
    if (read_bytes_len) HIPC(hipMemcpyAsync(ctx->d_reads.p, read_bytes, read_bytes_len, hipMemcpyHostToDevice, st));
    if (n_fusions) HIPC(hipMemcpyAsync(ctx->d_fusions.p, fusions, n_fusions * sizeof(dsa_fusion), hipMemcpyHostToDevice, st));
    if (n_pairs) HIPC(hipMemcpyAsync(ctx->d_pairs_in.p, pairs, n_pairs * sizeof(dsa_pair), hipMemcpyHostToDevice, st));
    if (!ctx->h_long.empty() || !ctx->h_long_fusions.empty()) {
        const int nl = (int)ctx->h_long.size(), nlf = (int)ctx->h_long_fusions.size();
        HIPC(ctx->d_long.reserve((size_t)nl + 1));
        HIPC(ctx->d_long_fusions.reserve((size_t)nlf + 1));
        if (nl) HIPC(hipMemcpyAsync(ctx->d_long.p, ctx->h_long.data(), (size_t)nl * sizeof(LongDesc), hipMemcpyHostToDevice, st));
        if (nlf) HIPC(hipMemcpyAsync(ctx->d_long_fusions.p, ctx->h_long_fusions.data(), (size_t)nlf * sizeof(int32_t), hipMemcpyHostToDevice, st));
        const int n = std::max(nl, nlf);
        hipLaunchKernelGGL(k_long_blank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ctx->d_pairs_in.p, ctx->d_long.p, nl, ctx->d_fusions.p,
                           ctx->d_long_fusions.p, nlf);
    }
    std::vector<int32_t>& tab = ctx->h_min_score;       // lives as long as the copy may be in flight
    tab.resize(lqmax + 1);
    for (int l = 0; l <= lqmax; ++l) tab[l] = min_score_for(l);
    HIPC(ctx->d_min_score.reserve(tab.size()));
    HIPC(hipMemcpyAsync(ctx->d_min_score.p, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    // geometry of the planning kernels: hash slots and packed words for the longest window that is planned at all
    {
        const int mw = std::min(maxwin, PLAN_MAXWIN - 1);
        int slots = 256;
        while (slots < 2 * mw) slots <<= 1;
        ctx->plan_prm = PlanParams{};
        ctx->plan_prm.n_fusions = n_fusions;
        ctx->plan_prm.slots = slots;
        ctx->plan_prm.wc = (mw + 2 * PLAN_PAD + 47) / 16 + 1;
        ctx->plan_prm.tile_cols = W;
    }
    make_slices(ctx, n_pairs, lqmax);
    HIPC(hipGetLastError());
    return DSA_OK;
}


}  // namespace

extern "C" {

int dsa_upload(dsa_ctx* ctx, const uint8_t* ref_bytes, int64_t ref_bytes_len, const dsa_fusion* fusions,
               int32_t n_fusions, const uint8_t* read_bytes, int64_t read_bytes_len, const dsa_pair* pairs,
               int64_t n_pairs)
{
    if (!ctx) return DSA_E_ARG;
    if (int rc = upload_enqueue(ctx, ctx->main_stream(), ref_bytes, ref_bytes_len, fusions, n_fusions, read_bytes, read_bytes_len, pairs, n_pairs)) return rc;
    if (int rc = enqueue_plan(ctx)) return rc;
    HIPC(hipStreamSynchronize(ctx->main_stream()));    // the caller's buffers are free again when dsa_upload returns
    HIPC(hipGetLastError());
    return DSA_OK;
}

// The sweep planning of the resident upload once more (everything dsa_upload does after its copies), queued on the
// context's stream; dsa_run follows it in stream order.
int dsa_plan(dsa_ctx* ctx)
{
    if (!ctx) return DSA_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    ctx->have_results = false;
    return enqueue_plan(ctx);
}

}  // extern "C"

namespace {

// Both lanes idle and without a slice: the state every dsa_run starts from and every failure leaves behind.  Lanes may be
// shared by several contexts (dsa_share_scratch): a run that failed half way must not leave a slice "in flight" that the next
// run — of this or of another context — would then finish with its own slice table.
void reset_lanes(dsa_ctx* ctx)
{
    for (PipeLane& L : ctx->lanes->lane) {
        (void)hipStreamSynchronize(L.stream);
        (void)hipStreamSynchronize(L.aux);
        L.slice = -1;
        L.emit_pending = false;
        L.emit_early = false;
    }
    (void)hipGetLastError();
}

// The two sweeps of the long pairs of the upload (dsa_long.hpp), on lane 0's stream in front of the slices: row maxima and
// kept splits, then — the host sizes the bitmaps in between, this path is rare — the columns of the kept rows.
int run_long(dsa_ctx* ctx)
{
    const int nl = (int)ctx->h_long.size();
    if (nl == 0) return DSA_OK;
    hipStream_t st = ctx->lanes->lane[0].stream;
    std::vector<LongState>& hs = ctx->h_long_state;
    hs.assign((size_t)nl, LongState{});
    int64_t work = 0, rows_max = 0;
    for (int k = 0; k < nl; ++k) {
        const LongDesc& d = ctx->h_long[(size_t)k];
        hs[(size_t)k].work_off = work;
        work += long_work_words(d.read_len);
        rows_max = std::max(rows_max, long_rows_words(d.ref0_len, d.ref1_len));
    }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nl, 512), ((int64_t)1 << 28) / std::max<int64_t>(rows_max, 1)));   // at most 1 GiB of row buffers
    HIPC(ctx->d_long_work.reserve((size_t)work + 1));
    HIPC(ctx->d_long_rows.reserve((size_t)(rows_max * grid) + 1));
    HIPC(ctx->d_long_state.reserve((size_t)nl));
    HIPC(hipMemcpyAsync(ctx->d_long_state.p, hs.data(), (size_t)nl * sizeof(LongState), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_long_rows, dim3((unsigned)grid), dim3(LONG_THREADS), 0, st, ctx->d_long.p, ctx->d_long_state.p, nl, ctx->d_ref.p, ctx->d_reads.p,
                       ctx->d_long_work.p, ctx->d_long_rows.p, rows_max);
    HIPC(hipMemcpyAsync(hs.data(), ctx->d_long_state.p, (size_t)nl * sizeof(LongState), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    HIPC(hipGetLastError());
    int64_t bits = 0;
    for (int k = 0; k < nl; ++k) {
        const LongDesc& d = ctx->h_long[(size_t)k];
        hs[(size_t)k].bits_off = bits;
        bits += (int64_t)hs[(size_t)k].n_kept * (long_bitmap_words(d.ref0_len) + long_bitmap_words(d.ref1_len));
    }
    HIPC(ctx->d_long_bits.reserve((size_t)bits + 1));
    HIPC(hipMemcpyAsync(ctx->d_long_state.p, hs.data(), (size_t)nl * sizeof(LongState), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_long_cols, dim3((unsigned)grid), dim3(LONG_THREADS), 0, st, ctx->d_long.p, ctx->d_long_state.p, nl, ctx->d_ref.p, ctx->d_reads.p,
                       ctx->d_long_work.p, ctx->d_long_rows.p, rows_max, ctx->d_long_bits.p);
    HIPC(hipGetLastError());
    ctx->timing.cells += ctx->long_cells - ctx->long_blank_cells;
    return DSA_OK;
}

int run_slices(dsa_ctx* ctx)
{
    if (int rc = run_long(ctx)) return rc;
    // two slices in flight: phase 1 of slice k+1 is queued before the host waits for slice k
    const int ns = (int)ctx->slices.size();
    if (ns > 0 && ctx->n_fusions > 0) {
        // reference bytes -> 16-bit codes for the whole upload, once per run, on lane 0's stream; lane 1 waits for it
        PipeLane& L0 = ctx->lanes->lane[0];
        Geom g = ctx->slices[0].g;
        const int64_t total = (int64_t)g.n_fusions * g.lrp;
        HIPC(hipEventRecord(ctx->ev_pack[0], L0.stream));
        hipLaunchKernelGGL(k_pack_refs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, L0.stream, ctx->d_ref.p, ctx->d_fusions.p,
                           ctx->d_refcodes.p, g);
        if (ns > 1) {       // lane 1 waits for the codes; a run of one slice times the pack up to the start of its fill instead
            HIPC(hipEventRecord(ctx->ev_pack[1], L0.stream));
            HIPC(hipStreamWaitEvent(ctx->lanes->lane[1].stream, ctx->ev_pack[1], 0));
        }
    }
    // room for two records per candidate before the first run, so that its first slice can write its records early too
    if (ns > 0 && ctx->d_records.cap == 0) HIPC(ctx->d_records.reserve((size_t)ctx->n_pairs * 2 + 1024));
    for (int k = 0; k < ns; ++k) {
        PipeLane& L = ctx->lanes->lane[k & 1];
        if (L.slice >= 0)
            if (int rc = phase2(ctx, L)) return rc;
        if (int rc = phase1(ctx, L, k)) return rc;
    }
    for (int k = ns; k < ns + 2; ++k) {
        PipeLane& L = ctx->lanes->lane[k & 1];
        if (L.slice >= 0)
            if (int rc = phase2(ctx, L)) return rc;
    }
    for (PipeLane& L : ctx->lanes->lane) {
        account_emit(ctx, L);
        HIPC(hipStreamSynchronize(L.stream));
    }
    if (ns > 0 && ctx->n_fusions > 0) ctx->timing.pack_ms += elapsed(ctx->ev_pack[0], ns > 1 ? ctx->ev_pack[1] : ctx->lanes->lane[0].ev[1]);
    return DSA_OK;
}

}  // namespace

extern "C" {

int dsa_run(dsa_ctx* ctx, int64_t* out_n)
{
    if (!ctx) return DSA_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    ctx->n_records = 0;
    ctx->have_results = false;
    ctx->timing = dsa_timing{};
    for (const PipeLane& L : ctx->lanes->lane)
        if (L.slice >= 0 || L.emit_pending) {       // left behind by a run that did not end (another context's, on shared lanes)
            reset_lanes(ctx);
            break;
        }
    const auto t0 = std::chrono::steady_clock::now();
    if (const char* inj = getenv("DEFUSE_DSA_TEST_FAIL_RUN")) {   // tests: the N-th run of the process stops with its LAST slice in flight
        static int countdown = atoi(inj);                         // and leaves the lanes as they are (no clean-up at all)
        if (countdown > 0 && --countdown == 0 && !ctx->slices.empty()) {
            (void)phase1(ctx, ctx->lanes->lane[0], (int)ctx->slices.size() - 1);
            return fail(ctx, DSA_E_DEVICE, "injected failure (DEFUSE_DSA_TEST_FAIL_RUN)");
        }
    }
    if (int rc = run_slices(ctx)) {
        reset_lanes(ctx);                           // the error text of the failure stays in ctx->err
        return rc;
    }
    if (ctx->plan_timed) {                    // the planning that preceded this run, on the same stream
        ctx->last_plan_ms = elapsed(ctx->ev_plan[0], ctx->ev_plan[1]);
        ctx->plan_timed = false;
    }
    ctx->timing.plan_ms = ctx->last_plan_ms;
    // stage times are per-stream sums and overlap between the lanes; total_ms is the elapsed time
    ctx->timing.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->timing.n_records = ctx->n_records;
    ctx->have_results = true;
    if (out_n) *out_n = ctx->n_records;
    return DSA_OK;
}

int dsa_download(dsa_ctx* ctx, dsa_record* out, int64_t out_cap, int64_t* out_n)
{
    if (!ctx || !ctx->have_results) return fail(ctx, DSA_E_ARG, "dsa_download before dsa_run");
    if (out_n) *out_n = ctx->n_records;
    if (ctx->n_records > out_cap) return fail(ctx, DSA_E_CAPACITY, "need room for %lld records", (long long)ctx->n_records);
    if (ctx->n_records) {
        if (!out) return fail(ctx, DSA_E_ARG, "null output");
        HIPC(hipMemcpyAsync(out, ctx->d_records.p, ctx->n_records * sizeof(dsa_record), hipMemcpyDeviceToHost, ctx->main_stream()));
        HIPC(hipStreamSynchronize(ctx->main_stream()));
    }
    return DSA_OK;
}

int dsa_copy_records_device(dsa_ctx* ctx, void* out_device, int64_t out_cap, int64_t* out_n)
{
    if (!ctx || !ctx->have_results) return fail(ctx, DSA_E_ARG, "dsa_copy_records_device before dsa_run");
    if (out_n) *out_n = ctx->n_records;
    if (ctx->n_records > out_cap) return fail(ctx, DSA_E_CAPACITY, "need room for %lld records", (long long)ctx->n_records);
    if (ctx->n_records) {
        if (!out_device) return fail(ctx, DSA_E_ARG, "null output");
        HIPC(hipMemcpyAsync(out_device, ctx->d_records.p, ctx->n_records * sizeof(dsa_record), hipMemcpyDeviceToDevice, ctx->main_stream()));
        HIPC(hipStreamSynchronize(ctx->main_stream()));
    }
    return DSA_OK;
}

int dsa_get_timing(const dsa_ctx* ctx, dsa_timing* out)
{
    if (!ctx || !out) return DSA_E_ARG;
    *out = ctx->timing;
    return DSA_OK;
}

int dsa_align_batch(dsa_ctx* ctx, const uint8_t* ref_bytes, int64_t ref_bytes_len, const dsa_fusion* fusions,
                    int32_t n_fusions, const uint8_t* read_bytes, int64_t read_bytes_len, const dsa_pair* pairs,
                    int64_t n_pairs, dsa_record* out, int64_t out_cap, int64_t* out_n)
{
    if (int rc = dsa_upload(ctx, ref_bytes, ref_bytes_len, fusions, n_fusions, read_bytes, read_bytes_len, pairs, n_pairs))
        return rc;
    int64_t n = 0;
    if (int rc = dsa_run(ctx, &n)) return rc;
    return dsa_download(ctx, out, out_cap, out_n);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Streaming: batches submitted one after the other, up to `depth` in flight, collected in order.  Every batch lives in a
// context of its own (input buffers, plan, records), all of which share one set of pipeline lanes; the copies in run on
// one HIP stream (queued by the submitting thread), plan + run on the lanes (a worker thread, one batch after the other),
// the copies out on a third stream — so H2D(k+1), plan / run(k) and D2H(k-1) overlap.
// ---------------------------------------------------------------------------------------------------------------------
struct dsa_stream {
    int device = -1, depth = 0;
    std::vector<dsa_ctx*> slot;
    hipStream_t s_in = nullptr, s_out = nullptr;
    struct Job {
        hipEvent_t ev_in = nullptr, ev_out = nullptr;
        dsa_record* out = nullptr;
        int64_t out_cap = 0, n = 0;
        int rc = 0;
        bool copied = false;
        std::string err;
    };
    std::vector<Job> job;
    uint64_t n_submitted = 0, n_run = 0, n_collected = 0;
    bool stop = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread worker;
    std::string err;
};

namespace {

void stream_worker(dsa_stream* s)
{
    (void)hipSetDevice(s->device);
    for (;;) {
        uint64_t k;
        {
            std::unique_lock<std::mutex> lk(s->m);
            s->cv.wait(lk, [&] { return s->stop || s->n_run < s->n_submitted; });
            if (s->n_run >= s->n_submitted) return;       // stop, nothing left
            k = s->n_run;
        }
        dsa_ctx* ctx = s->slot[k % s->depth];
        dsa_stream::Job& j = s->job[k % s->depth];
        int64_t n = 0;
        static const bool trace = [] { const char* e = getenv("DEFUSE_DSA_STREAM_TRACE"); return e && atoi(e) != 0; }();
        const auto t0 = std::chrono::steady_clock::now();
        int rc = hipStreamWaitEvent(ctx->main_stream(), j.ev_in, 0) == hipSuccess ? DSA_OK : DSA_E_DEVICE;
        if (rc == DSA_OK) rc = enqueue_plan(ctx);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == DSA_OK) rc = dsa_run(ctx, &n);
        const auto t2 = std::chrono::steady_clock::now();
        j.copied = false;
        if (rc == DSA_OK && n <= j.out_cap) {
            if (n && hipMemcpyAsync(j.out, ctx->d_records.p, (size_t)n * sizeof(dsa_record), hipMemcpyDeviceToHost, s->s_out) != hipSuccess) rc = DSA_E_DEVICE;
            if (hipEventRecord(j.ev_out, s->s_out) != hipSuccess) rc = DSA_E_DEVICE;
            j.copied = rc == DSA_OK;
        }
        if (trace) {
            const auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[dsa_stream] batch %llu: %lld pairs, plan queued %.2f ms, run %.2f ms (device: plan %.2f pack %.2f fill %.2f finish %.2f; %d fill launches), "
                            "records copy issued %.2f ms, %lld records\n", (unsigned long long)k, (long long)ctx->n_pairs, ms(t0, t1), ms(t1, t2), ctx->timing.plan_ms,
                    ctx->timing.pack_ms, ctx->timing.fill_ms, ctx->timing.finish_ms, ctx->timing.fill_launches, ms(t2, t3), (long long)n);
        }
        j.n = n;
        j.rc = rc;
        j.err = rc ? ctx->err : std::string();
        {
            std::lock_guard<std::mutex> lk(s->m);
            ++s->n_run;
        }
        s->cv.notify_all();
    }
}

}  // namespace

extern "C" {

int dsa_stream_create(dsa_stream** out, int device, int depth)
{
    if (!out || depth < 1 || depth > 8) return DSA_E_ARG;
    *out = nullptr;
    dsa_stream* s = new dsa_stream();
    s->device = device;
    s->depth = depth;
    s->job.resize(depth);
    bool ok = true;
    for (int k = 0; k < depth && ok; ++k) {
        dsa_ctx* c = nullptr;
        ok = create_ctx(&c, device, k > 0 ? s->slot[0] : nullptr) == DSA_OK;       // the slots share slot 0's pipeline lanes
        if (ok) s->slot.push_back(c);
    }
    ok = ok && hipStreamCreateWithFlags(&s->s_in, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&s->s_out, hipStreamNonBlocking) == hipSuccess;
    for (auto& j : s->job) {
        ok = ok && hipEventCreateWithFlags(&j.ev_in, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&j.ev_out, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        dsa_stream_destroy(s);
        return DSA_E_DEVICE;
    }
    s->worker = std::thread(stream_worker, s);
    *out = s;
    return DSA_OK;
}

void dsa_stream_destroy(dsa_stream* s)
{
    if (!s) return;
    if (s->worker.joinable()) {
        {
            std::lock_guard<std::mutex> lk(s->m);
            s->stop = true;
        }
        s->cv.notify_all();
        s->worker.join();                       // finishes what was submitted
    }
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (auto& j : s->job) {
        if (j.ev_in) (void)hipEventDestroy(j.ev_in);
        if (j.ev_out) (void)hipEventDestroy(j.ev_out);
    }
    if (s->s_in) (void)hipStreamDestroy(s->s_in);
    if (s->s_out) (void)hipStreamDestroy(s->s_out);
    for (size_t k = s->slot.size(); k-- > 0;) dsa_destroy(s->slot[k]);
    delete s;
}

const char* dsa_stream_last_error(const dsa_stream* s) { return s ? s->err.c_str() : "no stream"; }

int dsa_stream_submit(dsa_stream* s, const uint8_t* ref_bytes, int64_t ref_bytes_len, const dsa_fusion* fusions, int32_t n_fusions,
                      const uint8_t* read_bytes, int64_t read_bytes_len, const dsa_pair* pairs, int64_t n_pairs, dsa_record* out,
                      int64_t out_cap)
{
    if (!s || out_cap < 0 || (out_cap && !out)) return DSA_E_ARG;
    uint64_t k;
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->n_submitted - s->n_collected >= (uint64_t)s->depth) {
            s->err = "dsa_stream_submit: all slots in flight, collect first";
            return DSA_E_BUSY;
        }
        k = s->n_submitted;
    }
    dsa_ctx* ctx = s->slot[k % s->depth];
    dsa_stream::Job& j = s->job[k % s->depth];
    if (hipSetDevice(s->device) != hipSuccess) return DSA_E_DEVICE;
    if (int rc = upload_enqueue(ctx, s->s_in, ref_bytes, ref_bytes_len, fusions, n_fusions, read_bytes, read_bytes_len, pairs, n_pairs)) {
        s->err = ctx->err;
        return rc;
    }
    if (hipEventRecord(j.ev_in, s->s_in) != hipSuccess) return DSA_E_DEVICE;
    j.out = out;
    j.out_cap = out_cap;
    {
        std::lock_guard<std::mutex> lk(s->m);
        ++s->n_submitted;
    }
    s->cv.notify_all();
    return DSA_OK;
}

int dsa_stream_collect(dsa_stream* s, int64_t* out_n)
{
    if (!s) return DSA_E_ARG;
    uint64_t k;
    {
        std::unique_lock<std::mutex> lk(s->m);
        if (s->n_collected >= s->n_submitted) {
            s->err = "dsa_stream_collect: nothing submitted";
            return DSA_E_ARG;
        }
        k = s->n_collected;
        s->cv.wait(lk, [&] { return s->n_run > k; });
    }
    dsa_stream::Job& j = s->job[k % s->depth];
    if (out_n) *out_n = j.n;
    int rc = j.rc;
    if (rc == DSA_OK && !j.copied) {             // the records did not fit: the batch stays the oldest until dsa_stream_recollect took them
        s->err = "dsa_stream_collect: out_cap too small";
        return DSA_E_CAPACITY;
    }
    if (rc == DSA_OK && hipEventSynchronize(j.ev_out) != hipSuccess) rc = DSA_E_DEVICE;
    if (rc != DSA_OK) s->err = j.err.empty() ? "device error" : j.err;
    {
        std::lock_guard<std::mutex> lk(s->m);
        ++s->n_collected;
    }
    return rc;
}

int dsa_stream_recollect(dsa_stream* s, dsa_record* out, int64_t out_cap, int64_t* out_n)
{
    if (!s) return DSA_E_ARG;
    uint64_t k;
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->n_collected >= s->n_submitted || s->n_run <= s->n_collected) return DSA_E_ARG;
        k = s->n_collected;
    }
    dsa_ctx* ctx = s->slot[k % s->depth];
    const int rc = dsa_download(ctx, out, out_cap, out_n);      // the worker is not touching this slot: its run is over
    if (rc == DSA_E_CAPACITY) return rc;
    if (rc != DSA_OK) s->err = ctx->err;
    {
        std::lock_guard<std::mutex> lk(s->m);
        ++s->n_collected;
    }
    return rc;
}

void* dsa_host_alloc(size_t bytes)
{
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void dsa_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int dsa_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return DSA_E_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? DSA_OK : DSA_E_DEVICE;
}

int dsa_host_unregister(void* p)
{
    if (!p) return DSA_E_ARG;
    return hipHostUnregister(p) == hipSuccess ? DSA_OK : DSA_E_DEVICE;
}

}  // extern "C"
