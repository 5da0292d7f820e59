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
#include <memory>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "dsa_kernels.hpp"

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

struct HostResult {            // pinned: filled by async copies at the end of a slice's first phase
    Counters ctr;
    int64_t n_rec;
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
    bool copied_descriptors = false;   // phase 1 of the slice in flight uploaded its descriptors (ev[0]..ev[4] time that)
    DevBuf<WaveInfo> d_waves;
    DevBuf<WgInfo> d_wgs;
    DevBuf<uint32_t> d_wg_generic;
    DevBuf<uint32_t> d_rowcodes, d_bnd, d_cmax, d_rmax, d_tmask;
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
        d_waves.release(); d_wgs.release(); d_wg_generic.release(); d_rowcodes.release();
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

struct Slice {
    int64_t pair_begin = 0, pair_end = 0;
    std::vector<WaveInfo> waves;     // one per 64 pairs
    std::vector<WgInfo> wgs;         // one per 256 pairs
    std::vector<uint32_t> wg_flags;  // initial generic flag (1 = more than GSPLIT2 fusions)
    Geom g{};
    // workgroups per table tier of k_fill_fast (an instantiation without workgroups is not launched)
    void tiers(int64_t (&n)[3]) const
    {
        n[0] = n[1] = n[2] = 0;
        for (size_t k = 0; k < wgs.size(); ++k)
            if (!wg_flags[k]) ++n[tier_of(wgs[k].n_groups)];
    }
};

}  // namespace

struct dsa_ctx {
    int device = -1;
    hipStream_t stream = nullptr;    // = lane 0's stream unless dsa_set_stream gave another
    hipStream_t user_stream = nullptr;
    std::string err;
    size_t scratch_budget = (size_t)16 << 30;

    // resident batch
    int64_t n_pairs = 0, ref_bytes_len = 0, read_bytes_len = 0;
    int32_t n_fusions = 0;
    DevBuf<uint8_t> d_ref, d_reads;
    DevBuf<dsa_fusion> d_fusions;
    int64_t upload_serial = 0;       // serial of the resident upload (unique over all contexts)
    int nch_all = 1;                 // tiles of the widest window of the upload: refcodes stride = nch_all * W
    DevBuf<uint32_t> d_refcodes;     // packed once per run for the whole upload
    hipEvent_t ev_pack[2] = {};      // around k_pack_refs
    DevBuf<int32_t> d_orig;          // sweep order -> caller's pair index (Geom::orig), when pairs were reordered
    DevBuf<dsa_pair> d_pairs_sweep;  // second pair buffer: the permutation is written here, then the two are swapped
    DevBuf<dsa_pair> d_pairs;
    DevBuf<int32_t> d_min_score;
    DevBuf<FusionStat> plan_stat;    // sweep planning (plan_sweep): per-fusion statistics, probe votes, tiles / flips, starts, ranks
    DevBuf<int32_t> plan_votes, plan_start, plan_rank;
    DevBuf<uint8_t> plan_tiles;
    std::vector<Slice> slices;
    int64_t total_cells = 0;

    // per-slice scratch lives in two pipeline lanes so that the latency-bound finish stage of one
    // slice overlaps the fill of the next (separate HIP streams)
    std::shared_ptr<LaneSet> lanes;
    DevBuf<dsa_record> d_records;
    int64_t n_records = 0;
    bool have_results = false;

    dsa_timing timing{};
};

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
    return 3 * rows + 2 * rows * (size_t)nch;
}

// Slices bound the scratch footprint; inside a slice pair p lives in wave p/64, lane p%64.
// Per wave the loop bounds, per workgroup (256 pairs) the distinct fusions for the fast path.
int build_slices(dsa_ctx* ctx, const dsa_fusion* fusions, const dsa_pair* pairs, int64_t n_pairs);

// Sweep plan (speed only; records always come out in the caller's pair order).  Workgroups are 256
// consecutive pairs, waves 64: fusions with many reads go first (table tiers), and inside a size class
// fusions whose alignments end in the same tiles (device probe) sit next to each other, so the lanes of a
// wave that straddles two fusions are alive in the same tiles.  Everything that touches pairs runs on the
// device (statistics per fusion, probe, permutation); the host sorts the fusions and derives the wave and
// workgroup descriptors from the runs.  A batch larger than the scratch budget is cut into contiguous ranges
// of the caller's order, each planned on its own (records of a slice follow those of the slices before it, so
// the offsets of a reordered slice only need the scan over that slice).  Needs the pairs of every fusion to
// be one run inside a slice; otherwise, and with DEFUSE_DSA_NO_REORDER=1, the caller's order is swept.
// Returns 1 if the plan was made, 0 if not.
// plans pairs [begin, end) of the caller's order as one slice; 0 = some fusion is not one run in it
int plan_chunk(dsa_ctx* ctx, const dsa_fusion* fusions, int64_t begin, int64_t end, int lq1, DevBuf<FusionStat>& d_stat,
               DevBuf<int32_t>& d_votes, DevBuf<uint8_t>& d_tiles, DevBuf<int32_t>& d_start, DevBuf<int32_t>& d_rank, Slice& cur)
{
    const int nf = ctx->n_fusions;
    const int64_t n = end - begin;
    hipStream_t st = ctx->stream;
    const dsa_pair* pairs = ctx->d_pairs.p + begin;
    std::vector<FusionStat> stat((size_t)nf);
    std::vector<uint8_t> tiles((size_t)2 * nf);
    HIPC(hipMemsetAsync(d_votes.p, 0, (size_t)nf * 2 * PROBE_TILES * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_fusion_stats_init, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, d_stat.p, nf);
    hipLaunchKernelGGL(k_fusion_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pairs, n, d_stat.p);
    hipLaunchKernelGGL(k_probe_wave, dim3((unsigned)nf * PROBE_READS), dim3(WAVE), 0, st, ctx->d_ref.p, ctx->d_fusions.p, ctx->d_reads.p,
                       pairs, d_stat.p, d_votes.p);
    hipLaunchKernelGGL(k_probe_pick, dim3((unsigned)((2 * nf + 255) / 256)), dim3(256), 0, st, d_votes.p, 2 * nf, d_tiles.p);
    HIPC(hipMemcpyAsync(stat.data(), d_stat.p, (size_t)nf * sizeof(FusionStat), hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(tiles.data(), d_tiles.p, tiles.size(), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    HIPC(hipGetLastError());
    for (int f = 0; f < nf; ++f)
        if (stat[f].count > 0 && stat[f].last - stat[f].first + 1 != stat[f].count) return 0;     // not one run per fusion

    auto window_tiles = [&](int f) { return std::max(cdiv(fusions[f].ref0_len, W), cdiv(fusions[f].ref1_len, W)); };
    auto size_class = [&](int f) {
        const int c = stat[f].count;
        return c >= WAVE ? 0 : c >= WG_LANES / GSPLIT ? 1 : c >= (WG_LANES + GSPLIT2 - 1) / GSPLIT2 ? 2 : 3;
    };
    std::vector<int32_t> forder;
    for (int f = 0; f < nf; ++f)
        if (stat[f].count > 0) forder.push_back(f);
    // inside a size class the expensive fusions first (longest-processing-time order: the workgroups that are still
    // running when the launch drains are then the cheap ones).  Cost proxy: the number of distinct tiles in which either
    // matrix is alive, {t1-1, t1} and {t2-1, t2} with t1, t2 the probed end tiles — 2 when they coincide, up to 4.
    auto alive_tiles = [&](int f) {
        const int t1 = tiles[2 * f], t2 = tiles[2 * f + 1];
        if (t1 == 255 || t2 == 255) return 4;                  // no vote: assume the worst
        const int d = t1 > t2 ? t1 - t2 : t2 - t1;
        return d == 0 ? 2 : d == 1 ? 3 : 4;
    };
    static const bool lpt = [] { const char* e = getenv("DEFUSE_DSA_NO_LPT"); return !(e && atoi(e) != 0); }();
    std::stable_sort(forder.begin(), forder.end(), [&](int a, int b) {
        const int ca = size_class(a), cb = size_class(b);
        if (ca != cb) return ca < cb;
        if (lpt) {
            const int ka = alive_tiles(a), kb = alive_tiles(b);
            if (ka != kb) return ka > kb;
        }
        return tiles[2 * a] * 256 + tiles[2 * a + 1] < tiles[2 * b] * 256 + tiles[2 * b + 1];
    });

    // geometry and descriptors from the runs
    cur.pair_begin = begin;
    cur.pair_end = end;
    int nch = 1;
    for (int f : forder) nch = std::max(nch, window_tiles(f));
    const int64_t n_waves = (n + WAVE - 1) / WAVE, n_wgs = (n + WG_LANES - 1) / WG_LANES;
    cur.waves.assign((size_t)n_waves, WaveInfo{0, 0});
    cur.wgs.assign((size_t)n_wgs, WgInfo{});
    cur.wg_flags.assign((size_t)n_wgs, 0u);
    std::vector<int32_t> new_start((size_t)nf, 0);
    int64_t pos = 0;
    for (int f : forder) {
        const int64_t c = stat[f].count;
        new_start[f] = (int32_t)pos;
        const int tl = window_tiles(f);
        for (int64_t w = pos / WAVE; w <= (pos + c - 1) / WAVE; ++w) {
            cur.waves[w].lq_max = std::max(cur.waves[w].lq_max, stat[f].max_lq);    // upper bound of the wave's reads
            cur.waves[w].nch_max = std::max(cur.waves[w].nch_max, tl);
        }
        for (int64_t g = pos / WG_LANES; g <= (pos + c - 1) / WG_LANES; ++g) {
            WgInfo& wg = cur.wgs[g];
            if (cur.wg_flags[g]) continue;
            if (wg.n_groups < GSPLIT2)
                wg.group_f[wg.n_groups++] = f;
            else {
                wg.n_groups = 0;
                cur.wg_flags[g] = 1u;
            }
        }
        ctx->total_cells += (int64_t)(fusions[f].ref0_len + 1 + fusions[f].ref1_len + 1) * (stat[f].sum_lq + c);
        pos += c;
    }
    cur.g.n_waves = (int32_t)n_waves;
    cur.g.n_wgs = (int32_t)n_wgs;
    cur.g.lq1 = lq1;
    cur.g.nch = nch;
    cur.g.lrp = ctx->nch_all * W;
    cur.g.n_fusions = nf;
    cur.g.n_pairs = n;
    cur.g.orig = ctx->d_orig.p + begin;        // slice-relative indices of the caller's order

    // pairs into sweep order on the device: the fusions in the order just made, the pairs of a fusion by the estimated
    // read split (k_rank_in_fusion), alternate fusions in opposite directions (DEFUSE_DSA_NO_RANK=1: caller's order inside)
    std::vector<uint8_t> flip((size_t)nf, 0);
    {
        int pos_in_order = 0;
        for (int f : forder) flip[f] = (uint8_t)(pos_in_order++ & 1);
    }
    static const bool no_rank = [] { const char* e = getenv("DEFUSE_DSA_NO_RANK"); return e && atoi(e) != 0; }();
    // DEFUSE_DSA_NO_TIGHTEN=1: no per-pair score bound, pruning against minScore alone (round 1's behaviour)
    static const bool no_tighten = [] { const char* e = getenv("DEFUSE_DSA_NO_TIGHTEN"); return e && atoi(e) != 0; }();
    HIPC(hipMemcpyAsync(d_start.p, new_start.data(), (size_t)nf * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(d_tiles.p, flip.data(), (size_t)nf, hipMemcpyHostToDevice, st));        // the tile votes were read above: the buffer is free
    HIPC(d_rank.reserve((size_t)n));
    if (no_rank)
        hipLaunchKernelGGL(k_rank_identity, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pairs, n, d_stat.p, d_rank.p);
    else
        hipLaunchKernelGGL(k_rank_in_fusion, dim3((unsigned)nf), dim3(RANK_THREADS), 0, st, ctx->d_ref.p, ctx->d_fusions.p, ctx->d_reads.p,
                           ctx->d_pairs.p + begin, d_stat.p, d_tiles.p, d_rank.p, no_tighten ? 0 : 1);
    hipLaunchKernelGGL(k_permute_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pairs, n, d_stat.p, d_start.p, d_rank.p,
                       ctx->d_pairs_sweep.p + begin, ctx->d_orig.p + begin);
    HIPC(hipStreamSynchronize(st));             // new_start is reused by the next chunk
    HIPC(hipGetLastError());
    return 1;
}

int plan_sweep(dsa_ctx* ctx, const dsa_fusion* fusions, int64_t n_pairs, int lqmax)
{
    const int nf = ctx->n_fusions;
    const char* off = getenv("DEFUSE_DSA_NO_REORDER");
    if (nf < 2 || n_pairs < 2 * WG_LANES || (off && atoi(off) != 0)) return 0;
    // slices: contiguous ranges of the caller's order that fit the scratch budget (and DEFUSE_DSA_SLICE_PAIRS)
    int nch_all = 1;
    for (int f = 0; f < nf; ++f) nch_all = std::max(nch_all, std::max(cdiv(fusions[f].ref0_len, W), cdiv(fusions[f].ref1_len, W)));
    const int lq1 = (lqmax + 1 + 3) & ~3;
    const size_t per_wg = slice_scratch_bytes(WG_WAVES, lq1, nch_all);
    int64_t chunk = (int64_t)std::min<size_t>((size_t)1 << 40, ctx->scratch_budget / std::max<size_t>(per_wg, 1)) * WG_LANES;
    // A batch that fits one slice runs as ONE fill launch.  Cutting it in two so that the latency-bound finish kernels of the
    // first half run beside the fill of the second (DEFUSE_DSA_SPLIT_LARGE=1) measured +1 % aligns/s on BASELINE configs[1]
    // (4.04 against 4.08 ms per step) with the two fills overlapping on the two lanes — which blurs the per-launch times the
    // roofline is computed from — and -10 % with the fills serialised (4.52 ms: two launch tails instead of one); four or
    // eight slices lose as well (4.66, 5.61 ms).  So it stays off.
    if (const char* e = getenv("DEFUSE_DSA_SPLIT_LARGE"))
        if (atoi(e) != 0 && chunk >= n_pairs && n_pairs >= ((int64_t)1 << 19)) chunk = ((n_pairs + 1) / 2 + WG_LANES - 1) / WG_LANES * WG_LANES;
    if (const char* e = getenv("DEFUSE_DSA_SLICE_PAIRS")) chunk = std::min<int64_t>(chunk, std::max<int64_t>(WG_LANES, atoll(e) / WG_LANES * WG_LANES));
    if (chunk < 2 * WG_LANES) return 0;
    // planning buffers live in the context: an upload after the first finds them allocated
    DevBuf<FusionStat>& d_stat = ctx->plan_stat;
    DevBuf<int32_t>&d_votes = ctx->plan_votes, &d_start = ctx->plan_start, &d_rank = ctx->plan_rank;
    DevBuf<uint8_t>& d_tiles = ctx->plan_tiles;
    HIPC(d_stat.reserve((size_t)nf));
    HIPC(d_votes.reserve((size_t)nf * 2 * PROBE_TILES));
    HIPC(d_tiles.reserve((size_t)nf * 2));
    HIPC(d_start.reserve((size_t)nf));
    HIPC(ctx->d_orig.reserve((size_t)n_pairs));
    HIPC(ctx->d_pairs_sweep.reserve((size_t)n_pairs + 1));
    HIPC(d_rank.reserve((size_t)std::min<int64_t>(n_pairs, chunk)));
    std::vector<Slice> slices;
    ctx->total_cells = 0;
    for (int64_t b = 0; b < n_pairs; b += chunk) {
        Slice cur;
        const int rc = plan_chunk(ctx, fusions, b, std::min(n_pairs, b + chunk), lq1, d_stat, d_votes, d_tiles, d_start, d_rank, cur);
        if (rc != 1) return rc;                 // the caller's pairs are untouched: the unplanned path takes over
        slices.push_back(std::move(cur));
    }
    std::swap(ctx->d_pairs.p, ctx->d_pairs_sweep.p);
    std::swap(ctx->d_pairs.cap, ctx->d_pairs_sweep.cap);
    ctx->slices = std::move(slices);
    return 1;
}

int build_slices(dsa_ctx* ctx, const dsa_fusion* fusions, const dsa_pair* pairs, int64_t n_pairs)
{
    ctx->slices.clear();
    ctx->total_cells = 0;
    // Slices are cut by the scratch budget only: cutting finer to overlap two slices on the two lanes
    // measured no faster than one big slice (the finish work already hides inside the fill kernel).
    // DEFUSE_DSA_SLICE_PAIRS caps the pairs per slice (tests, experiments).
    int64_t slice_pair_cap = n_pairs + 1;
    if (const char* e = getenv("DEFUSE_DSA_SLICE_PAIRS")) slice_pair_cap = std::max<int64_t>(WG_LANES, atoll(e));
    int64_t p = 0;
    while (p < n_pairs) {
        Slice cur;
        cur.pair_begin = p;
        int lq1 = 1, nch = 1;
        while (p < n_pairs) {
            // one workgroup worth of pairs at a time
            const int64_t e = std::min<int64_t>(n_pairs, p + WG_LANES);
            int nlq1 = lq1, nnch = nch;
            for (int64_t q = p; q < e; ++q) {
                const dsa_fusion& fu = fusions[pairs[q].fusion_idx];
                nlq1 = std::max(nlq1, (int)pairs[q].read_len + 1);
                nnch = std::max(nnch, std::max(cdiv(fu.ref0_len, W), cdiv(fu.ref1_len, W)));
            }
            const int64_t waves_after = (int64_t)cur.waves.size() + cdiv((int)(e - p), WAVE);
            if (!cur.waves.empty() && slice_scratch_bytes(waves_after, (nlq1 + 3) & ~3, nnch) > ctx->scratch_budget) break;
            if (!cur.waves.empty() && (int64_t)cur.waves.size() * WAVE >= slice_pair_cap) break;
            lq1 = nlq1;
            nch = nnch;
            WgInfo wg{};
            bool too_many = false;
            for (int64_t wq = p; wq < e; wq += WAVE) {
                WaveInfo wi{0, 0};
                const int64_t we = std::min<int64_t>(e, wq + WAVE);
                for (int64_t q = wq; q < we; ++q) {
                    const int f = pairs[q].fusion_idx;
                    const dsa_fusion& fu = fusions[f];
                    wi.lq_max = std::max(wi.lq_max, (int)pairs[q].read_len);
                    wi.nch_max = std::max(wi.nch_max, std::max(cdiv(fu.ref0_len, W), cdiv(fu.ref1_len, W)));
                    ctx->total_cells += (int64_t)(fu.ref0_len + 1 + fu.ref1_len + 1) * (pairs[q].read_len + 1);
                    bool found = false;
                    for (int k = 0; k < wg.n_groups; ++k) found |= wg.group_f[k] == f;
                    if (!found) {
                        if (wg.n_groups < GSPLIT2)
                            wg.group_f[wg.n_groups++] = f;
                        else
                            too_many = true;
                    }
                }
                cur.waves.push_back(wi);
            }
            if (too_many) wg.n_groups = 0;
            cur.wgs.push_back(wg);
            cur.wg_flags.push_back(too_many ? 1u : 0u);
            p = e;
        }
        cur.pair_end = p;
        cur.g.n_waves = (int32_t)cur.waves.size();
        cur.g.n_wgs = (int32_t)cur.wgs.size();
        cur.g.lq1 = (lq1 + 3) & ~3;     // row planes are stored four rows per 16-byte word
        cur.g.nch = nch;
        cur.g.lrp = ctx->nch_all * W;
        cur.g.n_fusions = ctx->n_fusions;
        cur.g.n_pairs = cur.pair_end - cur.pair_begin;
        cur.g.orig = nullptr;
        ctx->slices.push_back(std::move(cur));
    }
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
                       (uint64_t)cap_left, (int64_t)s.pair_begin, s.g);
    (void)hipEventRecord(L.ev_join, L.aux);
    hipLaunchKernelGGL(k_emit_counted, dim3((unsigned)((np + EMIT_BLOCK - 1) / EMIT_BLOCK)), dim3(EMIT_BLOCK), 0, L.stream, pairs, ctx->d_fusions.p,
                       L.d_state.p, L.d_kept.p, L.d_tasks.p, L.d_masks.p, (const int64_t*)L.d_rec_offset.p, out, (uint64_t)cap_left,
                       (int64_t)s.pair_begin, L.d_ctr.p, (uint64_t)L.d_kept.cap, (uint64_t)L.d_tasks.cap, (uint64_t)(L.d_masks.cap / 2),
                       (uint64_t)L.d_gtasks.cap, s.g);
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
    // every workgroup is run by exactly one of the fill kernels
    int64_t n_tier[3];
    s.tiers(n_tier);
    if (n_tier[0])
        hipLaunchKernelGGL(k_fill_fast<0>, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, L.d_waves.p, L.d_wgs.p, L.d_wg_generic.p,
                           ctx->d_refcodes.p, ctx->d_reads.p, L.d_rowcodes.p, ctx->d_min_score.p, ctx->d_fusions.p, L.d_bnd.p, L.d_cmax.p, L.d_rmax.p,
                           L.d_tmask.p, fb, g);
    if (n_tier[1])
        hipLaunchKernelGGL(k_fill_fast<1>, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, L.d_waves.p, L.d_wgs.p, L.d_wg_generic.p,
                           ctx->d_refcodes.p, ctx->d_reads.p, L.d_rowcodes.p, ctx->d_min_score.p, ctx->d_fusions.p, L.d_bnd.p, L.d_cmax.p, L.d_rmax.p,
                           L.d_tmask.p, fb, g);
    if (n_tier[2])
        hipLaunchKernelGGL(k_fill_fast<2>, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, L.d_waves.p, L.d_wgs.p, L.d_wg_generic.p,
                           ctx->d_refcodes.p, ctx->d_reads.p, L.d_rowcodes.p, ctx->d_min_score.p, ctx->d_fusions.p, L.d_bnd.p, L.d_cmax.p, L.d_rmax.p,
                           L.d_tmask.p, fb, g);
    hipLaunchKernelGGL(k_fill_generic, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, L.d_waves.p, L.d_wgs.p, ctx->d_fusions.p,
                       L.d_wg_generic.p, ctx->d_refcodes.p, ctx->d_reads.p, L.d_rowcodes.p, ctx->d_min_score.p, L.d_bnd.p, L.d_cmax.p, L.d_rmax.p,
                       L.d_tmask.p, fb, g);
    HIPC(hipEventRecord(L.ev[2], st));
    hipLaunchKernelGGL(k_replay, dim3(2048), dim3(REPLAY_BLOCK), 0, st, L.d_tasks.p, (uint64_t)L.d_tasks.cap, L.d_gtasks.p,
                       (uint64_t)L.d_gtasks.cap, L.d_ctr.p, L.d_state.p, L.d_kept.p, (uint64_t)L.d_kept.cap, pairs, ctx->d_fusions.p,
                       ctx->d_refcodes.p, L.d_rowcodes.p, L.d_bnd.p, L.d_tstop.p, L.d_masks.p, (uint64_t)(L.d_masks.cap / 2), g);
    hipLaunchKernelGGL(k_emit_listed<false>, dim3(listed_grid(L, np)), dim3(EMIT_BLOCK), 0, st, L.d_gtasks.p, (uint64_t)L.d_gtasks.cap, L.d_ctr.p,
                       pairs, ctx->d_fusions.p, L.d_state.p, L.d_kept.p, L.d_tasks.p, (uint64_t)L.d_tasks.cap, L.d_masks.p,
                       (uint64_t)(L.d_masks.cap / 2), (uint64_t)L.d_kept.cap, L.d_rec_count.p, (const int64_t*)nullptr, (dsa_record*)nullptr,
                       (uint64_t)0, (int64_t)s.pair_begin, g);
    if (int rc = exclusive_scan(ctx, L, L.d_rec_count.p, L.d_rec_offset.p, np + 1)) return rc;
    // the cursors and the record total go to the lane's pinned result words
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, L.d_ctr.p, L.d_rec_offset.p + np, &L.host->ctr, &L.host->n_rec);
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
    if (g.lq1 > ((7600 + 1 + 3) & ~3)) return fail(ctx, DSA_E_LIMIT, "reads longer than 7600 are not supported");   // lq1 is padded to a multiple of 4
    const size_t n_rows = (size_t)g.n_waves * g.lq1 * WAVE;
    HIPC(L.d_waves.reserve(s.waves.size()));
    HIPC(L.d_wgs.reserve(s.wgs.size()));
    HIPC(L.d_wg_generic.reserve(s.wg_flags.size()));
    HIPC(L.d_rowcodes.reserve(n_rows));
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
    // The wave / workgroup descriptors are inputs of the batch like the pairs themselves: a batch that is run
    // again finds them on the device (the generic flags the fill kernel may have set for it stay valid too).
    // All packing (reference codes here, row codes inside the fill) is redone by every run.
    L.copied_descriptors = L.resident_upload != ctx->upload_serial || L.resident_slice != slice_idx;
    if (L.copied_descriptors) {
        HIPC(hipEventRecord(L.ev[0], st));
        HIPC(hipMemcpyAsync(L.d_waves.p, s.waves.data(), s.waves.size() * sizeof(WaveInfo), hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(L.d_wgs.p, s.wgs.data(), s.wgs.size() * sizeof(WgInfo), hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(L.d_wg_generic.p, s.wg_flags.data(), s.wg_flags.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        L.resident_upload = ctx->upload_serial;
        L.resident_slice = slice_idx;
        HIPC(hipEventRecord(L.ev[4], st));      // end of the slice's descriptor copies (ev[1] is re-recorded by every launch_compute)
    }
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
        if (c.n_kept <= L.d_kept.cap && c.n_tasks <= L.d_tasks.cap && c.n_masks <= L.d_masks.cap / 2 && c.n_gtasks <= L.d_gtasks.cap) break;
        if (attempt >= 3) return fail(ctx, DSA_E_DEVICE, "finish stage did not converge");
        HIPC(L.d_kept.reserve(c.n_kept + 1024));
        HIPC(L.d_tasks.reserve(c.n_tasks + 1024));
        HIPC(L.d_masks.reserve(2 * c.n_masks + 1024));
        HIPC(L.d_gtasks.reserve(c.n_gtasks + 1024));
        if (int rc = launch_compute(ctx, L, s)) return rc;
    }
#ifdef DSA_PRUNE_STATS
    {
        // the generic replay's tasks by last row and sidedness
        const size_t ng = (size_t)L.host->ctr.n_gtasks, nt = (size_t)L.host->ctr.n_tasks;
        std::vector<uint2> gt(ng);
        std::vector<ReplayTask> tk(nt);
        if (ng) HIPC(hipMemcpy(gt.data(), L.d_gtasks.p, ng * sizeof(uint2), hipMemcpyDeviceToHost));
        if (nt) HIPC(hipMemcpy(tk.data(), L.d_tasks.p, nt * sizeof(ReplayTask), hipMemcpyDeviceToHost));
        long hist[3][10] = {};
        for (size_t i = 0; i < ng; ++i) {
            const ReplayTask& t = tk[gt[i].x & ~GTASK_OWNER];
            const int kind = t.chunk0 != NO_CHUNK && t.chunk1 != NO_CHUNK ? 2 : t.chunk0 != NO_CHUNK ? 0 : 1;
            hist[kind][std::min(9, (t.last_row & TASK_ROW) / 8)]++;
        }
        {
            const size_t np_ = (size_t)s.g.n_pairs, nk = (size_t)L.host->ctr.n_kept, nm = (size_t)L.host->ctr.n_masks;
            std::vector<PairState> stv(np_);
            std::vector<KeptRow> kv(nk);
            std::vector<uint64_t> mv(nm * 2);
            HIPC(hipMemcpy(stv.data(), L.d_state.p, np_ * sizeof(PairState), hipMemcpyDeviceToHost));
            if (nk) HIPC(hipMemcpy(kv.data(), L.d_kept.p, nk * sizeof(KeptRow), hipMemcpyDeviceToHost));
            if (nm) HIPC(hipMemcpy(mv.data(), L.d_masks.p, nm * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
            int shown = 0;
            for (size_t i = 0; i < ng && shown < 8; ++i) {
                const ReplayTask& t = tk[gt[i].x & ~GTASK_OWNER];
                const int R = t.last_row & TASK_ROW;
                if (t.chunk1 != NO_CHUNK || R < 24 || R > 56) continue;
                const PairState& st = stv[t.pair];
                fprintf(stderr, "[stats] task pair %u tile0 %d R %d: n_kept %d n_tasks %d tiles0 %x tiles1 %x first %d %d;", t.pair, t.chunk0, R, st.n_kept, st.n_tasks,
                        st.tiles0, st.tiles1, st.first0, st.first1);
                for (int k = 0; k < st.n_kept; ++k)
                    fprintf(stderr, " [a %d m1 %d m2 %d mask0 %llx mask1 %llx]", kv[st.kept_begin + k].a, kv[st.kept_begin + k].m1, kv[st.kept_begin + k].m2,
                            (unsigned long long)mv[((size_t)t.mask_begin + k) * 2], (unsigned long long)mv[((size_t)t.mask_begin + k) * 2 + 1]);
                fprintf(stderr, "\n");
                ++shown;
            }
        }
        for (int k = 0; k < 3; ++k) {
            fprintf(stderr, "[stats] generic tasks, %s, by last row / 8:", k == 0 ? "M1 side only" : k == 1 ? "M2 side only" : "both sides");
            for (int b = 0; b < 10; ++b) fprintf(stderr, " %ld", hist[k][b]);
            fprintf(stderr, "\n");
        }
    }
    {
        unsigned long long h[16];
        HIPC(hipMemcpy(h, L.d_stats.p, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[stats] row groups skipped %llu of %llu; wave cycles: total %llu, at tile barriers %llu, table build %llu, tail %llu (row max %llu, combine %llu, replay %llu); generic replay: %llu waves, setup %llu, sweep %llu cycles, %llu row groups (lane 0); row groups skipped between live parts %llu (dead groups followed at once by a live boundary %llu, of which through the diagonal %llu); slowest lane of the listed count: %llu cycles, %llu kept rows, %llu tasks\n",
                h[0], h[1], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[12], h[10], h[11], h[13], h[14], h[2] & 0xFFFFFFFFull, h[2] >> 32, h[15] >> 24, (h[15] >> 8) & 0xFFFF, h[15] & 0xFF);
    }
#endif
    L.last_gtasks = L.host->ctr.n_gtasks;
    const int64_t n_rec = L.host->n_rec;
    if (L.copied_descriptors) ctx->timing.pack_ms += elapsed(L.ev[0], L.ev[4]);
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
const char* dsa_version(void) { return "defuse_amd dsa 0.2 (gfx950) src " DSA_BUILD_HASH; }

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
    const int start = (int)((unsigned long)getpid() % (unsigned long)n);
    const char* dir = getenv("DEFUSE_GPU_LOCK_DIR");
    if (!dir) dir = "/tmp";
    for (int k = 0; k < n; ++k) {
        const int d = (start + k) % n;
        char path[512];
        snprintf(path, sizeof path, "%s/defuse_gpu.%d.lock", dir, d);
        const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd < 0) continue;
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) return d;        // fd stays open on purpose: it is the claim
        close(fd);
    }
    return start;
}

int dsa_pick_device(void)
{
    if (const char* e = getenv("DEFUSE_GPU")) return atoi(e);
    return dsa_pick_device_among(dsa_device_count());
}

int dsa_set_scratch_budget(dsa_ctx* ctx, int64_t bytes)
{
    if (!ctx || bytes <= 0) return DSA_E_ARG;
    ctx->scratch_budget = (size_t)bytes;
    return DSA_OK;
}

int dsa_create(dsa_ctx** out, int device)
{
    if (!out) return DSA_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DSA_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DSA_E_DEVICE;
    dsa_ctx* ctx = new dsa_ctx();
    ctx->device = device;
    ctx->lanes = std::make_shared<LaneSet>();
    ctx->lanes->device = device;
    bool ok = true;
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
    for (auto& e : ctx->ev_pack) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) {
        dsa_destroy(ctx);
        return DSA_E_DEVICE;
    }
    ctx->stream = ctx->lanes->lane[0].stream;
    if (const char* mb = getenv("DEFUSE_DSA_SCRATCH_MB")) {
        long v = atol(mb);
        if (v > 0) ctx->scratch_budget = (size_t)v << 20;
    }
    *out = ctx;
    return DSA_OK;
}

void dsa_destroy(dsa_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->d_ref.release(); ctx->d_reads.release(); ctx->d_fusions.release(); ctx->d_pairs.release(); ctx->d_orig.release(); ctx->d_pairs_sweep.release();
    ctx->d_min_score.release(); ctx->d_records.release(); ctx->d_refcodes.release();
    ctx->plan_stat.release(); ctx->plan_votes.release(); ctx->plan_start.release(); ctx->plan_rank.release(); ctx->plan_tiles.release();
    for (auto& e : ctx->ev_pack)
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
    ctx->stream = ctx->user_stream ? ctx->user_stream : ctx->lanes->lane[0].stream;
    ctx->scratch_budget = donor->scratch_budget;
    return DSA_OK;
}

int dsa_get_limits(const dsa_ctx*, dsa_limits* out)
{
    if (!out) return DSA_E_ARG;
    out->max_read_len = 7600;       // V + 1024 = H + 2j + 1024 <= 4*Lq + 1024 must stay a finite fp16 pattern (< 0x7C00)
    out->max_ref_len = 255 * W;     // chunk index is 8 bits in ReplayTask
    out->tile_cols = W;
    return DSA_OK;
}

const char* dsa_last_error(const dsa_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int dsa_set_stream(dsa_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DSA_E_ARG;
    ctx->user_stream = (hipStream_t)hip_stream;
    ctx->lanes->lane[0].stream = hip_stream ? (hipStream_t)hip_stream : ctx->lanes->own[0];      // lane 1 keeps its private stream
    ctx->stream = ctx->lanes->lane[0].stream;
    return DSA_OK;
}

int dsa_synchronize(dsa_ctx* ctx)
{
    if (!ctx) return DSA_E_ARG;
    HIPC(hipStreamSynchronize(ctx->stream));
    return DSA_OK;
}

int dsa_upload(dsa_ctx* ctx, const uint8_t* ref_bytes, int64_t ref_bytes_len, const dsa_fusion* fusions,
               int32_t n_fusions, const uint8_t* read_bytes, int64_t read_bytes_len, const dsa_pair* pairs,
               int64_t n_pairs)
{
    if (!ctx) return DSA_E_ARG;
    if (n_fusions < 0 || n_pairs < 0 || ref_bytes_len < 0 || read_bytes_len < 0)
        return fail(ctx, DSA_E_ARG, "negative size");
    if (n_pairs >= ((int64_t)1 << 31)) return fail(ctx, DSA_E_LIMIT, "more than 2^31-1 pairs in one batch");
    if ((n_fusions && !fusions) || (n_pairs && !pairs) || (ref_bytes_len && !ref_bytes) || (read_bytes_len && !read_bytes))
        return fail(ctx, DSA_E_ARG, "null pointer with non-zero size");
    dsa_limits lim;
    dsa_get_limits(ctx, &lim);
    for (int32_t f = 0; f < n_fusions; ++f) {
        const dsa_fusion& fu = fusions[f];
        if (fu.ref0_len < 0 || fu.ref1_len < 0 || fu.ref0_off < 0 || fu.ref1_off < 0 ||
            (int64_t)fu.ref0_off + fu.ref0_len > ref_bytes_len || (int64_t)fu.ref1_off + fu.ref1_len > ref_bytes_len)
            return fail(ctx, DSA_E_ARG, "fusion %d: reference window outside ref_bytes", f);
        if (fu.ref0_len > lim.max_ref_len || fu.ref1_len > lim.max_ref_len)
            return fail(ctx, DSA_E_LIMIT, "fusion %d: reference window longer than %d", f, lim.max_ref_len);
    }
    int nch_all = 1;
    for (int32_t f = 0; f < n_fusions; ++f) nch_all = std::max(nch_all, std::max(cdiv(fusions[f].ref0_len, W), cdiv(fusions[f].ref1_len, W)));
    int lqmax = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        const dsa_pair& pr = pairs[p];
        if (pr.fusion_idx < 0 || pr.fusion_idx >= n_fusions) return fail(ctx, DSA_E_ARG, "pair %lld: bad fusion_idx", (long long)p);
        if (pr.read_len < 0 || pr.read_off < 0 || (int64_t)pr.read_off + pr.read_len > read_bytes_len)
            return fail(ctx, DSA_E_ARG, "pair %lld: read outside read_bytes", (long long)p);
        if (pr.read_len > lim.max_read_len) return fail(ctx, DSA_E_LIMIT, "pair %lld: read longer than %d", (long long)p, lim.max_read_len);
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
    HIPC(ctx->d_ref.reserve((size_t)ref_bytes_len + 1));
    HIPC(ctx->d_reads.reserve((size_t)read_bytes_len + 1));
    HIPC(ctx->d_fusions.reserve((size_t)n_fusions + 1));
    HIPC(ctx->d_pairs.reserve((size_t)n_pairs + 1));
    ctx->nch_all = nch_all;
    HIPC(ctx->d_refcodes.reserve((size_t)n_fusions * nch_all * W + 1));
    hipStream_t st = ctx->stream;
    if (ref_bytes_len) HIPC(hipMemcpyAsync(ctx->d_ref.p, ref_bytes, ref_bytes_len, hipMemcpyHostToDevice, st));
    if (read_bytes_len) HIPC(hipMemcpyAsync(ctx->d_reads.p, read_bytes, read_bytes_len, hipMemcpyHostToDevice, st));
    if (n_fusions) HIPC(hipMemcpyAsync(ctx->d_fusions.p, fusions, n_fusions * sizeof(dsa_fusion), hipMemcpyHostToDevice, st));
    if (n_pairs) {
        HIPC(hipMemcpyAsync(ctx->d_pairs.p, pairs, n_pairs * sizeof(dsa_pair), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_clear_pad, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, ctx->d_pairs.p, n_pairs);
    }
    std::vector<int32_t> tab(lqmax + 1);
    for (int l = 0; l <= lqmax; ++l) tab[l] = min_score_for(l);
    HIPC(ctx->d_min_score.reserve(tab.size()));
    HIPC(hipMemcpyAsync(ctx->d_min_score.p, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    const int planned = plan_sweep(ctx, fusions, n_pairs, lqmax);
    if (planned < 0) return planned;
    if (planned == 1) return DSA_OK;
    return build_slices(ctx, fusions, pairs, n_pairs);
}

int dsa_run(dsa_ctx* ctx, int64_t* out_n)
{
    if (!ctx) return DSA_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    ctx->n_records = 0;
    ctx->timing = dsa_timing{};
    ctx->timing.cells = ctx->total_cells;
    // two slices in flight: phase 1 of slice k+1 is queued before the host waits for slice k
    const int ns = (int)ctx->slices.size();
    const auto t0 = std::chrono::steady_clock::now();
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
        HIPC(hipMemcpyAsync(out, ctx->d_records.p, ctx->n_records * sizeof(dsa_record), hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
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
        HIPC(hipMemcpyAsync(out_device, ctx->d_records.p, ctx->n_records * sizeof(dsa_record), hipMemcpyDeviceToDevice, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
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
