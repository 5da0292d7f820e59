// dsa_api.hip — extern "C" ABI of include/defuse_dsa.h on top of the gfx950 kernels.
//
// Replaces the per-candidate call SplitAlignmentTask::Align (tools/SplitAlignment.cpp:371-444 of the
// reference) for whole batches.  There is deliberately no CPU fallback in this file: without a HIP
// device dsa_create fails and every other entry point needs a ctx.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dsa_kernels.hpp"

using namespace dsa;

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
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

struct Slice {
    int64_t pair_begin = 0, pair_end = 0;
    std::vector<WaveInfo> waves;     // one per 64 pairs
    std::vector<WgInfo> wgs;         // one per 256 pairs
    std::vector<uint32_t> wg_flags;  // initial generic flag (1 = more than GMAX fusions)
    Geom g{};
};

}  // namespace

struct dsa_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    size_t scratch_budget = (size_t)16 << 30;

    // resident batch
    int64_t n_pairs = 0, ref_bytes_len = 0, read_bytes_len = 0;
    int32_t n_fusions = 0;
    DevBuf<uint8_t> d_ref, d_reads;
    DevBuf<dsa_fusion> d_fusions;
    DevBuf<dsa_pair> d_pairs;
    DevBuf<int32_t> d_min_score;
    std::vector<Slice> slices;
    int64_t total_cells = 0;

    // scratch
    DevBuf<WaveInfo> d_waves;
    DevBuf<WgInfo> d_wgs;
    DevBuf<uint32_t> d_wg_generic;
    DevBuf<uint32_t> d_refcodes, d_rowcodes, d_bnd, d_cmax, d_rmax, d_tmask;
    DevBuf<PairState> d_state;
    DevBuf<KeptRow> d_kept;
    DevBuf<int64_t> d_rec_count, d_rec_offset;
    DevBuf<ReplayTask> d_tasks;
    DevBuf<uint64_t> d_masks;
    DevBuf<uint32_t> d_gtasks;
    DevBuf<int32_t> d_wgtile;
    DevBuf<Counters> d_ctr;
    DevBuf<int16_t> d_mscratch;
    DevBuf<uint8_t> d_scan_tmp;
    DevBuf<dsa_record> d_records;
    int64_t n_records = 0;
    bool have_results = false;

    hipEvent_t ev[8] = {};
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
int build_slices(dsa_ctx* ctx, const dsa_fusion* fusions, const dsa_pair* pairs, int64_t n_pairs)
{
    ctx->slices.clear();
    ctx->total_cells = 0;
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
                        if (wg.n_groups < GMAX)
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
        cur.g.lrp = nch * W;
        cur.g.n_fusions = ctx->n_fusions;
        cur.g.n_pairs = cur.pair_end - cur.pair_begin;
        ctx->slices.push_back(std::move(cur));
    }
    return DSA_OK;
}

int exclusive_scan(dsa_ctx* ctx, const int64_t* in, int64_t* out, int64_t n)
{
    size_t tmp = 0;
    HIPC(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, in, out, (int)n, ctx->stream));
    HIPC(ctx->d_scan_tmp.reserve(tmp));
    HIPC(hipcub::DeviceScan::ExclusiveSum(ctx->d_scan_tmp.p, tmp, in, out, (int)n, ctx->stream));
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
    if (ctx->n_records)
        HIPC(hipMemcpyAsync(bigger.p, ctx->d_records.p, ctx->n_records * sizeof(dsa_record), hipMemcpyDeviceToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->d_records.release();
    ctx->d_records = bigger;
    return DSA_OK;
}

int run_slice(dsa_ctx* ctx, const Slice& s)
{
    Geom g = s.g;
#ifdef DSA_PRUNE_STATS
    static unsigned long long* d_stats = nullptr;
    if (!d_stats) (void)hipMalloc((void**)&d_stats, 4 * sizeof(unsigned long long));
    (void)hipMemset(d_stats, 0, 4 * sizeof(unsigned long long));
    g.stats = d_stats;
#endif
    hipStream_t st = ctx->stream;
    const int64_t np = g.n_pairs;
    const dsa_pair* pairs = ctx->d_pairs.p + s.pair_begin;

    const size_t n_rows = (size_t)g.n_waves * g.lq1 * WAVE;
    HIPC(ctx->d_waves.reserve(s.waves.size()));
    HIPC(ctx->d_wgs.reserve(s.wgs.size()));
    HIPC(ctx->d_wg_generic.reserve(s.wg_flags.size()));
    HIPC(ctx->d_refcodes.reserve((size_t)g.n_fusions * g.lrp));
    HIPC(ctx->d_rowcodes.reserve(n_rows));
    HIPC(ctx->d_bnd.reserve(n_rows * g.nch));
    HIPC(ctx->d_cmax.reserve(n_rows * g.nch));
    HIPC(ctx->d_rmax.reserve(n_rows));
    HIPC(ctx->d_tmask.reserve(n_rows));
    HIPC(ctx->d_state.reserve(np));
    HIPC(ctx->d_rec_count.reserve(np + 1));
    HIPC(ctx->d_rec_offset.reserve(np + 1));
    HIPC(ctx->d_ctr.reserve(1));
    HIPC(ctx->d_kept.reserve((size_t)np * 2 + 1024));
    HIPC(ctx->d_tasks.reserve((size_t)np * 4 + 1024));
    HIPC(ctx->d_masks.reserve((size_t)np * 8 + 1024));
    HIPC(ctx->d_gtasks.reserve((size_t)np * 2 + 1024));
    HIPC(ctx->d_wgtile.reserve((size_t)g.n_wgs * GMAX + 16));
    if (int rc = grow_records(ctx, (size_t)ctx->n_records + (size_t)np * 2 + 1024)) return rc;
    HIPC(hipMemcpyAsync(ctx->d_waves.p, s.waves.data(), s.waves.size() * sizeof(WaveInfo), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(ctx->d_wgs.p, s.wgs.data(), s.wgs.size() * sizeof(WgInfo), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(ctx->d_wg_generic.p, s.wg_flags.data(), s.wg_flags.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));

    // ---- pack -------------------------------------------------------------------------------
    HIPC(hipEventRecord(ctx->ev[0], st));
    {
        int64_t total = (int64_t)g.n_fusions * g.lrp;
        hipLaunchKernelGGL(k_pack_refs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ctx->d_ref.p,
                           ctx->d_fusions.p, ctx->d_refcodes.p, g);
        total = (int64_t)n_rows / 4;
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ctx->d_reads.p, pairs,
                           ctx->d_rowcodes.p, ctx->d_wg_generic.p, g);
    }
    HIPC(hipEventRecord(ctx->ev[1], st));
    // ---- fill: every workgroup is run by exactly one of the two kernels ---------------------------
    hipLaunchKernelGGL(k_fill_fast, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, ctx->d_waves.p, ctx->d_wgs.p,
                       ctx->d_wg_generic.p, ctx->d_refcodes.p, ctx->d_rowcodes.p, ctx->d_min_score.p, ctx->d_bnd.p, ctx->d_cmax.p,
                       ctx->d_rmax.p, ctx->d_tmask.p, g);
    hipLaunchKernelGGL(k_fill_generic, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, ctx->d_waves.p,
                       ctx->d_fusions.p, ctx->d_wg_generic.p, ctx->d_refcodes.p, ctx->d_rowcodes.p, ctx->d_bnd.p,
                       ctx->d_cmax.p, ctx->d_rmax.p, ctx->d_tmask.p, g);
    HIPC(hipEventRecord(ctx->ev[2], st));
    HIPC(hipGetLastError());
    // ---- finish: combine -> replay -> emit, no host round trip unless a buffer overflowed --------
    if (g.lq1 > 7601) return fail(ctx, DSA_E_LIMIT, "reads longer than 7600 are not supported");
    const unsigned pair_grid = (unsigned)((np + 255) / 256);
    Counters ctr{};
    int64_t n_rec = 0;
    bool redo_combine = true;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (redo_combine) {
            HIPC(hipMemsetAsync(ctx->d_ctr.p, 0, sizeof(Counters), st));
            hipLaunchKernelGGL(k_combine, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, pairs, ctx->d_fusions.p,
                               ctx->d_cmax.p, ctx->d_rmax.p, ctx->d_tmask.p, ctx->d_min_score.p, ctx->d_wgs.p, ctx->d_wg_generic.p,
                               ctx->d_state.p, ctx->d_kept.p, (uint64_t)ctx->d_kept.cap, ctx->d_tasks.p,
                               (uint64_t)ctx->d_tasks.cap, (uint64_t)(ctx->d_masks.cap / 2), ctx->d_gtasks.p,
                               (uint64_t)ctx->d_gtasks.cap, ctx->d_wgtile.p, ctx->d_ctr.p, g);
            hipLaunchKernelGGL(k_replay_fast, dim3((unsigned)g.n_wgs), dim3(WG_LANES), 0, st, ctx->d_tasks.p,
                               (uint64_t)ctx->d_tasks.cap, ctx->d_ctr.p, ctx->d_state.p, ctx->d_kept.p,
                               (uint64_t)ctx->d_kept.cap, pairs, ctx->d_fusions.p, ctx->d_wgs.p, ctx->d_wgtile.p,
                               ctx->d_refcodes.p, ctx->d_rowcodes.p, ctx->d_bnd.p, ctx->d_masks.p,
                               (uint64_t)(ctx->d_masks.cap / 2), (uint64_t)ctx->d_gtasks.cap, g);
            hipLaunchKernelGGL(k_replay, dim3(256 * 4), dim3(256), 0, st, ctx->d_tasks.p, (uint64_t)ctx->d_tasks.cap,
                               ctx->d_gtasks.p, (uint64_t)ctx->d_gtasks.cap, ctx->d_ctr.p, ctx->d_state.p, ctx->d_kept.p,
                               (uint64_t)ctx->d_kept.cap, pairs, ctx->d_fusions.p, ctx->d_refcodes.p, ctx->d_rowcodes.p,
                               ctx->d_bnd.p, ctx->d_masks.p, (uint64_t)(ctx->d_masks.cap / 2), g);
            hipLaunchKernelGGL(k_emit<false>, dim3(pair_grid), dim3(256), 0, st, pairs, ctx->d_fusions.p, ctx->d_state.p,
                               ctx->d_kept.p, ctx->d_tasks.p, ctx->d_masks.p, ctx->d_rec_count.p, (const int64_t*)nullptr,
                               (dsa_record*)nullptr, (uint64_t)0, (int64_t)s.pair_begin, g);
            HIPC(hipMemsetAsync(ctx->d_rec_count.p + np, 0, sizeof(int64_t), st));
            if (int rc = exclusive_scan(ctx, ctx->d_rec_count.p, ctx->d_rec_offset.p, np + 1)) return rc;
        }
        hipLaunchKernelGGL(k_emit<true>, dim3(pair_grid), dim3(256), 0, st, pairs, ctx->d_fusions.p, ctx->d_state.p,
                           ctx->d_kept.p, ctx->d_tasks.p, ctx->d_masks.p, ctx->d_rec_count.p,
                           (const int64_t*)ctx->d_rec_offset.p, ctx->d_records.p + ctx->n_records,
                           (uint64_t)(ctx->d_records.cap - ctx->n_records), (int64_t)s.pair_begin, g);
        HIPC(hipEventRecord(ctx->ev[3], st));
        HIPC(hipMemcpyAsync(&ctr, ctx->d_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPC(hipMemcpyAsync(&n_rec, ctx->d_rec_offset.p + np, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        HIPC(hipGetLastError());
        redo_combine = ctr.n_kept > ctx->d_kept.cap || ctr.n_tasks > ctx->d_tasks.cap || ctr.n_masks > ctx->d_masks.cap / 2 ||
                       ctr.n_gtasks > ctx->d_gtasks.cap;
        if (redo_combine) {
            HIPC(ctx->d_kept.reserve(ctr.n_kept + 1024));
            HIPC(ctx->d_tasks.reserve(ctr.n_tasks + 1024));
            HIPC(ctx->d_masks.reserve(2 * ctr.n_masks + 1024));
            HIPC(ctx->d_gtasks.reserve(ctr.n_gtasks + 1024));
            continue;
        }
        if ((size_t)(ctx->n_records + n_rec) > ctx->d_records.cap) {
            if (int rc = grow_records(ctx, (size_t)(ctx->n_records + n_rec))) return rc;
            continue;
        }
        break;
    }
    if (redo_combine || (size_t)(ctx->n_records + n_rec) > ctx->d_records.cap)
        return fail(ctx, DSA_E_DEVICE, "finish stage did not converge");
#ifdef DSA_PRUNE_STATS
    {
        unsigned long long h[4] = {0, 0, 0, 0};
        (void)hipMemcpy(h, g.stats, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[prune] skipped row groups %llu of %llu, sum l_in %llu\n", h[0], h[1], h[2]);
    }
#endif
    ctx->n_records += n_rec;
    ctx->timing.pack_ms += elapsed(ctx->ev[0], ctx->ev[1]);
    ctx->timing.fill_ms += elapsed(ctx->ev[1], ctx->ev[2]);
    ctx->timing.finish_ms += elapsed(ctx->ev[2], ctx->ev[3]);
    ctx->timing.total_ms += elapsed(ctx->ev[0], ctx->ev[3]);
    ctx->timing.fill_launches += 1;
    ctx->timing.n_replay_tasks += (int64_t)ctr.n_tasks;
    return DSA_OK;
}

}  // namespace

extern "C" {

const char* dsa_version(void) { return "defuse_amd dsa 0.1 (gfx950)"; }

int dsa_create(dsa_ctx** out, int device)
{
    if (!out) return DSA_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DSA_E_DEVICE;
    dsa_ctx* ctx = new dsa_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess) {
        delete ctx;
        return DSA_E_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    for (auto& e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return DSA_E_DEVICE;
        }
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
    (void)hipStreamSynchronize(ctx->stream);
    ctx->d_ref.release(); ctx->d_reads.release(); ctx->d_fusions.release(); ctx->d_pairs.release();
    ctx->d_min_score.release(); ctx->d_waves.release(); ctx->d_wgs.release(); ctx->d_wg_generic.release();
    ctx->d_refcodes.release(); ctx->d_rowcodes.release(); ctx->d_bnd.release(); ctx->d_cmax.release(); ctx->d_rmax.release(); ctx->d_tmask.release();
    ctx->d_state.release(); ctx->d_kept.release(); ctx->d_rec_count.release(); ctx->d_rec_offset.release();
    ctx->d_tasks.release(); ctx->d_masks.release(); ctx->d_gtasks.release(); ctx->d_wgtile.release(); ctx->d_ctr.release(); ctx->d_mscratch.release();
    ctx->d_scan_tmp.release(); ctx->d_records.release();
    for (auto& e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
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
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
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
    ctx->n_pairs = n_pairs;
    ctx->n_fusions = n_fusions;
    ctx->ref_bytes_len = ref_bytes_len;
    ctx->read_bytes_len = read_bytes_len;
    HIPC(ctx->d_ref.reserve((size_t)ref_bytes_len + 1));
    HIPC(ctx->d_reads.reserve((size_t)read_bytes_len + 1));
    HIPC(ctx->d_fusions.reserve((size_t)n_fusions + 1));
    HIPC(ctx->d_pairs.reserve((size_t)n_pairs + 1));
    hipStream_t st = ctx->stream;
    if (ref_bytes_len) HIPC(hipMemcpyAsync(ctx->d_ref.p, ref_bytes, ref_bytes_len, hipMemcpyHostToDevice, st));
    if (read_bytes_len) HIPC(hipMemcpyAsync(ctx->d_reads.p, read_bytes, read_bytes_len, hipMemcpyHostToDevice, st));
    if (n_fusions) HIPC(hipMemcpyAsync(ctx->d_fusions.p, fusions, n_fusions * sizeof(dsa_fusion), hipMemcpyHostToDevice, st));
    if (n_pairs) HIPC(hipMemcpyAsync(ctx->d_pairs.p, pairs, n_pairs * sizeof(dsa_pair), hipMemcpyHostToDevice, st));
    std::vector<int32_t> tab(lqmax + 1);
    for (int l = 0; l <= lqmax; ++l) tab[l] = min_score_for(l);
    HIPC(ctx->d_min_score.reserve(tab.size()));
    HIPC(hipMemcpyAsync(ctx->d_min_score.p, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return build_slices(ctx, fusions, pairs, n_pairs);
}

int dsa_run(dsa_ctx* ctx, int64_t* out_n)
{
    if (!ctx) return DSA_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    ctx->n_records = 0;
    ctx->timing = dsa_timing{};
    ctx->timing.cells = ctx->total_cells;
    for (const Slice& s : ctx->slices)
        if (int rc = run_slice(ctx, s)) return rc;
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
