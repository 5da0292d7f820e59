// cmp_api.hip — the bin pairs of clustermatepairs on gfx950 (include/defuse_cmp.h): CheckConcordant
// (tools/clustermatepairs.cpp:211-244), AddBinPairs (:246-290) with Binning::GetBins (:146-176), PackAlignment (:178-192) and
// RefBinPacked (:28-65) for all fragments of the input at once, and the unordered_map<RefBinPackedPair, ...> they fill as one
// stable radix sort.
//
// What has to come out the same as a serial reader's map: for every bin pair (a.id, b.id), a <= b, the list `first` (packed
// alignments lying in bin a) and the list `second` (in bin b), each in the order of the reference's insert() calls.  Inside
// one fragment those calls run over the end-0 bins (ascending id) x the end-1 bins (ascending id); an alignment of end 0 in
// bin X is appended once for every distinct end-1 bin Y — to `first` of (X,Y) if X < Y, else to `second` of (Y,X) — and an
// alignment of end 1 in bin X once for every distinct end-0 bin Y — to `second` of (Y,X) if Y < X, else to `first` of (X,Y).
// One list of one bin pair receives entries from at most two of a fragment's (X,Y) combinations, and the loop order puts
// the end-0 alignments in front for `first` and the end-1 alignments in front for `second` (derivation in DESIGN.md
// section 7).  So a fragment's entries are written end by end in that order, fragments in file order, and a STABLE sort by
// key alone reproduces every list.  No atomics decide any order; the result does not depend on the launch shape.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <climits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/defuse_cmp.h"
#include "../../include/defuse_dsa.h"
#include "hip_raii.hpp"

namespace {

thread_local std::string g_cmp_err;

#define CMP_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[256];                                                                         \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_cmp_err = b_;                                                                       \
            return DSA_E_DEVICE;                                                                  \
        }                                                                                         \
    } while (0)

constexpr int BIN_LENGTH = 1 << 15;            // tools/clustermatepairs.cpp:385 (binLength)
constexpr uint32_t REF_MASK = 0x0FFFFFFFu;

__host__ __device__ inline int rec_ref(const cmp_record& a) { return (int)(a.meta & REF_MASK); }
__host__ __device__ inline int rec_strand(const cmp_record& a) { return (int)((a.meta >> 28) & 1u); }
__host__ __device__ inline int rec_end(const cmp_record& a) { return (int)((a.meta >> 29) & 1u); }
// RefBinPacked as one word (:28-65): the caller has checked ref < 2^18 and bin < 2^13
__host__ __device__ inline uint32_t pack_ref_bin(int ref, int strand, int bin) { return (uint32_t)ref | ((uint32_t)strand << 18) | (((uint32_t)bin & 0x1FFFu) << 19); }

// the first (alignment, bin) of a fragment on which AddBinPairs would stop, in its loop order (:252-266); 0 = none
__host__ __device__ inline int fragment_error(const cmp_record* r, int n, int mfr, int& bad, int& value)
{
    for (int i = 0; i < n; ++i) {
        const int sb = (r[i].start - mfr) / BIN_LENGTH, eb = (r[i].end + mfr) / BIN_LENGTH;      // C++ int division, as GetBins
        for (int b = sb; b <= eb; ++b) {
            const int rs = r[i].start - b * BIN_LENGTH + BIN_LENGTH / 2, re = r[i].end - b * BIN_LENGTH + BIN_LENGTH / 2;
            bad = i;
            if (rs < 0 || re < 0 || rs >= (1 << 16) || re >= (1 << 16)) return 1;
            if (rec_ref(r[i]) >= (1 << 18)) { value = rec_ref(r[i]); return 2; }
            if (b >= (1 << 13)) { value = b; return 3; }
        }
    }
    return 0;
}

// CheckConcordant (:211-244): a (reference, bin of length minFusionRange) that both ends reach.  Every alignment covers a
// contiguous run of bins, so two alignments share one iff their runs overlap.
__host__ __device__ inline bool fragment_concordant(const cmp_record* r, int n, int mfr)
{
    for (int i = 0; i < n; ++i) {
        if (rec_end(r[i]) != 0) continue;
        const int s0 = (r[i].start - mfr) / mfr, e0 = (r[i].end + mfr) / mfr;
        if (s0 > e0) continue;
        for (int j = 0; j < n; ++j) {
            if (rec_end(r[j]) != 1 || rec_ref(r[j]) != rec_ref(r[i])) continue;
            const int s1 = (r[j].start - mfr) / mfr, e1 = (r[j].end + mfr) / mfr;
            if (s1 <= e1 && s0 <= e1 && s1 <= e0) return true;
        }
    }
    return false;
}

// the (alignment, bin) entries of one read end of a fragment in arrival order: f(position in the list, record, bin, id)
template <class F>
__host__ __device__ inline void end_list(const cmp_record* r, int n, int mfr, int end, F f)
{
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (rec_end(r[i]) != end) continue;
        const int sb = (r[i].start - mfr) / BIN_LENGTH, eb = (r[i].end + mfr) / BIN_LENGTH;
        for (int b = sb; b <= eb; ++b) f(k++, i, b, pack_ref_bin(rec_ref(r[i]), rec_strand(r[i]), b));
    }
}

// which entries of an end's list are the first with their id (the distinct bins of that end): a bitmap for the first 256
// entries, looked up again the slow way beyond
struct Distinct {
    unsigned long long bits[4];
    __host__ __device__ bool first(const cmp_record* r, int n, int mfr, int end, int k, uint32_t id) const
    {
        if (k < 256) return (bits[k >> 6] >> (k & 63)) & 1ull;
        bool seen = false;
        end_list(r, n, mfr, end, [&](int k2, int, int, uint32_t id2) { if (k2 < k && id2 == id) seen = true; });
        return !seen;
    }
};
__host__ __device__ inline Distinct distinct_of(const cmp_record* r, int n, int mfr, int end)
{
    Distinct d;
    d.bits[0] = d.bits[1] = d.bits[2] = d.bits[3] = 0;
    end_list(r, n, mfr, end, [&](int k, int, int, uint32_t id) {
        if (k >= 256) return;
        bool seen = false;
        end_list(r, n, mfr, end, [&](int k2, int, int, uint32_t id2) { if (k2 < k && id2 == id) seen = true; });
        if (!seen) d.bits[k >> 6] |= 1ull << (k & 63);
    });
    return d;
}

// every entry the fragment adds to the bin pairs: sink(side 0 = first / 1 = second, read end of the alignment, key, record, bin)
template <class Sink>
__host__ __device__ inline void fragment_entries(const cmp_record* r, int n, int mfr, Sink sink)
{
    const Distinct d0 = distinct_of(r, n, mfr, 0), d1 = distinct_of(r, n, mfr, 1);
    for (int ea = 0; ea <= 1; ++ea) {
        const Distinct& dopp = ea == 0 ? d1 : d0;
        end_list(r, n, mfr, ea, [&](int, int ia, int ba, uint32_t X) {
            end_list(r, n, mfr, 1 - ea, [&](int ky, int, int, uint32_t Y) {
                if (!dopp.first(r, n, mfr, 1 - ea, ky, Y)) return;
                const bool fwd = ea == 0 ? X < Y : Y < X;                     // of the reference's (end-0 bin, end-1 bin) pair
                const int side = ea == 0 ? (fwd ? 0 : 1) : (fwd ? 1 : 0);
                const unsigned long long key = X < Y ? ((unsigned long long)X << 32) | Y : ((unsigned long long)Y << 32) | X;
                sink(side, ea, key, ia, ba);
            });
        });
    }
}

struct Counts { uint32_t first, second, first_e0, second_e1; };

__global__ __launch_bounds__(256) void k_cmp_count(const cmp_record* __restrict__ recs, const uint32_t* __restrict__ frag_start, int64_t n_fragments,
                                                    int mfr, uint32_t* __restrict__ cnt_first, uint32_t* __restrict__ cnt_second,
                                                    uint32_t* __restrict__ cnt_first_e0, uint32_t* __restrict__ cnt_second_e1,
                                                    unsigned long long* __restrict__ globals)      // [0] first error (record << 2 | kind), [1] concordant fragments
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fragments) return;
    const uint32_t lo = frag_start[f], hi = frag_start[f + 1];
    const cmp_record* r = recs + lo;
    const int n = (int)(hi - lo);
    Counts c{0, 0, 0, 0};
    if (fragment_concordant(r, n, mfr)) {
        atomicAdd(&globals[1], 1ull);
    } else {
        int bad = 0, value = 0;
        const int kind = fragment_error(r, n, mfr, bad, value);
        if (kind) atomicMin(&globals[0], ((unsigned long long)(lo + (uint32_t)bad) << 2) | (unsigned long long)kind);
        else
            fragment_entries(r, n, mfr, [&](int side, int ea, unsigned long long, int, int) {
                if (side == 0) { ++c.first; if (ea == 0) ++c.first_e0; }
                else { ++c.second; if (ea == 1) ++c.second_e1; }
            });
    }
    cnt_first[f] = c.first;
    cnt_second[f] = c.second;
    cnt_first_e0[f] = c.first_e0;
    cnt_second_e1[f] = c.second_e1;
}

__global__ __launch_bounds__(256) void k_cmp_emit(const cmp_record* __restrict__ recs, const uint32_t* __restrict__ frag_start, int64_t n_fragments,
                                                   int mfr, const uint32_t* __restrict__ cnt_first, const uint32_t* __restrict__ off_first,
                                                   const uint32_t* __restrict__ off_second, const uint32_t* __restrict__ cnt_first_e0,
                                                   const uint32_t* __restrict__ cnt_second_e1, unsigned long long* __restrict__ key_first,
                                                   cmp_packed* __restrict__ pay_first, unsigned long long* __restrict__ key_second,
                                                   cmp_packed* __restrict__ pay_second, const uint32_t* __restrict__ cnt_second)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fragments) return;
    if (cnt_first[f] == 0 && cnt_second[f] == 0) return;            // concordant, stopped, or one-ended
    const uint32_t lo = frag_start[f];
    const cmp_record* r = recs + lo;
    const int n = (int)(frag_start[f + 1] - lo);
    // `first` lists: the end-0 alignments, then the end-1 alignments; `second` lists: end 1, then end 0
    uint32_t at[2][2];
    at[0][0] = off_first[f];
    at[0][1] = off_first[f] + cnt_first_e0[f];
    at[1][1] = off_second[f];
    at[1][0] = off_second[f] + cnt_second_e1[f];
    fragment_entries(r, n, mfr, [&](int side, int ea, unsigned long long key, int ia, int b) {
        const uint32_t k = at[side][ea]++;
        cmp_packed p;
        p.fragment = r[ia].fragment;
        p.read_end = rec_end(r[ia]);
        p.rel_start = (uint16_t)(r[ia].start - b * BIN_LENGTH + BIN_LENGTH / 2);
        p.rel_end = (uint16_t)(r[ia].end - b * BIN_LENGTH + BIN_LENGTH / 2);
        if (side == 0) { key_first[k] = key; pay_first[k] = p; }
        else { key_second[k] = key; pay_second[k] = p; }
    });
}

__global__ void k_cmp_iota(uint32_t* __restrict__ v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

__global__ void k_cmp_gather(const cmp_packed* __restrict__ in, const uint32_t* __restrict__ idx, cmp_packed* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

__global__ void k_cmp_same_keys(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, int64_t n, unsigned long long* __restrict__ globals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) globals[2] = 1;
}

template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        release();
        const hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
};

struct Side {
    DBuf<unsigned long long> key, key_sorted, uniq;
    DBuf<cmp_packed> pay, pay_sorted;
    DBuf<uint32_t> idx, idx_sorted, off;
    DBuf<int> run_len;
    int64_t n = 0;
};

}  // namespace

struct cmp_binner {
    int device = -1;
    hipraii::Stream st;
    hipraii::Event ev[2];
    int64_t n_records = 0, n_fragments = 0;
    DBuf<cmp_record> recs;
    DBuf<uint32_t> frag_start, cnt[4], off_first, off_second;
    DBuf<unsigned long long> globals;
    DBuf<uint8_t> tmp;
    DBuf<int> n_runs;
    Side side[2];
    int64_t n_keys = 0;
    bool ran = false;
};

extern "C" {

const char* cmp_last_error(void) { return g_cmp_err.c_str(); }

int cmp_bin_create(cmp_binner** out, int device)
{
    if (!out) return DSA_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) { g_cmp_err = "no usable HIP device"; return DSA_E_DEVICE; }
    CMP_HIP(hipSetDevice(device));
    cmp_binner* b = new cmp_binner();
    b->device = device;
    if (b->st.create(hipStreamNonBlocking) != hipSuccess || b->ev[0].create() != hipSuccess || b->ev[1].create() != hipSuccess) {
        delete b;
        g_cmp_err = "cannot create a stream";
        return DSA_E_DEVICE;
    }
    *out = b;
    return DSA_OK;
}

void cmp_bin_destroy(cmp_binner* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)hipDeviceSynchronize();
    delete b;
}

int cmp_bin_reserve(cmp_binner* b, int64_t n_records, int64_t n_fragments)
{
    if (!b || n_records < 0 || n_fragments < 0) return DSA_E_ARG;
    if (n_records >= ((int64_t)1 << 32) - 1) { g_cmp_err = "more than 2^32 alignment records in one call"; return DSA_E_LIMIT; }
    CMP_HIP(hipSetDevice(b->device));
    b->n_records = n_records;
    b->n_fragments = n_fragments;
    b->ran = false;
    CMP_HIP(b->recs.reserve((size_t)n_records));
    CMP_HIP(b->frag_start.reserve((size_t)n_fragments + 1));
    return DSA_OK;
}

int cmp_bin_upload_records(cmp_binner* b, const cmp_record* recs, int64_t n, int64_t at)
{
    if (!b || n < 0 || at < 0 || at + n > b->n_records || (n && !recs)) return DSA_E_ARG;
    CMP_HIP(hipSetDevice(b->device));
    if (n) CMP_HIP(hipMemcpy(b->recs.p + at, recs, (size_t)n * sizeof(cmp_record), hipMemcpyHostToDevice));
    return DSA_OK;
}

int cmp_bin_upload_fragments(cmp_binner* b, const uint32_t* frag_start, int64_t n, int64_t at)
{
    if (!b || n < 0 || at < 0 || at + n > b->n_fragments + 1 || (n && !frag_start)) return DSA_E_ARG;
    CMP_HIP(hipSetDevice(b->device));
    if (n) CMP_HIP(hipMemcpy(b->frag_start.p + at, frag_start, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    return DSA_OK;
}

int cmp_bin_run(cmp_binner* b, int32_t mfr, cmp_stats* stats)
{
    if (!b || !stats || mfr <= 0) return DSA_E_ARG;
    CMP_HIP(hipSetDevice(b->device));
    hipStream_t st = b->st;
    const int64_t nf = b->n_fragments;
    *stats = cmp_stats{};
    stats->n_fragments = nf;
    stats->err_record = -1;
    for (auto& c : b->cnt) CMP_HIP(c.reserve((size_t)nf + 1));
    CMP_HIP(b->off_first.reserve((size_t)nf + 1));
    CMP_HIP(b->off_second.reserve((size_t)nf + 1));
    CMP_HIP(b->globals.reserve(4));
    CMP_HIP(b->n_runs.reserve(2));
    const unsigned long long init[4] = {~0ull, 0, 0, 0};
    CMP_HIP(hipMemcpyAsync(b->globals.p, init, sizeof init, hipMemcpyHostToDevice, st));
    CMP_HIP(hipEventRecord(b->ev[0], st));
    const unsigned grid = (unsigned)((nf + 255) / 256);
    if (nf) {
        hipLaunchKernelGGL(k_cmp_count, dim3(grid), dim3(256), 0, st, b->recs.p, b->frag_start.p, nf, (int)mfr, b->cnt[0].p, b->cnt[1].p, b->cnt[2].p,
                           b->cnt[3].p, b->globals.p);
        CMP_HIP(hipMemsetAsync(b->cnt[0].p + nf, 0, sizeof(uint32_t), st));      // one more element: the scans' last output is the total
        CMP_HIP(hipMemsetAsync(b->cnt[1].p + nf, 0, sizeof(uint32_t), st));
    } else {
        CMP_HIP(hipMemsetAsync(b->cnt[0].p, 0, sizeof(uint32_t), st));
        CMP_HIP(hipMemsetAsync(b->cnt[1].p, 0, sizeof(uint32_t), st));
    }
    size_t tmp_bytes = 0, need = 0;
    CMP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, b->cnt[0].p, b->off_first.p, (int)(nf + 1), st));
    CMP_HIP(b->tmp.reserve(tmp_bytes));
    need = b->tmp.cap;
    CMP_HIP(hipcub::DeviceScan::ExclusiveSum(b->tmp.p, need, b->cnt[0].p, b->off_first.p, (int)(nf + 1), st));
    need = b->tmp.cap;
    CMP_HIP(hipcub::DeviceScan::ExclusiveSum(b->tmp.p, need, b->cnt[1].p, b->off_second.p, (int)(nf + 1), st));
    uint32_t totals[2] = {0, 0};
    unsigned long long glob[4];
    CMP_HIP(hipMemcpyAsync(&totals[0], b->off_first.p + nf, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CMP_HIP(hipMemcpyAsync(&totals[1], b->off_second.p + nf, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CMP_HIP(hipMemcpyAsync(glob, b->globals.p, sizeof glob, hipMemcpyDeviceToHost, st));
    CMP_HIP(hipStreamSynchronize(st));
    CMP_HIP(hipGetLastError());
    stats->n_concordant = (int64_t)glob[1];
    if (glob[0] != ~0ull) {
        // the reference stops at this alignment: the caller reports it (kind and value recomputed on the host from the record)
        stats->err_record = (int64_t)(glob[0] >> 2);
        stats->err_kind = (int32_t)(glob[0] & 3u);
        b->ran = false;
        return DSA_OK;
    }
    // (a 32-bit scan: the totals must fit; the counts of one fragment are far below that)
    if ((uint64_t)totals[0] >= ((uint64_t)1 << 31) || (uint64_t)totals[1] >= ((uint64_t)1 << 31)) {
        g_cmp_err = "more than 2^31 bin-pair entries in one call";
        return DSA_E_LIMIT;
    }
    for (int s = 0; s < 2; ++s) {
        Side& S = b->side[s];
        S.n = totals[s];
        CMP_HIP(S.key.reserve((size_t)S.n));
        CMP_HIP(S.key_sorted.reserve((size_t)S.n));
        CMP_HIP(S.pay.reserve((size_t)S.n));
        CMP_HIP(S.pay_sorted.reserve((size_t)S.n));
        CMP_HIP(S.idx.reserve((size_t)S.n));
        CMP_HIP(S.idx_sorted.reserve((size_t)S.n));
        CMP_HIP(S.uniq.reserve((size_t)S.n));
        CMP_HIP(S.run_len.reserve((size_t)S.n));
        CMP_HIP(S.off.reserve((size_t)S.n + 1));
    }
    if (nf)
        hipLaunchKernelGGL(k_cmp_emit, dim3(grid), dim3(256), 0, st, b->recs.p, b->frag_start.p, nf, (int)mfr, b->cnt[0].p, b->off_first.p, b->off_second.p,
                           b->cnt[2].p, b->cnt[3].p, b->side[0].key.p, b->side[0].pay.p, b->side[1].key.p, b->side[1].pay.p, b->cnt[1].p);
    int runs[2] = {0, 0};
    for (int s = 0; s < 2; ++s) {
        Side& S = b->side[s];
        const int n = (int)S.n;
        if (n == 0) continue;
        hipLaunchKernelGGL(k_cmp_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S.idx.p, (int64_t)n);
        // stable: entries with one key keep the order they were written in, which is the reference's order of appends
        size_t tb = 0;
        CMP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, S.key.p, S.key_sorted.p, S.idx.p, S.idx_sorted.p, n, 0, 64, st));
        CMP_HIP(b->tmp.reserve(tb));
        tb = b->tmp.cap;
        CMP_HIP(hipcub::DeviceRadixSort::SortPairs(b->tmp.p, tb, S.key.p, S.key_sorted.p, S.idx.p, S.idx_sorted.p, n, 0, 64, st));
        hipLaunchKernelGGL(k_cmp_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S.pay.p, S.idx_sorted.p, S.pay_sorted.p, (int64_t)n);
        tb = 0;
        CMP_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb, S.key_sorted.p, S.uniq.p, S.run_len.p, b->n_runs.p + s, n, st));
        CMP_HIP(b->tmp.reserve(tb));
        tb = b->tmp.cap;
        CMP_HIP(hipcub::DeviceRunLengthEncode::Encode(b->tmp.p, tb, S.key_sorted.p, S.uniq.p, S.run_len.p, b->n_runs.p + s, n, st));
        CMP_HIP(hipMemcpyAsync(&runs[s], b->n_runs.p + s, sizeof(int), hipMemcpyDeviceToHost, st));
        CMP_HIP(hipStreamSynchronize(st));
        CMP_HIP(hipMemsetAsync(S.run_len.p + runs[s], 0, sizeof(int), st));
        tb = 0;
        CMP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, S.run_len.p, (int*)S.off.p, runs[s] + 1, st));
        CMP_HIP(b->tmp.reserve(tb));
        tb = b->tmp.cap;
        CMP_HIP(hipcub::DeviceScan::ExclusiveSum(b->tmp.p, tb, S.run_len.p, (int*)S.off.p, runs[s] + 1, st));
    }
    // every bin pair has entries on both sides (each insert() pair of the reference feeds both lists)
    if (runs[0] != runs[1]) { g_cmp_err = "internal: the two sides disagree on the bin pairs"; return DSA_E_DEVICE; }
    if (runs[0])
        hipLaunchKernelGGL(k_cmp_same_keys, dim3((unsigned)((runs[0] + 255) / 256)), dim3(256), 0, st, b->side[0].uniq.p, b->side[1].uniq.p, (int64_t)runs[0], b->globals.p);
    CMP_HIP(hipEventRecord(b->ev[1], st));
    CMP_HIP(hipMemcpyAsync(glob, b->globals.p, sizeof glob, hipMemcpyDeviceToHost, st));
    CMP_HIP(hipStreamSynchronize(st));
    CMP_HIP(hipGetLastError());
    if (glob[2]) { g_cmp_err = "internal: the two sides disagree on the bin pairs"; return DSA_E_DEVICE; }
    b->n_keys = runs[0];
    stats->n_keys = runs[0];
    stats->n_first = b->side[0].n;
    stats->n_second = b->side[1].n;
    (void)hipEventElapsedTime(&stats->device_ms, b->ev[0], b->ev[1]);
    b->ran = true;
    return DSA_OK;
}

int cmp_bin_fetch(cmp_binner* b, uint64_t* keys, int64_t* off_first, int64_t* off_second, cmp_packed* first, cmp_packed* second)
{
    if (!b || !b->ran) { g_cmp_err = "cmp_bin_fetch before a successful cmp_bin_run"; return DSA_E_ARG; }
    if (!off_first || !off_second) return DSA_E_ARG;
    CMP_HIP(hipSetDevice(b->device));
    const int64_t nk = b->n_keys;
    std::vector<int32_t> tmp((size_t)nk + 1);
    int64_t* outs[2] = {off_first, off_second};
    cmp_packed* pays[2] = {first, second};
    for (int s = 0; s < 2; ++s) {
        Side& S = b->side[s];
        if (nk) CMP_HIP(hipMemcpy(tmp.data(), S.off.p, ((size_t)nk + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
        else tmp[0] = 0;
        for (int64_t k = 0; k <= nk; ++k) outs[s][k] = tmp[(size_t)k];
        if (S.n && pays[s])           // (NULL: the caller takes the lists in parts, cmp_bin_fetch_part)
            CMP_HIP(hipMemcpy(pays[s], S.pay_sorted.p, (size_t)S.n * sizeof(cmp_packed), hipMemcpyDeviceToHost));
    }
    if (nk) {
        if (!keys) return DSA_E_ARG;
        CMP_HIP(hipMemcpy(keys, b->side[0].uniq.p, (size_t)nk * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return DSA_OK;
}


int cmp_bin_fetch_part(cmp_binner* b, int which, int64_t from, int64_t n, cmp_packed* out)
{
    if (!b || !b->ran) { g_cmp_err = "cmp_bin_fetch_part before a successful cmp_bin_run"; return DSA_E_ARG; }
    if (which < 0 || which > 1 || from < 0 || n < 0 || from + n > b->side[which].n || (n && !out)) return DSA_E_ARG;
    CMP_HIP(hipSetDevice(b->device));
    if (n) CMP_HIP(hipMemcpy(out, b->side[which].pay_sorted.p + from, (size_t)n * sizeof(cmp_packed), hipMemcpyDeviceToHost));
    return DSA_OK;
}

}  // extern "C"
