// la_api.hip — batched SimpleAligner scores on gfx950 behind the C ABI of include/defuse_la.h.
//
// Layout mirrors the split-read DP (dsa_kernels.hpp): a lane owns its pairs for the whole matrix, the
// reference is cut into tiles of W = 64 columns that live in registers, rows stream through, and the
// last column of a tile is handed to the next tile through a per-wave plane in HBM (two planes,
// ping-pong).  Four consecutive rows of a lane share one 16-byte word of every plane.
//
//   k_la16: two pairs per lane (lo / hi 16-bit fields).  Works on Y = H - gap*j (>= 0, kept biased so
//           that every field is the bit pattern of a positive normal fp16 and v_pk_maximum3_f16 is an
//           exact integer maximum, as in dsa_kernels.hpp):
//               Y(i,j) = max3(Y(i-1,j-1) + s - gap, Y(i,j-1), Y(i-1,j) + gap)
//           per column: xor + pk_min + pk_mad (score term), add (diagonal), sub (left), max3, and half a
//           max3 for the row maximum.  Needs gap <= mismatch <= 0 and a bounded value range.
//           With a minimum score per pair (the tool's -t threshold) it prunes exactly, as the split-read fill
//           does: a cell with H + max(match,0)*(rows left) < need cannot lie on the path to a score >= need, so
//           once a whole row group of a tile and the boundary entering it are dead the tile is left.
//   k_la32: one pair per lane in int32, any scores, padded columns masked out of the maximum.
//
// Pairs are sorted by (sequence length, reference length) so that the lanes of a wave sweep about the
// same number of rows and tiles.
#include <hip/hip_runtime.h>

#include "hip_raii.hpp"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/defuse_dsa.h"
#include "../../include/defuse_la.h"

namespace la {

constexpr int W = 64;
constexpr int WAVE = 64;
constexpr int WG_WAVES = 4;
constexpr uint32_t REF_PAD = 0x0100u;      // never equals a byte nor ROW_PAD
constexpr uint32_t ROW_PAD = 0x0200u;
constexpr uint32_t BIAS16 = 0x0800u;       // Y = 0
constexpr uint32_t BIAS2 = BIAS16 * 0x00010001u;
constexpr int Y_LIMIT = 30000;             // largest biased value the packed kernel may reach (< 0x7C00)

typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t max3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(
                                            __builtin_elementwise_maximum(__builtin_bit_cast(v2h, a), __builtin_bit_cast(v2h, b)),
                                            __builtin_bit_cast(v2h, c)));
}

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)   // per-field unsigned maximum
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b)));
}

struct Lane {              // what one lane aligns: ITEMS pairs (second one unused by the int32 kernel)
    int32_t lr[2], ls[2];
    int32_t out[2];        // index into scores, -1 = padding
    int32_t need[2];       // scores below this need not be exact (NO_NEED: always exact)
};
constexpr int32_t NO_NEED = -0x40000000;
struct Wave {
    int64_t ref_off;       // dwords into refcodes: [nch*W columns][64 lanes]
    int64_t row_off;       // dwords into rowcodes: [rows4][64 lanes], rowidx layout
    int64_t bnd_off;       // dwords into bnd: 2 planes of [rows4][64 lanes]
    int32_t nch, ls_max, rows4, pad_;
};
struct Params {
    int32_t match, mismatch, gap;      // as given (int32 kernel)
    int32_t dm, dx, gd;                // packed kernel: match - gap, mismatch - gap (after the 2*gap clamp), -gap
    int32_t mp;                        // max(match, 0): the most a further row can add to a score
};

__host__ __device__ __forceinline__ int64_t rowidx(int j, int lane) { return ((int64_t)(j >> 2) * WAVE + lane) * 4 + (j & 3); }

// codes of both planes; ITEMS pairs per lane share a dword (16 bits each)
template <int ITEMS>
__global__ void k_pack(const uint8_t* __restrict__ pool, const la_item* __restrict__ items, const Lane* __restrict__ lanes,
                       const Wave* __restrict__ waves, uint32_t* __restrict__ refcodes, uint32_t* __restrict__ rowcodes)
{
    const int w = blockIdx.x;
    const Wave wv = waves[w];
    const int lane = threadIdx.x & 63;
    const Lane ln = lanes[(int64_t)w * WAVE + lane];
    const uint8_t* ref[2] = {nullptr, nullptr};
    const uint8_t* seq[2] = {nullptr, nullptr};
#pragma unroll
    for (int f = 0; f < ITEMS; ++f)
        if (ln.out[f] >= 0) {
            ref[f] = pool + items[ln.out[f]].ref_off;
            seq[f] = pool + items[ln.out[f]].seq_off;
        }
    for (int i = threadIdx.x >> 6; i < wv.nch * W; i += blockDim.x >> 6) {
        uint32_t code = 0;
#pragma unroll
        for (int f = 0; f < ITEMS; ++f) code |= ((ref[f] != nullptr && i < ln.lr[f]) ? (uint32_t)ref[f][i] : REF_PAD) << (16 * f);
        refcodes[wv.ref_off + (int64_t)i * WAVE + lane] = code;
    }
    for (int j = threadIdx.x >> 6; j < wv.rows4; j += blockDim.x >> 6) {
        uint32_t code = 0;
#pragma unroll
        for (int f = 0; f < ITEMS; ++f)
            code |= ((seq[f] != nullptr && j >= 1 && j <= ln.ls[f]) ? (uint32_t)seq[f][j - 1] : ROW_PAD) << (16 * f);
        rowcodes[wv.row_off + rowidx(j, lane)] = code;
    }
}

// ---------------------------------------------------------------------------------------------
// packed kernel: two pairs per lane
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_WAVES * WAVE, 2) void k_la16(const Lane* __restrict__ lanes, const Wave* __restrict__ waves, int n_waves,
                                                           const uint32_t* __restrict__ refcodes,
                                                           const uint32_t* __restrict__ rowcodes, uint32_t* __restrict__ bnd,
                                                           Params prm, int32_t* __restrict__ scores)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WG_WAVES + (threadIdx.x >> 6)));
    if (w >= n_waves) return;
    const int lane = threadIdx.x & 63;
    const Wave wv = waves[w];
    const Lane ln = lanes[(int64_t)w * WAVE + lane];
    const uint32_t gd2 = (uint32_t)prm.gd * 0x00010001u;
    const uint32_t dm2 = (uint32_t)prm.dm * 0x00010001u;
    const uint32_t nd2 = (uint32_t)((prm.dx - prm.dm) & 0xFFFF) * 0x00010001u;     // mod 2^16 per field
    const uint32_t one2 = 0x00010001u;
    // score term of a cell: dm where the codes are equal, dx elsewhere (per field): xor, then min(.,1) and a
    // multiply-add on both fields at once.  Written as the two packed instructions: from the C expression
    // the compiler builds per-field compares, selects and a permute (11 instructions per column instead of 7).
    auto term = [&](uint32_t r, uint32_t q) -> uint32_t {
        uint32_t t, d;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(r ^ q), "v"(one2));
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(t), "v"(nd2), "v"(dm2));
        return d;
    };
    const uint4* rows4 = reinterpret_cast<const uint4*>(rowcodes + wv.row_off) + lane;
    const int64_t plane = (int64_t)wv.rows4 * WAVE;
    const uint4 bias4 = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
    const int ngq = (wv.ls_max >> 2) + 1;
    int best0 = 0, best1 = 0;
    // pruning: field f is alive at row j while Y (biased) >= need_f - mp*ls_f + (mp + gd)*j + BIAS
    const int slope = prm.mp + prm.gd;
    const int base0 = ln.need[0] - prm.mp * ln.ls[0] + (int)BIAS16, base1 = ln.need[1] - prm.mp * ln.ls[1] + (int)BIAS16;
    auto wave_max = [](int v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
        return v;
    };
    auto rows_alive_at_zero = [&](int base, int ls) {          // rows at which Y = 0 (column 0, fresh starts) is alive
        if (ls <= 0) return 0;
        if (base - (int)BIAS16 <= 0 && slope <= 0) return ls;
        if (base - (int)BIAS16 > 0) return 0;
        return min(ls, ((int)BIAS16 - base) / slope);
    };
    int l_in = wave_max(max(rows_alive_at_zero(base0, ln.ls[0]), rows_alive_at_zero(base1, ln.ls[1])));
    int stop_prev = 0;

    for (int c = 0; c < wv.nch; ++c) {
        uint32_t r[W];
        const uint32_t* rc = refcodes + wv.ref_off + (int64_t)c * W * WAVE + lane;
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] = rc[(int64_t)i * WAVE];
        uint32_t X[W];
#pragma unroll
        for (int i = 0; i < W; ++i) X[i] = BIAS2;              // row 0: Y = 0
        const uint4* in4 = reinterpret_cast<const uint4*>(bnd + wv.bnd_off + ((c - 1) & 1) * plane) + lane;
        uint4* out4 = reinterpret_cast<uint4*>(bnd + wv.bnd_off + (c & 1) * plane) + lane;
        uint32_t bprev = BIAS2;
        uint4 rc_n = rows4[0];
        uint4 b_n = c == 0 ? bias4 : in4[0];                    // every tile stores at least its first row group
        int last_bnd = 0;
        int gq = 0;
        for (; gq < ngq; ++gq) {
            const uint4 rcq = rc_n, b = b_n;
            const int gn = gq + 1 < ngq ? gq + 1 : gq;
            rc_n = rows4[(int64_t)gn * WAVE];
            b_n = (c > 0 && gn < stop_prev) ? in4[(int64_t)gn * WAVE] : bias4;   // past the left tile's stop: dead, Y = 0
            const uint32_t rcv[4] = {rcq.x, rcq.y, rcq.z, rcq.w}, bv[4] = {b.x, b.y, b.z, b.w};
            uint32_t bov[4] = {BIAS2, BIAS2, BIAS2, BIAS2};
            uint32_t alive_bits = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int j = 4 * gq + s;
                const uint32_t bcur = bv[s];            // Y(c*W, j); column 0 of the matrix: H = j*gap, Y = 0
                if (j >= 1 && j <= wv.ls_max) {         // wave-uniform
                    const uint32_t q = rcv[s];
                    uint32_t a = bprev + term(r[0], q);
                    uint32_t left = bcur - gd2;
                    uint32_t m0 = BIAS2, m1 = BIAS2;
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        uint32_t an = 0;
                        if (i + 1 < W) an = X[i] + term(r[i + 1], q);
                        X[i] = max3(a, X[i], left);
                        left = X[i] - gd2;
                        a = an;
                        if (i & 1) {
                            if (i & 2) m1 = max3(m1, X[i - 1], X[i]);
                            else m0 = max3(m0, X[i - 1], X[i]);
                        }
                    }
                    // padded columns and rows never beat the true maximum here (gap, mismatch <= 0): the
                    // row maximum needs no column mask, rows past a pair's sequence are skipped
                    const uint32_t m = max3(m0, m1, BIAS2);
                    const int off = (int)BIAS16 + prm.gd * j;           // H = Y + gap*j
                    const int h0 = (int)(m & 0xFFFFu) - off, h1 = (int)(m >> 16) - off;
                    if (j <= ln.ls[0]) best0 = max(best0, h0);
                    if (j <= ln.ls[1]) best1 = max(best1, h1);
                    bov[s] = X[W - 1];
                    // alive fields: x >= thr <=> max(x, thr - 1) != thr - 1; rows past a field's sequence compare with 0xFFFF
                    const uint32_t t0 = j <= ln.ls[0] ? (uint32_t)min(max(base0 + slope * j - 1, 0), 0xFFFE) : 0xFFFFu;
                    const uint32_t t1 = j <= ln.ls[1] ? (uint32_t)min(max(base1 + slope * j - 1, 0), 0xFFFE) : 0xFFFFu;
                    const uint32_t tm2 = t0 | (t1 << 16);
                    alive_bits |= pk_max_u16(m, tm2) ^ tm2;
                    if ((pk_max_u16(bov[s], tm2) ^ tm2) != 0u) last_bnd = j;
                }
                bprev = bcur;
            }
            if (c + 1 < wv.nch) out4[(int64_t)gq * WAVE] = make_uint4(bov[0], bov[1], bov[2], bov[3]);
            // a live boundary value at row l_in also enters row l_in + 1 (diagonal move): sweep past it
            if (4 * gq + 3 > l_in && __builtin_amdgcn_ballot_w64(alive_bits != 0u) == 0) { ++gq; break; }   // wave-uniform
        }
        stop_prev = gq;
        l_in = wave_max(last_bnd);
    }
    if (ln.out[0] >= 0) scores[ln.out[0]] = best0;
    if (ln.out[1] >= 0) scores[ln.out[1]] = best1;
}

// ---------------------------------------------------------------------------------------------
// int32 kernel: one pair per lane, any scores
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_WAVES * WAVE) void k_la32(const Lane* __restrict__ lanes, const Wave* __restrict__ waves, int n_waves,
                                                         const uint32_t* __restrict__ refcodes,
                                                         const uint32_t* __restrict__ rowcodes, uint32_t* __restrict__ bnd,
                                                         Params prm, int32_t* __restrict__ scores)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WG_WAVES + (threadIdx.x >> 6)));
    if (w >= n_waves) return;
    const int lane = threadIdx.x & 63;
    const Wave wv = waves[w];
    const Lane ln = lanes[(int64_t)w * WAVE + lane];
    const int64_t plane = (int64_t)wv.rows4 * WAVE;
    const uint32_t* rows = rowcodes + wv.row_off;
    int best = 0;
    for (int c = 0; c < wv.nch; ++c) {
        uint32_t r[W];
        const uint32_t* rc = refcodes + wv.ref_off + (int64_t)c * W * WAVE + lane;
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] = rc[(int64_t)i * WAVE];
        int X[W];
#pragma unroll
        for (int i = 0; i < W; ++i) X[i] = 0;                  // H(i,0) = 0
        const int nv = min(W, max(0, ln.lr[0] - c * W));        // columns of this tile that exist
        const int* in = reinterpret_cast<const int*>(bnd + wv.bnd_off + ((c - 1) & 1) * plane);
        int* out = reinterpret_cast<int*>(bnd + wv.bnd_off + (c & 1) * plane);
        int bprev = 0;
        for (int j = 1; j <= wv.ls_max; ++j) {                  // wave-uniform
            const uint32_t q = rows[rowidx(j, lane)];
            const int bcur = c == 0 ? j * prm.gap : in[rowidx(j, lane)];
            int a = bprev + (r[0] == q ? prm.match : prm.mismatch);
            int left = bcur + prm.gap;
            const bool count = j <= ln.ls[0];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                int an = 0;
                if (i + 1 < W) an = X[i] + (r[i + 1] == q ? prm.match : prm.mismatch);
                X[i] = max(a, max(X[i] + prm.gap, left));
                left = X[i] + prm.gap;
                a = an;
                if (count && i < nv) best = max(best, X[i]);
            }
            if (c + 1 < wv.nch) out[rowidx(j, lane)] = X[W - 1];
            bprev = bcur;
        }
    }
    if (ln.out[0] >= 0) scores[ln.out[0]] = best;
}

}  // namespace la

namespace {

using namespace la;

thread_local std::string g_err;
int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPL(call)                                                                                    \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(DSA_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_));      \
    } while (0)

template <class T>
struct Buf {
    T* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

// One launch group: the waves built from `order[begin..end)`, ITEMS pairs per lane.
struct Group {
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    int64_t ref_dwords = 0, row_dwords = 0, bnd_dwords = 0;
};

void add_wave(Group& g, const la_item* items, const int32_t* min_score, const int64_t* order, int64_t n, int items_per_lane)
{
    Wave wv{};
    wv.ref_off = g.ref_dwords;
    wv.row_off = g.row_dwords;
    wv.bnd_off = g.bnd_dwords;
    int lr_max = 0, ls_max = 0;
    for (int lane = 0; lane < WAVE; ++lane) {
        Lane ln{};
        for (int f = 0; f < 2; ++f) {
            ln.out[f] = -1;
            ln.need[f] = NO_NEED;
            const int64_t k = (int64_t)lane * items_per_lane + f;
            if (f < items_per_lane && k < n) {
                const la_item& it = items[order[k]];
                ln.out[f] = (int32_t)order[k];
                ln.lr[f] = it.ref_len;
                ln.ls[f] = it.seq_len;
                if (min_score) ln.need[f] = std::max(min_score[order[k]], NO_NEED);
                lr_max = std::max(lr_max, it.ref_len);
                ls_max = std::max(ls_max, it.seq_len);
            }
        }
        g.lanes.push_back(ln);
    }
    wv.nch = (lr_max + W - 1) / W;
    wv.ls_max = ls_max;
    wv.rows4 = (ls_max + 1 + 3) & ~3;
    g.ref_dwords += (int64_t)wv.nch * W * WAVE;
    g.row_dwords += (int64_t)wv.rows4 * WAVE;
    g.bnd_dwords += 2 * (int64_t)wv.rows4 * WAVE;
    g.waves.push_back(wv);
}

}  // namespace

extern "C" {

const char* la_last_error(void) { return g_err.c_str(); }

int la_align_batch(int device, int32_t match, int32_t mismatch, int32_t gap, const uint8_t* pool, int64_t pool_len,
                   const la_item* items, int64_t n_items, int32_t* scores, la_timing* timing)
{
    return la_align_batch_min(device, match, mismatch, gap, pool, pool_len, items, n_items, nullptr, scores, timing);
}

int la_align_batch_min(int device, int32_t match, int32_t mismatch, int32_t gap, const uint8_t* pool, int64_t pool_len,
                       const la_item* items, int64_t n_items, const int32_t* min_score, int32_t* scores, la_timing* timing)
{
    const auto t_begin = std::chrono::steady_clock::now();
    la_timing tm{};
    if (n_items < 0 || (n_items > 0 && (!items || !scores || !pool))) return fail(DSA_E_ARG, "null argument");
    if (n_items >= (int64_t)1 << 31) return fail(DSA_E_LIMIT, "more than 2^31-1 pairs");
    for (int64_t k = 0; k < n_items; ++k) {
        const la_item& it = items[k];
        if (it.ref_len < 0 || it.seq_len < 0 || it.ref_off < 0 || it.seq_off < 0 || it.ref_off + it.ref_len > pool_len ||
            it.seq_off + it.seq_len > pool_len)
            return fail(DSA_E_ARG, "pair %lld lies outside the pool", (long long)k);
        tm.cells += ((int64_t)it.ref_len + 1) * ((int64_t)it.seq_len + 1);
    }
    if (n_items == 0) {
        if (timing) *timing = tm;
        return DSA_OK;
    }
    HIPL(hipSetDevice(device));

    // Moves by mismatch are never better than two gaps when mismatch < 2*gap (same for match): clamping
    // gives the same matrix and keeps the packed kernel's diagonal term non-negative.
    Params prm{};
    prm.match = match;
    prm.mismatch = mismatch;
    prm.gap = gap;
    prm.gd = -gap;
    prm.dm = std::max(match, 2 * gap) - gap;
    prm.dx = std::max(mismatch, 2 * gap) - gap;
    prm.mp = std::max(match, 0);

    // longest sequences first; a wave takes consecutive pairs
    std::vector<int64_t> order((size_t)n_items);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        if (items[a].seq_len != items[b].seq_len) return items[a].seq_len > items[b].seq_len;
        if (items[a].ref_len != items[b].ref_len) return items[a].ref_len > items[b].ref_len;
        return a < b;
    });
    // the packed kernel takes the pairs whose values fit 16-bit fields
    const bool scores16 = gap <= 0 && mismatch <= 0 && prm.gd <= 1000 && prm.dm >= 0 && prm.dm <= 1000 && prm.dx >= 0 && prm.dx <= 1000;
    const int64_t step = std::max(prm.dm, std::max(prm.dx, 1));
    int64_t first16 = n_items;      // order[first16..) go to the packed kernel (sequence lengths descend)
    if (scores16) {
        first16 = 0;
        while (first16 < n_items && (int64_t)BIAS16 + (int64_t)items[order[first16]].seq_len * step > Y_LIMIT) ++first16;
    }
    tm.n_int32 = (int32_t)first16;
    tm.n_packed16 = (int32_t)(n_items - first16);

    Buf<uint8_t> d_pool;
    Buf<la_item> d_items;
    Buf<int32_t> d_scores;
    HIPL(d_pool.alloc((size_t)pool_len));
    HIPL(d_items.alloc((size_t)n_items));
    HIPL(d_scores.alloc((size_t)n_items));
    HIPL(hipMemcpy(d_pool.p, pool, (size_t)pool_len, hipMemcpyHostToDevice));
    HIPL(hipMemcpy(d_items.p, items, (size_t)n_items * sizeof(la_item), hipMemcpyHostToDevice));
    hipraii::Event ev[3];                 // destroyed on every return
    for (auto& e : ev) HIPL(e.create());

    size_t budget_dwords = (size_t)2 << 28;       // 2 GiB of planes per launch group
    if (const char* e = getenv("DEFUSE_LA_SCRATCH_MB")) budget_dwords = std::max<size_t>(1, (size_t)atoll(e)) << 18;

    auto run_range = [&](int64_t begin, int64_t end, int items_per_lane) -> int {
        const int64_t per_wave = (int64_t)WAVE * items_per_lane;
        int64_t k = begin;
        while (k < end) {
            Group g;
            while (k < end) {
                const int64_t n = std::min(per_wave, end - k);
                add_wave(g, items, min_score, order.data() + k, n, items_per_lane);
                k += n;
                if ((size_t)(g.ref_dwords + g.row_dwords + g.bnd_dwords) >= budget_dwords) break;
            }
            Buf<Lane> d_lanes;
            Buf<Wave> d_waves;
            Buf<uint32_t> d_ref, d_row, d_bnd;
            HIPL(d_lanes.alloc(g.lanes.size()));
            HIPL(d_waves.alloc(g.waves.size()));
            HIPL(d_ref.alloc((size_t)g.ref_dwords));
            HIPL(d_row.alloc((size_t)g.row_dwords));
            HIPL(d_bnd.alloc((size_t)g.bnd_dwords));
            HIPL(hipMemcpy(d_lanes.p, g.lanes.data(), g.lanes.size() * sizeof(Lane), hipMemcpyHostToDevice));
            HIPL(hipMemcpy(d_waves.p, g.waves.data(), g.waves.size() * sizeof(Wave), hipMemcpyHostToDevice));
            const int n_waves = (int)g.waves.size();
            HIPL(hipEventRecord(ev[0], nullptr));
            if (items_per_lane == 2)
                hipLaunchKernelGGL(k_pack<2>, dim3(n_waves), dim3(256), 0, nullptr, d_pool.p, d_items.p, d_lanes.p, d_waves.p, d_ref.p, d_row.p);
            else
                hipLaunchKernelGGL(k_pack<1>, dim3(n_waves), dim3(256), 0, nullptr, d_pool.p, d_items.p, d_lanes.p, d_waves.p, d_ref.p, d_row.p);
            HIPL(hipEventRecord(ev[1], nullptr));
            const unsigned grid = (unsigned)((n_waves + WG_WAVES - 1) / WG_WAVES);
            if (items_per_lane == 2)
                hipLaunchKernelGGL(k_la16, dim3(grid), dim3(WG_WAVES * WAVE), 0, nullptr, d_lanes.p, d_waves.p, n_waves, d_ref.p, d_row.p,
                                   d_bnd.p, prm, d_scores.p);
            else
                hipLaunchKernelGGL(k_la32, dim3(grid), dim3(WG_WAVES * WAVE), 0, nullptr, d_lanes.p, d_waves.p, n_waves, d_ref.p, d_row.p,
                                   d_bnd.p, prm, d_scores.p);
            HIPL(hipEventRecord(ev[2], nullptr));
            HIPL(hipEventSynchronize(ev[2]));
            HIPL(hipGetLastError());
            float ms = 0;
            HIPL(hipEventElapsedTime(&ms, ev[0], ev[1]));
            tm.pack_ms += ms;
            HIPL(hipEventElapsedTime(&ms, ev[1], ev[2]));
            tm.kernel_ms += ms;
        }
        return DSA_OK;
    };
    int rc = run_range(0, first16, 1);
    if (rc == DSA_OK) rc = run_range(first16, n_items, 2);
    if (rc != DSA_OK) return rc;
    HIPL(hipMemcpy(scores, d_scores.p, (size_t)n_items * sizeof(int32_t), hipMemcpyDeviceToHost));
    tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (timing) *timing = tm;
    return DSA_OK;
}

}  // extern "C"
