// dsa_kernels.hpp — gfx950 kernels of the split-read alignment path.
//
// What is computed (reference: tools/SplitReadAligner.cpp:24-75, :91-122, :156-298 and
// tools/SplitAlignment.cpp:371-400): for every candidate (fusion, read) two full semi-global DP
// matrices
//     M1 = Fill(ref1, read)            M2 = Fill(reverse(ref2), reverse(read))
//     H(i,0)=0, H(0,j)=-2j, H(i,j)=max(H(i-1,j-1)+(eq?2:-1), H(i-1,j)-2, H(i,j-1)-2)
// then for every read split a the row maxima m1(a), m2(Lq-a), the best a's, and for those rows
// every column that attains the maximum.
//
// How it is laid out for CDNA4 (DESIGN.md has the long version):
//   * one lane = one candidate pair; its two matrices are packed as 2 x int16 per VGPR (lo = M1,
//     hi = M2), so the recurrence needs no cross-lane traffic, every lane is busy whatever the
//     number of reads per fusion, and pair p simply lives in wave p/64, lane p%64.
//   * the matrices are swept in column tiles of W=64 reference positions held in 64 VGPRs; rows
//     (read bases) are the outer runtime loop.  State V(i,j) = H(i,j) + 2j makes the "left" move free:
//         V(i,j) = max( V(i-1,j-1) + (eq ? 4 : 1),  V(i-1,j) - 2,  V(i,j-1) )
//     pass 1 (descending i, in place)  X[i] = max(X[i-1] + d(i), X[i])
//     pass 2 (ascending i)             X[i] = max(X[i], X[i-1] - 2)
//   * fast path (reads over {A,C,G,T,N}): the substitution term d(i) of both matrices comes out of
//     a per-(fusion,tile) score table in LDS with one ds_read_b128 per 4 columns: 5 packed VALU ops
//     per 2 cells.  Generic path (any bytes): d(i) from xor/min: 8 ops.
//   * per (tile,row) the kernel stores the tile's row maximum and the tile's last column (the
//     boundary the next tile starts from).  The finish kernels pick the winning rows and replay only
//     the winning tiles from the stored boundaries to enumerate tied columns — exact, at a few
//     percent of the fill work instead of a second full pass.
//
// Everything here is integer; results are bit-exact with the reference by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/defuse_dsa.h"

namespace dsa {

constexpr int W = 64;                 // tile width (reference columns per register tile)
constexpr int WAVE = 64;
constexpr int WG_WAVES = 4;           // waves per workgroup of the fill kernels
constexpr int WG_LANES = WG_WAVES * WAVE;
constexpr uint32_t REF_PAD16 = 0x00AAu;    // never equals a read code (byte<<8) nor ROW_PAD16
constexpr uint32_t ROW_PAD16 = 0x0055u;

// fast path: read alphabet classes A,C,G,T,N -> 0..4 ; 25 (M1 base, M2 base) combinations per table
constexpr int NCLS = 5;
constexpr int NCOMBO = NCLS * NCLS;
constexpr int TROW = W + 4;           // table row stride in dwords (+4: rotate banks between rows)
constexpr int GMAX = 4;               // max distinct fusions per workgroup on the fast path
constexpr int TGROUP = NCOMBO * TROW; // dwords per fusion table
constexpr int T_PADCOL = -12000;      // substitution term of padded reference columns

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ v2u as_v2u(uint32_t x) { return __builtin_bit_cast(v2u, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2u vminu(v2u a, v2u b) { return __builtin_elementwise_min(a, b); }

struct WaveInfo {
    int32_t lq_max;        // longest read among the wave's pairs
    int32_t nch_max;       // most tiles among the wave's pairs (either matrix)
};

// Per workgroup (4 waves = 256 pairs): the distinct fusions of its pairs, for the fast path.
struct WgInfo {
    int32_t n_groups;              // 0 => not eligible for the fast path (more than GMAX fusions)
    int32_t group_f[GMAX];         // fusion_idx
};

// Geometry shared by all kernels of one run (one slice).  Pair p <-> wave p>>6, lane p&63.
struct Geom {
    int32_t n_waves;
    int32_t n_wgs;
    int32_t lq1;           // rows stride  = max read length + 1
    int32_t nch;           // chunk stride = max n_chunks
    int32_t lrp;           // refcodes stride = nch * W
    int32_t n_fusions;
    int64_t n_pairs;
};

struct KeptRow {           // one winning read split of a pair that has columns on both sides
    int16_t a;             // alignedToRef1
    int16_t m1, m2;        // row maxima (H units)
    int16_t pad_;
};

// A tile to re-run: all kept rows of one pair in one (matrix, chunk).
struct ReplayTask {
    uint32_t pair;         // slice-relative
    uint32_t mask_begin;   // masks[mask_begin + k] for kept row k of the pair
    uint16_t last_row;     // largest row (in this matrix) that needs a mask
    uint8_t  matrix;
    uint8_t  chunk;
};

struct PairState {
    int32_t max_score;     // best m1+m2 (0 = no output)
    int32_t n_kept;        // kept rows with columns on both sides
    uint32_t kept_begin;
    uint32_t task_begin;
    uint32_t n_tasks0;     // tasks of matrix 0 come first, then matrix 1
    uint32_t n_tasks1;
};

struct Counters {          // device-side allocation cursors (and overflow detection)
    unsigned long long n_kept, n_tasks, n_masks, pad_;
};

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int half_of(uint32_t v, int h) { return (int)(int16_t)(v >> (16 * h)); }

// ---------------------------------------------------------------------------------------------
// K0: byte -> packed code, code16 = byte<<8.
//   refcodes[f*lrp + i]        = { lo: ref0[i],            hi: ref1[len1-1-i] }   (pad beyond the end)
//   rowcodes[(w*lq1 + j)*64+l] = { lo: read[j-1],          hi: read[lq-j]     }   of pair w*64+l
// ---------------------------------------------------------------------------------------------
__global__ void k_pack_refs(const uint8_t* __restrict__ ref_bytes, const dsa_fusion* __restrict__ fusions,
                            uint32_t* __restrict__ refcodes, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.n_fusions * g.lrp;
    if (t >= total) return;
    const int i = (int)(t % g.lrp);
    const dsa_fusion f = fusions[t / g.lrp];
    uint32_t lo = REF_PAD16, hi = REF_PAD16;
    if (i < f.ref0_len) lo = (uint32_t)ref_bytes[(int64_t)f.ref0_off + i] << 8;
    if (i < f.ref1_len) hi = (uint32_t)ref_bytes[(int64_t)f.ref1_off + (f.ref1_len - 1 - i)] << 8;
    refcodes[t] = lo | (hi << 16);
}

__device__ __forceinline__ bool is_fast_base(uint32_t b)
{
    return b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N';
}

__global__ void k_pack_rows(const uint8_t* __restrict__ read_bytes, const dsa_pair* __restrict__ pairs,
                            uint32_t* __restrict__ rowcodes, uint32_t* __restrict__ wg_generic, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.n_waves * g.lq1 * WAVE;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int j = (int)((t >> 6) % g.lq1);
    const int w = (int)((t >> 6) / g.lq1);
    const int64_t p = (int64_t)w * WAVE + lane;
    uint32_t lo = ROW_PAD16, hi = ROW_PAD16;
    if (p < g.n_pairs && j >= 1) {
        const dsa_pair pr = pairs[p];
        if (j <= pr.read_len) {
            const uint32_t b0 = read_bytes[(int64_t)pr.read_off + (j - 1)];
            const uint32_t b1 = read_bytes[(int64_t)pr.read_off + (pr.read_len - j)];
            lo = b0 << 8;
            hi = b1 << 8;
            if (!is_fast_base(b0)) atomicOr(&wg_generic[w / WG_WAVES], 1u);   // b1 is some other row's b0
        }
    }
    rowcodes[t] = lo | (hi << 16);
}

// ---------------------------------------------------------------------------------------------
// Row step, generic scoring (any byte alphabet): per-lane reference codes in VGPRs.
// X[i] holds V(i0+i, j-1) on entry and V(i0+i, j) on exit.
//   bprev = V(i0-1, j-1), bcur = V(i0-1, j)   (the previous tile's last column; 0 for tile 0)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_step(v2s (&X)[W], const uint32_t (&r)[W], uint32_t cj, v2s bprev, v2s bcur)
{
    const v2u three = {3, 3};
    const v2s four = {4, 4};
    const v2s two = {2, 2};
    // pass 1: diagonal and left candidates, descending so X[i-1] is still the previous row
#pragma unroll
    for (int i = W - 1; i >= 1; --i) {
        v2u t = vminu(as_v2u(cj ^ r[i]), three);
        v2s a = (X[i - 1] - __builtin_bit_cast(v2s, t)) + four;
        X[i] = vmax(a, X[i]);
    }
    {
        v2u t = vminu(as_v2u(cj ^ r[0]), three);
        v2s a = (bprev - __builtin_bit_cast(v2s, t)) + four;
        X[0] = vmax(a, X[0]);
    }
    // pass 2: the gap-in-read chain along the reference
    X[0] = vmax(X[0], bcur - two);
#pragma unroll
    for (int i = 1; i < W; ++i) X[i] = vmax(X[i], X[i - 1] - two);
}

// tile row maximum with four interleaved accumulators (dependent packed ops are 1 wait state
// apart on gfx950, so keep them from sitting back to back).  MASKED: only columns < nv0 / nv1.
template <bool MASKED>
__device__ __forceinline__ v2s tile_row_max(const v2s (&X)[W], int nv0, int nv1)
{
    const v2s neg = {-32768, -32768};
    v2s acc4[4] = {neg, neg, neg, neg};
#pragma unroll
    for (int i = 0; i < W; ++i) {
        v2s x = X[i];
        if (MASKED) {
            if (i >= nv0) x.x = -32768;
            if (i >= nv1) x.y = -32768;
        }
        acc4[i & 3] = vmax(acc4[i & 3], x);
    }
    return vmax(vmax(acc4[0], acc4[1]), vmax(acc4[2], acc4[3]));
}

// ---------------------------------------------------------------------------------------------
// K1g: generic DP fill.  One wave = 64 pairs; 4 waves per workgroup.  Runs only the workgroups
// flagged generic (exotic read bytes or more than GMAX fusions in the workgroup).
//   cmax[((w*nch + c)*lq1 + j)*64 + lane] = max over the tile's valid columns of V(.,j)   (2 x i16)
//   bnd [((w*nch + c)*lq1 + j)*64 + lane] = V(last column of tile c, j)
// ---------------------------------------------------------------------------------------------
template <bool MASKED>
__device__ __forceinline__ void sweep_tile_generic(const uint32_t (&r)[W], const uint32_t* __restrict__ rows,
                                                   const uint32_t* __restrict__ bi, uint32_t* __restrict__ cm,
                                                   uint32_t* __restrict__ bo, int lq, bool first, int nv0, int nv1)
{
    v2s X[W];
#pragma unroll
    for (int i = 0; i < W; ++i) X[i] = (v2s){0, 0};
    v2s bprev = {0, 0};
    for (int j = 1; j <= lq; ++j) {
        const uint32_t cj = rows[(int64_t)j * WAVE];
        const v2s bcur = first ? (v2s){0, 0} : as_v2s(bi[(int64_t)j * WAVE]);
        row_step(X, r, cj, bprev, bcur);
        bprev = bcur;
        cm[(int64_t)j * WAVE] = as_u32(tile_row_max<MASKED>(X, nv0, nv1));
        bo[(int64_t)j * WAVE] = as_u32(X[W - 1]);
    }
}

__global__ __launch_bounds__(WG_LANES) void k_fill_generic(const dsa_pair* __restrict__ pairs,
                                                           const WaveInfo* __restrict__ winfo,
                                                           const dsa_fusion* __restrict__ fusions,
                                                           const uint32_t* __restrict__ wg_generic,
                                                           const uint32_t* __restrict__ refcodes,
                                                           const uint32_t* __restrict__ rowcodes,
                                                           uint32_t* __restrict__ bnd, uint32_t* __restrict__ cmax, Geom g)
{
    if (wg_generic[blockIdx.x] == 0) return;     // the fast kernel owns this workgroup
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WG_WAVES + (threadIdx.x >> 6)));
    if (w >= g.n_waves) return;
    const int lane = threadIdx.x & 63;
    const int64_t p = min((int64_t)w * WAVE + lane, g.n_pairs - 1);   // tail lanes shadow the last pair
    const WaveInfo wi = winfo[w];
    const int f = pairs[p].fusion_idx;
    const dsa_fusion fu = fusions[f];
    const uint32_t* rc = refcodes + (int64_t)f * g.lrp;
    const uint32_t* rows = rowcodes + (int64_t)w * g.lq1 * WAVE + lane;

    for (int c = 0; c < wi.nch_max; ++c) {
        uint32_t r[W];
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] = rc[c * W + i];
        const int nv0 = fu.ref0_len - c * W, nv1 = fu.ref1_len - c * W;   // per lane; may be <= 0
        uint32_t* cm = cmax + ((int64_t)w * g.nch + c) * g.lq1 * WAVE + lane;
        uint32_t* bo = bnd + ((int64_t)w * g.nch + c) * g.lq1 * WAVE + lane;
        const uint32_t* bi = bnd + ((int64_t)w * g.nch + (c - 1)) * g.lq1 * WAVE + lane;
        if (__builtin_amdgcn_ballot_w64(nv0 < W || nv1 < W) == 0)
            sweep_tile_generic<false>(r, rows, bi, cm, bo, wi.lq_max, c == 0, nv0, nv1);
        else
            sweep_tile_generic<true>(r, rows, bi, cm, bo, wi.lq_max, c == 0, nv0, nv1);
    }
}

// ---------------------------------------------------------------------------------------------
// K1f: fast DP fill (reads over {A,C,G,T,N}).  Per workgroup and tile, the substitution terms of
// every fusion present are tabulated in LDS:
//     T[g][k1*5+k2][i] = { d(ref0_g[i], base[k1]), d(rev(ref1_g)[i], base[k2]) },  d = eq ? 4 : 1
// where k1/k2 are the classes of the M1 / M2 read base of the row.  Padded reference columns get a
// large negative term, which keeps them strictly below every row maximum, so the tile row maximum
// needs no masking.  A row then costs one ds_read_b128 per 4 columns and 5 packed VALU ops per column.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_class(uint32_t byte)   // A,C,T,G,N -> 0,1,2,3,4
{
    uint32_t k = (byte >> 1) & 7u;       // A:0 C:1 T:2 G:3 N:7
    return k > 4u ? 4u : k;
}

__global__ __launch_bounds__(WG_LANES, 4) void k_fill_fast(const dsa_pair* __restrict__ pairs,
                                                           const WaveInfo* __restrict__ winfo,
                                                           const WgInfo* __restrict__ wginfo,
                                                           const uint32_t* __restrict__ wg_generic,
                                                           const uint32_t* __restrict__ refcodes,
                                                           const uint32_t* __restrict__ rowcodes,
                                                           uint32_t* __restrict__ bnd, uint32_t* __restrict__ cmax, Geom g)
{
    __shared__ __attribute__((aligned(16))) uint32_t T[GMAX * TGROUP];
    __shared__ int s_nch;
    if (wg_generic[blockIdx.x] != 0) return;     // the generic kernel owns this workgroup (uniform)
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WG_WAVES + (threadIdx.x >> 6)));
    const bool live = w < g.n_waves;             // whole waves past the end still join the barriers
    const int lane = threadIdx.x & 63;
    const WgInfo wgi = wginfo[blockIdx.x];
    WaveInfo wi = {0, 0};
    int f = 0;
    if (live) {
        wi = winfo[w];
        f = pairs[min((int64_t)w * WAVE + lane, g.n_pairs - 1)].fusion_idx;
    }
    if (threadIdx.x == 0) s_nch = 0;
    __syncthreads();
    if (lane == 0 && live) atomicMax(&s_nch, wi.nch_max);
    __syncthreads();
    const int nch_wg = s_nch;

    int gsel = 0;
#pragma unroll
    for (int k = 0; k < GMAX; ++k)
        if (k < wgi.n_groups && wgi.group_f[k] == f) gsel = k;
    const uint32_t* tb = T + gsel * TGROUP;
    const uint32_t* rows = rowcodes + (int64_t)w * g.lq1 * WAVE + lane;

    for (int c = 0; c < nch_wg; ++c) {
        __syncthreads();                          // previous tile's tables no longer in use
        for (int e = threadIdx.x; e < wgi.n_groups * NCOMBO * W; e += WG_LANES) {
            const int i = e & (W - 1);
            const int combo = (e >> 6) % NCOMBO;
            const int gi = (e >> 6) / NCOMBO;
            const uint32_t code = refcodes[(int64_t)wgi.group_f[gi] * g.lrp + c * W + i];
            const uint32_t c0 = code & 0xFFFFu, c1 = code >> 16;
            const uint32_t cls_byte[NCLS] = {'A', 'C', 'T', 'G', 'N'};   // inverse of base_class
            const int d0 = c0 == REF_PAD16 ? T_PADCOL : ((c0 >> 8) == cls_byte[combo / NCLS] ? 4 : 1);
            const int d1 = c1 == REF_PAD16 ? T_PADCOL : ((c1 >> 8) == cls_byte[combo % NCLS] ? 4 : 1);
            T[gi * TGROUP + combo * TROW + i] = (uint32_t)(uint16_t)d0 | ((uint32_t)(uint16_t)d1 << 16);
        }
        __syncthreads();
        if (!live || c >= wi.nch_max) continue;   // wave-uniform

        uint32_t* cm = cmax + ((int64_t)w * g.nch + c) * g.lq1 * WAVE + lane;
        uint32_t* bo = bnd + ((int64_t)w * g.nch + c) * g.lq1 * WAVE + lane;
        const uint32_t* bi = bnd + ((int64_t)w * g.nch + (c - 1)) * g.lq1 * WAVE + lane;
        v2s X[W];
#pragma unroll
        for (int i = 0; i < W; ++i) X[i] = (v2s){0, 0};
        v2s bprev = {0, 0};
        const v2s two = {2, 2};
        for (int j = 1; j <= wi.lq_max; ++j) {
            const uint32_t cj = rows[(int64_t)j * WAVE];
            const v2s bcur = (c == 0) ? (v2s){0, 0} : as_v2s(bi[(int64_t)j * WAVE]);
            const uint32_t combo = base_class((cj >> 8) & 0xFFu) * NCLS + base_class(cj >> 24);
            const uint4* trow = reinterpret_cast<const uint4*>(tb + combo * TROW);
            // pass 1, descending, four columns per LDS read
#pragma unroll
            for (int q = W / 4 - 1; q >= 0; --q) {
                const uint4 v = trow[q];
                X[4 * q + 3] = vmax(X[4 * q + 2] + as_v2s(v.w), X[4 * q + 3]);
                X[4 * q + 2] = vmax(X[4 * q + 1] + as_v2s(v.z), X[4 * q + 2]);
                X[4 * q + 1] = vmax(X[4 * q + 0] + as_v2s(v.y), X[4 * q + 1]);
                if (q > 0)
                    X[4 * q] = vmax(X[4 * q - 1] + as_v2s(v.x), X[4 * q]);
                else
                    X[0] = vmax(bprev + as_v2s(v.x), X[0]);
            }
            X[0] = vmax(X[0], bcur - two);
#pragma unroll
            for (int i = 1; i < W; ++i) X[i] = vmax(X[i], X[i - 1] - two);
            bprev = bcur;
            cm[(int64_t)j * WAVE] = as_u32(tile_row_max<false>(X, W, W));
            bo[(int64_t)j * WAVE] = as_u32(X[W - 1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Finish stage
// ---------------------------------------------------------------------------------------------
// Row maximum of one matrix of pair p in H units with FindMaxRowEntry's acceptance rule
// (tools/SplitReadAligner.cpp:91-102): values below minSplitScore (8) count as 0.
__device__ __forceinline__ int row_max_h(const uint32_t* __restrict__ cmax, const Geom& g, int64_t p, int h,
                                         int n_chunks, int row)
{
    if (row == 0 || n_chunks == 0) return 0;   // H(i,0)=0 < 8; empty reference: only column 0 (<=0)
    const int64_t w = p >> 6;
    const int lane = (int)(p & 63);
    int v = -32768;
    for (int c = 0; c < n_chunks; ++c) {
        int x = half_of(cmax[((w * g.nch + c) * g.lq1 + row) * WAVE + lane], h);
        v = x > v ? x : v;
    }
    v -= 2 * row;
    return v >= DSA_MIN_SPLIT ? v : 0;
}

// wave-aggregated allocation: every lane asks for n items, one atomic per wave
__device__ __forceinline__ unsigned long long wave_alloc(unsigned long long* counter, unsigned n)
{
    unsigned incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned y = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += y;
    }
    const unsigned total = __shfl(incl, 63, 64);
    unsigned long long base = 0;
    if ((threadIdx.x & 63) == 63 && total) base = atomicAdd(counter, (unsigned long long)total);
    base = __shfl(base, 63, 64);
    return base + (incl - n);
}

// K2: per pair, the winning read splits (tools/SplitReadAligner.cpp:194-223), the kept rows that
// have columns on both sides, and the tiles that hold a row maximum for them.  Single pass; space
// comes from device cursors (capacities are checked by the host afterwards).
template <int COMBINE_THREADS>
__global__ __launch_bounds__(COMBINE_THREADS) void k_combine(
    const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions, const uint32_t* __restrict__ cmax,
    const int32_t* __restrict__ min_score_tab, PairState* __restrict__ state, KeptRow* __restrict__ kept,
    uint64_t kept_cap, ReplayTask* __restrict__ tasks, uint64_t task_cap, uint64_t mask_cap,
    Counters* __restrict__ ctr, Geom g)
{
    extern __shared__ int16_t s_m[];   // [2][lq1][COMBINE_THREADS]: m1(a), m2(b) per thread
    const int tid = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * COMBINE_THREADS + tid;
    const bool active = p < g.n_pairs;
    int lq = 0, nc0 = 0, nc1 = 0, max_score = 0, n_kept = 0;
    unsigned n_t0 = 0, n_t1 = 0;
    int16_t* m1s = s_m + tid;
    int16_t* m2s = s_m + (size_t)g.lq1 * COMBINE_THREADS + tid;
    constexpr int S = COMBINE_THREADS;
    if (active) {
        const dsa_pair pr = pairs[p];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        lq = pr.read_len;
        nc0 = cdiv_dev(fu.ref0_len, W);
        nc1 = cdiv_dev(fu.ref1_len, W);
        const int min_score = min_score_tab[lq];
        for (int a = 0; a <= lq; ++a) {
            m1s[(size_t)a * S] = (int16_t)row_max_h(cmax, g, p, 0, nc0, a);
            m2s[(size_t)a * S] = (int16_t)row_max_h(cmax, g, p, 1, nc1, a);
        }
        for (int a = 0; a <= lq; ++a) {
            const int s = m1s[(size_t)a * S] + m2s[(size_t)(lq - a) * S];
            if (s >= min_score && s > max_score) max_score = s;
        }
        if (max_score != 0) {
            for (int a = 0; a <= lq; ++a) {
                const int m1 = m1s[(size_t)a * S], m2 = m2s[(size_t)(lq - a) * S];
                if (m1 + m2 == max_score && m1 != 0 && m2 != 0) ++n_kept;   // an empty side emits nothing
            }
        }
    }
    // tiles that attain the maximum at some kept row (bitmaps; references with more than 64 tiles
    // replay every tile instead: exact, just not minimal)
    uint64_t tiles0 = 0, tiles1 = 0;
    const bool small = nc0 <= 64 && nc1 <= 64;
    if (active && n_kept > 0) {
        const int64_t w = p >> 6;
        const int lane = (int)(p & 63);
        if (small) {
            for (int a = 0; a <= lq; ++a) {
                const int b = lq - a;
                const int m1 = m1s[(size_t)a * S], m2 = m2s[(size_t)b * S];
                if (m1 + m2 != max_score || m1 == 0 || m2 == 0) continue;
                for (int c = 0; c < nc0; ++c)
                    if (half_of(cmax[((w * g.nch + c) * g.lq1 + a) * WAVE + lane], 0) == m1 + 2 * a) tiles0 |= 1ull << c;
                for (int c = 0; c < nc1; ++c)
                    if (half_of(cmax[((w * g.nch + c) * g.lq1 + b) * WAVE + lane], 1) == m2 + 2 * b) tiles1 |= 1ull << c;
            }
            n_t0 = (unsigned)__builtin_popcountll(tiles0);
            n_t1 = (unsigned)__builtin_popcountll(tiles1);
        } else {
            n_t0 = (unsigned)nc0;
            n_t1 = (unsigned)nc1;
        }
    }
    const unsigned n_tasks = n_t0 + n_t1;
    const unsigned long long kb = wave_alloc(&ctr->n_kept, (unsigned)n_kept);
    const unsigned long long tb = wave_alloc(&ctr->n_tasks, n_tasks);
    const unsigned long long mb = wave_alloc(&ctr->n_masks, n_tasks * (unsigned)n_kept);
    if (!active) return;
    PairState st;
    st.max_score = max_score;
    st.n_kept = n_kept;
    st.kept_begin = (uint32_t)kb;
    st.task_begin = (uint32_t)tb;
    st.n_tasks0 = n_t0;
    st.n_tasks1 = n_t1;
    if (n_kept > 0 &&
        (kb + n_kept > kept_cap || tb + n_tasks > task_cap || mb + (unsigned long long)n_tasks * n_kept > mask_cap)) {
        st.n_kept = 0;     // overflow: the host sees the cursors, grows the buffers and reruns the finish stage
        state[p] = st;
        return;
    }
    state[p] = st;
    if (n_kept == 0) return;
    int k = 0, last_a = 0, first_a = lq;
    for (int a = 0; a <= lq; ++a) {
        const int m1 = m1s[(size_t)a * S], m2 = m2s[(size_t)(lq - a) * S];
        if (m1 + m2 != max_score || m1 == 0 || m2 == 0) continue;
        KeptRow kr;
        kr.a = (int16_t)a;
        kr.m1 = (int16_t)m1;
        kr.m2 = (int16_t)m2;
        kr.pad_ = 0;
        kept[kb + k] = kr;
        if (k == 0) first_a = a;
        last_a = a;
        ++k;
    }
    unsigned t = 0;
    for (int m = 0; m < 2; ++m) {
        const int nc = m ? nc1 : nc0;
        const uint64_t tiles = m ? tiles1 : tiles0;
        for (int c = 0; c < nc; ++c) {
            if (small && !((tiles >> c) & 1ull)) continue;
            ReplayTask rt;
            rt.pair = (uint32_t)p;
            rt.mask_begin = (uint32_t)(mb + (unsigned long long)t * n_kept);
            rt.last_row = (uint16_t)(m ? (lq - first_a) : last_a);
            rt.matrix = (uint8_t)m;
            rt.chunk = (uint8_t)c;
            tasks[tb + t] = rt;
            ++t;
        }
    }
}

// K3: replay one tile per lane from the stored boundary; for every kept row of the pair report, as a
// 64-bit mask, the valid columns whose value equals the row maximum.  Grid-stride over the device
// task counter (no host round trip between combine and replay).
__global__ __launch_bounds__(256) void k_replay(const ReplayTask* __restrict__ tasks, uint64_t task_cap,
                                                const Counters* __restrict__ ctr, const PairState* __restrict__ state,
                                                const KeptRow* __restrict__ kept, uint64_t kept_cap,
                                                const dsa_pair* __restrict__ pairs,
                                                const dsa_fusion* __restrict__ fusions,
                                                const uint32_t* __restrict__ refcodes,
                                                const uint32_t* __restrict__ rowcodes,
                                                const uint32_t* __restrict__ bnd, uint64_t* __restrict__ masks,
                                                uint64_t mask_cap, Geom g)
{
    const unsigned long long n_tasks = ctr->n_tasks;
    if (n_tasks > task_cap || ctr->n_masks > mask_cap || ctr->n_kept > kept_cap) return;   // overflow run
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_tasks;
         t += (unsigned long long)gridDim.x * blockDim.x) {
        const ReplayTask rt = tasks[t];
        const int64_t p = rt.pair;
        const int64_t w = p >> 6;
        const int lane = (int)(p & 63), h = rt.matrix, c = rt.chunk;
        const dsa_pair pr = pairs[p];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        const int lr = h ? fu.ref1_len : fu.ref0_len;
        const PairState st = state[p];
        const int lq = pr.read_len;
        const uint32_t* rc = refcodes + (int64_t)pr.fusion_idx * g.lrp + c * W;
        const uint32_t* rows = rowcodes + w * g.lq1 * WAVE + lane;
        const uint32_t* bi = bnd + (w * g.nch + (c - 1)) * g.lq1 * WAVE + lane;
        const KeptRow* kr = kept + st.kept_begin;

        uint32_t r[W];
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] = rc[i];
        v2s X[W];
#pragma unroll
        for (int i = 0; i < W; ++i) X[i] = (v2s){0, 0};
        v2s bprev = {0, 0};
        const int R = rt.last_row;
        const int nvalid = min(W, lr - c * W);
        // kept rows ascend in a: matrix 0 meets them in order k=0.., matrix 1 (row = lq-a) in reverse
        int k = h ? st.n_kept - 1 : 0;
        const int kstep = h ? -1 : 1;
        for (int j = 1; j <= R; ++j) {
            const uint32_t cj = rows[(int64_t)j * WAVE];
            const v2s bcur = (c > 0) ? as_v2s(bi[(int64_t)j * WAVE]) : (v2s){0, 0};
            row_step(X, r, cj, bprev, bcur);
            bprev = bcur;
            if (k >= 0 && k < st.n_kept) {
                const KeptRow kk = kr[k];
                const int row = h ? lq - kk.a : kk.a;
                if (row == j) {
                    const int target = (h ? kk.m2 : kk.m1) + 2 * j;
                    uint64_t mask = 0;
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        const int v = h ? (int)X[i].y : (int)X[i].x;
                        if (i < nvalid && v == target) mask |= (1ull << i);
                    }
                    masks[rt.mask_begin + k] = mask;
                    k += kstep;
                }
            }
        }
    }
}

// K4: emit.  For every kept split a (ascending) the cross product columns1 x columns2 in ascending
// order (tools/SplitReadAligner.cpp:233-269), then the refSplit de-duplication of
// tools/SplitAlignment.cpp:381-391 (first occurrence wins).  WRITE=false counts.
__device__ __forceinline__ bool col_in(const ReplayTask* tasks, const uint64_t* masks, uint32_t tb, uint32_t te,
                                       int k, int col /*1-based matrix column*/)
{
    const int c = (col - 1) / W, bit = (col - 1) % W;
    for (uint32_t q = tb; q < te; ++q)
        if (tasks[q].chunk == c) return (masks[tasks[q].mask_begin + k] >> bit) & 1ull;
    return false;
}

template <bool WRITE>
__global__ void k_emit(const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions,
                       const PairState* __restrict__ state, const KeptRow* __restrict__ kept,
                       const ReplayTask* __restrict__ tasks, const uint64_t* __restrict__ masks,
                       int64_t* __restrict__ rec_count, const int64_t* __restrict__ rec_offset,
                       dsa_record* __restrict__ out, uint64_t out_cap, Geom g)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.n_pairs) return;
    const PairState st = state[p];
    int64_t n = 0;
    if (st.n_kept > 0) {
        const dsa_pair pr = pairs[p];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        const uint32_t t0b = st.task_begin, t0e = t0b + st.n_tasks0, t1e = t0e + st.n_tasks1;
        int64_t wr = WRITE ? rec_offset[p] : 0;
        if (WRITE && (uint64_t)rec_offset[p + 1] > out_cap) return;   // host grows the buffer and reruns emit
        for (int k = 0; k < st.n_kept; ++k) {
            const KeptRow kr = kept[st.kept_begin + k];
            for (uint32_t q1 = t0b; q1 < t0e; ++q1) {
                uint64_t m1 = masks[tasks[q1].mask_begin + k];
                while (m1) {
                    const int i1 = tasks[q1].chunk * W + __builtin_ctzll(m1) + 1;
                    m1 &= m1 - 1;
                    for (uint32_t q2 = t0e; q2 < t1e; ++q2) {
                        uint64_t m2 = masks[tasks[q2].mask_begin + k];
                        while (m2) {
                            const int i2 = tasks[q2].chunk * W + __builtin_ctzll(m2) + 1;
                            m2 &= m2 - 1;
                            bool dup = false;     // same refSplit <=> same (i1,i2) at an earlier kept a
                            for (int k2 = 0; k2 < k && !dup; ++k2)
                                dup = col_in(tasks, masks, t0b, t0e, k2, i1) && col_in(tasks, masks, t0e, t1e, k2, i2);
                            if (dup) continue;
                            if (WRITE) {
                                dsa_record rec;
                                rec.fusion_id = fu.fusion_id;
                                rec.frag = pr.frag;
                                rec.read_end = pr.read_end;
                                rec.revcomp = pr.revcomp;
                                rec.ref_first = i1;
                                rec.ref_second = fu.ref1_len - i2 - 1;
                                rec.read_first = kr.a;
                                rec.read_second = pr.read_len - kr.a;
                                rec.score = kr.m1 < kr.m2 ? kr.m1 : kr.m2;
                                out[wr] = rec;
                            }
                            ++wr;
                            ++n;
                        }
                    }
                }
            }
        }
    }
    if (!WRITE) rec_count[p] = n;
}

}  // namespace dsa
