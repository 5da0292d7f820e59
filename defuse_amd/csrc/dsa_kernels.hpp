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
//   * inter-sequence parallelism: one lane = two DP problems (two reads of the same fusion) packed
//     as 2 x int16 in one VGPR, so the recurrence needs no cross-lane traffic and the reference
//     bases of a wave are wave-uniform (SGPR operands).
//   * the matrix is swept in column tiles of W reference positions held in W VGPRs; rows (read
//     bases) are the outer runtime loop.  State V(i,j) = H(i,j) + 2j makes the "left" move free:
//         V(i,j) = max( V(i-1,j-1) + (eq ? 4 : 1),  V(i-1,j) - 2,  V(i,j-1) )
//     pass 1 (descending i, in place)  X[i] = max(X[i-1] + d(i), X[i])
//     pass 2 (ascending i)             X[i] = max(X[i], X[i-1] - 2)
//   * per (tile,row) the kernel stores the tile's row maximum and the tile's last column (the
//     boundary the next tile starts from).  The finish kernels pick the winning rows and replay only
//     the winning tiles from the stored boundaries to enumerate tied columns — exact, and ~5 % of
//     the fill work instead of a second full pass.
//
// Everything here is integer; results are bit-exact with the reference by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/defuse_dsa.h"

namespace dsa {

constexpr int W = 64;                 // tile width (reference columns per register tile)
constexpr int WAVE = 64;
constexpr int PAIRS_PER_WTASK = 128;  // two reads per lane
constexpr uint32_t REF_PAD = 0x00AA00AAu;  // never equals a read code (byte<<8) nor ROW_PAD
constexpr uint32_t ROW_PAD = 0x00550055u;

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ v2u as_v2u(uint32_t x) { return __builtin_bit_cast(v2u, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2u vminu(v2u a, v2u b) { return __builtin_elementwise_min(a, b); }

// One wave task = one (block of <=128 pairs of one fusion) x (matrix 0 or 1).
struct WTask {
    int32_t pair_base;     // first pair of the block
    int32_t n_pairs;       // 1..128; lane l holds pairs base+l (lo half) and base+64+l (hi half)
    int32_t fusion_idx;
    int32_t matrix;        // 0: ref0 vs read; 1: reverse(ref1) vs reverse(read)
    int32_t lr;            // reference length of this matrix
    int32_t lq_max;        // longest read in the block
    int32_t n_chunks;      // ceil(lr / W)
    int32_t pad_;
};

// Geometry shared by all kernels of one run.
struct Geom {
    int32_t n_wtasks;
    int32_t n_blocks;      // n_wtasks / 2
    int32_t lq1;           // rows stride  = max read length + 1
    int32_t nch;           // chunk stride = max n_chunks
    int32_t lrp;           // refcodes stride = nch * W
    int32_t n_fusions;
    int64_t n_pairs;
};

// A tile to re-run: find all columns of (matrix, chunk) whose value at `row` equals `target`.
struct ReplayTask {
    uint32_t slot;         // block * 128 + (half*64 + lane)
    uint16_t row;          // 1..lq
    int16_t  target;       // V units (H + 2*row)
    uint8_t  matrix;
    uint8_t  chunk;
    uint8_t  pad_[2];
    uint32_t kept_idx;     // index of the kept split a within the pair (0-based)
};

struct PairState {
    int32_t max_score;     // best m1+m2 (0 = no output)
    int32_t n_kept;        // number of kept a's that have both sides non-zero
    int64_t task_begin;    // into ReplayTask[] (filled after the scan)
};

// ---------------------------------------------------------------------------------------------
// K0: byte -> packed code.  code = byte<<8 replicated into both int16 halves for references;
// for rows, lo half = read of lane l, hi half = read of lane l+64 (second pair of the lane).
// ---------------------------------------------------------------------------------------------
__global__ void k_pack_refs(const uint8_t* __restrict__ ref_bytes, const dsa_fusion* __restrict__ fusions,
                            uint32_t* __restrict__ refcodes, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.n_fusions * 2 * g.lrp;
    if (t >= total) return;
    int i = (int)(t % g.lrp);
    int fm = (int)(t / g.lrp);
    int m = fm & 1;
    const dsa_fusion f = fusions[fm >> 1];
    int len = m ? f.ref1_len : f.ref0_len;
    uint32_t code = REF_PAD;
    if (i < len) {
        uint32_t b = m ? ref_bytes[(int64_t)f.ref1_off + (len - 1 - i)] : ref_bytes[(int64_t)f.ref0_off + i];
        code = (b << 8) | (b << 24);
    }
    refcodes[t] = code;
}

__global__ void k_pack_rows(const uint8_t* __restrict__ read_bytes, const dsa_pair* __restrict__ pairs,
                            const WTask* __restrict__ wtasks, uint32_t* __restrict__ rowcodes, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.n_wtasks * g.lq1 * WAVE;
    if (t >= total) return;
    int lane = (int)(t & 63);
    int j = (int)((t >> 6) % g.lq1);
    int w = (int)((t >> 6) / g.lq1);
    const WTask wt = wtasks[w];
    uint32_t code = 0;
    for (int h = 0; h < 2; ++h) {
        uint32_t c = ROW_PAD & 0xFFFFu;
        int slot = h * 64 + lane;
        if (slot < wt.n_pairs && j >= 1) {
            const dsa_pair p = pairs[wt.pair_base + slot];
            if (j <= p.read_len) {
                uint32_t b = wt.matrix ? read_bytes[(int64_t)p.read_off + (p.read_len - j)]
                                       : read_bytes[(int64_t)p.read_off + (j - 1)];
                c = b << 8;
            }
        }
        code |= c << (16 * h);
    }
    rowcodes[t] = code;
}

// ---------------------------------------------------------------------------------------------
// The row step shared by the fill kernel (wave-uniform reference in SGPRs) and the replay kernel
// (per-lane reference in VGPRs).  X[i] holds V(i0+i, j-1) on entry and V(i0+i, j) on exit.
//   bprev = V(i0-1, j-1), bcur = V(i0-1, j)   (the previous tile's last column; 0 for tile 0)
// ---------------------------------------------------------------------------------------------
template <typename RefT>
__device__ __forceinline__ void row_step(v2s (&X)[W], const RefT (&r)[W], uint32_t cj, v2s bprev, v2s bcur)
{
    const v2u three = {3, 3};
    const v2s four = {4, 4};
    const v2s two = {2, 2};
    // pass 1: diagonal and left candidates, descending so X[i-1] is still the previous row
#pragma unroll
    for (int i = W - 1; i >= 1; --i) {
        v2u t = vminu(as_v2u(cj ^ (uint32_t)r[i]), three);
        v2s a = (X[i - 1] - __builtin_bit_cast(v2s, t)) + four;
        X[i] = vmax(a, X[i]);
    }
    {
        v2u t = vminu(as_v2u(cj ^ (uint32_t)r[0]), three);
        v2s a = (bprev - __builtin_bit_cast(v2s, t)) + four;
        X[0] = vmax(a, X[0]);
    }
    // pass 2: the gap-in-read chain along the reference
    X[0] = vmax(X[0], bcur - two);
#pragma unroll
    for (int i = 1; i < W; ++i) X[i] = vmax(X[i], X[i - 1] - two);
}

// ---------------------------------------------------------------------------------------------
// K1: DP fill.  One wave per WTask; 4 waves per workgroup (independent).
//   cmax[((w*nch + c)*lq1 + j)*64 + lane] = max over the tile's valid columns of V(.,j)   (2 x i16)
//   bnd [((w*nch + c)*lq1 + j)*64 + lane] = V(last column of tile c, j)
// ---------------------------------------------------------------------------------------------
template <bool TAIL>
__device__ __forceinline__ void sweep_tile(const uint32_t (&r)[W], const uint32_t* __restrict__ rows,
                                           const uint32_t* __restrict__ bi, uint32_t* __restrict__ cm,
                                           uint32_t* __restrict__ bo, int lq, bool first, int nvalid)
{
    v2s X[W];
#pragma unroll
    for (int i = 0; i < W; ++i) X[i] = (v2s){0, 0};
    v2s bprev = {0, 0};
    for (int j = 1; j <= lq; ++j) {
        const uint32_t cj = rows[(int64_t)j * WAVE];
        const v2s bcur = first ? (v2s){0, 0} : as_v2s(bi[(int64_t)j * WAVE]);
        row_step<uint32_t>(X, r, cj, bprev, bcur);
        bprev = bcur;
        // tile row maximum over the valid columns; four interleaved accumulators so that dependent
        // packed ops (1 wait state apart on gfx950) never sit back to back
        v2s acc4[4] = {X[0], X[0], X[0], X[0]};
#pragma unroll
        for (int i = 1; i < W; ++i) {
            if (!TAIL) {
                acc4[i & 3] = vmax(acc4[i & 3], X[i]);
            } else if (i < nvalid) {   // wave-uniform
                acc4[i & 3] = vmax(acc4[i & 3], X[i]);
            }
        }
        const v2s acc = vmax(vmax(acc4[0], acc4[1]), vmax(acc4[2], acc4[3]));
        cm[(int64_t)j * WAVE] = as_u32(acc);
        bo[(int64_t)j * WAVE] = as_u32(X[W - 1]);
    }
}

__global__ __launch_bounds__(256) void k_fill(const WTask* __restrict__ wtasks,
                                              const uint32_t* __restrict__ refcodes,
                                              const uint32_t* __restrict__ rowcodes,
                                              uint32_t* __restrict__ bnd, uint32_t* __restrict__ cmax, Geom g)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (w >= g.n_wtasks) return;
    const int lane = threadIdx.x & 63;
    const WTask wt = wtasks[w];
    const int lq = wt.lq_max;
    const uint32_t* rc = refcodes + ((int64_t)wt.fusion_idx * 2 + wt.matrix) * g.lrp;
    const uint32_t* rows = rowcodes + (int64_t)w * g.lq1 * WAVE + lane;

    for (int c = 0; c < wt.n_chunks; ++c) {
        // The tile's reference codes are wave-uniform; they are kept in VGPRs on purpose: 64 SGPRs
        // would not fit next to the rest of the kernel's scalar state (the compiler then spills
        // them to lanes and pays v_readlane + hazard nops per column).
        uint32_t r[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            r[i] = rc[c * W + i];
            asm volatile("" : "+v"(r[i]));
        }
        const int nvalid = min(W, wt.lr - c * W);     // wave-uniform; < W only in the last tile
        uint32_t* cm = cmax + ((int64_t)w * g.nch + c) * g.lq1 * WAVE + lane;
        uint32_t* bo = bnd + ((int64_t)w * g.nch + c) * g.lq1 * WAVE + lane;
        const uint32_t* bi = bnd + ((int64_t)w * g.nch + (c - 1)) * g.lq1 * WAVE + lane;
        if (nvalid == W)
            sweep_tile<false>(r, rows, bi, cm, bo, lq, c == 0, nvalid);
        else
            sweep_tile<true>(r, rows, bi, cm, bo, lq, c == 0, nvalid);
    }
}

// ---------------------------------------------------------------------------------------------
// Finish stage helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int half_of(uint32_t v, int h) { return (int)(int16_t)(v >> (16 * h)); }

// Row maximum of one matrix in H units with FindMaxRowEntry's acceptance rule
// (tools/SplitReadAligner.cpp:91-102): values below minSplitScore (8) count as 0.
__device__ __forceinline__ int row_max_h(const uint32_t* __restrict__ cmax, const Geom& g, int w, int n_chunks,
                                         int lane, int h, int row)
{
    if (row == 0 || n_chunks == 0) return 0;   // H(i,0)=0 < 8; empty reference: only column 0 (<=0)
    int v = -32768;
    for (int c = 0; c < n_chunks; ++c) {
        int x = half_of(cmax[(((int64_t)w * g.nch + c) * g.lq1 + row) * WAVE + lane], h);
        v = x > v ? x : v;
    }
    v -= 2 * row;
    return v >= DSA_MIN_SPLIT ? v : 0;
}

// K2a/K2b: per pair, pick the winning read splits (tools/SplitReadAligner.cpp:194-223) and list
// the tiles that hold a row maximum for them.  WRITE=false counts, WRITE=true fills.
template <bool WRITE>
__global__ void k_combine(const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions,
                          const WTask* __restrict__ wtasks, const uint32_t* __restrict__ cmax,
                          const int32_t* __restrict__ min_score_tab, PairState* __restrict__ state,
                          int64_t* __restrict__ task_count, const int64_t* __restrict__ task_offset,
                          ReplayTask* __restrict__ tasks, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)g.n_blocks * PAIRS_PER_WTASK) return;
    const int block = (int)(t >> 7), slot = (int)(t & 127);
    const int lane = slot & 63, h = slot >> 6;
    const WTask w0 = wtasks[2 * block], w1 = wtasks[2 * block + 1];
    if (slot >= w0.n_pairs) return;
    const int64_t p = (int64_t)w0.pair_base + slot;
    const int lq = pairs[p].read_len;
    const int min_score = min_score_tab[lq];

    int max_score = 0;
    for (int a = 0; a <= lq; ++a) {
        int s = row_max_h(cmax, g, 2 * block, w0.n_chunks, lane, h, a) +
                row_max_h(cmax, g, 2 * block + 1, w1.n_chunks, lane, h, lq - a);
        if (s >= min_score && s > max_score) max_score = s;
    }
    int64_t n_tasks = 0;
    int n_kept = 0;
    int64_t out = WRITE ? task_offset[p] : 0;
    if (max_score != 0) {
        for (int a = 0; a <= lq; ++a) {
            const int b = lq - a;
            const int m1 = row_max_h(cmax, g, 2 * block, w0.n_chunks, lane, h, a);
            const int m2 = row_max_h(cmax, g, 2 * block + 1, w1.n_chunks, lane, h, b);
            if (m1 + m2 != max_score) continue;
            if (m1 == 0 || m2 == 0) continue;     // empty column list on one side: no output for this a
            for (int m = 0; m < 2; ++m) {
                const int w = 2 * block + m, row = m ? b : a, target = (m ? m2 : m1) + 2 * row;
                const int nc = m ? w1.n_chunks : w0.n_chunks;
                for (int c = 0; c < nc; ++c) {
                    int x = half_of(cmax[(((int64_t)w * g.nch + c) * g.lq1 + row) * WAVE + lane], h);
                    if (x == target) {
                        if (WRITE) {
                            ReplayTask rt;
                            rt.slot = (uint32_t)t;
                            rt.row = (uint16_t)row;
                            rt.target = (int16_t)target;
                            rt.matrix = (uint8_t)m;
                            rt.chunk = (uint8_t)c;
                            rt.pad_[0] = rt.pad_[1] = 0;
                            rt.kept_idx = (uint32_t)n_kept;
                            tasks[out] = rt;
                        }
                        ++out;
                        ++n_tasks;
                    }
                }
            }
            ++n_kept;
        }
    }
    if (!WRITE) {
        PairState st;
        st.max_score = max_score;
        st.n_kept = n_kept;
        st.task_begin = 0;
        state[p] = st;
        task_count[p] = n_tasks;
    } else {
        state[p].task_begin = task_offset[p];
    }
}

// K3: replay one tile per lane from the stored boundary and report, as a 64-bit mask, the valid
// columns whose value at the task's row equals the row maximum.
__global__ __launch_bounds__(256) void k_replay(const ReplayTask* __restrict__ tasks, int64_t n_tasks,
                                                const WTask* __restrict__ wtasks,
                                                const uint32_t* __restrict__ refcodes,
                                                const uint32_t* __restrict__ rowcodes,
                                                const uint32_t* __restrict__ bnd,
                                                uint64_t* __restrict__ colmask, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const ReplayTask rt = tasks[t];
    const int block = (int)(rt.slot >> 7), slot = (int)(rt.slot & 127);
    const int lane = slot & 63, h = slot >> 6;
    const int w = 2 * block + rt.matrix, c = rt.chunk;
    const WTask wt = wtasks[w];
    const uint32_t* rc = refcodes + ((int64_t)wt.fusion_idx * 2 + wt.matrix) * g.lrp + c * W;
    const uint32_t* rows = rowcodes + (int64_t)w * g.lq1 * WAVE + lane;
    const uint32_t* bi = bnd + ((int64_t)w * g.nch + (c - 1)) * g.lq1 * WAVE + lane;

    uint32_t r[W];
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = rc[i];
    v2s X[W];
#pragma unroll
    for (int i = 0; i < W; ++i) X[i] = (v2s){0, 0};
    v2s bprev = {0, 0};
    const int R = rt.row;
    for (int j = 1; j <= R; ++j) {
        const uint32_t cj = rows[(int64_t)j * WAVE];
        const v2s bcur = (c > 0) ? as_v2s(bi[(int64_t)j * WAVE]) : (v2s){0, 0};
        row_step<uint32_t>(X, r, cj, bprev, bcur);
        bprev = bcur;
    }
    const int nvalid = min(W, wt.lr - c * W);
    uint64_t mask = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        int v = h ? (int)X[i].y : (int)X[i].x;
        if (i < nvalid && v == (int)rt.target) mask |= (1ull << i);
    }
    colmask[t] = mask;
}

// K4: emit.  For every kept split a (ascending) the cross product columns1 x columns2 in ascending
// order (tools/SplitReadAligner.cpp:233-269), then the refSplit de-duplication of
// tools/SplitAlignment.cpp:381-391 (first occurrence wins).  WRITE=false counts.
struct ColIter {   // walks the set bits of the replay masks of one (kept a, matrix), ascending column
    const ReplayTask* tasks;
    const uint64_t* masks;
    int64_t begin, end;    // task range of this (kept, matrix)
};

__device__ __forceinline__ bool cols_contains(const ReplayTask* tasks, const uint64_t* masks, int64_t begin,
                                              int64_t end, int col /*1-based matrix column*/)
{
    const int c = (col - 1) / W, bit = (col - 1) % W;
    for (int64_t k = begin; k < end; ++k)
        if (tasks[k].chunk == c) return (masks[k] >> bit) & 1ull;
    return false;
}

template <bool WRITE>
__global__ void k_emit(const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions,
                       const WTask* __restrict__ wtasks, const PairState* __restrict__ state,
                       const int64_t* __restrict__ task_count, const ReplayTask* __restrict__ tasks,
                       const uint64_t* __restrict__ colmask, int64_t* __restrict__ rec_count,
                       const int64_t* __restrict__ rec_offset, dsa_record* __restrict__ out, Geom g)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)g.n_blocks * PAIRS_PER_WTASK) return;
    const int block = (int)(t >> 7), slot = (int)(t & 127);
    const WTask w0 = wtasks[2 * block];
    if (slot >= w0.n_pairs) return;
    const int64_t p = (int64_t)w0.pair_base + slot;
    const PairState st = state[p];
    int64_t n = 0;
    if (st.max_score != 0 && st.n_kept > 0) {
        const dsa_pair pr = pairs[p];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        const int64_t tb = st.task_begin, te = tb + task_count[p];
        int64_t wr = WRITE ? rec_offset[p] : 0;
        // task ranges: tasks are ordered by kept_idx, then matrix, then chunk
        int64_t k = tb;
        while (k < te) {
            const uint32_t kept = tasks[k].kept_idx;
            int64_t b0 = k;
            while (k < te && tasks[k].kept_idx == kept && tasks[k].matrix == 0) ++k;
            int64_t b1 = k;
            while (k < te && tasks[k].kept_idx == kept && tasks[k].matrix == 1) ++k;
            int64_t e1 = k;
            const int a = tasks[b0].row;
            const int b = tasks[b1].row;
            const int s1 = (int)tasks[b0].target - 2 * a, s2 = (int)tasks[b1].target - 2 * b;
            for (int64_t k1 = b0; k1 < b1; ++k1) {
                uint64_t m1 = colmask[k1];
                while (m1) {
                    const int i1 = tasks[k1].chunk * W + __builtin_ctzll(m1) + 1;
                    m1 &= m1 - 1;
                    for (int64_t k2 = b1; k2 < e1; ++k2) {
                        uint64_t m2 = colmask[k2];
                        while (m2) {
                            const int i2 = tasks[k2].chunk * W + __builtin_ctzll(m2) + 1;
                            m2 &= m2 - 1;
                            // duplicate of an earlier kept a?  same refSplit <=> same (i1,i2)
                            bool dup = false;
                            int64_t q = tb;
                            while (q < b0 && !dup) {
                                const uint32_t kq = tasks[q].kept_idx;
                                int64_t q0 = q;
                                while (q < b0 && tasks[q].kept_idx == kq && tasks[q].matrix == 0) ++q;
                                int64_t q1 = q;
                                while (q < b0 && tasks[q].kept_idx == kq && tasks[q].matrix == 1) ++q;
                                dup = cols_contains(tasks, colmask, q0, q1, i1) &&
                                      cols_contains(tasks, colmask, q1, q, i2);
                            }
                            if (dup) continue;
                            if (WRITE) {
                                dsa_record rec;
                                rec.fusion_id = fu.fusion_id;
                                rec.frag = pr.frag;
                                rec.read_end = pr.read_end;
                                rec.revcomp = pr.revcomp;
                                rec.ref_first = i1;
                                rec.ref_second = fu.ref1_len - i2 - 1;
                                rec.read_first = a;
                                rec.read_second = b;
                                rec.score = s1 < s2 ? s1 : s2;
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
