// dsa_long.hpp — the split-read alignment of pairs that do not fit the 16-bit tile kernels: reads longer than FAST_MAX_READ
// bases or windows longer than FAST_MAX_REF (the packed fields of dsa_kernels.hpp hold V + 1024 <= 4 Lq + 1024 as a finite fp16
// bit pattern, and a tile index in eight bits).  The reference's matrix simply grows (tools/Matrix.h:98-107); here such a pair
// is swept in 32-bit integers by ONE WORKGROUP, row by row, all columns of a row side by side:
//     H(i,j) = max( c(i), H(i-1,j) - 2 ),  c(i) = max( H(i-1,j-1) + (eq ? 2 : -1), H(i,j-1) - 2 )
//  => H(i,j) + 2i = max_{k <= i} ( c(k) + 2k )        (a prefix maximum; H(0,j) = -2j enters as k = 0)
// Every thread owns a run of whole 32-column words, forms c(k) + 2k and its running maximum there, the workgroup scans the
// runs' maxima, and a second touch of the run finishes the row.  Two sweeps per pair, both matrices in the same row loop:
//   k_long_rows   row maxima of M1 and M2 (tools/SplitReadAligner.cpp:91-102), then the winning read splits (:194-223)
//   k_long_cols   the same sweep again; at the rows of the kept splits the columns that attain the row maximum, as bitmaps
//   k_long_emit   the records of the pair — cross product per kept split, refSplit de-duplication
//                 (tools/SplitReadAligner.cpp:233-269, tools/SplitAlignment.cpp:381-400) — counted, then written
// This path is rare (long-read experiments, very wide windows) and built for being right, not fast: a few microseconds per
// row and matrix.  The regular kernels see these pairs as empty reads, which emit nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/defuse_dsa.h"

namespace dsa {

constexpr int FAST_MAX_READ = 7600;            // V + 1024 = H + 2j + 1024 <= 4 * Lq + 1024 must stay a finite fp16 pattern (< 0x7C00)
constexpr int FAST_MAX_REF = 255 * 64;         // tile index in eight bits (ReplayTask)
constexpr int LONG_THREADS = 256;
constexpr int LONG_MAX_READ = 1 << 20;         // scores stay far inside 32 bits
constexpr int LONG_MAX_REF = 1 << 24;

struct LongDesc {                              // one long pair, everything the kernels need (the regular copies are blanked)
    int32_t pair_idx;                          // in the caller's order, batch-wide
    int32_t read_off, read_len;
    int32_t ref0_off, ref0_len, ref1_off, ref1_len;
    int32_t fusion_id, frag;
    int32_t read_end, revcomp;
    int32_t min_score;
};
struct LongState {                             // per long pair, written by k_long_rows
    int32_t n_kept, max_score;
    int64_t work_off;                          // int32 words: [rm1 (lq+1)][rm2 (lq+1)][kept_a (lq+1)][rowmap1 (lq+1)][rowmap2 (lq+1)]
    int64_t rows_off;                          // int32 words of the sweep's row buffers (per BLOCK, not per pair): see long_rows_words
    int64_t bits_off;                          // uint32 words: per kept split [bitmap1 (w1)][bitmap2 (w2)], set by the host between the sweeps
};
__host__ __device__ inline int64_t long_work_words(int lq) { return 5 * (int64_t)(lq + 1); }
__host__ __device__ inline int long_bitmap_words(int len) { return (len + 1 + 31) / 32; }
// row buffers of one workgroup: previous and current row of either matrix, padded to whole words of columns
__host__ __device__ inline int64_t long_rows_words(int l0, int l1) { return 2 * 32 * ((int64_t)long_bitmap_words(l0) + long_bitmap_words(l1)); }

__device__ __forceinline__ int long_accept(int v) { return v >= DSA_MIN_SPLIT ? v : 0; }   // FindMaxRowEntry: below minSplitScore counts as 0

// One row of one matrix.  prev / cur: rows j-1 / j (columns 0..len), ref(i) = reference base of column i (1-based), b = read base of
// the row.  Returns this thread's maximum of the row; with `bits`, sets the bits of the columns whose value equals `target`.
template <bool REVERSED>
__device__ __forceinline__ int long_row(const int32_t* __restrict__ prev, int32_t* __restrict__ cur, const uint8_t* __restrict__ ref, int len,
                                        uint32_t b, int j, int* s_carry, uint32_t* bits, int target)
{
    const int nw = long_bitmap_words(len);                         // words of 32 columns
    const int per = (nw + LONG_THREADS - 1) / LONG_THREADS;         // words per thread
    const int w0 = threadIdx.x * per, w1 = min(nw, w0 + per);
    const int i0 = 32 * w0, i1 = min(len + 1, 32 * w1);            // columns [i0, i1)
    // first touch: g(i) = c(i) + 2i and its running maximum inside the run
    int run = INT32_MIN;
    for (int i = i0; i < i1; ++i) {
        int g;
        if (i == 0)
            g = -2 * j;                                            // H(0,j) = -2j
        else {
            const uint32_t r = REVERSED ? ref[len - i] : ref[i - 1];
            const int diag = prev[i - 1] + (r == b ? DSA_MATCH : DSA_MISMATCH);
            const int up = prev[i] + DSA_GAP;
            g = (diag > up ? diag : up) + 2 * i;
        }
        run = g > run ? g : run;
        cur[i] = run;
    }
    // exclusive prefix maximum of the runs' maxima over the workgroup: inside a wave by shuffles, across the waves through LDS
    int carry;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int v = run;
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(v, d, 64);
            if (lane >= d) v = y > v ? y : v;
        }
        carry = __shfl_up(v, 1, 64);
        if (lane == 0) carry = INT32_MIN;
        if (lane == 63) s_carry[wave] = v;
        __syncthreads();
        for (int w = 0; w < wave; ++w) carry = s_carry[w] > carry ? s_carry[w] : carry;
    }
    // second touch: H(i,j) = max(run maximum up to i, carry) - 2i
    int mine = INT32_MIN;
    for (int w = w0; w < w1; ++w) {
        uint32_t word = 0;
        const int e = min(len + 1, 32 * w + 32);
        for (int i = 32 * w; i < e; ++i) {
            int h = cur[i];
            h = (h > carry ? h : carry) - 2 * i;
            cur[i] = h;
            mine = h > mine ? h : mine;
            if (bits && h == target) word |= 1u << (i & 31);
        }
        if (bits) bits[w] = word;
    }
    __syncthreads();                                               // s_carry is free again, the row is complete
    return mine;
}

__device__ __forceinline__ int long_block_max(int v, int* s_red)
{
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int m = s_red[0];
    for (int w = 1; w < LONG_THREADS / 64; ++w) m = max(m, s_red[w]);
    __syncthreads();
    return m;
}

// the sweep of both matrices of one pair: COLS = false stores the row maxima, COLS = true the column bitmaps of the kept rows
template <bool COLS>
__device__ __forceinline__ void long_sweep(const LongDesc& d, const LongState& st, const uint8_t* __restrict__ ref_bytes,
                                           const uint8_t* __restrict__ read_bytes, int32_t* __restrict__ work, int32_t* __restrict__ rows,
                                           uint32_t* __restrict__ bits, int* s_carry, int* s_red)
{
    const int lq = d.read_len, l0 = d.ref0_len, l1 = d.ref1_len;
    const int nw0 = long_bitmap_words(l0), nw1 = long_bitmap_words(l1);
    int32_t* rm1 = work;
    int32_t* rm2 = rm1 + (lq + 1);
    const int32_t* rowmap1 = rm2 + 2 * (int64_t)(lq + 1);
    const int32_t* rowmap2 = rowmap1 + (lq + 1);
    int32_t* a_prev = rows;
    int32_t* a_cur = a_prev + 32 * (int64_t)nw0;
    int32_t* b_prev = a_cur + 32 * (int64_t)nw0;
    int32_t* b_cur = b_prev + 32 * (int64_t)nw1;
    const uint8_t* r0 = ref_bytes + d.ref0_off;
    const uint8_t* r1 = ref_bytes + d.ref1_off;
    const uint8_t* rd = read_bytes + d.read_off;
    // row 0: H(i,0) = 0
    for (int i = threadIdx.x; i <= l0; i += LONG_THREADS) a_prev[i] = 0;
    for (int i = threadIdx.x; i <= l1; i += LONG_THREADS) b_prev[i] = 0;
    if (!COLS && threadIdx.x == 0) { rm1[0] = 0; rm2[0] = 0; }
    if (COLS) {                                                    // row 0 never has a value >= 8: no kept row maps to it with columns
        (void)bits;
    }
    __syncthreads();
    for (int j = 1; j <= lq; ++j) {
        const uint32_t bq1 = rd[j - 1], bq2 = rd[lq - j];          // M2 runs on the reversed read
        uint32_t* bm1 = nullptr;
        uint32_t* bm2 = nullptr;
        int t1 = 0, t2 = 0;
        if (COLS) {
            const int k1 = rowmap1[j], k2 = rowmap2[j];
            if (k1 >= 0) { bm1 = bits + (int64_t)k1 * (nw0 + nw1); t1 = rm1[j]; }
            if (k2 >= 0) { bm2 = bits + (int64_t)k2 * (nw0 + nw1) + nw0; t2 = rm2[j]; }
        }
        const int m1 = long_row<false>(a_prev, a_cur, r0, l0, bq1, j, s_carry, bm1, t1);
        const int m2 = long_row<true>(b_prev, b_cur, r1, l1, bq2, j, s_carry, bm2, t2);
        if (!COLS) {
            const int x1 = long_block_max(m1, s_red), x2 = long_block_max(m2, s_red);
            if (threadIdx.x == 0) { rm1[j] = x1; rm2[j] = x2; }
        }
        int32_t* t = a_prev; a_prev = a_cur; a_cur = t;
        t = b_prev; b_prev = b_cur; b_cur = t;
    }
    (void)st;
}

// sweep 1 + the winning read splits.  One workgroup per long pair (grid-stride), row buffers per workgroup.
__global__ __launch_bounds__(LONG_THREADS) void k_long_rows(const LongDesc* __restrict__ desc, LongState* __restrict__ state, int n_long,
                                                            const uint8_t* __restrict__ ref_bytes, const uint8_t* __restrict__ read_bytes,
                                                            int32_t* __restrict__ work, int32_t* __restrict__ rows, int64_t rows_stride)
{
    __shared__ int s_carry[LONG_THREADS];
    __shared__ int s_red[LONG_THREADS / 64];
    for (int p = blockIdx.x; p < n_long; p += gridDim.x) {
        const LongDesc d = desc[p];
        LongState st = state[p];
        int32_t* w = work + st.work_off;
        long_sweep<false>(d, st, ref_bytes, read_bytes, w, rows + (int64_t)blockIdx.x * rows_stride, nullptr, s_carry, s_red);
        __syncthreads();
        if (threadIdx.x == 0) {
            // tools/SplitReadAligner.cpp:194-223: the read splits of maximal m1(a) + m2(lq - a) >= minScore; a split with an empty
            // side competes but emits nothing
            const int lq = d.read_len;
            const int32_t* rm1 = w;
            const int32_t* rm2 = rm1 + (lq + 1);
            int32_t* kept_a = w + 2 * (int64_t)(lq + 1);
            int32_t* rowmap1 = kept_a + (lq + 1);
            int32_t* rowmap2 = rowmap1 + (lq + 1);
            int max_score = 0;
            for (int a = 0; a <= lq; ++a) {
                const int s = long_accept(rm1[a]) + long_accept(rm2[lq - a]);
                if (s >= d.min_score && s > max_score) max_score = s;
            }
            int n = 0;
            for (int j = 0; j <= lq; ++j) { rowmap1[j] = -1; rowmap2[j] = -1; }
            if (max_score != 0)
                for (int a = 0; a <= lq; ++a) {
                    const int m1 = long_accept(rm1[a]), m2 = long_accept(rm2[lq - a]);
                    if (m1 + m2 != max_score || m1 == 0 || m2 == 0) continue;
                    kept_a[n] = a;
                    rowmap1[a] = n;
                    rowmap2[lq - a] = n;
                    ++n;
                }
            st.n_kept = n;
            st.max_score = max_score;
            state[p] = st;
        }
        __syncthreads();
    }
}

// sweep 2: the columns of the kept rows
__global__ __launch_bounds__(LONG_THREADS) void k_long_cols(const LongDesc* __restrict__ desc, const LongState* __restrict__ state, int n_long,
                                                            const uint8_t* __restrict__ ref_bytes, const uint8_t* __restrict__ read_bytes,
                                                            int32_t* __restrict__ work, int32_t* __restrict__ rows, int64_t rows_stride,
                                                            uint32_t* __restrict__ bits)
{
    __shared__ int s_carry[LONG_THREADS];
    __shared__ int s_red[LONG_THREADS / 64];
    for (int p = blockIdx.x; p < n_long; p += gridDim.x) {
        const LongDesc d = desc[p];
        const LongState st = state[p];
        if (st.n_kept == 0) continue;                               // uniform
        long_sweep<true>(d, st, ref_bytes, read_bytes, work + st.work_off, rows + (int64_t)blockIdx.x * rows_stride, bits + st.bits_off,
                         s_carry, s_red);
        __syncthreads();
    }
}

// The records of the long pairs of one slice: counted into rec_count (WRITE = false, before the slice's scan) or written behind
// rec_offset (WRITE = true, after it).  One thread per pair: per kept split (ascending) the cross product of its two column
// sets in ascending order, minus the refSplits an earlier kept split already has.
template <bool WRITE>
__global__ void k_long_emit(const LongDesc* __restrict__ desc, const LongState* __restrict__ state, int n_long, const int32_t* __restrict__ work,
                            const uint32_t* __restrict__ bits, int64_t pair_begin, int64_t pair_end, int64_t* __restrict__ rec_count,
                            const int64_t* __restrict__ rec_offset, dsa_record* __restrict__ out, uint64_t out_cap)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_long) return;
    const LongDesc d = desc[p];
    if (d.pair_idx < pair_begin || d.pair_idx >= pair_end) return;  // another slice's
    const LongState st = state[p];
    const int64_t o = d.pair_idx - pair_begin;
    const int lq = d.read_len, nw0 = long_bitmap_words(d.ref0_len), nw1 = long_bitmap_words(d.ref1_len);
    const int32_t* w = work + st.work_off;
    const int32_t* rm1 = w;
    const int32_t* rm2 = rm1 + (lq + 1);
    const int32_t* kept_a = w + 2 * (int64_t)(lq + 1);
    const uint32_t* bm = bits + st.bits_off;
    int64_t n = 0, at = 0;
    if (WRITE) {
        at = rec_offset[o];
        if ((uint64_t)rec_offset[o + 1] > out_cap) return;          // the host grows the buffer and runs the emit again
    }
    for (int k = 0; k < st.n_kept; ++k) {
        const int a = kept_a[k];
        const uint32_t* b1 = bm + (int64_t)k * (nw0 + nw1);
        const uint32_t* b2 = b1 + nw0;
        for (int wa = 0; wa < nw0; ++wa)
            for (uint32_t ra = b1[wa]; ra; ra &= ra - 1) {
                const int i1 = 32 * wa + __builtin_ctz(ra);
                for (int wb = 0; wb < nw1; ++wb)
                    for (uint32_t rb = b2[wb]; rb; rb &= rb - 1) {
                        const int i2 = 32 * wb + __builtin_ctz(rb);
                        bool dup = false;
                        for (int k2 = 0; k2 < k && !dup; ++k2) {
                            const uint32_t* c1 = bm + (int64_t)k2 * (nw0 + nw1);
                            dup = ((c1[i1 >> 5] >> (i1 & 31)) & 1u) && ((c1[nw0 + (i2 >> 5)] >> (i2 & 31)) & 1u);
                        }
                        if (dup) continue;
                        if (WRITE) {
                            dsa_record r;
                            r.fusion_id = d.fusion_id;
                            r.frag = d.frag;
                            r.read_end = d.read_end;
                            r.revcomp = d.revcomp;
                            r.ref_first = i1;
                            r.ref_second = d.ref1_len - i2 - 1;
                            r.read_first = a;
                            r.read_second = lq - a;
                            const int m1 = rm1[a], m2 = rm2[lq - a];
                            r.score = m1 < m2 ? m1 : m2;
                            r.pair_idx = d.pair_idx;
                            out[at + n] = r;
                        }
                        ++n;
                    }
            }
    }
    if (!WRITE) rec_count[o] = n;
}

// the regular kernels' copies of the long pairs become empty reads, of fusions with an over-long window empty windows
__global__ void k_long_blank(dsa_pair* __restrict__ pairs, const LongDesc* __restrict__ desc, int n_long, dsa_fusion* __restrict__ fusions,
                             const int32_t* __restrict__ long_fusions, int n_long_fusions)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_long) pairs[desc[t].pair_idx].read_len = 0;
    if (t < n_long_fusions) {
        dsa_fusion& f = fusions[long_fusions[t]];
        f.ref0_len = 0;
        f.ref1_len = 0;
    }
}

}  // namespace dsa
