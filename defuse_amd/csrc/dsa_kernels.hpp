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
//     one ascending pass per row, in place:  X[i] = max3(Xold[i-1] + d(i), Xold[i], Xnew[i-1] - 2)
//     Register i of a tile holds V + 2i ("drift"), which turns the -2 of the along-the-reference move
//     into 0: the serial chain of a row is then one max3 per column and nothing else; the drift is
//     taken out again (one off-chain add) where columns are compared for the row maximum.
//   * instruction mix chosen from measured gfx950 issue rates (profiles/microbench): 32-bit add/sub/xor
//     issue in 2 cycles, every packed (VOP3P) op and every 32-bit min/max in 4.  So the two int16
//     fields are added with plain v_add_u32 (SWAR: fields are kept in [1024, 31743] so no carry or
//     borrow ever crosses), and both maxima of a cell pair come from ONE v_pk_maximum3_f16: positive
//     normal fp16 bit patterns order exactly like the integers they spell, and maximum3 returns one
//     of its inputs unchanged.  Per column (2 cells): 2 adds + 1 max3 + 1/2 max3 for the row maximum
//     = 10 issue cycles, against 20 for the obvious pk_add/pk_max formulation.
//   * fast path (reads over {A,C,G,T,N}): the substitution term d(i) of both matrices comes out of
//     a per-(fusion,tile) score table in LDS with one ds_read_b128 per 4 columns.  Generic path (any
//     bytes): d(i) from xor + pk_min.
//   * per (tile,row) the kernel stores the tile's row maximum and the tile's last column (the
//     boundary the next tile starts from).  In its tail the same workgroup picks the winning rows
//     (combine_wg) and replays only the winning tile pair from the stored boundaries to enumerate tied
//     columns (replay_fast_wg) — exact, at a fraction of the fill work instead of a second full pass,
//     and in the shadow of the other resident workgroups' sweeps; a pair with no other tile also gets its
//     record count there.  k_replay takes the left-over tiles (four lanes per task), k_emit_listed<false>
//     counts the pairs that had some, a scan places the records, and k_emit_counted / k_emit_listed<true>
//     write them side by side.
//
// Everything here is integer; results are bit-exact with the reference by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/defuse_dsa.h"
#include "dsa_diag.hpp"

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
constexpr int TROW = W + 4;           // table row stride in dwords: table row t starts at bank quad t mod 16, so
                                      // the 16 rows of the A/C/G/T x A/C/G/T combinations never collide under
                                      // ds_read_b128 (rows 16..24, the combinations with an N, are rare)
#ifndef FILL_PF
#define FILL_PF 2               // LDS table reads kept in flight ahead of the column being updated
#endif
constexpr int GMAX = 4;               // max distinct fusions per workgroup with one 25-row table per fusion
constexpr int TGROUP = NCOMBO * TROW; // dwords per fusion table
// Workgroups of 5..GSPLIT fusions keep two 5-row tables of 16-bit terms per fusion in the same LDS instead
// (M1 terms by M1 class, M2 terms by M2 class): four columns come from two ds_read_b64 and one v_perm_b32
// per column joins the two fields (one instruction more per cell pair than the 25-row tables).
constexpr int GSPLIT = 20;                       // at four workgroups per CU
constexpr int GSPLIT2 = 40;                      // at two workgroups per CU (twice the LDS): still ahead of the generic kernel
constexpr int TROW_S = TROW / 2;                  // split table row stride in dwords (TROW 16-bit terms)
constexpr int TGROUP_SPLIT = 2 * NCLS * TROW_S;   // dwords per fusion
static_assert(GSPLIT * TGROUP_SPLIT <= GMAX * TGROUP, "split tables must fit the LDS of the combined ones");
static_assert((TROW_S % 2) == 0, "split table rows are read with ds_read_b64");

// terms of columns 4q..4q+3 for M1 class k1 / M2 class k2 of one fusion's split tables
__device__ __forceinline__ uint4 split_terms(const uint32_t* tb, uint32_t k1, uint32_t k2, int q)
{
    const uint2 L = reinterpret_cast<const uint2*>(tb + k1 * TROW_S)[q];
    const uint2 H = reinterpret_cast<const uint2*>(tb + (NCLS + k2) * TROW_S)[q];
    return make_uint4(__builtin_amdgcn_perm(H.x, L.x, 0x05040100u), __builtin_amdgcn_perm(H.x, L.x, 0x07060302u),
                      __builtin_amdgcn_perm(H.y, L.y, 0x05040100u), __builtin_amdgcn_perm(H.y, L.y, 0x07060302u));
}
// Stored values are V + 1024 per int16 field: always a positive normal fp16 bit pattern.
constexpr uint32_t BIAS16 = 0x0400u;
constexpr uint32_t BIAS2 = 0x04000400u;
constexpr uint32_t TWO2 = 0x00020002u;
constexpr uint32_t FOUR2 = 0x00040004u;
constexpr uint32_t SIX2 = 0x00060006u;
__host__ __device__ constexpr uint32_t drift2(int i) { return (uint32_t)(2 * i) * 0x00010001u; }

typedef unsigned short v2u __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// packed maxima on biased fields, via fp16 maximum (v_pk_maximum3_f16 on gfx950)
__device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_bit_cast(v2h, a), __builtin_bit_cast(v2h, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)   // per-field unsigned maximum
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b)));
}
__device__ __forceinline__ uint32_t max3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t,
                              __builtin_elementwise_maximum(
                                  __builtin_elementwise_maximum(__builtin_bit_cast(v2h, a), __builtin_bit_cast(v2h, b)),
                                  __builtin_bit_cast(v2h, c)));
}
__device__ __forceinline__ uint32_t min3u(uint32_t x)   // per field min(x, 3)
{
    const v2u three = {3, 3};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2u, x), three));
}

// lower bound of the pair's final score found at upload (k_rank_in_fusion), 0 = none
__device__ __forceinline__ int pair_bound(const dsa_pair& pr) { return (int)pr.pad_[0] | ((int)pr.pad_[1] << 8); }

struct WaveInfo {
    int32_t lq_max;        // longest read among the wave's pairs
    int32_t nch_max;       // most tiles among the wave's pairs (either matrix)
};

// Which kernel sweeps a workgroup (wg_tier[], written by k_fill_fast<0>, which is always launched first): the table tiers 0..2
// of k_fill_fast or TIER_GENERIC; the bits of Counters::need_tiers say which kernels a slice needs at all.
constexpr uint8_t TIER_GENERIC = 3;

// Per workgroup (WG_WAVES waves = WG_LANES pairs): the fusions of its pairs, for the fast path.  A "group" is a run of
// consecutive pairs of one fusion (in a planned sweep every fusion is one run, so groups are the distinct fusions; a fusion
// that comes back later in an unplanned order simply gets a second table).  The fill kernels find the runs themselves, in
// their prologue (wg_groups): every lane knows its group, the list of the groups' fusions lives in LDS.
struct WgGroupsLds {
    int32_t wave_starts[WG_WAVES];     // runs that begin in each wave
    int32_t group_f[GSPLIT2];          // fusion_idx of group k
};
// What a kernel keeps of its workgroup's groups: the first GMAX fusions in registers (all there are for the 25-row-table
// tier) and a pointer to the full list for the split-table tiers.
struct WgView {
    int32_t n_groups;      // 0: more than GSPLIT2 runs (generic kernel)
    int32_t f4[GMAX];
    const int32_t* list;
};
__device__ __forceinline__ int group_fusion(const WgView& v, int k)      // fusion_idx of group k < n_groups
{
    if (v.n_groups > GMAX) return v.list[k];                          // uniform
    int f = v.f4[0];
#pragma unroll
    for (int j = 1; j < GMAX; ++j)
        if (k == j) f = v.f4[j];
    return f;
}
// Prologue of the fill kernels, called by all threads of the workgroup (barrier inside): f = fusion of the lane's pair (lanes
// past the end shadow the last pair), f_before = fusion of the pair before the workgroup's first (any value for workgroup 0's
// first lane: it starts a run anyway).  Returns the view; my_group = the lane's group (meaningful while n_groups > 0).
__device__ __forceinline__ WgView wg_groups(WgGroupsLds* gl, int f, int f_prev_of_wave_lane0, int& my_group)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int fprev = __shfl_up(f, 1, 64);
    if (lane == 0) fprev = f_prev_of_wave_lane0;
    const bool start = tid == 0 || f != fprev;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(start);
    if (lane == 0) gl->wave_starts[wv] = __builtin_popcountll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < WG_WAVES; ++k) {
        const int c = gl->wave_starts[k];
        if (k < wv) base += c;
        total += c;
    }
    my_group = base + __builtin_popcountll(m & ((2ull << lane) - 1ull)) - 1;
    if (start && my_group < GSPLIT2) gl->group_f[my_group] = f;
    __syncthreads();
    WgView v;
    v.n_groups = total <= GSPLIT2 ? total : 0;
    v.list = gl->group_f;
#pragma unroll
    for (int k = 0; k < GMAX; ++k) v.f4[k] = k < total ? gl->group_f[k] : 0;
    return v;
}

// Geometry shared by all kernels of one run (one slice).  Pair p <-> wave p>>6, lane p&63.
struct Geom {
    int32_t n_waves;
    int32_t n_wgs;
    int32_t lq1;           // rows stride  = max read length + 1
    int32_t nch;           // chunk stride = max n_chunks
    int32_t lrp;           // refcodes stride = nch * W
    int32_t n_fusions;
    int64_t n_pairs;
    const int32_t* orig;   // pair order of the sweep -> the caller's pair index (nullptr: the same order)
    uint32_t tiers_launched;   // fill kernels this slice was given (bit t: k_fill_fast<t>, bit 3: k_fill_generic)
#ifdef DSA_PRUNE_STATS
    unsigned long long* stats;   // diagnostic builds only
#endif
};

struct KeptRow {           // one winning read split of a pair that has columns on both sides
    int16_t a;             // alignedToRef1
    int16_t m1, m2;        // row maxima (H units)
    int16_t pad_;
};

// A tile pair to re-run: all kept rows of one pair in tile chunk0 of M1 (lo half) and tile chunk1 of
// M2 (hi half) in one sweep — both matrices share the row index, so they replay together.
constexpr uint8_t NO_CHUNK = 0xFF;
struct ReplayTask {
    uint32_t pair;         // slice-relative
    uint32_t mask_begin;   // masks[(mask_begin + k)*2 + h] for kept row k of the pair, matrix h
    uint16_t last_row;     // largest row either matrix needs; bit 15 = replayed by the table-driven kernel
    uint8_t  chunk0;       // NO_CHUNK = nothing to replay on that side
    uint8_t  chunk1;
    int32_t fusion;        // the pair's fusion: the generic replay starts fetching its reference codes at once
};
constexpr uint32_t GTASK_OWNER = 0x80000000u;   // gtasks[].x: the pair's first task in the list (that lane counts / writes the pair's records)
constexpr uint16_t TASK_FAST = 0x8000u;
constexpr uint16_t TASK_ONLY = 0x4000u;   // LaneInfo only: the pair has no other task, the replaying lane also counts its records
constexpr uint16_t TASK_ROW = 0x3FFFu;    // rows <= 7600

struct alignas(8) PairState {
    uint32_t mask_begin;   // masks[(mask_begin + t * n_kept + k) * 2 + h]: columns of kept row k in the tile of task t, matrix h
    uint32_t kept_begin;
    uint32_t task_begin;
    uint16_t n_kept;       // kept rows with columns on both sides
    uint8_t n_tasks;       // a window has at most 255 tiles
    uint8_t flags;
    uint16_t tiles0, tiles1;   // STATE_TILES: the tiles of M1 / M2 that hold a column of a kept row
    uint8_t first0, first1;    // tiles of task 0 (NO_CHUNK: none); task t > 0 has the (t-1)-th set bit of the other tiles
    uint16_t pad_;
};
// task index of tile c of a side whose tiles are `tiles` and whose task 0 has tile `first`
__device__ __forceinline__ int task_of_tile(uint32_t tiles, int first, int c)
{
    if (c == first) return 0;
    const uint32_t rest = first < 16 ? tiles & ~(1u << first) : tiles;
    return 1 + __builtin_popcount(rest & ((1u << c) - 1u));
}
static_assert(sizeof(PairState) == 24, "three 8-byte loads");
constexpr uint8_t STATE_COUNTED = 1;   // rec_count already holds the pair's record count (combine: no records; fill tail: its only task)
constexpr uint8_t STATE_TILES = 2;     // tiles0 / tiles1 are valid (windows of at most 16 tiles): emit needs no task list

struct Counters {          // device-side allocation cursors (and overflow detection)
    unsigned long long n_kept, n_tasks, n_masks, n_gtasks;
    unsigned need_tiers;   // bit t: some workgroup of the slice belongs to fill kernel t (1, 2: split-table tiers, 3: generic)
    unsigned pad_;
};

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// Per-row planes ([rows][64 lanes] dwords: rowcodes, bnd, cmax, rmax) keep four consecutive rows of a
// lane in one 16-byte word, so the fill kernels move them with dwordx4 loads and stores:
// element (row j, lane) of a plane sits at rowidx(j, lane); geometry lq1 is a multiple of 4.
__host__ __device__ __forceinline__ int64_t rowidx(int j, int lane) { return ((int64_t)(j >> 2) * WAVE + lane) * 4 + (j & 3); }
constexpr uint32_t CODE_MASK = 0xFF00FF00u;   // rowcodes: byte codes; bytes 0 and 2 carry the table row / the two classes
// unbiased V of one field
__device__ __forceinline__ int half_of(uint32_t v, int h) { return (int)((v >> (16 * h)) & 0xFFFFu) - (int)BIAS16; }

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

__device__ __forceinline__ uint32_t base_class(uint32_t byte)   // A,C,T,G,N -> 0,1,2,3,4
{
    uint32_t k = (byte >> 1) & 7u;       // A:0 C:1 T:2 G:3 N:7
    return k > 4u ? 4u : k;
}
// table row of a (M1 class, M2 class) combination: the 16 N-free combinations first
__device__ __forceinline__ uint32_t table_row(uint32_t k1, uint32_t k2)
{
    if (k1 < 4u && k2 < 4u) return 4u * k1 + k2;
    return k1 == 4u ? 16u + k2 : 21u + k1;
}
__device__ __forceinline__ void table_row_classes(int t, int& k1, int& k2)   // inverse of table_row
{
    if (t < 16) { k1 = t >> 2; k2 = t & 3; }
    else if (t < 21) { k1 = 4; k2 = t - 16; }
    else { k1 = t - 21; k2 = 4; }
}

// Score tables of one tile for the fusion groups of a workgroup (layout: k_fill_fast).  One thread per
// (group, column): the five per-class terms of either field are formed once and combined into the 25
// table rows, so a tile's tables cost ~100 instructions per thread.  code_of(gi, i, q0, q1) returns the
// 16-bit reference codes of column i (REF_PAD16 = padding) for M1 / M2.
template <bool SPLIT, class CodeFn>
__device__ __forceinline__ void build_tables(uint32_t* __restrict__ T, int n_groups, CodeFn&& code_of)
{
    for (int e = threadIdx.x; e < n_groups * W; e += WG_LANES) {
        const int i = e & (W - 1), gi = e >> 6;
        uint32_t q0, q1;
        code_of(gi, i, q0, q1);
        const uint32_t cls_byte[NCLS] = {'A', 'C', 'T', 'G', 'N'};   // inverse of base_class
        uint32_t lo[NCLS], hi[NCLS];
#pragma unroll
        for (int k = 0; k < NCLS; ++k) {
            lo[k] = q0 == REF_PAD16 ? 0u : ((q0 >> 8) == cls_byte[k] ? 4u : 1u);
            hi[k] = (q1 == REF_PAD16 ? 0u : ((q1 >> 8) == cls_byte[k] ? 4u : 1u)) << 16;
        }
        // +2 per field for i > 0: the diagonal move from column i-1 to i picks up the drift
        const uint32_t drift = i > 0 ? TWO2 : 0u;
        if (SPLIT) {                   // rows 0..4: M1 term by M1 class, rows 5..9: M2 term by M2 class, 16 bits each
            unsigned short* col = reinterpret_cast<unsigned short*>(T + gi * TGROUP_SPLIT) + i;
#pragma unroll
            for (int k = 0; k < NCLS; ++k) {
                col[k * TROW] = (unsigned short)(lo[k] + (drift & 0xFFFFu));
                col[(NCLS + k) * TROW] = (unsigned short)((hi[k] + (drift & 0xFFFF0000u)) >> 16);
            }
        } else {
            uint32_t* col = T + gi * TGROUP + i;
#pragma unroll
            for (int t = 0; t < NCOMBO; ++t) {
                int k1, k2;
                table_row_classes(t, k1, k2);
                col[t * TROW] = (lo[k1] | hi[k2]) + drift;
            }
        }
    }
}

// Row codes of one wave's pairs, written by the wave itself in the prologue of the fill kernels (each lane
// its own pair, four rows per dwordx4).  Returns whether the lane met a read byte outside {A,C,G,T,N}.
// The table kernels sweep from a plane of ONE BYTE per row (row_bytes: four rows per dword and lane, [wave][lq1/4][64]): all a
// table sweep needs of a row is its table row (25-row tables) or its two classes (split tables), and the sweeps read the row
// codes once per tile — a quarter of the bytes of the full codes, which the replays still take from `rowcodes`.
template <int BYTE_KIND>     // 0: no byte plane (generic kernel), 1: table row, 2: classes k1 | k2 << 4
__device__ __forceinline__ bool pack_rows_wave(const uint8_t* __restrict__ read_bytes, const dsa_pair* __restrict__ pairs,
                                               uint32_t* __restrict__ rowcodes, uint32_t* __restrict__ row_bytes, const Geom& g, int w,
                                               int lane, int lq_wave)
{
    const int64_t p = (int64_t)w * WAVE + lane;
    const bool valid = p < g.n_pairs;
    dsa_pair pr{};
    if (valid) pr = pairs[p];
    uint4* out = reinterpret_cast<uint4*>(rowcodes + (int64_t)w * g.lq1 * WAVE) + lane;
    uint32_t* out1 = BYTE_KIND ? row_bytes + (int64_t)w * (g.lq1 >> 2) * WAVE + lane : nullptr;
    bool exotic = false;
    for (int gq = 0; gq <= (lq_wave >> 2); ++gq) {
        uint32_t code[4];
        uint32_t bytes = 0;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int j = 4 * gq + sidx;
            code[sidx] = ROW_PAD16 | (ROW_PAD16 << 16);
            if (valid && j >= 1 && j <= pr.read_len) {
                const uint32_t b0 = read_bytes[(int64_t)pr.read_off + (j - 1)];
                const uint32_t b1 = read_bytes[(int64_t)pr.read_off + (pr.read_len - j)];
                exotic |= !is_fast_base(b0);                       // b1 is some other row's b0
                const uint32_t k1 = base_class(b0), k2 = base_class(b1);
                code[sidx] = (b0 << 8) | (b1 << 24) | table_row(k1, k2) | ((k1 | (k2 << 4)) << 16);   // byte 2: classes, split tables
                if (BYTE_KIND) bytes |= (BYTE_KIND == 1 ? table_row(k1, k2) : (k1 | (k2 << 4))) << (8 * sidx);
            }
        }
        out[(int64_t)gq * WAVE] = make_uint4(code[0], code[1], code[2], code[3]);
        if (BYTE_KIND) out1[(int64_t)gq * WAVE] = bytes;
    }
    return exotic;
}

// ---------------------------------------------------------------------------------------------
// Row step, generic scoring (any byte alphabet): per-lane reference codes in VGPRs.
// X[i] holds V(i0+i, j-1) on entry and V(i0+i, j) on exit (biased).
//   bprev = V(i0-1, j-1), bcur = V(i0-1, j)   (the previous tile's last column; V=0 for tile 0)
// The next column's diagonal term is formed before X[i] is overwritten, so the update is in place.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_step(uint32_t (&X)[W], const uint32_t (&r)[W], uint32_t cj, uint32_t bprev,
                                         uint32_t bcur)
{
    // X[i] carries drift 2i; bprev/bcur are drift-free.  Diagonal into column i>0 gains +2 of drift.
    // The substitution terms do not depend on X: they are formed 16 columns at a time, ahead of the
    // chain, so that only max3 -> max3 is serial (left to itself the compiler recycles two registers
    // and serialises xor/min/sub/add/max3 of every column).
    cj &= CODE_MASK;
    constexpr int BLK = 16;
    uint32_t a = (bprev + FOUR2) - min3u(cj ^ r[0]);
    uint32_t up = bcur - TWO2;
#pragma unroll
    for (int b = 0; b < W; b += BLK) {
        uint32_t d[BLK];
#pragma unroll
        for (int k = 0; k < BLK; ++k)
            d[k] = b + k + 1 < W ? SIX2 - min3u(cj ^ r[b + k + 1]) : 0u;
#pragma unroll
        for (int k = 0; k < BLK; ++k) {
            const int i = b + k;
            const uint32_t a_next = X[i] + d[k];
            X[i] = max3(a, X[i], up);
            up = X[i];                  // drift makes the next column's "up - 2" equal to this value
            a = a_next;
        }
    }
}

// tile row maximum: one max3 per two columns, two interleaved accumulators.
// MASKED: only columns < nv0 (lo field) / nv1 (hi field) count.
template <bool MASKED>
__device__ __forceinline__ uint32_t tile_row_max(const uint32_t (&X)[W], int nv0, int nv1)
{
    uint32_t acc0 = BIAS2, acc1 = BIAS2;      // V >= 0 everywhere, so V = 0 is neutral
#pragma unroll
    for (int i = 0; i < W; i += 4) {
        uint32_t x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = X[i + k] - drift2(i + k);
            if (MASKED) {
                if (i + k >= nv0) x[k] = (x[k] & 0xFFFF0000u) | BIAS16;
                if (i + k >= nv1) x[k] = (x[k] & 0x0000FFFFu) | (BIAS16 << 16);
            }
        }
        acc0 = max3(acc0, x[0], x[1]);
        acc1 = max3(acc1, x[2], x[3]);
    }
    return max2(acc0, acc1);
}

// ---------------------------------------------------------------------------------------------
// Finish stage
// ---------------------------------------------------------------------------------------------
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

__device__ __forceinline__ int nth_set_bit(uint64_t m, int n)   // index of the n-th (0-based) set bit, -1 if none
{
    for (int k = 0; k < n && m; ++k) m &= m - 1;
    return m ? __builtin_ctzll(m) : -1;
}

// Combine (tail of the fill kernels): per pair, the winning read splits (tools/SplitReadAligner.cpp:194-223),
// the kept rows that have columns on both sides, and the tiles that hold a row maximum for them.  Space
// comes from device cursors (capacities are checked by the host afterwards).  The first tile pair of
// every pair is offered to the table-driven replay: per fusion of the workgroup the first offer fixes
// the (M1 tile, M2 tile) the tables will be built for.
constexpr int TMASK_TILES = 16;   // tmask covers references of at most 16 tiles

// device buffers of the finish stage, handed to the fill kernels as one argument
struct FinishBufs {
    PairState* state;
    KeptRow* kept;
    ReplayTask* tasks;
    uint64_t* masks;
    uint2* gtasks;         // {task | GTASK_OWNER, pair} of every task the table-driven replay leaves to k_replay
    Counters* ctr;
    uint64_t kept_cap, task_cap, mask_cap, gtask_cap;
    int32_t* tstop;        // [wave][tile]: row groups of the tile whose bnd / cmax were stored (the rest is dead)
    int64_t* rec_count;    // [pair in the caller's order]: records of the pair
};

// tstop is read in the kernel that writes it, by other lanes and waves of the workgroup: bypass the
// vector L1, whose lines may predate the store
__device__ __forceinline__ int load_tstop(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// What combine hands to the table-driven replay of the same workgroup through LDS, so that the replay
// starts without a chain of dependent global loads (state -> task -> kept row).
struct LaneInfo {
    uint32_t kept_begin, mask_begin;
    uint16_t n_kept;
    uint16_t last_row;        // bit 15 (TASK_FAST): the pair's first task is replayed by the workgroup
    uint8_t  c0, c1;          // its tile pair (NO_CHUNK: side not replayed)
    uint16_t key_group;       // sort key min(a, lq - a, 255) of the first kept a | fusion group of the pair << 8
    uint16_t lq;
    uint8_t  nv0, nv1;        // valid columns of the two tiles
};
constexpr int KCACHE = 3;     // kept rows per pair that travel through LDS as well
struct FinishLds {
    int tile[GSPLIT2];
    int hist[258];
    unsigned short order[WG_LANES];
    LaneInfo info[WG_LANES];
    uint64_t kc[KCACHE * WG_LANES];    // [k][thread], a KeptRow each
};

// four wave-aggregated allocations at once: the four atomics are in flight together
__device__ __forceinline__ void wave_alloc4(Counters* ctr, const unsigned (&n)[4], unsigned long long (&base)[4])
{
    unsigned incl[4] = {n[0], n[1], n[2], n[3]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned y = __shfl_up(incl[q], d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl[q] += y;
        }
    }
    unsigned long long* const counter[4] = {&ctr->n_kept, &ctr->n_tasks, &ctr->n_masks, &ctr->n_gtasks};
    unsigned long long b[4] = {0, 0, 0, 0};
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (incl[q]) b[q] = atomicAdd(counter[q], (unsigned long long)incl[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) base[q] = __shfl(b[q], 63, 64) + (incl[q] - n[q]);
}

// Runs in the workgroup that filled these pairs, right after its row maxima are written: every lane
// re-reads only what it stored itself.  All threads of the workgroup must call it (barriers inside).
// fl->tile[k] returns the tile pair agreed for the table-driven replay of fusion group k (-1: none);
// HANDOFF also fills fl->info / fl->kc for replay_fast_wg.
template <bool HANDOFF>
__device__ __forceinline__ void combine_wg(
    const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions, const uint32_t* __restrict__ cmax,
    const uint32_t* __restrict__ rmax, const uint32_t* __restrict__ tmask, const int32_t* __restrict__ min_score_tab,
    const WgView& wgi, int my_group, bool fast_wg, FinishLds* fl, const FinishBufs& fb, const Geom& g)
{
    PairState* __restrict__ state = fb.state;
    KeptRow* __restrict__ kept = fb.kept;
    ReplayTask* __restrict__ tasks = fb.tasks;
    uint2* __restrict__ gtasks = fb.gtasks;
    const uint64_t kept_cap = fb.kept_cap, task_cap = fb.task_cap, mask_cap = fb.mask_cap, gtask_cap = fb.gtask_cap;
    const int tid = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * WG_LANES + tid;
    const bool active = p < g.n_pairs;
    const int64_t w = p >> 6;
    const int lane = (int)(p & 63);
    if (tid < GSPLIT2) fl->tile[tid] = -1;
    const uint32_t* rm = rmax + w * g.lq1 * WAVE;
    const uint4* rm4 = reinterpret_cast<const uint4*>(rm) + lane;
    const uint4* tm4 = reinterpret_cast<const uint4*>(tmask + w * g.lq1 * WAVE) + lane;
    // FindMaxRowEntry's acceptance rule (tools/SplitReadAligner.cpp:91-102): below minSplitScore counts as 0
    auto accept = [](uint32_t word, int h, int n_chunks, int row) -> int {
        if (row == 0 || n_chunks == 0) return 0;   // H(i,0)=0 < 8; empty reference: only column 0 (<=0)
        const int v = half_of(word, h) - 2 * row;
        return v >= DSA_MIN_SPLIT ? v : 0;
    };
    auto rowmax = [&](int h, int n_chunks, int row) -> int { return accept(rm[rowidx(row, lane)], h, n_chunks, row); };
    auto pick = [](const uint4& v, int k) -> uint32_t { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; };
    int lq = 0, nc0 = 0, nc1 = 0, max_score = 0, n_kept = 0, fidx = -1, len0 = 0, len1 = 0;
    unsigned n_t0 = 0, n_t1 = 0;
    uint64_t tiles0 = 0, tiles1 = 0;     // tiles that attain the maximum at some kept row
    bool small = true;                   // references with more than 64 tiles replay every tile (exact, not minimal)
    int first_a = 0, last_a = 0;
    uint64_t kcr[KCACHE] = {};           // the first kept rows (one-pass scan only)
    int n_cached = 0;
    if (active) {
        const dsa_pair pr = pairs[p];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        fidx = pr.fusion_idx;
        lq = pr.read_len;
        len0 = fu.ref0_len;
        len1 = fu.ref1_len;
        nc0 = cdiv_dev(fu.ref0_len, W);
        nc1 = cdiv_dev(fu.ref1_len, W);
        small = nc0 <= 64 && nc1 <= 64;
        const int min_score = min_score_tab[lq];
        if (g.nch <= TMASK_TILES) {
            // One pass over the read splits a, four per step: row maxima and winning-tile masks of the rows
            // a and lq-a come from six dwordx4 loads, issued two steps ahead (no dependent round trips).
            struct Step { uint4 x, yh, yl, tx, th, tl; };
            const int ng = lq >> 2;
            auto load = [&](int gq) -> Step {
                gq = gq < ng ? gq : ng;
                const int b_hi = lq - 4 * gq, b_lo = b_hi - 3;
                const int64_t ih = (int64_t)(b_hi >> 2) * WAVE, il = (int64_t)((b_lo > 0 ? b_lo : 0) >> 2) * WAVE;
                Step s;
                s.x = rm4[(int64_t)gq * WAVE];
                s.yh = rm4[ih];
                s.yl = rm4[il];
                s.tx = tm4[(int64_t)gq * WAVE];
                s.th = tm4[ih];
                s.tl = tm4[il];
                return s;
            };
            Step s0 = load(0), s1 = load(1);
            for (int gq = 0; gq <= ng; ++gq) {
                const Step cur = s0;
                s0 = s1;
                s1 = load(gq + 2);
                const int b_hi = lq - 4 * gq;
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int a = 4 * gq + sidx, b = lq - a;
                    if (b < 0) continue;
                    const bool in_hi = (b >> 2) == (b_hi >> 2);
                    const uint32_t yw = in_hi ? pick(cur.yh, b & 3) : pick(cur.yl, b & 3);
                    const int m1 = accept(pick(cur.x, sidx), 0, nc0, a), m2 = accept(yw, 1, nc1, b);
                    const int sc = m1 + m2;
                    if (sc < min_score || sc < max_score) continue;
                    if (sc > max_score) {        // a better split: forget the kept rows so far
                        max_score = sc;
                        n_kept = 0;
                        tiles0 = tiles1 = 0;
                    }
                    if (m1 == 0 || m2 == 0) continue;   // an empty side emits nothing
                    if (n_kept == 0) first_a = a;
                    last_a = a;
                    KeptRow kr;
                    kr.a = (int16_t)a;
                    kr.m1 = (int16_t)m1;
                    kr.m2 = (int16_t)m2;
                    kr.pad_ = 0;
#pragma unroll
                    for (int k = 0; k < KCACHE; ++k)
                        if (n_kept == k) kcr[k] = __builtin_bit_cast(uint64_t, kr);
                    ++n_kept;
                    tiles0 |= pick(cur.tx, sidx) & 0xFFFFu;
                    tiles1 |= (in_hi ? pick(cur.th, b & 3) : pick(cur.tl, b & 3)) >> 16;
                }
            }
            n_cached = n_kept < KCACHE ? n_kept : KCACHE;
        } else {
            // long references: two passes, the winning tiles looked up in the per-tile maxima
            auto for_each_split = [&](auto&& fn) {
                for (int gq = 0; 4 * gq <= lq; ++gq) {
                    const int b_hi = lq - 4 * gq, b_lo = b_hi - 3;
                    const uint4 x1 = rm4[(int64_t)gq * WAVE];
                    const uint4 y_hi = rm4[(int64_t)(b_hi >> 2) * WAVE];
                    const uint4 y_lo = rm4[(int64_t)((b_lo > 0 ? b_lo : 0) >> 2) * WAVE];
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) {
                        const int a = 4 * gq + sidx, b = lq - a;
                        if (b >= 0) {
                            const uint32_t yw = (b >> 2) == (b_hi >> 2) ? pick(y_hi, b & 3) : pick(y_lo, b & 3);
                            fn(a, b, accept(pick(x1, sidx), 0, nc0, a), accept(yw, 1, nc1, b));
                        }
                    }
                }
            };
            for_each_split([&](int, int, int m1, int m2) {
                const int sc = m1 + m2;
                if (sc >= min_score && sc > max_score) max_score = sc;
            });
            if (max_score != 0) {
                first_a = lq;
                for_each_split([&](int a, int b, int m1, int m2) {
                    if (m1 + m2 != max_score || m1 == 0 || m2 == 0) return;   // an empty side emits nothing
                    if (n_kept == 0) first_a = a;
                    last_a = a;
                    ++n_kept;
                    if (small) {
                        auto tile_max = [&](int c, int row) -> uint32_t {      // rows past the tile's stop are dead (V = 0)
                            return (row >> 2) < load_tstop(fb.tstop + w * g.nch + c) ? cmax[(w * g.nch + c) * g.lq1 * WAVE + rowidx(row, lane)] : BIAS2;
                        };
                        for (int c = 0; c < nc0; ++c)
                            if (half_of(tile_max(c, a), 0) == m1 + 2 * a) tiles0 |= 1ull << c;
                        for (int c = 0; c < nc1; ++c)
                            if (half_of(tile_max(c, b), 1) == m2 + 2 * b) tiles1 |= 1ull << c;
                    }
                });
            }
        }
        if (n_kept > 0) {
            n_t0 = small ? (unsigned)__builtin_popcountll(tiles0) : (unsigned)nc0;
            n_t1 = small ? (unsigned)__builtin_popcountll(tiles1) : (unsigned)nc1;
        }
    }
    unsigned n_tasks = n_t0 > n_t1 ? n_t0 : n_t1;
    // Which tile pair does the table-driven replay of this workgroup take for each of its fusions?  All reads of a
    // fusion have the junction's tiles among theirs, a tie elsewhere is a matter of one read: every pair votes for
    // each of its tiles, per side, and the pairs that have both winners replay them as their task 0 (any choice
    // gives the same records: what is left goes to the generic replay, and a tie's tile alone is a short task there).
    // Votes are 16-bit counters in the LDS words that later serve the replay's counting sort; windows of more
    // than VOTE_TILES tiles take the first offer of a (lowest M1 tile, lowest M2 tile) instead.
    const int gsel = my_group;            // the lane's run of pairs of one fusion (wg_groups)
    int f0 = small ? nth_set_bit(tiles0, 0) : (n_t0 > 0 ? 0 : -1), f1 = small ? nth_set_bit(tiles1, 0) : (n_t1 > 0 ? 0 : -1);   // tiles of task 0
    const bool eligible = n_tasks > 0 && small && fast_wg && gsel >= 0 && f0 >= 0 && f1 >= 0;
    const int key = eligible ? ((f0 << 8) | f1) : -1;
    constexpr int VOTE_TILES = 8;
    static_assert(GMAX * 2 * VOTE_TILES <= 2 * 258, "votes must fit the histogram words");
    const bool vote = g.nch <= VOTE_TILES && wgi.n_groups <= GMAX;   // uniform
    if (vote)
        for (int e = tid; e < 258; e += WG_LANES) fl->hist[e] = 0;
    __syncthreads();
    if (eligible) {
        if (vote) {
            for (uint64_t r = tiles0; r; r &= r - 1) {
                const int slot = (gsel * 2 + 0) * VOTE_TILES + __builtin_ctzll(r);
                atomicAdd(&fl->hist[slot >> 1], 1 << (16 * (slot & 1)));
            }
            for (uint64_t r = tiles1; r; r &= r - 1) {
                const int slot = (gsel * 2 + 1) * VOTE_TILES + __builtin_ctzll(r);
                atomicAdd(&fl->hist[slot >> 1], 1 << (16 * (slot & 1)));
            }
        } else
            atomicCAS(&fl->tile[gsel], -1, key);
    }
    __syncthreads();
    if (vote && tid < GMAX) {                                  // one thread per group picks its winners
        int win[2] = {-1, -1};
        for (int h = 0; h < 2; ++h) {
            int best = 0;
            for (int c = 0; c < VOTE_TILES; ++c) {
                const int slot = (tid * 2 + h) * VOTE_TILES + c;
                const int n = (fl->hist[slot >> 1] >> (16 * (slot & 1))) & 0xFFFF;
                if (n > best) { best = n; win[h] = c; }
            }
        }
        if (win[0] >= 0 && win[1] >= 0) fl->tile[tid] = (win[0] << 8) | win[1];
    }
    if (vote) __syncthreads();
    bool fast = false;
    if (eligible) {
        const int agreed = fl->tile[gsel];
        if (vote) {
            // One agreed tile is enough: the side that lacks its tile (a read whose anchor on that side is a few bases and
            // scores best somewhere else) sits out task 0, and its tiles follow as tasks of their own - short ones, where a
            // task that pairs them with the other side's junction tile would run the whole read.
            const bool h0 = agreed >= 0 && ((tiles0 >> (agreed >> 8)) & 1ull), h1 = agreed >= 0 && ((tiles1 >> (agreed & 0xFF)) & 1ull);
            fast = h0 || h1;
            if (fast) {
                f0 = h0 ? agreed >> 8 : -1;
                f1 = h1 ? agreed & 0xFF : -1;
                const unsigned s0 = n_t0 + (h0 ? 0u : 1u), s1 = n_t1 + (h1 ? 0u : 1u);
                n_tasks = s0 > s1 ? s0 : s1;
            }
        } else
            fast = agreed == key;
    }
    const unsigned n_gen = n_tasks - (fast ? 1u : 0u);

    // Sort key of the replays: a pair meets its kept rows at row a in M1 and at row lq - a in M2, so lanes
    // sorted by min(a, lq - a) share both hit rows with their neighbours (a and lq - a just swap sides).
    const int fold_a = first_a < lq - first_a ? first_a : lq - first_a;
    const unsigned want[4] = {(unsigned)n_kept, n_tasks, n_tasks * (unsigned)n_kept, n_gen};
    unsigned long long base[4];
    wave_alloc4(fb.ctr, want, base);
    const unsigned long long kb = base[0], tb = base[1], mb = base[2], gb = base[3];
    bool ok = active && n_kept > 0;
    if (ok && (kb + n_kept > kept_cap || tb + n_tasks > task_cap || mb + (unsigned long long)n_tasks * n_kept > mask_cap ||
               gb + n_gen > gtask_cap))
        ok = false;        // overflow: the host sees the cursors, grows the buffers and reruns the slice
    if (HANDOFF) {
        LaneInfo li;
        li.kept_begin = (uint32_t)kb;
        li.mask_begin = (uint32_t)mb;
        li.n_kept = (uint16_t)(ok ? n_kept : 0);
        const int c0 = f0 >= 0 ? f0 : (int)NO_CHUNK, c1 = f1 >= 0 ? f1 : (int)NO_CHUNK;
        const bool here = ok && fast;
        const int r0 = c0 != NO_CHUNK ? last_a : 0, r1 = c1 != NO_CHUNK ? lq - first_a : 0;
        li.last_row = here ? (uint16_t)((r0 > r1 ? r0 : r1) | TASK_FAST | (n_tasks == 1 ? TASK_ONLY : 0)) : (uint16_t)0;
        li.c0 = here ? (uint8_t)c0 : NO_CHUNK;
        li.c1 = here ? (uint8_t)c1 : NO_CHUNK;
        li.key_group = (uint16_t)((fold_a < 255 ? fold_a : 255) | ((gsel >= 0 ? gsel : 0) << 8));
        li.lq = (uint16_t)lq;
        const int v0 = len0 - c0 * W, v1 = len1 - c1 * W;
        li.nv0 = (uint8_t)(here && c0 != NO_CHUNK ? (v0 < W ? (v0 > 0 ? v0 : 0) : W) : 0);
        li.nv1 = (uint8_t)(here && c1 != NO_CHUNK ? (v1 < W ? (v1 > 0 ? v1 : 0) : W) : 0);
        fl->info[tid] = li;
    }
    if (!active) return;
    PairState st;
    st.mask_begin = (uint32_t)mb;
    st.kept_begin = (uint32_t)kb;
    st.task_begin = (uint32_t)tb;
    st.n_kept = (uint16_t)(ok ? n_kept : 0);
    st.n_tasks = (uint8_t)(n_tasks < 255u ? n_tasks : 255u);
    const bool tiles_fit = small && g.nch <= 16;
    // no kept row: no records, said here; a pair whose only task is replayed by this workgroup is counted by that lane
    st.flags = (uint8_t)(((!ok || (HANDOFF && fast && n_tasks == 1)) ? STATE_COUNTED : 0) | (tiles_fit ? STATE_TILES : 0));
    st.tiles0 = (uint16_t)(tiles_fit ? tiles0 : 0);
    st.tiles1 = (uint16_t)(tiles_fit ? tiles1 : 0);
    st.first0 = f0 >= 0 ? (uint8_t)f0 : NO_CHUNK;
    st.first1 = f1 >= 0 ? (uint8_t)f1 : NO_CHUNK;
    st.pad_ = 0;
    state[p] = st;
    if (!ok) {
        fb.rec_count[g.orig ? g.orig[p] : p] = 0;
        return;
    }
    // kept rows: the first ones are still in registers, the rest is re-read
    int k = 0, a_next = first_a;
    for (; k < n_cached; ++k) {
        const KeptRow kr = __builtin_bit_cast(KeptRow, k == 0 ? kcr[0] : k == 1 ? kcr[1] : kcr[KCACHE - 1]);
        kept[kb + k] = kr;
        if (HANDOFF) fl->kc[k * WG_LANES + tid] = __builtin_bit_cast(uint64_t, kr);
        a_next = kr.a + 1;
    }
    for (int a = a_next; k < n_kept && a <= last_a; ++a) {
        const int m1 = rowmax(0, nc0, a), m2 = rowmax(1, nc1, lq - a);
        if (m1 + m2 != max_score || m1 == 0 || m2 == 0) continue;
        KeptRow kr;
        kr.a = (int16_t)a;
        kr.m1 = (int16_t)m1;
        kr.m2 = (int16_t)m2;
        kr.pad_ = 0;
        kept[kb + k] = kr;
        if (HANDOFF && k < KCACHE) fl->kc[k * WG_LANES + tid] = __builtin_bit_cast(uint64_t, kr);
        ++k;
    }
    unsigned gi = 0;
    const uint64_t rest0 = f0 >= 0 && small ? tiles0 & ~(1ull << f0) : tiles0, rest1 = f1 >= 0 && small ? tiles1 & ~(1ull << f1) : tiles1;
    for (unsigned t = 0; t < n_tasks; ++t) {
        // task 0 has the pair's first tiles (the agreed ones if it has them), the others follow in ascending order
        const int c0 = small ? (t == 0 ? f0 : nth_set_bit(rest0, (int)t - 1)) : (t < n_t0 ? (int)t : -1);
        const int c1 = small ? (t == 0 ? f1 : nth_set_bit(rest1, (int)t - 1)) : (t < n_t1 ? (int)t : -1);
        ReplayTask rt;
        rt.pair = (uint32_t)p;
        rt.mask_begin = (uint32_t)(mb + (unsigned long long)t * n_kept);
        const int r0 = c0 >= 0 ? last_a : 0, r1 = c1 >= 0 ? lq - first_a : 0;
        rt.last_row = (uint16_t)(r0 > r1 ? r0 : r1);
        rt.chunk0 = c0 >= 0 ? (uint8_t)c0 : NO_CHUNK;
        rt.chunk1 = c1 >= 0 ? (uint8_t)c1 : NO_CHUNK;
        rt.fusion = fidx;
        if (t == 0 && fast)
            rt.last_row |= TASK_FAST;
        else {
            gtasks[gb + gi] = make_uint2((uint32_t)(tb + t) | (gi == 0 ? GTASK_OWNER : 0u), (uint32_t)p);
            ++gi;
        }
        tasks[tb + t] = rt;
    }
}

// Replay bookkeeping of one lane: the next kept row of either matrix, the row at which the sweep
// meets it and the (biased) value its row maximum has there.  Kept rows ascend in a: M1 (row a) meets
// them in order k = 0.., M2 (row lq-a) in reverse.  The first KCACHE kept rows may come from LDS
// (kc != nullptr: kc[k * WG_LANES] is kept row k of the pair).
struct HitCursor {
    int k0, k1;
    int row0, row1;        // -1: none left on that side
    uint32_t t0, t1;
    uint64_t valid0, valid1;   // columns of the tile that exist in the reference
};
__device__ __forceinline__ KeptRow kept_row(const KeptRow* __restrict__ kr, const uint64_t* kc, int k)
{
    if (kc != nullptr && k < KCACHE) return __builtin_bit_cast(KeptRow, kc[k * WG_LANES]);
    return kr[k];
}
__device__ __forceinline__ void cursor_next0(HitCursor& hc, const KeptRow* __restrict__ kr, const uint64_t* kc, int n_kept, bool has0)
{
    hc.row0 = -1;
    if (has0 && hc.k0 < n_kept) {
        const KeptRow r = kept_row(kr, kc, hc.k0);
        hc.row0 = r.a;
        hc.t0 = (uint32_t)(r.m1 + 2 * r.a) + BIAS16;
    }
}
__device__ __forceinline__ void cursor_next1(HitCursor& hc, const KeptRow* __restrict__ kr, const uint64_t* kc, int lq, bool has1)
{
    hc.row1 = -1;
    if (has1 && hc.k1 >= 0) {
        const KeptRow r = kept_row(kr, kc, hc.k1);
        hc.row1 = lq - r.a;
        hc.t1 = (uint32_t)(r.m2 + 2 * (lq - r.a)) + BIAS16;
    }
}
__device__ __forceinline__ HitCursor cursor_init(const KeptRow* __restrict__ kr, const uint64_t* kc, int n_kept, int lq,
                                                 bool has0, bool has1, int nv0, int nv1)
{
    HitCursor hc;
    hc.k0 = 0;
    hc.k1 = n_kept - 1;
    hc.t0 = hc.t1 = 0;
    hc.valid0 = nv0 >= W ? ~0ull : ((1ull << (nv0 > 0 ? nv0 : 0)) - 1ull);
    hc.valid1 = nv1 >= W ? ~0ull : ((1ull << (nv1 > 0 ? nv1 : 0)) - 1ull);
    cursor_next0(hc, kr, kc, n_kept, has0);
    cursor_next1(hc, kr, kc, lq, has1);
    return hc;
}

// Columns of the tile whose value at this row equals the given targets, both fields in one pass.  A target
// is the row maximum (no column exceeds it) or NO_TARGET16 (above every value), so per column
// sat(X - (target - 1 + drift)) is 1 exactly on a hit: one add (drift), one saturating packed subtract
// and one shift-or into a hit-bit accumulator.
constexpr uint32_t NO_TARGET16 = 0x7F00u;   // above BIAS16 + 4 * 7600 + drift, and + drift still fits the field
__device__ __forceinline__ void equal_columns(const uint32_t (&X)[W], uint32_t target2, uint64_t& m0, uint64_t& m1)
{
    uint32_t hit[W / 16];
    const uint32_t below = target2 - 0x00010001u;
#pragma unroll
    for (int q = 0; q < W / 16; ++q) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 15; k >= 0; --k) {
            const int i = 16 * q + k;
            const uint32_t c = below + drift2(i);
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const uint32_t y = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, X[i]), __builtin_bit_cast(us2, c)));   // v_pk_sub_u16 clamp
            asm("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(acc) : "v"(y));   // bit k: lo field hits, bit 16+k: hi field hits (left to
                                                                        // itself the compiler builds shift + or3 trees, 1.3 per column)
        }
        hit[q] = acc;
    }
    const uint32_t lo_a = (hit[0] & 0xFFFFu) | (hit[1] << 16), lo_b = (hit[2] & 0xFFFFu) | (hit[3] << 16);
    const uint32_t hi_a = (hit[0] >> 16) | (hit[1] & 0xFFFF0000u), hi_b = (hit[2] >> 16) | (hit[3] & 0xFFFF0000u);
    m0 = (uint64_t)lo_a | ((uint64_t)lo_b << 32);
    m1 = (uint64_t)hi_a | ((uint64_t)hi_b << 32);
}

// At row j: report, for the kept rows met here, the valid columns that attain the row maximum.
__device__ __forceinline__ void record_hits(const uint32_t (&X)[W], int j, int lq, const KeptRow* __restrict__ kr,
                                            const uint64_t* kc, int n_kept, bool has0, bool has1, HitCursor& hc,
                                            uint64_t* __restrict__ masks, uint32_t mask_begin)
{
    const bool hit0 = hc.row0 == j, hit1 = hc.row1 == j;
    if (!(hit0 || hit1)) return;
    uint64_t m0, m1;
    equal_columns(X, (hit0 ? hc.t0 : NO_TARGET16) | ((hit1 ? hc.t1 : NO_TARGET16) << 16), m0, m1);
    if (hit0) {
        masks[((uint64_t)mask_begin + hc.k0) * 2] = m0 & hc.valid0;
        ++hc.k0;
        cursor_next0(hc, kr, kc, n_kept, has0);
    }
    if (hit1) {
        masks[((uint64_t)mask_begin + hc.k1) * 2 + 1] = m1 & hc.valid1;
        --hc.k1;
        cursor_next1(hc, kr, kc, lq, has1);
    }
}

// Records of a pair whose columns all lie in one tile pair: per kept split the cross product of the two column
// sets minus the refSplits an earlier kept split already has (tools/SplitAlignment.cpp:381-391); mk[2k + h] is
// the column mask of kept row k in matrix h.  Same count as k_emit<false>.
__device__ __forceinline__ int64_t count_one_tile_pair(const uint64_t* mk, int n_kept)
{
    int64_t n = 0;
    for (int k = 0; k < n_kept; ++k) {
        const uint64_t m0 = mk[2 * k], m1 = mk[2 * k + 1];
        if (k == 0) {
            n += (int64_t)__builtin_popcountll(m0) * __builtin_popcountll(m1);
            continue;
        }
        for (uint64_t r0 = m0; r0; r0 &= r0 - 1)
            for (uint64_t r1 = m1; r1; r1 &= r1 - 1) {
                const int i1 = __builtin_ctzll(r0), i2 = __builtin_ctzll(r1);
                bool dup = false;
                for (int k2 = 0; k2 < k && !dup; ++k2) dup = ((mk[2 * k2] >> i1) & 1ull) && ((mk[2 * k2 + 1] >> i2) & 1ull);
                n += dup ? 0 : 1;
            }
    }
    return n;
}

// Table-driven replay (second tail of the fast fill kernel).  The first task of every pair of the
// workgroup whose tile pair matches its fusion's agreed (M1 tile, M2 tile) is replayed with the same
// LDS score tables as the fill (no row maxima, no stores: add + max3 per column).  Called by all
// threads of the workgroup after combine_wg<true>, whose LDS hand-off (fl) says what to replay; T is
// free by then (barriers inside combine_wg).
template <bool SPLIT>
__device__ __forceinline__ void replay_fast_wg(uint32_t* T, FinishLds* fl, const WgView& wgi, const FinishBufs& fb,
                                               const uint32_t* __restrict__ refcodes, const uint32_t* __restrict__ rowcodes,
                                               const uint32_t* __restrict__ bnd, const Geom& g)
{
    const KeptRow* __restrict__ kept = fb.kept;
    uint64_t* __restrict__ masks = fb.masks;
    bool any = false;
    for (int k = 0; k < wgi.n_groups; ++k) any |= fl->tile[k] >= 0;
    if (!any) return;                                   // uniform
    // tables for the agreed tile pair of every fusion of the workgroup
    build_tables<SPLIT>(T, wgi.n_groups, [&](int gi, int i, uint32_t& q0, uint32_t& q1) {
        const int key = fl->tile[gi];
        q0 = q1 = REF_PAD16;                 // no tile agreed: nobody reads this group's table
        if (key >= 0) {
            const int c0 = key >> 8, c1 = key & 0xFF;
            const uint32_t* rc = refcodes + (int64_t)group_fusion(wgi, gi) * g.lrp;
            if (c0 != NO_CHUNK) q0 = rc[c0 * W + i] & 0xFFFFu;
            if (c1 != NO_CHUNK) q1 = rc[c1 * W + i] >> 16;
        }
    });
    // Lanes take the workgroup's tasks in order of their first kept read split, so that the lanes of
    // a wave reach their kept rows (where the column masks are extracted) together and sweep about
    // the same number of rows.  Counting sort in LDS; any tie order gives the same output.
    for (int e = threadIdx.x; e < 258; e += WG_LANES) fl->hist[e] = 0;
    __syncthreads();                                    // also: info / kc of combine_wg are complete
    int my_key = 256, my_rank = 0;
    {
        const LaneInfo mine = fl->info[threadIdx.x];
        if (mine.last_row & TASK_FAST) my_key = mine.key_group & 0xFF;
        my_rank = atomicAdd(&fl->hist[my_key], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int e = 0; e < 257; ++e) {
            const int n = fl->hist[e];
            fl->hist[e] = run;
            run += n;
        }
    }
    __syncthreads();
    fl->order[fl->hist[my_key] + my_rank] = (unsigned short)threadIdx.x;
    __syncthreads();

    const int src = fl->order[threadIdx.x];
    const LaneInfo li = fl->info[src];
    const int64_t p = (int64_t)blockIdx.x * WG_LANES + src;
    const int lane = (int)(p & 63);
    const int64_t w = p >> 6;
    const bool has = (li.last_row & TASK_FAST) != 0;
    const int R = has ? (li.last_row & TASK_ROW) : 0;
    int Rw = R;                                          // wave maximum
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) Rw = max(Rw, __shfl_xor(Rw, d, 64));
    if (Rw == 0) return;                                  // wave-uniform; no barriers below
    const bool has0 = has && li.c0 != NO_CHUNK, has1 = has && li.c1 != NO_CHUNK;
    const int c0 = has0 ? li.c0 : 0, c1 = has1 ? li.c1 : 0;
    const uint32_t* tb = T + (li.key_group >> 8) * (SPLIT ? TGROUP_SPLIT : TGROUP);
    const int64_t wr = w < g.n_waves ? w : (int64_t)blockIdx.x * WG_WAVES;   // idle lanes read a valid plane
    const uint4* rows4 = reinterpret_cast<const uint4*>(rowcodes + wr * g.lq1 * WAVE) + lane;
    const uint4* bi0 = reinterpret_cast<const uint4*>(bnd + (wr * g.nch + (c0 - 1)) * g.lq1 * WAVE) + lane;
    const uint4* bi1 = reinterpret_cast<const uint4*>(bnd + (wr * g.nch + (c1 - 1)) * g.lq1 * WAVE) + lane;
    const KeptRow* kr = kept + li.kept_begin;
    const uint64_t* kc = fl->kc + src;
    const int lq = li.lq;
    const int n_kept = has ? li.n_kept : 0;
    HitCursor hc = cursor_init(kr, kc, n_kept, lq, has0, has1, li.nv0, li.nv1);
    const int stop0 = c0 > 0 ? load_tstop(fb.tstop + wr * g.nch + (c0 - 1)) : 0;   // stored row groups of the tiles to the left
    const int stop1 = c1 > 0 ? load_tstop(fb.tstop + wr * g.nch + (c1 - 1)) : 0;

    uint32_t X[W];
#pragma unroll
    for (int i = 0; i < W; ++i) X[i] = BIAS2 + drift2(i);
    const uint4 bias4 = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
    uint32_t bprev = BIAS2;
    const int ngq = (Rw >> 2) + 1;
    auto boundary = [&](int gq) -> uint4 {
        const uint4 x0 = gq < stop0 ? bi0[(int64_t)gq * WAVE] : bias4;
        const uint4 x1 = gq < stop1 ? bi1[(int64_t)gq * WAVE] : bias4;
        return make_uint4((x0.x & 0xFFFFu) | (x1.x & 0xFFFF0000u), (x0.y & 0xFFFFu) | (x1.y & 0xFFFF0000u),
                          (x0.z & 0xFFFFu) | (x1.z & 0xFFFF0000u), (x0.w & 0xFFFFu) | (x1.w & 0xFFFF0000u));
    };
    uint4 rc_n = rows4[0];
    uint4 b_n = boundary(0);
    for (int gq = 0; gq < ngq; ++gq) {
        const uint4 rc = rc_n, b = b_n;
        const int gn = gq + 1 < ngq ? gq + 1 : gq;
        rc_n = rows4[(int64_t)gn * WAVE];
        b_n = boundary(gn);
        const uint32_t rcv[4] = {rc.x, rc.y, rc.z, rc.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int j = 4 * gq + sidx;
            const uint32_t bcur = bv[sidx];
            if (j >= 1 && j <= Rw) {                     // wave-uniform
                const uint4* trow = reinterpret_cast<const uint4*>(tb + (rcv[sidx] & 0xFFu) * TROW);       // combined tables
                auto table4 = [&](int q) -> uint4 {
                    if (SPLIT) return split_terms(tb, (rcv[sidx] >> 16) & 0xFu, (rcv[sidx] >> 20) & 0xFu, q);
                    return trow[q];
                };
                uint4 v = table4(0);
                uint32_t a = bprev + v.x;
                uint32_t up = bcur - TWO2;
#pragma unroll
                for (int q = 0; q < W / 4; ++q) {
                    uint4 vn = v;
                    if (q + 1 < W / 4) vn = table4(q + 1);
                    uint32_t an;
                    an = X[4 * q + 0] + v.y;
                    X[4 * q + 0] = max3(a, X[4 * q + 0], up);
                    a = an;
                    an = X[4 * q + 1] + v.z;
                    X[4 * q + 1] = max3(a, X[4 * q + 1], X[4 * q + 0]);
                    a = an;
                    an = X[4 * q + 2] + v.w;
                    X[4 * q + 2] = max3(a, X[4 * q + 2], X[4 * q + 1]);
                    a = an;
                    an = X[4 * q + 3] + vn.x;
                    X[4 * q + 3] = max3(a, X[4 * q + 3], X[4 * q + 2]);
                    a = an;
                    up = X[4 * q + 3];
                    v = vn;
                }
                record_hits(X, j, lq, kr, kc, n_kept, has0, has1, hc, masks, li.mask_begin);
            }
            bprev = bcur;
        }
    }
    if (has && !has0)
        for (int k = 0; k < n_kept; ++k) masks[((uint64_t)li.mask_begin + k) * 2] = 0;
    if (has && !has1)
        for (int k = 0; k < n_kept; ++k) masks[((uint64_t)li.mask_begin + k) * 2 + 1] = 0;
    // a pair with no other task has all its columns here: count its records now (k_emit<false> skips it)
    if (has && (li.last_row & TASK_ONLY) && p < g.n_pairs)
        fb.rec_count[g.orig ? g.orig[p] : p] = count_one_tile_pair(masks + (uint64_t)li.mask_begin * 2, n_kept);
}

// ---------------------------------------------------------------------------------------------
// K1g: generic DP fill.  One wave = 64 pairs; 4 waves per workgroup.  Runs only the workgroups
// flagged generic (exotic read bytes or more than GMAX fusions in the workgroup).
//   cmax[((w*nch + c)*lq1 + j)*64 + lane] = max over the tile's valid columns of V(.,j)   (2 x u16, biased)
//   bnd [((w*nch + c)*lq1 + j)*64 + lane] = V(last column of tile c, j)
// ---------------------------------------------------------------------------------------------
// Pruning as in k_fill_fast (exact): lq_lane / slack are the lane's read length and
// 2*Lq - minScore, l_in the last row at which a live value can still enter from the left, stop_prev the
// stored row groups of the tile to the left.  Returns the row groups stored for this tile; last_bnd is
// the last row whose outgoing boundary is alive for this lane.
template <bool MASKED>
__device__ __forceinline__ int sweep_tile_generic(const uint32_t (&r)[W], const uint4* __restrict__ rows4,
                                                  const uint4* __restrict__ bi4, uint4* __restrict__ cm4,
                                                  uint4* __restrict__ bo4, int lq, bool first, int nv0, int nv1,
                                                  int lq_lane, int slack, int l_in, int stop_prev, int& last_bnd)
{
    uint32_t X[W];
#pragma unroll
    for (int i = 0; i < W; ++i) X[i] = BIAS2 + drift2(i);
    const uint4 bias4 = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
    uint32_t bprev = BIAS2;
    const int ngq = (lq >> 2) + 1;
    uint4 rc_n = rows4[0];
    uint4 b_n = first ? bias4 : bi4[0];                     // every tile stores at least its first row group
    last_bnd = 0;
    int gq = 0;
    for (; gq < ngq; ++gq) {
        const uint4 rc = rc_n, b = b_n;
        const int gn = gq + 1 < ngq ? gq + 1 : gq;          // prefetch the next four rows' operands
        rc_n = rows4[(int64_t)gn * WAVE];
        b_n = (!first && gn < stop_prev) ? bi4[(int64_t)gn * WAVE] : bias4;   // past the left tile's stop: dead, V = 0
        const uint32_t rcv[4] = {rc.x, rc.y, rc.z, rc.w}, bv[4] = {b.x, b.y, b.z, b.w};
        uint32_t cmv[4] = {BIAS2, BIAS2, BIAS2, BIAS2}, bov[4] = {BIAS2, BIAS2, BIAS2, BIAS2};
        bool alive = false;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int j = 4 * gq + sidx;
            if (j >= 1 && j <= lq) {                        // wave-uniform
                row_step(X, r, rcv[sidx], bprev, bv[sidx]);
                cmv[sidx] = tile_row_max<MASKED>(X, nv0, nv1);
                bov[sidx] = X[W - 1] - drift2(W - 1);
                const int thr = 4 * j - slack + (int)BIAS16;
                const bool in_read = j <= lq_lane;
                alive |= in_read && ((int)(cmv[sidx] & 0xFFFFu) >= thr || (int)(cmv[sidx] >> 16) >= thr);
                if (in_read && ((int)(bov[sidx] & 0xFFFFu) >= thr || (int)(bov[sidx] >> 16) >= thr)) last_bnd = j;
            }
            bprev = bv[sidx];
        }
        cm4[(int64_t)gq * WAVE] = make_uint4(cmv[0], cmv[1], cmv[2], cmv[3]);
        bo4[(int64_t)gq * WAVE] = make_uint4(bov[0], bov[1], bov[2], bov[3]);
        if (DIAG_PRUNE && 4 * gq + 3 > l_in && __builtin_amdgcn_ballot_w64(alive) == 0) { ++gq; break; }   // wave-uniform; a live boundary at row l_in also enters row l_in + 1 (diagonal)
    }
    return gq;
}

// After the last tile: rmax = max over tiles of cmax (both fields), so the combine step reads one
// dword per row instead of one per tile; tmask = which tiles attain it (bit c: M1 tile c, bit 16+c:
// M2 tile c; only meaningful while a reference has at most 16 tiles, the combine step falls back to
// cmax otherwise).  cmax of this wave is L2-hot.
__device__ __forceinline__ uint32_t eq_bits(uint32_t v, uint32_t m, int c)
{
    // bit c if the lo fields are equal, bit 16+c if the hi fields are: min(v ^ m, 1) per field is the
    // "differs" flag of both fields at once (one packed instruction instead of two compares and selects)
    uint32_t differs;
    const uint32_t one2 = 0x00010001u;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(differs) : "v"(v ^ m), "v"(one2));
    return (differs ^ one2) << c;
}
struct TileStops { int v[8]; };    // stored row groups of a wave's first tiles (wave-uniform)
// WIDE: with the in-register path for 9 to 16 tiles.  It is a template parameter of the fill kernels (the host picks the
// instantiation by the slice's tile count) because its mere presence in a kernel costs the sweep of the common, narrower
// shapes 4 % — the compiler's schedule of the sweep loop is sensitive to what else the kernel holds (profiles/r04/ab_rm16.txt).
template <bool WIDE>
__device__ __forceinline__ void reduce_row_max(const uint32_t* __restrict__ cmax, uint32_t* __restrict__ rmax,
                                               uint32_t* __restrict__ tmask, const int32_t* tstop, const TileStops& stops,
                                               const Geom& g, int w, int lane, int nch_wave, int lq)
{
    const int ngq = (lq >> 2) + 1;
    const uint4 dead4 = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);   // what the rows past a tile's stop stand for
    uint4* out = reinterpret_cast<uint4*>(rmax + (int64_t)w * g.lq1 * WAVE) + lane;
    uint4* tout = reinterpret_cast<uint4*>(tmask + (int64_t)w * g.lq1 * WAVE) + lane;
    const uint4* src = reinterpret_cast<const uint4*>(cmax + (int64_t)w * g.nch * g.lq1 * WAVE) + lane;
    const int64_t cstride = (int64_t)(g.lq1 >> 2) * WAVE;       // uint4 elements between two tiles
    constexpr int NC = 8;
    if (nch_wave <= NC) {
        // all tiles of a row group in registers: one round of independent loads, then max and masks
        for (int gq = 0; gq < ngq; ++gq) {
            uint4 v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                v[c] = c < nch_wave ? (gq < stops.v[c] ? src[c * cstride + (int64_t)gq * WAVE] : dead4) : make_uint4(0, 0, 0, 0);
            uint4 m = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < nch_wave) {
                    m.x = max2(m.x, v[c].x);
                    m.y = max2(m.y, v[c].y);
                    m.z = max2(m.z, v[c].z);
                    m.w = max2(m.w, v[c].w);
                }
            uint4 t = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < nch_wave) {
                    t.x |= eq_bits(v[c].x, m.x, c);
                    t.y |= eq_bits(v[c].y, m.y, c);
                    t.z |= eq_bits(v[c].z, m.z, c);
                    t.w |= eq_bits(v[c].w, m.w, c);
                }
            out[(int64_t)gq * WAVE] = m;
            tout[(int64_t)gq * WAVE] = t;
        }
        return;
    }
    if (WIDE && nch_wave <= 2 * NC) {
        // 9 to 16 tiles (2x150 bp: windows of 590 bases are ten tiles): the same in two rounds of eight, the stops of the tiles
        // beyond the eighth read back (wave-uniform).  With the tile-by-tile loop below this reduction was 37 % of the fill
        // kernel's wave cycles at 2x150 (profiles/r04/mix_stats.txt).
        int stop_hi[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) stop_hi[c] = NC + c < nch_wave ? load_tstop(tstop + (int64_t)w * g.nch + NC + c) : 0;
        for (int gq = 0; gq < ngq; ++gq) {
            uint4 v[NC], u[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                v[c] = gq < stops.v[c] ? src[c * cstride + (int64_t)gq * WAVE] : dead4;
                u[c] = NC + c < nch_wave ? (gq < stop_hi[c] ? src[(NC + c) * cstride + (int64_t)gq * WAVE] : dead4) : make_uint4(0, 0, 0, 0);
            }
            uint4 m = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                m.x = max2(m.x, v[c].x); m.y = max2(m.y, v[c].y); m.z = max2(m.z, v[c].z); m.w = max2(m.w, v[c].w);
                if (NC + c < nch_wave) { m.x = max2(m.x, u[c].x); m.y = max2(m.y, u[c].y); m.z = max2(m.z, u[c].z); m.w = max2(m.w, u[c].w); }
            }
            uint4 t = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                t.x |= eq_bits(v[c].x, m.x, c); t.y |= eq_bits(v[c].y, m.y, c); t.z |= eq_bits(v[c].z, m.z, c); t.w |= eq_bits(v[c].w, m.w, c);
                if (NC + c < nch_wave) {
                    t.x |= eq_bits(u[c].x, m.x, NC + c); t.y |= eq_bits(u[c].y, m.y, NC + c); t.z |= eq_bits(u[c].z, m.z, NC + c); t.w |= eq_bits(u[c].w, m.w, NC + c);
                }
            }
            out[(int64_t)gq * WAVE] = m;
            tout[(int64_t)gq * WAVE] = t;
        }
        return;
    }
    for (int gq = 0; gq < ngq; ++gq) {
        uint4 m = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
        for (int c = 0; c < nch_wave; ++c) {
            const uint4 v = gq < load_tstop(tstop + (int64_t)w * g.nch + c) ? src[c * cstride + (int64_t)gq * WAVE] : dead4;
            m.x = max2(m.x, v.x);
            m.y = max2(m.y, v.y);
            m.z = max2(m.z, v.z);
            m.w = max2(m.w, v.w);
        }
        out[(int64_t)gq * WAVE] = m;
        if (nch_wave <= TMASK_TILES) {
            uint4 t = make_uint4(0, 0, 0, 0);
            for (int c = 0; c < nch_wave; ++c) {
                const uint4 v = gq < load_tstop(tstop + (int64_t)w * g.nch + c) ? src[c * cstride + (int64_t)gq * WAVE] : dead4;
                t.x |= eq_bits(v.x, m.x, c);
                t.y |= eq_bits(v.y, m.y, c);
                t.z |= eq_bits(v.z, m.z, c);
                t.w |= eq_bits(v.w, m.w, c);
            }
            tout[(int64_t)gq * WAVE] = t;
        }
    }
}

__global__ __launch_bounds__(WG_LANES) void k_fill_generic(const dsa_pair* __restrict__ pairs,
                                                           const dsa_fusion* __restrict__ fusions,
                                                           const uint8_t* __restrict__ wg_tier,
                                                           const uint32_t* __restrict__ refcodes,
                                                           const uint8_t* __restrict__ read_bytes,
                                                           uint32_t* __restrict__ rowcodes,
                                                           const int32_t* __restrict__ min_score_tab,
                                                           uint32_t* __restrict__ bnd, uint32_t* __restrict__ cmax,
                                                           uint32_t* __restrict__ rmax, uint32_t* __restrict__ tmask,
                                                           FinishBufs fb, Geom g)
{
    __shared__ FinishLds fl;
    if (wg_tier[blockIdx.x] != TIER_GENERIC) return;     // a table kernel owns this workgroup (uniform)
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WG_WAVES + (threadIdx.x >> 6)));
    if (w < g.n_waves) {                         // whole waves past the end only join the combine barriers
        const int lane = threadIdx.x & 63;
        const int64_t p = min((int64_t)w * WAVE + lane, g.n_pairs - 1);   // tail lanes shadow the last pair
        const bool in_batch = (int64_t)w * WAVE + lane < g.n_pairs;
        const int f = pairs[p].fusion_idx;
        const dsa_fusion fu = fusions[f];
        auto wave_max = [](int v) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
            return v;
        };
        WaveInfo wi;                             // rows and tiles of the wave, from its own pairs
        wi.lq_max = wave_max(in_batch ? (int)pairs[p].read_len : 0);
        wi.nch_max = wave_max(in_batch ? max(cdiv_dev(fu.ref0_len, W), cdiv_dev(fu.ref1_len, W)) : 0);
        (void)pack_rows_wave<0>(read_bytes, pairs, rowcodes, nullptr, g, w, lane, wi.lq_max);   // the fast kernel packs only what it keeps
        const uint32_t* rc = refcodes + (int64_t)f * g.lrp;
        const uint4* rows4 = reinterpret_cast<const uint4*>(rowcodes + (int64_t)w * g.lq1 * WAVE) + lane;
        int lq_lane = 0, slack = 0;
        if (in_batch) {
            lq_lane = pairs[p].read_len;
            slack = 2 * lq_lane - max(min_score_tab[lq_lane], pair_bound(pairs[p]));
        }
        int l_in = wave_max(lq_lane > 0 ? min(slack >> 2, lq_lane) : 0);
        int stop_prev = 0;
        TileStops stops = {};

        for (int c = 0; c < wi.nch_max; ++c) {
            uint32_t r[W];
#pragma unroll
            for (int i = 0; i < W; ++i) r[i] = rc[c * W + i];
            const int nv0 = fu.ref0_len - c * W, nv1 = fu.ref1_len - c * W;   // per lane; may be <= 0
            uint4* cm4 = reinterpret_cast<uint4*>(cmax + ((int64_t)w * g.nch + c) * g.lq1 * WAVE) + lane;
            uint4* bo4 = reinterpret_cast<uint4*>(bnd + ((int64_t)w * g.nch + c) * g.lq1 * WAVE) + lane;
            const uint4* bi4 = reinterpret_cast<const uint4*>(bnd + ((int64_t)w * g.nch + (c - 1)) * g.lq1 * WAVE) + lane;
            int last_bnd, stop;
            if (__builtin_amdgcn_ballot_w64(nv0 < W || nv1 < W) == 0)
                stop = sweep_tile_generic<false>(r, rows4, bi4, cm4, bo4, wi.lq_max, c == 0, nv0, nv1, lq_lane, slack, l_in, stop_prev, last_bnd);
            else
                stop = sweep_tile_generic<true>(r, rows4, bi4, cm4, bo4, wi.lq_max, c == 0, nv0, nv1, lq_lane, slack, l_in, stop_prev, last_bnd);
            if (lane == 0) fb.tstop[(int64_t)w * g.nch + c] = stop;
            stop_prev = stop;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k == c) stops.v[k] = stop;
            l_in = wave_max(last_bnd);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // own stores before own re-reads
        reduce_row_max<false>(cmax, rmax, tmask, fb.tstop, stops, g, w, lane, wi.nch_max, wi.lq_max);
    }
    WgView wgi = {};                             // no tables, no groups: every task goes to k_replay
    wgi.list = nullptr;
    combine_wg<false>(pairs, fusions, cmax, rmax, tmask, min_score_tab, wgi, 0, false, &fl, fb, g);
}

// The sweep of one tile by one wave of the table kernels (TIER 0: 25-row tables, SPLIT: two 5-row tables per fusion): rows in
// groups of four, per row one ascending pass over the tile's NW columns — four columns per table read, the diagonal term of the
// next column formed from X[i] before X[i] is overwritten, the chain max3 -> max3 — the tile's row maximum and, unless the tile
// is the wave's last (LAST: nothing reads it), its boundary column.  NW = 64; a 16-column instantiation for a last tile of which
// no lane has more than 16 columns (windows of 389 = 6 x 64 + 5 bases) exists behind -DDSA_NARROW only: measured on one box
// against the same build without it, the fill took 3.100 instead of 3.050 ms — the second copy of the loop costs more
// (instruction cache, register allocation) than the quarter-width seventh pass saves.  Exact pruning as described at the
// kernel.  Returns the row groups stored for the tile; last_bnd = last row whose outgoing boundary is alive (this lane).
template <int NW, bool SPLIT, bool LAST>
__device__ __forceinline__ int sweep_tile_fast(const uint32_t* __restrict__ tb, const uint32_t* __restrict__ rows1, const uint4* __restrict__ bi4,
                                               uint4* __restrict__ cm4, uint4* __restrict__ bo4, bool first_tile, int lq_max, int lq_lane,
                                               int slack, int l_in, int stop_prev, int& last_bnd, const Geom& g)
{
    static_assert(NW % 4 == 0 && NW / 4 > FILL_PF, "table reads in flight");
    (void)g;
    uint32_t X[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) X[i] = BIAS2 + drift2(i);
    const uint4 bias4 = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
    uint32_t bprev = BIAS2;
    const int ngq = (lq_max >> 2) + 1;
    uint32_t rc_n = rows1[0];                               // four rows' bytes: table row / classes
    uint4 b_n = first_tile ? bias4 : bi4[0];                // every tile stores at least its first row group
    last_bnd = 0;
    int gq = 0;
    int n_gap_groups = 0;
    for (; gq < ngq; ++gq) {
        const uint32_t rc = rc_n;
        const uint4 b = b_n;
        const int gn = gq + 1 < ngq ? gq + 1 : gq;          // prefetch the next four rows' operands
        rc_n = rows1[(int64_t)gn * WAVE];
        b_n = gn < stop_prev ? bi4[(int64_t)gn * WAVE] : bias4;   // past the left tile's stop: dead, V = 0
        const uint32_t bv[4] = {b.x, b.y, b.z, b.w};
        uint32_t cmv[4] = {BIAS2, BIAS2, BIAS2, BIAS2}, bov[4] = {BIAS2, BIAS2, BIAS2, BIAS2};
        uint32_t alive_bits = 0;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int j = 4 * gq + sidx;
            const uint32_t bcur = bv[sidx];
            if (j >= 1 && j <= lq_max) {                    // wave-uniform
                if constexpr (SPLIT) {
                    const uint32_t k1 = (rc >> (8 * sidx)) & 0xFu, k2 = (rc >> (8 * sidx + 4)) & 0xFu;
                    uint4 v = split_terms(tb, k1, k2, 0);
                    uint32_t a = bprev + v.x;
                    uint32_t up = bcur - TWO2;
#pragma unroll
                    for (int q = 0; q < NW / 4; ++q) {
                        uint4 vn = v;                         // one pair of table reads in flight
                        if (q + 1 < NW / 4) vn = split_terms(tb, k1, k2, q + 1);
                        uint32_t an;
                        an = X[4 * q + 0] + v.y;
                        X[4 * q + 0] = max3(a, X[4 * q + 0], up);
                        a = an;
                        an = X[4 * q + 1] + v.z;
                        X[4 * q + 1] = max3(a, X[4 * q + 1], X[4 * q + 0]);
                        a = an;
                        an = X[4 * q + 2] + v.w;
                        X[4 * q + 2] = max3(a, X[4 * q + 2], X[4 * q + 1]);
                        a = an;
                        an = X[4 * q + 3] + vn.x;
                        X[4 * q + 3] = max3(a, X[4 * q + 3], X[4 * q + 2]);
                        a = an;
                        up = X[4 * q + 3];
                        v = vn;
                    }
                } else {
                    const uint4* trow = reinterpret_cast<const uint4*>(tb + ((rc >> (8 * sidx)) & 0xFFu) * TROW);
                    uint4 vq[FILL_PF + 1];                  // table reads in flight
#pragma unroll
                    for (int k = 0; k <= FILL_PF; ++k) vq[k] = trow[k];
                    uint32_t a = bprev + vq[0].x;
                    uint32_t up = bcur - TWO2;
#pragma unroll
                    for (int q = 0; q < NW / 4; ++q) {
                        const uint4 v = vq[0];
#pragma unroll
                        for (int k = 0; k < FILL_PF; ++k) vq[k] = vq[k + 1];
                        if (q + 1 + FILL_PF < NW / 4) vq[FILL_PF] = trow[q + 1 + FILL_PF];
                        const uint4 vn = vq[0];
                        uint32_t an;
                        an = X[4 * q + 0] + v.y;
                        X[4 * q + 0] = max3(a, X[4 * q + 0], up);
                        a = an;
                        an = X[4 * q + 1] + v.z;
                        X[4 * q + 1] = max3(a, X[4 * q + 1], X[4 * q + 0]);
                        a = an;
                        an = X[4 * q + 2] + v.w;
                        X[4 * q + 2] = max3(a, X[4 * q + 2], X[4 * q + 1]);
                        a = an;
                        an = X[4 * q + 3] + vn.x;
                        X[4 * q + 3] = max3(a, X[4 * q + 3], X[4 * q + 2]);
                        a = an;
                        up = X[4 * q + 3];
                    }
                }
                {   // the tile's row maximum: one max3 per two columns, two interleaved accumulators (padded columns have
                    // substitution term 0 and never exceed the valid ones: no masking)
                    uint32_t acc0 = BIAS2, acc1 = BIAS2;
#pragma unroll
                    for (int i = 0; i < NW; i += 4) {
                        acc0 = max3(acc0, X[i] - drift2(i), X[i + 1] - drift2(i + 1));
                        acc1 = max3(acc1, X[i + 2] - drift2(i + 2), X[i + 3] - drift2(i + 3));
                    }
                    cmv[sidx] = max2(acc0, acc1);
                }
                // alive: a field >= thr = 4j - slack (biased).  Both fields at once: x >= thr <=> max(x, thr-1) != thr-1;
                // rows past the lane's read compare against 0xFFFF, which nothing exceeds.
                const int t1 = max(4 * j - slack + (int)BIAS16 - 1, 0);
                const uint32_t tm2 = j <= lq_lane ? (uint32_t)t1 * 0x00010001u : 0xFFFFFFFFu;
                alive_bits |= pk_max_u16(cmv[sidx], tm2) ^ tm2;
                if (!LAST) {
                    bov[sidx] = X[NW - 1] - drift2(NW - 1);
                    if ((pk_max_u16(bov[sidx], tm2) ^ tm2) != 0u) last_bnd = j;
                }
            }
            bprev = bcur;
        }
        const bool alive = alive_bits != 0u;
        cm4[(int64_t)gq * WAVE] = make_uint4(cmv[0], cmv[1], cmv[2], cmv[3]);
        if (!LAST) bo4[(int64_t)gq * WAVE] = make_uint4(bov[0], bov[1], bov[2], bov[3]);
        if (DIAG_PRUNE && __builtin_amdgcn_ballot_w64(alive) == 0) {      // wave-uniform
            if (4 * gq + 3 > l_in) { ++gq; break; }             // a live boundary at row l_in also enters row l_in + 1 (diagonal)
            if (DIAG_GAP_SKIP) {
                // Nothing in these four rows is alive, so rows further down can only come alive through the boundary
                // column (a live cell's best predecessor is alive: within the tile that chain would cross these rows).
                // Until a row group has a live incoming boundary value the sweep is skipped: those groups are all dead
                // and store V = 0, the lower bound every reader substitutes for dead cells anyway.
                int g2 = gq + 1;
                bool resume = false;
                uint4 bb = b_n;                                     // group gq + 1, already on its way
                uint32_t in_prev = bv[3];                           // the boundary value of the row above the group: it enters by the diagonal
                for (; g2 < ngq; ++g2) {
                    if (g2 > gq + 1) bb = g2 < stop_prev ? bi4[(int64_t)g2 * WAVE] : bias4;
                    const uint32_t bbv[5] = {in_prev, bb.x, bb.y, bb.z, bb.w};
                    uint32_t in_bits = 0;
#pragma unroll
                    for (int sidx = 0; sidx < 5; ++sidx) {
                        const int j = 4 * g2 + sidx - 1;
                        const int t1 = max(4 * j - slack + (int)BIAS16 - 1, 0);
                        const uint32_t tm2 = j <= lq_lane ? (uint32_t)t1 * 0x00010001u : 0xFFFFFFFFu;
                        in_bits |= pk_max_u16(bbv[sidx], tm2) ^ tm2;
                    }
                    if (__builtin_amdgcn_ballot_w64(in_bits != 0u) != 0) { resume = true; break; }
                    cm4[(int64_t)g2 * WAVE] = bias4;
                    if (!LAST) bo4[(int64_t)g2 * WAVE] = bias4;
                    in_prev = bb.w;
                    if (4 * g2 + 3 > l_in) { ++g2; break; }
                }
                n_gap_groups += g2 - (gq + 1);
                if (!resume) { gq = g2; break; }                    // g2 groups are stored
#pragma unroll
                for (int i = 0; i < NW; ++i) X[i] = BIAS2 + drift2(i);
                bprev = in_prev;
                b_n = bb;
                rc_n = rows1[(int64_t)g2 * WAVE];
                gq = g2 - 1;
            }
        }
    }
    if ((threadIdx.x & 63) == 0) {
        DSA_STAT_ADD(g, DS_GAP_GROUPS, n_gap_groups);
        DSA_STAT_ADD(g, DS_GROUPS_SKIPPED, ngq - gq);
        DSA_STAT_ADD(g, DS_GROUPS, ngq);
    }
    return gq;
}

// ---------------------------------------------------------------------------------------------
// K1f: fast DP fill (reads over {A,C,G,T,N}).  Per workgroup and tile, the substitution terms of
// every fusion present are tabulated in LDS:
//     T[g][k1*5+k2][i] = { d(ref0_g[i], base[k1]), d(rev(ref1_g)[i], base[k2]) },  d = eq ? 4 : 1
// where k1/k2 are the classes of the M1 / M2 read base of the row.  Padded reference columns get
// d = 0 (worse than a mismatch): their values can then reach but never exceed the row maximum of the
// valid columns, so the tile row maximum needs no masking (the replay masks exclude them by index).
// A row costs one ds_read_b128 per 4 columns and 2 adds + 1.5 max3 per column; four rows share one
// dwordx4 load of row codes / boundary and one dwordx4 store of tile maxima / boundary.
// ---------------------------------------------------------------------------------------------
// TIER 0: one 25-row table per fusion, at most GMAX fusions; 1: split tables, at most GSPLIT; 2: split tables in
// twice the LDS (two workgroups per CU), at most GSPLIT2.
__host__ __device__ constexpr int tier_of(int n_groups) { return n_groups <= GMAX ? 0 : n_groups <= GSPLIT ? 1 : 2; }
#ifndef DSA_FAST_WGS
#define DSA_FAST_WGS 4        // workgroups (of four waves) per CU the table tiers 0 and 1 are compiled for
#endif
template <int TIER, bool WIDE = false>
__global__ __launch_bounds__(WG_LANES, TIER == 2 ? 2 : DSA_FAST_WGS) void k_fill_fast(const dsa_pair* __restrict__ pairs,
                                                           uint8_t* __restrict__ wg_tier,
                                                           const uint32_t* __restrict__ refcodes,
                                                           const uint8_t* __restrict__ read_bytes,
                                                           uint32_t* __restrict__ rowcodes, uint32_t* __restrict__ row_bytes,
                                                           const int32_t* __restrict__ min_score_tab,
                                                           const dsa_fusion* __restrict__ fusions,
                                                           uint32_t* __restrict__ bnd, uint32_t* __restrict__ cmax,
                                                           uint32_t* __restrict__ rmax, uint32_t* __restrict__ tmask,
                                                           FinishBufs fb, Geom g)
{
    constexpr bool SPLIT = TIER > 0;
    __shared__ __attribute__((aligned(16))) uint32_t T[TIER == 2 ? GSPLIT2 * TGROUP_SPLIT : GMAX * TGROUP];
    __shared__ int s_nch, s_exotic;
    __shared__ FinishLds fl;
    __shared__ WgGroupsLds gl;
    // k_fill_fast<0> is launched first and says which kernel owns every workgroup; the others read that (uniform)
    if (TIER != 0 && wg_tier[blockIdx.x] != TIER) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WG_WAVES + (threadIdx.x >> 6)));
    const bool live = w < g.n_waves;             // whole waves past the end still join the barriers
    const int lane = threadIdx.x & 63;
    // The descriptors of the workgroup come from its own pairs: the runs of pairs of one fusion (table groups), the wave's
    // longest read and widest window.
    const int64_t p = (int64_t)w * WAVE + lane;
    const bool in_batch = p < g.n_pairs;
    const dsa_pair pr = pairs[min(p, g.n_pairs - 1)];      // lanes past the end shadow the last pair
    const int f = pr.fusion_idx;
    int my_group;
    const WgView wgi = wg_groups(&gl, f, (lane == 0 && threadIdx.x != 0) ? pairs[min(p - 1, g.n_pairs - 1)].fusion_idx : -1, my_group);
    if (TIER == 0) {
        const int tier = wgi.n_groups == 0 ? (int)TIER_GENERIC : tier_of(wgi.n_groups);
        if (threadIdx.x == 0) {
            wg_tier[blockIdx.x] = (uint8_t)tier;
            if (tier != 0) atomicOr(&fb.ctr->need_tiers, 1u << tier);
        }
        if (tier != 0) return;                   // uniform
    }
    // Exact pruning: a cell with V(i,j) < 4j - slack, slack = 2*Lq - minScore, can never
    // feed a row maximum that takes part in a split of score >= minScore (each further row adds at most
    // 4), and no live cell's value comes from a dead cell.  Once a whole tile row and everything that
    // can still enter from the left are dead, the rest of the tile is dead: the sweep stops there and
    // stores "V = 0" for the remaining rows (a lower bound, which is all dead cells need to be).
    int lq_lane = 0, slack = 0, tiles_lane = 0, tail_cols_lane = 0;
    if (in_batch) {
        const dsa_fusion fu = fusions[f];
        lq_lane = pr.read_len;
        slack = 2 * lq_lane - max(min_score_tab[lq_lane], pair_bound(pr));
        tiles_lane = max(cdiv_dev(fu.ref0_len, W), cdiv_dev(fu.ref1_len, W));
        tail_cols_lane = max(fu.ref0_len, fu.ref1_len) - (tiles_lane - 1) * W;       // columns of the longer window in its last tile
    }
    auto wave_max = [](int v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
        return v;
    };
    WaveInfo wi;
    wi.lq_max = wave_max(lq_lane);
    wi.nch_max = wave_max(tiles_lane);
    // rows up to which the boundary entering the next tile may still be alive (tile 0: column 0, V = 0)
    int l_in = wave_max(lq_lane > 0 ? min(slack >> 2, lq_lane) : 0);
    // Row codes are packed here, by the wave that uses them.  A read byte outside {A,C,G,T,N} hands the
    // whole workgroup to the generic kernel, which is launched after this one and re-packs its rows.
    const bool exotic = live && pack_rows_wave<SPLIT ? 2 : 1>(read_bytes, pairs, rowcodes, row_bytes, g, w, lane, wi.lq_max);
    if (threadIdx.x == 0) { s_nch = 0; s_exotic = 0; }
    __syncthreads();
    if (lane == 0 && live) atomicMax(&s_nch, wi.nch_max);
    if (__builtin_amdgcn_ballot_w64(exotic) != 0 && lane == 0) atomicOr(&s_exotic, 1);
    __syncthreads();
    if (s_exotic != 0) {                         // uniform
        if (threadIdx.x == 0) {
            wg_tier[blockIdx.x] = TIER_GENERIC;
            atomicOr(&fb.ctr->need_tiers, 1u << TIER_GENERIC);
        }
        return;
    }
    const int nch_wg = s_nch;

    const uint32_t* tb = T + my_group * (SPLIT ? TGROUP_SPLIT : TGROUP);
    const uint32_t* rows1 = row_bytes + (int64_t)w * (g.lq1 >> 2) * WAVE + lane;
    int stop_prev = 0;                   // stored row groups of the tile to the left
    TileStops stops = {};

    // columns the lane has in the wave's last tile (wave-uniform maximum): 16 or fewer in every lane -> the narrow sweep
    const int tail_cols = wave_max(in_batch && tiles_lane == wi.nch_max ? tail_cols_lane : 0);
    DiagClock clk_all, clk;
    unsigned long long t_bar = 0, t_tab = 0;
    for (int c = 0; c < nch_wg; ++c) {
        clk.lap();
        __syncthreads();                          // previous tile's tables no longer in use
        t_bar += clk.lap();
        build_tables<SPLIT>(T, wgi.n_groups, [&](int gi, int i, uint32_t& q0, uint32_t& q1) {
            const uint32_t code = refcodes[(int64_t)group_fusion(wgi, gi) * g.lrp + c * W + i];
            q0 = code & 0xFFFFu;
            q1 = code >> 16;
        });
        t_tab += clk.lap();
        __syncthreads();
        t_bar += clk.lap();
        if (!live || c >= wi.nch_max) continue;   // wave-uniform

        uint4* cm4 = reinterpret_cast<uint4*>(cmax + ((int64_t)w * g.nch + c) * g.lq1 * WAVE) + lane;
        uint4* bo4 = reinterpret_cast<uint4*>(bnd + ((int64_t)w * g.nch + c) * g.lq1 * WAVE) + lane;
        const uint4* bi4 = reinterpret_cast<const uint4*>(bnd + ((int64_t)w * g.nch + (c - 1)) * g.lq1 * WAVE) + lane;
        int last_bnd = 0, gq;
        if (c + 1 < wi.nch_max)
            gq = sweep_tile_fast<W, SPLIT, false>(tb, rows1, bi4, cm4, bo4, c == 0, wi.lq_max, lq_lane, slack, l_in, stop_prev, last_bnd, g);
        else if (!DIAG_NARROW || tail_cols > 16)
            gq = sweep_tile_fast<W, SPLIT, true>(tb, rows1, bi4, cm4, bo4, c == 0, wi.lq_max, lq_lane, slack, l_in, stop_prev, last_bnd, g);
        else       // (-DDSA_NARROW builds only, see dsa_diag.hpp)
            gq = sweep_tile_fast<16, SPLIT, true>(tb, rows1, bi4, cm4, bo4, c == 0, wi.lq_max, lq_lane, slack, l_in, stop_prev, last_bnd, g);
        // the dead remainder of the tile is not stored: its readers substitute V = 0 past the stop
        if (lane == 0) fb.tstop[(int64_t)w * g.nch + c] = gq;
        stop_prev = gq;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k == c) stops.v[k] = gq;
        l_in = wave_max(last_bnd);
    }
    clk.lap();
    unsigned long long t_rowmax = 0, t_combine = 0, t_replay = 0;
    if (DIAG_TAIL) {
        if (live) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // own stores before own re-reads
            reduce_row_max<WIDE>(cmax, rmax, tmask, fb.tstop, stops, g, w, lane, wi.nch_max, wi.lq_max);
        }
        t_rowmax = clk.lap();
        // The latency-bound finish work of this workgroup runs here, in the shadow of the other resident
        // workgroups' sweeps, instead of in kernels of its own.
        combine_wg<true>(pairs, fusions, cmax, rmax, tmask, min_score_tab, wgi, my_group, true, &fl, fb, g);
        t_combine = clk.lap();
        if (DIAG_REPLAY) replay_fast_wg<SPLIT>(T, &fl, wgi, fb, refcodes, rowcodes, bnd, g);
        t_replay = clk.lap();
    }
    if (lane == 0 && live) {
        DSA_STAT_ADD(g, DS_WAVE_CYCLES, clk_all.lap());
        DSA_STAT_ADD(g, DS_BARRIER, t_bar);
        DSA_STAT_ADD(g, DS_TABLES, t_tab);
        DSA_STAT_ADD(g, DS_TAIL, t_rowmax + t_combine + t_replay);
        DSA_STAT_ADD(g, DS_ROWMAX, t_rowmax);
        DSA_STAT_ADD(g, DS_COMBINE, t_combine);
        DSA_STAT_ADD(g, DS_REPLAY, t_replay);
    }
    (void)t_bar; (void)t_tab; (void)t_rowmax; (void)t_combine; (void)t_replay;
}

// K3g: replay the tile pairs the table-driven replay left over, from the stored boundaries (generic scoring, any pair
// mix); for every kept row of the pair report, as 64-bit masks, the valid columns whose value equals the row maximum
// (lo field: M1 tile chunk0, hi field: M2 tile chunk1).
// FOUR LANES PER TASK: the kernel is as long as its longest task (a junction on a tile boundary needs the neighbour tile up
// to the read's last rows), and one lane sweeping 64 columns takes 1.7 us per row.  Lane q of a task holds columns
// 16q..16q+15 and runs one row behind lane q-1, whose last column it receives through a cross-lane move: a quarter is a
// 16-column tile whose left boundary comes from its neighbour (quarter 0: from the stored boundary column), so the
// recurrence is that of the fill, and a task of R rows takes R + 3 steps of a quarter of the work.
// A block takes REPLAY_TASKS consecutive tasks and hands them to its lane groups in order of their last row (counting
// sort in LDS): two tasks in three are the second tile of a tie on a short side and need a dozen rows, and a wave steps
// as long as its longest task.
constexpr int REPLAY_BLOCK = 256;
constexpr int RQ = 4;                          // lanes per task
constexpr int RW = W / RQ;                     // columns per lane
constexpr int REPLAY_TASKS = REPLAY_BLOCK / RQ;

// row step of RW columns, generic scoring (row_step of the fill's generic kernel, narrower)
__device__ __forceinline__ void row_step_quarter(uint32_t (&X)[RW], const uint32_t (&r)[RW], uint32_t cj, uint32_t bprev, uint32_t bcur)
{
    cj &= CODE_MASK;
    uint32_t d[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) d[k] = k + 1 < RW ? SIX2 - min3u(cj ^ r[k + 1]) : 0u;
    uint32_t a = (bprev + FOUR2) - min3u(cj ^ r[0]);
    uint32_t up = bcur - TWO2;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const uint32_t a_next = X[i] + d[i];
        X[i] = max3(a, X[i], up);
        up = X[i];
        a = a_next;
    }
}

// hit bits of the quarter's columns: bit k = column k equals the lo target, bit 16 + k = the hi target (equal_columns)
__device__ __forceinline__ uint32_t equal_columns_quarter(const uint32_t (&X)[RW], uint32_t target2)
{
    static_assert(RW == 16, "one accumulator word");
    const uint32_t below = target2 - 0x00010001u;
    uint32_t acc = 0;
#pragma unroll
    for (int k = RW - 1; k >= 0; --k) {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const uint32_t c = below + drift2(k);
        const uint32_t y = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, X[k]), __builtin_bit_cast(us2, c)));
        asm("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(acc) : "v"(y));
    }
    return acc;
}

__global__ __launch_bounds__(REPLAY_BLOCK, 5) void k_replay(const ReplayTask* __restrict__ tasks, uint64_t task_cap,
                                                   const uint2* __restrict__ gtasks, uint64_t gtask_cap,
                                                   const Counters* __restrict__ ctr,
                                                   const PairState* __restrict__ state,
                                                   const KeptRow* __restrict__ kept, uint64_t kept_cap,
                                                   const dsa_pair* __restrict__ pairs,
                                                   const dsa_fusion* __restrict__ fusions,
                                                   const uint32_t* __restrict__ refcodes,
                                                   const uint32_t* __restrict__ rowcodes,
                                                   const uint32_t* __restrict__ bnd, const int32_t* __restrict__ tstop,
                                                   uint64_t* __restrict__ masks, uint64_t mask_cap, Geom g)
{
    __shared__ int s_hist[258];
    __shared__ unsigned short s_order[REPLAY_TASKS];
    __shared__ ReplayTask s_task[REPLAY_TASKS];
    __shared__ uint64_t s_kc[KCACHE * REPLAY_BLOCK];   // the first kept rows of every lane's pair: the cursors advance without a global load
    static_assert(REPLAY_BLOCK == WG_LANES, "kept_row() strides the cache by WG_LANES");
    const unsigned long long n_g = ctr->n_gtasks;
    if (ctr->n_tasks > task_cap || ctr->n_masks > mask_cap || ctr->n_kept > kept_cap || n_g > gtask_cap) return;
    if (ctr->need_tiers & ~g.tiers_launched) return;      // some workgroups were not swept: the host runs the slice again with every fill kernel
    const int q = threadIdx.x & (RQ - 1);             // quarter of the tile
    const int slot = threadIdx.x / RQ;                // task slot of the block
    for (unsigned long long base = (unsigned long long)blockIdx.x * REPLAY_TASKS; base < n_g;
         base += (unsigned long long)gridDim.x * REPLAY_TASKS) {                 // uniform per block
        __syncthreads();                                  // previous round's LDS no longer in use
        for (int e = threadIdx.x; e < 258; e += REPLAY_BLOCK) s_hist[e] = 0;
        __syncthreads();
        int my_key = 256, my_rank = 0;
        if (threadIdx.x < REPLAY_TASKS) {
            if (base + threadIdx.x < n_g) {
                const ReplayTask mine = tasks[gtasks[base + threadIdx.x].x & ~GTASK_OWNER];
                s_task[threadIdx.x] = mine;
                my_key = min((int)(mine.last_row & TASK_ROW), 255);
            }
            my_rank = atomicAdd(&s_hist[my_key], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int e = 0; e < 257; ++e) {
                const int n = s_hist[e];
                s_hist[e] = run;
                run += n;
            }
        }
        __syncthreads();
        if (threadIdx.x < REPLAY_TASKS) s_order[s_hist[my_key] + my_rank] = (unsigned short)threadIdx.x;
        __syncthreads();
        const int src = s_order[slot];
        const bool valid = base + src < n_g;              // sorted last; a lane group without a task steps along idle
        if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;
        ReplayTask rt = s_task[valid ? src : s_order[0]];
        if (!valid) { rt.last_row = 0; rt.chunk0 = rt.chunk1 = NO_CHUNK; }
        DiagClock clk;
        const int64_t p = rt.pair;
        const int64_t w = p >> 6;
        const int lane = (int)(p & 63);
        const bool has0 = rt.chunk0 != NO_CHUNK, has1 = rt.chunk1 != NO_CHUNK;
        const int c0 = has0 ? rt.chunk0 : 0, c1 = has1 ? rt.chunk1 : 0;
        const uint32_t* rc = refcodes + (int64_t)rt.fusion * g.lrp;      // the loads below need nothing but the task
        const dsa_fusion fu = fusions[rt.fusion];
        PairState st = state[p];
        if (!valid) st.n_kept = 0;
        const int lq = pairs[p].read_len;
        const uint32_t* rows = rowcodes + w * g.lq1 * WAVE;
        const uint32_t* bi0 = bnd + (w * g.nch + (c0 - 1)) * g.lq1 * WAVE;
        const uint32_t* bi1 = bnd + (w * g.nch + (c1 - 1)) * g.lq1 * WAVE;
        const KeptRow* kr = kept + st.kept_begin;

        uint32_t r[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const uint32_t lo = has0 ? (rc[c0 * W + q * RW + i] & 0xFFFFu) : REF_PAD16;
            const uint32_t hi = has1 ? (rc[c1 * W + q * RW + i] & 0xFFFF0000u) : (REF_PAD16 << 16);
            r[i] = lo | hi;
        }
        uint32_t X[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) X[i] = BIAS2 + drift2(i);
        const int R = rt.last_row & TASK_ROW;
        const int nv0 = has0 ? min(W, fu.ref0_len - c0 * W) : 0;
        const int nv1 = has1 ? min(W, fu.ref1_len - c1 * W) : 0;
        // kept rows ascend in a: M1 (row a) meets them in order k=0.., M2 (row lq-a) in reverse
        {
            const uint2* kr2 = reinterpret_cast<const uint2*>(kr);
            uint2 first[KCACHE];
#pragma unroll
            for (int k = 0; k < KCACHE; ++k) first[k] = kr2[k < (int)st.n_kept ? k : (st.n_kept ? (int)st.n_kept - 1 : 0)];
#pragma unroll
            for (int k = 0; k < KCACHE; ++k) s_kc[k * REPLAY_BLOCK + threadIdx.x] = ((uint64_t)first[k].y << 32) | first[k].x;
        }
        const uint64_t* kc = s_kc + threadIdx.x;
        HitCursor hc = cursor_init(kr, kc, st.n_kept, lq, has0, has1, nv0, nv1);
        const uint32_t my_valid0 = (uint32_t)(hc.valid0 >> (q * RW)) & 0xFFFFu, my_valid1 = (uint32_t)(hc.valid1 >> (q * RW)) & 0xFFFFu;
        const int stop0 = c0 > 0 ? tstop[w * g.nch + (c0 - 1)] : 0;   // stored row groups of the tiles to the left
        const int stop1 = c1 > 0 ? tstop[w * g.nch + (c1 - 1)] : 0;
        // the planes hold four rows per 16-byte word: row j of this pair is word (j / 4) * 64 + lane, element j % 4
        auto plane_at = [&](int j) -> int64_t { return ((int64_t)(j >> 2) * WAVE + lane) * 4 + (j & 3); };
        auto row_code = [&](int j) -> uint32_t { return rows[plane_at(j < 0 ? 0 : j > R ? R : j)]; };
        auto boundary = [&](int j) -> uint32_t {          // quarter 0 only: the tiles to the left (V = 0 past their stop)
            const int jj = j < 0 ? 0 : j > R ? R : j;
            const uint32_t x0 = (jj >> 2) < stop0 ? bi0[plane_at(jj)] : BIAS2;
            const uint32_t x1 = (jj >> 2) < stop1 ? bi1[plane_at(jj)] : BIAS2;
            return (x0 & 0xFFFFu) | (x1 & 0xFFFF0000u);
        };
        int Rw = R;                                        // the wave steps as long as its longest task
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) Rw = max(Rw, __shfl_xor(Rw, d, 64));
        // lane q is at row t - q in step t; operands are fetched two steps ahead
        uint32_t rc_n = row_code(1 - q), rc_nn = row_code(2 - q);
        uint32_t b_n = q == 0 ? boundary(1) : BIAS2, b_nn = q == 0 ? boundary(2) : BIAS2;
        uint32_t bprev = BIAS2;                            // V(left neighbour column, row 0) = 0
        uint32_t xlast = BIAS2;                            // this lane's last column after its latest row, drift removed (row 0: V = 0)
        const unsigned long long t_setup = clk.lap();
        for (int t = 1; t <= Rw + RQ - 1; ++t) {           // uniform
            const int j = t - q;
            const uint32_t cj = rc_n;
            const uint32_t from_left = __shfl_up(xlast, 1, RQ);     // lane q-1 finished row j in the step before
            const uint32_t bcur = q == 0 ? b_n : from_left;
            rc_n = rc_nn;
            b_n = b_nn;
            rc_nn = row_code(j + 2);
            if (q == 0) b_nn = boundary(j + 2);
            if (j >= 1 && j <= R) {
                row_step_quarter(X, r, cj, bprev, bcur);
                const bool hit0 = hc.row0 == j, hit1 = hc.row1 == j;
                if (hit0 || hit1) {
                    const uint32_t bits = equal_columns_quarter(X, (hit0 ? hc.t0 : NO_TARGET16) | ((hit1 ? hc.t1 : NO_TARGET16) << 16));
                    if (hit0) {
                        reinterpret_cast<unsigned short*>(masks + ((uint64_t)rt.mask_begin + hc.k0) * 2)[q] = (unsigned short)(bits & my_valid0);
                        ++hc.k0;
                        cursor_next0(hc, kr, kc, st.n_kept, has0);
                    }
                    if (hit1) {
                        reinterpret_cast<unsigned short*>(masks + ((uint64_t)rt.mask_begin + hc.k1) * 2 + 1)[q] = (unsigned short)((bits >> 16) & my_valid1);
                        --hc.k1;
                        cursor_next1(hc, kr, kc, lq, has1);
                    }
                }
                bprev = bcur;
                xlast = X[RW - 1] - drift2(RW - 1);
            }
        }
        // sides that were not replayed report no columns
        if (valid && q == 0 && !has0)
            for (int k = 0; k < st.n_kept; ++k) masks[((uint64_t)rt.mask_begin + k) * 2] = 0;
        if (valid && q == 0 && !has1)
            for (int k = 0; k < st.n_kept; ++k) masks[((uint64_t)rt.mask_begin + k) * 2 + 1] = 0;
        if ((threadIdx.x & 63) == 0) {
            DSA_STAT_ADD(g, DS_GREPLAY_SETUP, t_setup);
            DSA_STAT_ADD(g, DS_GREPLAY_SWEEP, clk.lap());
            DSA_STAT_ADD(g, DS_GREPLAY_WAVES, 1);
            DSA_STAT_ADD(g, DS_GREPLAY_STEPS, Rw + RQ - 1);
        }
        (void)t_setup;
    }
}

// the device cursors of a slice and the slot behind its last record count, zeroed before the fill
__global__ void k_reset_finish(Counters* ctr, int64_t* rec_count_end)
{
    if (threadIdx.x == 0) {
        ctr->n_kept = ctr->n_tasks = ctr->n_masks = ctr->n_gtasks = 0;
        ctr->need_tiers = 0;
        *rec_count_end = 0;
    }
}

// the cursors and the record total of a slice, stored into pinned host memory at the end of its phase 1
struct PlanGlobals;
__global__ void k_publish(const Counters* __restrict__ ctr, const int64_t* __restrict__ n_rec, const unsigned long long* __restrict__ plan,
                          Counters* host_ctr, int64_t* host_n_rec, unsigned long long* host_plan)
{
    if (threadIdx.x == 0) {
        *host_ctr = *ctr;
        *host_n_rec = *n_rec;
        host_plan[0] = plan[0];          // PlanGlobals of the slice: cells, identity flag
        host_plan[1] = plan[1];
        __threadfence_system();
    }
}

// K4: emit.  For every kept split a (ascending) the cross product columns1 x columns2 in ascending
// order (tools/SplitReadAligner.cpp:233-269), then the refSplit de-duplication of
// tools/SplitAlignment.cpp:381-391 (first occurrence wins).  WRITE=false counts.
__device__ __forceinline__ bool col_in(const ReplayTask* tasks, const uint64_t* masks, uint32_t tb, uint32_t te,
                                       int k, int h, int col /*1-based matrix column*/)
{
    const int c = (col - 1) / W, bit = (col - 1) % W;
    for (uint32_t q = tb; q < te; ++q)
        if ((h ? tasks[q].chunk1 : tasks[q].chunk0) == c)
            return (masks[((uint64_t)tasks[q].mask_begin + k) * 2 + h] >> bit) & 1ull;
    return false;
}

constexpr int EMIT_BLOCK = 128;
// (task, kept row) mask pairs a lane stages in LDS, all loads in flight at once; a pair with more walks global memory, a
// chain of dependent loads per column pair.  Few pairs have many, but the slowest lane sets the time of these kernels.
constexpr int EMIT_SLOTS_COUNTED = 1, EMIT_SLOTS_LISTED = 16;
constexpr int EMIT_REG_ROWS = 4;   // kept rows of a one-tile-pair pair that are handled in registers
template <int SLOTS>
struct EmitLds {
    uint4 mask[SLOTS][EMIT_BLOCK];   // [task * n_kept + kept row]: {M1 mask, M2 mask}
    uint2 kept[SLOTS][EMIT_BLOCK];   // the kept rows
};

// The records of one pair (WRITE) or their number.
template <bool WRITE, int SLOTS>
__device__ __forceinline__ int64_t emit_pair(EmitLds<SLOTS>* lds, int64_t p, int64_t o, const PairState& st, const dsa_pair* __restrict__ pairs,
                                             const dsa_fusion* __restrict__ fusions, const KeptRow* __restrict__ kept,
                                             const ReplayTask* __restrict__ tasks, const uint64_t* __restrict__ masks,
                                             const int64_t* __restrict__ rec_offset, dsa_record* __restrict__ out, uint64_t out_cap,
                                             int64_t pair_base)
{
    const int tid = threadIdx.x;
    int64_t n = 0;
    int64_t wr = 0;
    dsa_record rec = {};
    int ref1_len = 0;
    if (WRITE) {
        const dsa_pair pr = pairs[p];                // issued together with the offsets: the chain is pair -> fusion
        wr = rec_offset[o];
        const int64_t wr_end = rec_offset[o + 1];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        if ((uint64_t)wr_end > out_cap) return 0;    // host grows the buffer and reruns emit
        rec.fusion_id = fu.fusion_id;
        rec.frag = pr.frag;
        rec.read_end = pr.read_end;
        rec.revcomp = pr.revcomp;
        rec.read_second = pr.read_len;               // minus a below
        rec.pair_idx = (int32_t)(pair_base + o);
        ref1_len = fu.ref1_len;
    }
    // 40-byte records at a multiple of 40 bytes from a 256-byte aligned base: five 8-byte stores
    auto put = [&](int i1, int i2, const KeptRow& kr) {
        if (WRITE) {
            dsa_record r = rec;
            r.ref_first = i1;
            r.ref_second = ref1_len - i2 - 1;
            r.read_first = kr.a;
            r.read_second = rec.read_second - kr.a;
            r.score = kr.m1 < kr.m2 ? kr.m1 : kr.m2;
            uint2* dst = reinterpret_cast<uint2*>(out + wr);
            const uint2* src = reinterpret_cast<const uint2*>(&r);
#pragma unroll
            for (int q = 0; q < 5; ++q) dst[q] = src[q];
        }
        ++wr;
        ++n;
    };
    const int K = st.n_kept, T = (int)st.n_tasks;
    const bool tiles = (st.flags & STATE_TILES) != 0;
    if (tiles && T == 1 && K <= EMIT_REG_ROWS) {
        // one tile pair and a few kept rows - nearly every pair: masks and rows in registers, two loads deep
        const uint4* masks4 = reinterpret_cast<const uint4*>(masks) + st.mask_begin;
        const uint2* kept2 = reinterpret_cast<const uint2*>(kept) + st.kept_begin;
        uint4 m[EMIT_REG_ROWS];
        uint2 kr2[EMIT_REG_ROWS];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            m[e] = masks4[e < K ? e : K - 1];
            kr2[e] = kept2[e < K ? e : K - 1];
        }
#pragma unroll
        for (int e = 2; e < EMIT_REG_ROWS; ++e) {
            m[e] = m[1];
            kr2[e] = kr2[1];
        }
        if (K > 2) {
#pragma unroll
            for (int e = 2; e < EMIT_REG_ROWS; ++e) {
                m[e] = masks4[e < K ? e : K - 1];
                kr2[e] = kept2[e < K ? e : K - 1];
            }
        }
        const bool both = st.tiles0 != 0 && st.tiles1 != 0;
        const int base1 = __builtin_ctz(st.tiles0 | 0x10000u) * W + 1, base2 = __builtin_ctz(st.tiles1 | 0x10000u) * W + 1;
        auto lo64 = [](const uint4& v) -> uint64_t { return ((uint64_t)v.y << 32) | v.x; };
        auto hi64 = [](const uint4& v) -> uint64_t { return ((uint64_t)v.w << 32) | v.z; };
#pragma unroll
        for (int k = 0; k < EMIT_REG_ROWS; ++k) {
            if (both && k < K) {
                const KeptRow kr = __builtin_bit_cast(KeptRow, kr2[k]);
                for (uint64_t r1 = lo64(m[k]); r1; r1 &= r1 - 1) {
                    const int b1 = __builtin_ctzll(r1);
                    for (uint64_t r2 = hi64(m[k]); r2; r2 &= r2 - 1) {
                        const int b2 = __builtin_ctzll(r2);
                        bool dup = false;     // same refSplit <=> same (i1,i2) at an earlier kept a
#pragma unroll
                        for (int k2 = 0; k2 < k; ++k2) dup = dup || (((lo64(m[k2]) >> b1) & 1ull) && ((hi64(m[k2]) >> b2) & 1ull));
                        if (!dup) put(base1 + b1, base2 + b2, kr);
                    }
                }
            }
        }
    } else if (tiles) {
        // Every tile is in one task only, so column i1 of a tile can only recur in the same task's mask of an earlier
        // kept row: the de-dup (same refSplit <=> same (i1,i2) at an earlier kept a) is two bit tests per earlier
        // row.  The masks of the pair are one block of memory, [task][kept row]; tiles are walked in ascending order.
        const uint4* masks4 = reinterpret_cast<const uint4*>(masks) + st.mask_begin;
        const uint2* kept2 = reinterpret_cast<const uint2*>(kept) + st.kept_begin;
        const int slots = K * T;
        const bool staged = slots <= SLOTS;
        if (staged) {
            // in growing groups, each with all its loads in flight: most pairs have one or two
#pragma unroll
            for (int e0 = 0, e1 = 2; e0 < SLOTS; e0 = e1, e1 *= 2) {
                if (e0 == 0 || slots > e0) {
#pragma unroll
                    for (int e = e0; e < e1 && e < SLOTS; ++e) {
                        lds->mask[e][tid] = masks4[e < slots ? e : slots - 1];
                        lds->kept[e][tid] = kept2[e < K ? e : K - 1];
                    }
                }
            }
        }
        // pairs with more masks than a lane stages (and that no wave took) walk global memory
        auto mask_of = [&](int k, int t, int h) -> uint64_t {
            const uint4 v = staged ? lds->mask[t * K + k][tid] : masks4[t * K + k];
            return h ? ((uint64_t)v.w << 32) | v.z : ((uint64_t)v.y << 32) | v.x;
        };
        for (int k = 0; k < K; ++k) {
            const KeptRow kr = __builtin_bit_cast(KeptRow, staged ? lds->kept[k][tid] : kept2[k]);
            for (uint32_t w0 = st.tiles0; w0; w0 &= w0 - 1) {
                const int c0 = __builtin_ctz(w0), t1 = task_of_tile(st.tiles0, st.first0, c0);
                for (uint64_t r1 = mask_of(k, t1, 0); r1; r1 &= r1 - 1) {
                    const int b1 = __builtin_ctzll(r1);
                    for (uint32_t w1 = st.tiles1; w1; w1 &= w1 - 1) {
                        const int c1 = __builtin_ctz(w1), t2 = task_of_tile(st.tiles1, st.first1, c1);
                        for (uint64_t r2 = mask_of(k, t2, 1); r2; r2 &= r2 - 1) {
                            const int b2 = __builtin_ctzll(r2);
                            bool dup = false;
                            for (int k2 = 0; k2 < k && !dup; ++k2) dup = ((mask_of(k2, t1, 0) >> b1) & 1ull) && ((mask_of(k2, t2, 1) >> b2) & 1ull);
                            if (!dup) put(c0 * W + b1 + 1, c1 * W + b2 + 1, kr);
                        }
                    }
                }
            }
        }
    } else {
        // windows of more than 16 tiles: the task list, whose tiles ascend (task 0 has the lowest of either side)
        const uint32_t tb = st.task_begin, te = tb + st.n_tasks;
        for (int k = 0; k < K; ++k) {
            const KeptRow kr = kept[st.kept_begin + k];
            for (uint32_t q1 = tb; q1 < te; ++q1) {        // tasks hold M1 tiles in ascending order
                if (tasks[q1].chunk0 == NO_CHUNK) continue;
                uint64_t m1 = masks[((uint64_t)tasks[q1].mask_begin + k) * 2];
                while (m1) {
                    const int i1 = tasks[q1].chunk0 * W + __builtin_ctzll(m1) + 1;
                    m1 &= m1 - 1;
                    for (uint32_t q2 = tb; q2 < te; ++q2) {
                        if (tasks[q2].chunk1 == NO_CHUNK) continue;
                        uint64_t m2 = masks[((uint64_t)tasks[q2].mask_begin + k) * 2 + 1];
                        while (m2) {
                            const int i2 = tasks[q2].chunk1 * W + __builtin_ctzll(m2) + 1;
                            m2 &= m2 - 1;
                            bool dup = false;     // same refSplit <=> same (i1,i2) at an earlier kept a
                            for (int k2 = 0; k2 < k && !dup; ++k2)
                                dup = col_in(tasks, masks, tb, te, k2, 0, i1) && col_in(tasks, masks, tb, te, k2, 1, i2);
                            if (!dup) put(i1, i2, kr);
                        }
                    }
                }
            }
        }
    }
    return n;
}

// A pair with more (task, kept row) mask pairs than a lane stages: the whole wave takes it.  Its lanes copy the masks
// and kept rows into the wave's part of the staging area (SLOTS entries per lane, so 64 * SLOTS in all), then lane k
// counts kept row k (the de-dup only reads earlier rows' masks), a prefix sum places the rows, and every lane writes
// its row's records: the order is that of emit_pair.  Called by all 64 lanes with the same arguments.
template <bool WRITE, int SLOTS>
__device__ __forceinline__ int64_t emit_pair_wave(EmitLds<SLOTS>* lds, int64_t p, int64_t o, uint32_t mask_begin, uint32_t kept_begin, int K, int T,
                                                  uint32_t tiles0, uint32_t tiles1, int first0, int first1, const dsa_pair* __restrict__ pairs,
                                                  const dsa_fusion* __restrict__ fusions, const KeptRow* __restrict__ kept,
                                                  const uint64_t* __restrict__ masks, const int64_t* __restrict__ rec_offset,
                                                  dsa_record* __restrict__ out, uint64_t out_cap, int64_t pair_base)
{
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    auto slot_m = [&](int e) -> uint4& { return lds->mask[e >> 6][wbase + (e & 63)]; };
    auto slot_k = [&](int e) -> uint2& { return lds->kept[e >> 6][wbase + (e & 63)]; };
    const uint4* masks4 = reinterpret_cast<const uint4*>(masks) + mask_begin;
    const uint2* kept2 = reinterpret_cast<const uint2*>(kept) + kept_begin;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < K * T; e += 64) slot_m(e) = masks4[e];
    for (int e = lane; e < K; e += 64) slot_k(e) = kept2[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto mask_of = [&](int k, int t, int h) -> uint64_t {
        const uint4 v = slot_m(t * K + k);
        return h ? ((uint64_t)v.w << 32) | v.z : ((uint64_t)v.y << 32) | v.x;
    };
    int64_t wr = 0;
    dsa_record rec = {};
    int ref1_len = 0;
    if (WRITE) {
        wr = rec_offset[o];
        if ((uint64_t)rec_offset[o + 1] > out_cap) return 0;   // uniform: host grows the buffer and reruns emit
        const dsa_pair pr = pairs[p];
        const dsa_fusion fu = fusions[pr.fusion_idx];
        rec.fusion_id = fu.fusion_id;
        rec.frag = pr.frag;
        rec.read_end = pr.read_end;
        rec.revcomp = pr.revcomp;
        rec.read_second = pr.read_len;
        rec.pair_idx = (int32_t)(pair_base + o);
        ref1_len = fu.ref1_len;
    }
    // the records of kept row k, handed to fn(i1, i2) in output order
    auto row = [&](int k, auto&& fn) {
        for (uint32_t w0 = tiles0; w0; w0 &= w0 - 1) {
            const int c0 = __builtin_ctz(w0), t1 = task_of_tile(tiles0, first0, c0);
            for (uint64_t r1 = mask_of(k, t1, 0); r1; r1 &= r1 - 1) {
                const int b1 = __builtin_ctzll(r1);
                for (uint32_t w1 = tiles1; w1; w1 &= w1 - 1) {
                    const int c1 = __builtin_ctz(w1), t2 = task_of_tile(tiles1, first1, c1);
                    for (uint64_t r2 = mask_of(k, t2, 1); r2; r2 &= r2 - 1) {
                        const int b2 = __builtin_ctzll(r2);
                        bool dup = false;
                        for (int k2 = 0; k2 < k && !dup; ++k2) dup = ((mask_of(k2, t1, 0) >> b1) & 1ull) && ((mask_of(k2, t2, 1) >> b2) & 1ull);
                        if (!dup) fn(c0 * W + b1 + 1, c1 * W + b2 + 1);
                    }
                }
            }
        }
    };
    int64_t total = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {           // uniform
        const int k = k0 + lane;
        int n_row = 0;
        if (k < K) row(k, [&](int, int) { ++n_row; });
        int incl = n_row;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
        if (WRITE && k < K) {
            const KeptRow kr = __builtin_bit_cast(KeptRow, slot_k(k));
            int64_t at = wr + total + (incl - n_row);
            row(k, [&](int i1, int i2) {
                dsa_record r = rec;
                r.ref_first = i1;
                r.ref_second = ref1_len - i2 - 1;
                r.read_first = kr.a;
                r.read_second = rec.read_second - kr.a;
                r.score = kr.m1 < kr.m2 ? kr.m1 : kr.m2;
                uint2* dst = reinterpret_cast<uint2*>(out + at);
                const uint2* src = reinterpret_cast<const uint2*>(&r);
#pragma unroll
                for (int q = 0; q < 5; ++q) dst[q] = src[q];
                ++at;
            });
        }
        total += __shfl(incl, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    return total;
}

// does the wave take the pair (emit_pair_wave) rather than its lane?
template <int SLOTS>
__device__ __forceinline__ bool emit_is_heavy(const PairState& st)
{
    const int slots = (int)st.n_kept * (int)st.n_tasks;
    if (st.n_tasks == 1 && st.n_kept <= EMIT_REG_ROWS) return false;
    return (st.flags & STATE_TILES) != 0 && slots > SLOTS && slots <= 64 * SLOTS;
}

// the heavy pairs of a wave, one after the other; fn_done(lane's own pair result) is told the count of the lane's pair
template <bool WRITE, int SLOTS>
__device__ __forceinline__ int64_t emit_heavy_of_wave(EmitLds<SLOTS>* lds, bool heavy, int64_t p, int64_t o, const PairState& st,
                                                      const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions,
                                                      const KeptRow* __restrict__ kept, const uint64_t* __restrict__ masks,
                                                      const int64_t* __restrict__ rec_offset, dsa_record* __restrict__ out, uint64_t out_cap,
                                                      int64_t pair_base)
{
    int64_t mine = 0;
    uint64_t todo = __builtin_amdgcn_ballot_w64(heavy);
    while (todo) {                                   // uniform
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t sp = __shfl((int)p, src, 64), so = __shfl((int)o, src, 64);       // pair indices are below 2^31
        const uint32_t mb = (uint32_t)__shfl((int)st.mask_begin, src, 64), kb = (uint32_t)__shfl((int)st.kept_begin, src, 64);
        const int packed = __shfl((int)st.n_kept | ((int)st.n_tasks << 16), src, 64);
        const int tl = __shfl((int)st.tiles0 | ((int)st.tiles1 << 16), src, 64);
        const int firsts = __shfl((int)st.first0 | ((int)st.first1 << 8), src, 64);
        const int64_t n = emit_pair_wave<WRITE, SLOTS>(lds, sp, so, mb, kb, packed & 0xFFFF, (packed >> 16) & 0xFF, (uint32_t)tl & 0xFFFFu,
                                                        (uint32_t)tl >> 16, firsts & 0xFF, (firsts >> 8) & 0xFF, pairs, fusions, kept, masks, rec_offset, out, out_cap, pair_base);
        if ((int)(threadIdx.x & 63) == src) mine = n;
    }
    return mine;
}

// K4a: the records of the pairs the fill kernel's tail has counted (one tile pair, replayed there) - one lane per pair.
__global__ __launch_bounds__(EMIT_BLOCK) void k_emit_counted(const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions,
                       const PairState* __restrict__ state, const KeptRow* __restrict__ kept,
                       const ReplayTask* __restrict__ tasks, const uint64_t* __restrict__ masks,
                       const int64_t* __restrict__ rec_offset, dsa_record* __restrict__ out, uint64_t out_cap, int64_t pair_base,
                       const Counters* __restrict__ ctr, uint64_t kept_cap, uint64_t task_cap, uint64_t mask_cap, uint64_t gtask_cap, Geom g)
{
    __shared__ EmitLds<EMIT_SLOTS_COUNTED> lds;
    if (ctr->n_tasks > task_cap || ctr->n_masks > mask_cap || ctr->n_kept > kept_cap || ctr->n_gtasks > gtask_cap) return;   // counts are void: the host reruns the slice
    if (ctr->need_tiers & ~g.tiers_launched) return;      // pairs of workgroups that were not swept have no state yet
    const int64_t p = (int64_t)blockIdx.x * EMIT_BLOCK + threadIdx.x;
    PairState st = {};
    int64_t o = 0;                                   // records are counted and written in the caller's pair order
    if (p < g.n_pairs) {
        st = state[p];
        o = g.orig ? g.orig[p] : p;
    }
    const bool mine = p < g.n_pairs && (st.flags & STATE_COUNTED) && st.n_kept != 0;
    // Pairs with more kept rows than the register path takes come in runs (a repeat in one fusion's window gives all its
    // reads the same ties): handing them to the whole wave one after the other (emit_pair_wave) would serialise a run
    // that sits in one wave, so here they stay with their lanes and walk global memory.
    if (mine) emit_pair<true, EMIT_SLOTS_COUNTED>(&lds, p, o, st, pairs, fusions, kept, tasks, masks, rec_offset, out, out_cap, pair_base);
}

// K4b: the other pairs - every one of them has a task in the generic replay's list, and the lane that finds the pair's
// first such task there counts (WRITE = false) or writes the pair's records.  These lanes all work through several
// tiles, which one lane in ten of a per-pair launch would do while the rest of its wave waits.
template <bool WRITE>
__global__ __launch_bounds__(EMIT_BLOCK) void k_emit_listed(const uint2* __restrict__ gtasks, uint64_t gtask_cap, const Counters* __restrict__ ctr,
                       const dsa_pair* __restrict__ pairs, const dsa_fusion* __restrict__ fusions,
                       const PairState* __restrict__ state, const KeptRow* __restrict__ kept,
                       const ReplayTask* __restrict__ tasks, uint64_t task_cap, const uint64_t* __restrict__ masks, uint64_t mask_cap,
                       uint64_t kept_cap, int64_t* __restrict__ rec_count, const int64_t* __restrict__ rec_offset,
                       dsa_record* __restrict__ out, uint64_t out_cap, int64_t pair_base, Geom g)
{
    __shared__ EmitLds<EMIT_SLOTS_LISTED> lds;
    const unsigned long long n_g = ctr->n_gtasks;
    if (ctr->n_tasks > task_cap || ctr->n_masks > mask_cap || ctr->n_kept > kept_cap || n_g > gtask_cap) return;   // the host reruns the slice
    if (ctr->need_tiers & ~g.tiers_launched) return;
    // Consecutive list entries go to different blocks (entry = thread * blocks + block within a round of the grid): the
    // pairs with many kept rows come in runs, and a wave takes its heavy pairs one after the other.
    const unsigned long long stride = (unsigned long long)gridDim.x * EMIT_BLOCK;
    for (unsigned long long e0 = 0; e0 < n_g; e0 += stride) {      // uniform
        const unsigned long long e = e0 + (unsigned long long)threadIdx.x * gridDim.x + blockIdx.x;
        uint2 entry = make_uint2(0u, 0u);
        if (e < n_g) entry = gtasks[e];
        const bool mine = (entry.x & GTASK_OWNER) != 0;
        const int64_t p = entry.y;
        PairState st = {};
        if (mine) st = state[p];
        const int64_t o = mine ? (g.orig ? g.orig[p] : p) : 0;
        const bool heavy = mine && emit_is_heavy<EMIT_SLOTS_LISTED>(st);
        DiagClock clk;
        int64_t n = 0;
        if (mine && !heavy) n = emit_pair<WRITE, EMIT_SLOTS_LISTED>(&lds, p, o, st, pairs, fusions, kept, tasks, masks, rec_offset, out, out_cap, pair_base);
        const int64_t nh = emit_heavy_of_wave<WRITE, EMIT_SLOTS_LISTED>(&lds, heavy, p, o, st, pairs, fusions, kept, masks, rec_offset, out, out_cap, pair_base);
        if (!WRITE && mine) rec_count[o] = heavy ? nh : n;
        if (!WRITE && mine)        // slowest lane: cycles << 24 | n_kept << 8 | n_tasks
            DSA_STAT_MAX(g, DS_SLOWEST_LISTED, (clk.lap() << 24) | ((unsigned long long)st.n_kept << 8) | st.n_tasks);
    }
}

}  // namespace dsa

#include "dsa_plan.hpp"
