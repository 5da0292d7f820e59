// dsa_plan.hpp — sweep planning of a batch, entirely on the device (no host round trip, no descriptor built by the host).
//
// The reference aligns its candidates one after the other in the order it meets them (tools/SplitAlignment.cpp:266-303);
// the order is free.  The plan is a SPEED heuristic and a pruning aid only — any order and any bound that is a true lower
// bound give the same records — made of
//   * the order of the fusions in the sweep: fusions with many reads first (table tiers of the fill), inside a size class
//     the expensive ones first (longest-processing-time order), then by the pair of tiles in which their alignments end,
//     so that a wave that straddles two fusions is alive in the same tiles;
//   * the order of the pairs inside a fusion: by the diagonal of the read in window 0, i.e. by the estimated read split,
//     alternate fusions in opposite directions;
//   * a per-pair LOWER BOUND T' of the final score, which tightens the exact pruning of the fill kernels.
//
// Kernels, in stream order (one slice of the caller's pair order at a time):
//   k_plan_runs      one thread per pair: where the run of every fusion begins and ends, how many runs it has
//   k_plan_fusion    one workgroup per fusion: the windows 2-bit packed and their 11-mers hashed in LDS; per read the
//                    diagonals d1 (first 11-mers in window 0) and d2 (last 11-mers in window 1), the vote of the fusion on
//                    d2 - d1, the bound T' from the two ungapped paths, the tiles in which the alignments end, the rank of
//                    the read inside the fusion; per fusion the sort key (size class, alive tiles, tile pair)
//   (hipcub radix sort of the fusion keys: dsa_api.hip)
//   k_plan_place_a/b the start of every fusion in the sweep order (scan of the counts in sorted order) and its direction
//   k_plan_permute   the pairs into sweep order, T' in the two padding bytes of the device copy of dsa_pair
// A fusion whose pairs are not ONE run inside the slice switches the slice to the caller's order (the identity flag); the
// fill kernels derive everything else (rows and tiles per wave, fusions per workgroup) from the pairs they are given.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/defuse_dsa.h"

// (included at the end of dsa_kernels.hpp: WAVE, WG_LANES, GSPLIT, GSPLIT2 come from there)
namespace dsa {

constexpr int PLAN_THREADS = 128;
constexpr int PLAN_K = 11;             // seed length
constexpr int PLAN_MAXWIN = 1000;      // windows of this many bases and more: no rank / bound for the fusion (10-bit positions)
constexpr int PLAN_MAX = 2048;         // pairs per fusion beyond which the caller's order stays
constexpr int PLAN_LQ = 160;           // reads longer than this get no bound (their match masks live in 5 + 5 registers)
constexpr int PLAN_CHUNKS = PLAN_LQ / 32;
constexpr int PLAN_PAD = 160;          // invalid bases on either side of a packed window: diagonals may leave the window
constexpr int PLAN_TILES = 16;         // tile votes: tiles 0..14, 15 = that and beyond
constexpr int PLAN_VOTE_BINS = 2048;   // histogram of d2 - d1 + 1024

// sort key of a fusion, ascending = sweep order: size class, cost rank (expensive first), the tiles in which the alignments of
// M1 / M2 end (255 = no vote)
constexpr int PLAN_KEY_BITS = 14;
constexpr uint32_t PLAN_KEY_NONE = (1u << PLAN_KEY_BITS) - 1u;
__host__ __device__ constexpr uint32_t plan_fusion_key(int cls, int cost_rank, int t0, int t1)
{
    return ((uint32_t)cls << 12) | ((uint32_t)cost_rank << 10) | ((uint32_t)(t0 == 255 ? 31 : t0) << 5) | (uint32_t)(t1 == 255 ? 31 : t1);
}

struct PlanRun {
    int32_t first, last, nruns, count;   // pairs of the fusion in the slice (slice-relative); count is filled by k_plan_fusion
};
struct PlanGlobals {
    unsigned long long cells;            // DP cells of the planned pairs (dsa_timing.cells)
    int32_t identity;                    // some fusion has more than one run in the slice: the caller's order is swept
    int32_t pad_;
};
struct PlanParams {
    int32_t n_fusions;
    int32_t slots;                       // hash slots per window (power of two >= 2 x the longest window below PLAN_MAXWIN)
    int32_t wc;                          // dwords of 2-bit codes per packed window (with padding)
    int32_t use_rank, use_bound, use_lpt;
    int32_t tile_cols;
};

__global__ void k_plan_runs(const dsa_pair* __restrict__ pairs, int64_t n, PlanRun* __restrict__ runs)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int f = pairs[p].fusion_idx;
    const int prev = p > 0 ? pairs[p - 1].fusion_idx : -1, next = p + 1 < n ? pairs[p + 1].fusion_idx : -1;
    if (f != prev) {
        atomicAdd(&runs[f].nruns, 1);
        runs[f].first = (int32_t)p;
    }
    if (f != next) runs[f].last = (int32_t)p;
}

// ---- 2-bit packing -------------------------------------------------------------------------------------------------
// code of a base: (byte >> 1) & 3 -> A 0, C 1, T 2, G 3; valid = the byte is one of those four upper-case letters.  Anything
// else (N, lower case, other bytes) is invalid and counts as a mismatch — conservative for a lower bound: the reference
// compares raw bytes, so an invalid base can only score better there.
__device__ __forceinline__ void plan_pack4(uint32_t x, uint32_t& code8, uint32_t& valid4)
{
    const uint32_t c = (x >> 1) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, c);       // byte k = "ACTG"[code k]
    const uint32_t diff = x ^ expect;
    const uint32_t nz = (diff | ((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u;   // 0x80 in every byte that differs
    uint32_t t = c | (c >> 6);
    t &= 0x000F000Fu;
    code8 = (t | (t >> 12)) & 0xFFu;
    uint32_t z = (~nz & 0x80808080u) >> 7;                                    // bit 0 / 8 / 16 / 24: valid
    z |= z >> 7;
    z |= z >> 14;
    valid4 = z & 0xFu;
}

// 32 bases of a read starting at base j0 (bytes src[j0 ..]), only bases < lq valid; reads up to 36 bytes past src + j0
// (the device buffers carry that slack, dsa_api.hip)
struct PlanChunk {
    uint32_t lo, hi, valid;            // codes of bases 0..15 / 16..31 of the chunk, validity bits
};
__device__ __forceinline__ PlanChunk plan_load_chunk(const uint8_t* __restrict__ src, int j0, int lq)
{
    const uint8_t* q = src + j0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(q);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3u);
    uint32_t raw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) raw[k] = w[k];
    PlanChunk ch{0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t x = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh);
        uint32_t c8, v4;
        plan_pack4(x, c8, v4);
        if (k < 4) ch.lo |= c8 << (8 * k);
        else ch.hi |= c8 << (8 * (k - 4));
        ch.valid |= v4 << (4 * k);
    }
    const int left = lq - j0;
    const uint32_t lenmask = left >= 32 ? 0xFFFFFFFFu : left <= 0 ? 0u : ((1u << left) - 1u);
    ch.valid &= lenmask;
    return ch;
}
// the 11-mer at base offset o (0..21) of a chunk; ok = all eleven bases valid
__device__ __forceinline__ uint32_t plan_kmer(const PlanChunk& ch, int o, bool& ok)
{
    const uint64_t codes = ((uint64_t)ch.hi << 32) | ch.lo;
    ok = ((ch.valid >> o) & 0x7FFu) == 0x7FFu;
    return (uint32_t)(codes >> (2 * o)) & 0x3FFFFFu;
}
// even bits of x (one per base) gathered into the low 16 bits
__device__ __forceinline__ uint32_t plan_even_bits(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0xFFFFu;
}

__device__ __forceinline__ void plan_table_insert(uint32_t* table, int slots, uint32_t km, int x)
{
    const uint32_t val = (km << 10) | (uint32_t)x;
    uint32_t h = ((km * 2654435761u) >> 16) & (uint32_t)(slots - 1);
    for (int probe = 0; probe < slots; ++probe, h = (h + 1) & (uint32_t)(slots - 1)) {
        const uint32_t old = atomicCAS(&table[h], 0xFFFFFFFFu, val);
        if (old == 0xFFFFFFFFu) break;
        if ((old >> 10) == km) { atomicMin(&table[h], val); break; }     // same 11-mer: the smallest position stays
    }
}
__device__ __forceinline__ int plan_table_find(const uint32_t* table, int slots, uint32_t km)
{
    uint32_t h = ((km * 2654435761u) >> 16) & (uint32_t)(slots - 1);
    for (int probe = 0; probe < slots; ++probe, h = (h + 1) & (uint32_t)(slots - 1)) {
        const uint32_t e = table[h];
        if (e == 0xFFFFFFFFu) return -1;
        if ((e >> 10) == km) return (int)(e & 1023u);
    }
    return -1;
}

// match bits of the 32 read bases of chunk `ch` (bases j0 .. j0 + 31) against the packed window along diagonal d: read base j
// lies on window position j + d.  codes / valid: the window's packed words, position x at padded index x + PLAN_PAD.
__device__ __forceinline__ uint32_t plan_match32(const PlanChunk& ch, const uint32_t* codes, const uint32_t* valid, int j0, int d)
{
    const int s = j0 + d + PLAN_PAD;                   // >= 0 by the caller's clamps
    const int wi = s >> 4;
    const uint32_t sh = (uint32_t)(2 * (s & 15));
    const uint32_t w0 = codes[wi], w1 = codes[wi + 1], w2 = codes[wi + 2];
    const uint32_t wlo = __builtin_amdgcn_alignbit(w1, w0, sh), whi = __builtin_amdgcn_alignbit(w2, w1, sh);
    const int vi = s >> 5;
    const uint32_t wv = __builtin_amdgcn_alignbit(valid[vi + 1], valid[vi], (uint32_t)(s & 31));
    const uint32_t xl = ch.lo ^ wlo, xh = ch.hi ^ whi;
    const uint32_t mm = plan_even_bits(xl | (xl >> 1)) | (plan_even_bits(xh | (xh >> 1)) << 16);
    return ~mm & ch.valid & wv;
}

// the 32 bases that begin at base `at` of a read whose chunks c[k] hold bases 32k .. 32k + 31 (at < 32 * PLAN_CHUNKS)
__device__ __forceinline__ PlanChunk plan_chunk_at(const PlanChunk (&c)[PLAN_CHUNKS], int at)
{
    const int k = at >> 5;
    const uint32_t sh = (uint32_t)(at & 31);
    PlanChunk a = c[0], b = c[1 < PLAN_CHUNKS ? 1 : 0];
#pragma unroll
    for (int q = 1; q < PLAN_CHUNKS; ++q)
        if (k == q) {
            a = c[q];
            b = q + 1 < PLAN_CHUNKS ? c[q + 1] : PlanChunk{0u, 0u, 0u};
        }
    if (sh == 0) return a;
    const uint64_t ca = ((uint64_t)a.hi << 32) | a.lo, cb = ((uint64_t)b.hi << 32) | b.lo;
    const uint64_t cc = (ca >> (2 * sh)) | (cb << (64 - 2 * sh));
    PlanChunk r;
    r.lo = (uint32_t)cc;
    r.hi = (uint32_t)(cc >> 32);
    r.valid = (a.valid >> sh) | (b.valid << (32 - sh));
    return r;
}

// Shared memory of k_plan_fusion (dynamic): [codes 2 x wc][valid 2 x (wc/2 + 2)][tables 2 x slots][votes slots][keys 2 x PLAN_THREADS]
__host__ __device__ inline size_t plan_lds_bytes(int wc, int slots)
{
    return sizeof(uint32_t) * ((size_t)2 * wc + 2 * (size_t)(wc / 2 + 2) + 3 * (size_t)slots + 2 * PLAN_THREADS);
}

#ifdef DSA_PRUNE_STATS
__device__ unsigned long long g_plan_phase[8];       // wave cycles of k_plan_fusion by phase (diagnostic builds, dsa_diag.hpp)
#define PLAN_STAMP(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_plan_phase[k], pclk.lap()); } while (0)
#else
#define PLAN_STAMP(k) ((void)0)
#endif
__global__ __launch_bounds__(PLAN_THREADS) void k_plan_fusion(const uint8_t* __restrict__ ref_bytes, const dsa_fusion* __restrict__ fusions,
                                                               const uint8_t* __restrict__ read_bytes, const dsa_pair* __restrict__ pairs,
                                                               PlanRun* __restrict__ runs, uint32_t* __restrict__ fkey, int32_t* __restrict__ fidx,
                                                               int32_t* __restrict__ rank, uint16_t* __restrict__ bound_out,
                                                               PlanGlobals* __restrict__ glob, PlanParams prm)
{
    extern __shared__ uint32_t plan_lds[];
    DiagClock pclk;
    (void)pclk;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int wc = prm.wc, vc = wc / 2 + 2, slots = prm.slots;
    uint32_t* codes0 = plan_lds;
    uint32_t* codes1 = codes0 + wc;
    uint32_t* valid0 = codes1 + wc;
    uint32_t* valid1 = valid0 + vc;
    uint32_t* table0 = valid1 + vc;
    uint32_t* table1 = table0 + slots;
    int* s_votes = reinterpret_cast<int*>(table1 + slots);     // histogram of d2 - d1 + slots / 2 (slots >= 2 x the longest window)
    uint32_t* keys_small = table1 + 2 * slots;                  // sort keys of fusions of at most 2 x PLAN_THREADS pairs
    uint32_t* keys_big = table0;                                // larger fusions: the keys take the tables' place once the lookups are done
    __shared__ int s_tile[2][PLAN_TILES];
    __shared__ unsigned s_best[PLAN_THREADS / 64];
    __shared__ int s_delta, s_delta_votes;

    PlanRun run = runs[f];
    if (tid == 0) fidx[f] = f;
    if (run.nruns != 1) {                               // no pair in the slice, or not one run
        if (tid == 0) {
            if (run.nruns > 1) glob->identity = 1;
            runs[f].count = 0;
            fkey[f] = PLAN_KEY_NONE;                    // sorts behind every fusion with pairs
        }
        return;
    }
    const int n = run.last - run.first + 1;
    const int64_t p0 = run.first;
    const dsa_fusion fu = fusions[f];
    const int len0 = fu.ref0_len, len1 = fu.ref1_len;
    const int tile_cols = prm.tile_cols;
    // size class of the fusion (the table tiers of the fill kernels: whole waves, split tables at 4 / 2 workgroups per CU, generic)
    const int cls = n >= WAVE ? 0 : n >= WG_LANES / GSPLIT ? 1 : n >= (WG_LANES + GSPLIT2 - 1) / GSPLIT2 ? 2 : 3;
    const bool windows_ok = len0 >= PLAN_K && len0 < PLAN_MAXWIN && len1 >= PLAN_K && len1 < PLAN_MAXWIN;
    const bool full = n >= 2 && n <= PLAN_MAX && n <= 2 * slots && windows_ok && (prm.use_rank || prm.use_bound);

    if (tid < 2 * PLAN_TILES) (&s_tile[0][0])[tid] = 0;
    if (!full) {
        // the caller's order inside the fusion, no bound, no tile vote
        for (int k = tid; k < n; k += PLAN_THREADS) {
            rank[p0 + k] = k;
            bound_out[p0 + k] = 0;
        }
        if (tid == 0) {
            runs[f].count = n;
            fkey[f] = plan_fusion_key(cls, 0, 255, 255);
        }
        return;
    }

    // The bases of a read: chunks of 32 bases, 2-bit packed (reads of at most PLAN_LQ bases: all of them, in registers; longer
    // reads get no bound and only their first and last 32 bases are looked at).
    struct ReadBits {
        PlanChunk c[PLAN_CHUNKS];
        PlanChunk tail;
        int lq, tail_base;
    };
    auto load_read = [&](const dsa_pair& pr) -> ReadBits {
        ReadBits rb;
        rb.lq = pr.read_len;
        rb.tail_base = rb.lq > 32 ? rb.lq - 32 : 0;
        const uint8_t* rd = read_bytes + pr.read_off;
#pragma unroll
        for (int c = 0; c < PLAN_CHUNKS; ++c) rb.c[c] = PlanChunk{0u, 0u, 0u};
        rb.tail = PlanChunk{0u, 0u, 0u};
        if (rb.lq < PLAN_K) return rb;
        if (rb.lq <= PLAN_LQ) {
#pragma unroll
            for (int c = 0; c < PLAN_CHUNKS; ++c)
                if (32 * c < rb.lq) rb.c[c] = plan_load_chunk(rd, 32 * c, rb.lq);
            rb.tail = plan_chunk_at(rb.c, rb.tail_base);
        } else {
            rb.c[0] = plan_load_chunk(rd, 0, rb.lq);
            rb.tail = plan_load_chunk(rd, rb.tail_base, rb.lq);
        }
        return rb;
    };
    // the first read of every thread is fetched while the windows are packed (the barrier below waits for it)
    dsa_pair pr_first{};
    ReadBits rb_first{};
    if (tid < n) {
        pr_first = pairs[p0 + tid];
        rb_first = load_read(pr_first);
    }

    PLAN_STAMP(0);
    // ---- the two windows, 2-bit packed with PLAN_PAD invalid bases on either side, and their 11-mers hashed
    for (int k = tid; k < 2 * slots; k += PLAN_THREADS) table0[k] = 0xFFFFFFFFu;
    for (int k = tid; k < slots; k += PLAN_THREADS) s_votes[k] = 0;
    {
        unsigned short* v16_0 = reinterpret_cast<unsigned short*>(valid0);
        unsigned short* v16_1 = reinterpret_cast<unsigned short*>(valid1);
        for (int e = tid; e < 2 * wc; e += PLAN_THREADS) {
            const int h = e >= wc, wd = h ? e - wc : e;
            const int len = h ? len1 : len0;
            const uint8_t* src = ref_bytes + (h ? fu.ref1_off : fu.ref0_off);
            uint32_t code = 0, val = 0;
            const int x0 = 16 * wd - PLAN_PAD;
            if (x0 >= 0 && x0 + 16 <= len) {                    // whole words: aligned dword loads
                const uintptr_t a = reinterpret_cast<uintptr_t>(src + x0);
                const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(a & 3u);
                uint32_t raw[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) raw[k] = w[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t c8, v4;
                    plan_pack4(__builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh), c8, v4);
                    code |= c8 << (8 * k);
                    val |= v4 << (4 * k);
                }
            } else if (x0 + 16 > 0 && x0 < len) {
                for (int b = 0; b < 16; ++b) {
                    const int x = x0 + b;
                    if (x >= 0 && x < len) {
                        const uint32_t by = src[x];
                        const bool ok = by == 'A' || by == 'C' || by == 'G' || by == 'T';
                        code |= ((by >> 1) & 3u) << (2 * b);
                        val |= (ok ? 1u : 0u) << b;
                    }
                }
            }
            (h ? codes1 : codes0)[wd] = code;
            (h ? v16_1 : v16_0)[wd] = (unsigned short)val;
        }
        for (int e = tid; e < 2 * (2 * vc - wc); e += PLAN_THREADS) {           // the valid words behind the last code word
            const int h = e >= (2 * vc - wc), k = h ? e - (2 * vc - wc) : e;
            (h ? v16_1 : v16_0)[wc + k] = 0;
        }
    }
    __syncthreads();
    PLAN_STAMP(1);
    auto window_kmer = [&](const uint32_t* codes, const uint32_t* valid, int x, bool& ok) -> uint32_t {
        const int s = x + PLAN_PAD;
        const uint64_t c = (((uint64_t)codes[(s >> 4) + 1] << 32) | codes[s >> 4]) >> (2 * (s & 15));
        const uint64_t v = (((uint64_t)valid[(s >> 5) + 1] << 32) | valid[s >> 5]) >> (s & 31);
        ok = ((uint32_t)v & 0x7FFu) == 0x7FFu;
        return (uint32_t)c & 0x3FFFFFu;
    };
    {
        const int n0 = len0 - PLAN_K + 1, n1 = len1 - PLAN_K + 1;       // 11-mer positions of the two windows, one after the other
        for (int x = tid; x < n0 + n1; x += PLAN_THREADS) {
            const bool second = x >= n0;
            const int xx = second ? x - n0 : x;
            bool ok;
            const uint32_t km = window_kmer(second ? codes1 : codes0, second ? valid1 : valid0, xx, ok);
            if (ok) plan_table_insert(second ? table1 : table0, slots, km, xx);
        }
    }
    __syncthreads();

    PLAN_STAMP(2);
    // ---- per read: the diagonals of its first 11-mers in window 0 (d1) and of its last ones in window 1 (d2)
    struct Diag { int d1, d2; bool have1, have2; };
    auto diagonals = [&](const ReadBits& rb) -> Diag {
        Diag dg{0, 0, false, false};
        const int lq = rb.lq;
        for (int off = 0; off <= 12 && off + PLAN_K <= lq && !dg.have1; off += 4) {
            bool ok;
            const uint32_t km = plan_kmer(rb.c[0], off, ok);
            if (!ok) continue;
            const int x = plan_table_find(table0, slots, km);
            if (x >= 0) { dg.d1 = x - off; dg.have1 = true; }
        }
        for (int off = 0; off <= 12 && off + PLAN_K <= lq && !dg.have2; off += 4) {
            const int at = lq - PLAN_K - off;
            bool ok;
            const uint32_t km = plan_kmer(rb.tail, at - rb.tail_base, ok);
            if (!ok) continue;
            const int y = plan_table_find(table1, slots, km);
            if (y >= 0) { dg.d2 = y - at; dg.have2 = true; }
        }
        return dg;
    };
    // the reads that vote on d2 - d1 (the same for every read that spans the junction): the first PLAN_THREADS of the fusion
    Diag dg_first{0, 0, false, false};
    if (tid < n) {
        dg_first = diagonals(rb_first);
        if (dg_first.have1 && dg_first.have2) {
            const int v = dg_first.d2 - dg_first.d1 + slots / 2;
            if (v >= 0 && v < slots) atomicAdd(&s_votes[v], 1);
        }
    }
    __syncthreads();
    {   // the mode of the votes; ties to the smaller difference (deterministic)
        unsigned best = 0;
        for (int v = tid; v < slots; v += PLAN_THREADS) {
            const unsigned c = (unsigned)s_votes[v];
            if (c) best = max(best, (c << 12) | (unsigned)(4095 - v));
        }
        for (int d = 32; d >= 1; d >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, d, 64));
        if ((tid & 63) == 0) s_best[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned b = 0;
            for (int w = 0; w < PLAN_THREADS / 64; ++w) b = max(b, s_best[w]);
            s_delta_votes = (int)(b >> 12);
            s_delta = b ? (4095 - (int)(b & 4095u)) - slots / 2 : 0;
        }
        __syncthreads();
    }
    const int delta = s_delta, delta_votes = s_delta_votes;
    PLAN_STAMP(3);

    // ---- per read: bound T', tile votes, sort key
    for (int k = tid; k < n; k += PLAN_THREADS) {
        const bool first = k == tid;
        const dsa_pair pr = first ? pr_first : pairs[p0 + k];
        const ReadBits rb = first ? rb_first : load_read(pr);
        Diag dg = first ? dg_first : diagonals(rb);
        const int lq = pr.read_len;
        // a read with one side only (its junction lies within a few bases of one end) takes the other diagonal from the
        // fusion's vote; any pair of diagonals gives a VALID bound below, a wrong guess only a weak one
        if (delta_votes > 0) {
            if (dg.have1 && !dg.have2) { dg.d2 = dg.d1 + delta; dg.have2 = true; }
            else if (dg.have2 && !dg.have1) { dg.d1 = dg.d2 - delta; dg.have1 = true; }
        }
        const int diag = dg.have1 ? min(1022, max(0, dg.d1 + 16)) : 1023;      // no diagonal at all: sorts to the small-a* end
        int tprime = 0;
        // read base j lies on window 0 position j + d1 (prefix side) and on window 1 position j + d2 (suffix side); the
        // ungapped paths along the two diagonals are valid DP paths (free start in the reference), so the best split scores
        // at least max_a P1(a) + P2(a) over the a where both sides reach the anchor minimum
        if (prm.use_bound && dg.have1 && dg.have2 && dg.d1 >= 0 && lq - 1 + dg.d2 < len1 && lq >= PLAN_K && lq <= PLAN_LQ && dg.d2 > -PLAN_PAD) {
            const int a_hi = min(lq, len0 - dg.d1);        // the prefix path stays inside window 0
            const int a_lo = max(0, -dg.d2);               // the suffix path stays inside window 1
            if (a_lo <= a_hi) {
                uint32_t m1[PLAN_CHUNKS], m2[PLAN_CHUNKS];
                int suf_matches = 0;
#pragma unroll
                for (int c = 0; c < PLAN_CHUNKS; ++c) {
                    m1[c] = m2[c] = 0;
                    if (32 * c < lq) {
                        m1[c] = plan_match32(rb.c[c], codes0, valid0, 32 * c, dg.d1);
                        m2[c] = plan_match32(rb.c[c], codes1, valid1, 32 * c, dg.d2);
                        // matches of the suffix side at read bases >= a_lo
                        const int lo = a_lo - 32 * c;
                        const uint32_t from = lo <= 0 ? 0xFFFFFFFFu : lo >= 32 ? 0u : ~((1u << lo) - 1u);
                        suf_matches += __builtin_popcount(m2[c] & from);
                    }
                }
                // P1(a) = 3 * (matches among read bases < a) - a,   P2(a) = 3 * (matches among bases >= a) - (lq - a); the best
                // (score, smallest a) travels as one word: score << 8 | 255 - a
                int pre = 0, suf = 3 * suf_matches - (lq - a_lo);
                uint32_t best = 0;
#pragma unroll
                for (int c = 0; c < PLAN_CHUNKS; ++c) {
                    if (32 * c <= a_hi) {
                        const int b_end = min(32, a_hi - 32 * c + 1);
                        uint32_t w1 = m1[c], w2 = m2[c];
                        for (int b = 0; b < b_end; ++b) {
                            const int a = 32 * c + b;
                            if (a >= a_lo) {
                                if (min(pre, suf) >= DSA_MIN_SPLIT) best = max(best, ((uint32_t)(pre + suf) << 8) | (uint32_t)(255 - a));
                                suf -= 3 * (int)(w2 & 1u) - 1;
                            }
                            pre += 3 * (int)(w1 & 1u) - 1;
                            w1 >>= 1;
                            w2 >>= 1;
                        }
                    }
                }
                tprime = min((int)(best >> 8), 65535);
                const int best_a = best ? 255 - (int)(best & 255u) : -1;
                if (best_a > 0) {
                    // the tiles in which the two alignments end: matrix column d1 + a* of M1, and of M2 (reversed window 1)
                    // the column len1 - (a* + d2)
                    const int t1 = (dg.d1 + best_a - 1) / tile_cols, s1 = best_a + dg.d2;
                    const int t2 = (len1 - s1 - 1) / tile_cols;
                    atomicAdd(&s_tile[0][min(max(t1, 0), PLAN_TILES - 1)], 1);
                    atomicAdd(&s_tile[1][min(max(t2, 0), PLAN_TILES - 1)], 1);
                }
            }
        }
        if (prm.use_bound && tprime == 0 && lq >= PLAN_K && lq <= PLAN_LQ) {
            // A read that lies on ONE side as a whole — most of the mates DoAlignment enumerates do not cross the junction
            // (tools/SplitAlignment.cpp:266-303) — has no split with both sides above the anchor minimum, but the split a = Lq
            // (everything in M1, the M2 side empty and counted as 0, tools/SplitReadAligner.cpp:156-298) or a = 0 competes with
            // its one side's score: the ungapped path of the WHOLE read along its diagonal is a lower bound of the final score
            // too.  Without it such a read is swept with the slack of minScore alone.
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const bool have = side ? dg.have2 : dg.have1;
                const int d = side ? dg.d2 : dg.d1, len = side ? len1 : len0;
                if (have && d >= 0 && d + lq <= len) {
                    int matches = 0;
#pragma unroll
                    for (int c = 0; c < PLAN_CHUNKS; ++c)
                        if (32 * c < lq) matches += __builtin_popcount(plan_match32(rb.c[c], side ? codes1 : codes0, side ? valid1 : valid0, 32 * c, d));
                    const int whole = 3 * matches - lq;
                    if (whole >= DSA_MIN_SPLIT) tprime = max(tprime, min(whole, 65535));
                }
            }
        }
        bound_out[p0 + k] = (uint16_t)tprime;
        const uint32_t key = ((uint32_t)(prm.use_rank ? diag : 0) << 16) | (uint32_t)k;
        if (n <= 2 * PLAN_THREADS) keys_small[k] = key;
        else rank[p0 + k] = (int32_t)key;                  // parked in the output array until the tables are free
    }
    __syncthreads();
    PLAN_STAMP(4);

    // ---- rank of every read inside the fusion: ascending (diagonal key, index) — the keys are distinct
    if (n <= 2 * PLAN_THREADS) {
        for (int k = tid; k < n; k += PLAN_THREADS) {
            const uint32_t me = keys_small[k];
            int r = 0;
            for (int i = 0; i < n; ++i) r += keys_small[i] < me ? 1 : 0;
            rank[p0 + k] = r;
        }
    } else {
        int npad = 1;
        while (npad < n) npad <<= 1;                        // <= 2 * slots: the keys fit where the tables were
        for (int k = tid; k < npad; k += PLAN_THREADS) keys_big[k] = k < n ? (uint32_t)rank[p0 + k] : (0xFFFF0000u | (uint32_t)k);
        __syncthreads();
        for (int size = 2; size <= npad; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < npad / 2; t += PLAN_THREADS) {
                    const int lo = (t / stride) * stride * 2 + (t % stride), hi = lo + stride;
                    const bool up = ((lo / size) & 1) == 0;
                    const uint32_t a = keys_big[lo], b = keys_big[hi];
                    if ((a > b) == up) { keys_big[lo] = b; keys_big[hi] = a; }
                }
                __syncthreads();
            }
        for (int r = tid; r < n; r += PLAN_THREADS) rank[p0 + (keys_big[r] & 0xFFFFu)] = r;
    }

    PLAN_STAMP(5);
    if (tid == 0) {
        int t[2];
        for (int h = 0; h < 2; ++h) {
            int best = 0;
            t[h] = 255;
            for (int c = 0; c < PLAN_TILES; ++c)
                if (s_tile[h][c] > best) { best = s_tile[h][c]; t[h] = c; }
        }
        // cost proxy: the number of distinct tiles in which either matrix is alive, {t1-1, t1} and {t2-1, t2} — 2 when they
        // coincide, up to 4; no vote: assume the worst.  Expensive fusions first inside a size class.
        int alive = 4;
        if (t[0] != 255 && t[1] != 255) {
            const int d = t[0] > t[1] ? t[0] - t[1] : t[1] - t[0];
            alive = d == 0 ? 2 : d == 1 ? 3 : 4;
        }
        runs[f].count = n;
        fkey[f] = plan_fusion_key(cls, prm.use_lpt ? 4 - alive : 0, t[0], t[1]);
    }
}

// ---- the start of every fusion in the sweep: exclusive scan of the counts in sorted order --------------------------
constexpr int PLACE_BLOCK = 1024;
__device__ __forceinline__ int place_block_sum(int v, int* s_part)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < PLACE_BLOCK / 64; ++w) total += s_part[w];
    __syncthreads();
    return total;
}
__global__ __launch_bounds__(PLACE_BLOCK) void k_plan_place_a(const int32_t* __restrict__ forder, const PlanRun* __restrict__ runs, int nf,
                                                              int32_t* __restrict__ bsum)
{
    __shared__ int s_part[PLACE_BLOCK / 64];
    const int i = blockIdx.x * PLACE_BLOCK + threadIdx.x;
    const int c = i < nf ? runs[forder[i]].count : 0;
    const int total = place_block_sum(c, s_part);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(PLACE_BLOCK) void k_plan_place_b(const int32_t* __restrict__ forder, const PlanRun* __restrict__ runs, int nf,
                                                              const int32_t* __restrict__ bsum, int32_t* __restrict__ new_start,
                                                              uint8_t* __restrict__ flip)
{
    __shared__ int s_part[PLACE_BLOCK / 64];
    __shared__ int s_wave[PLACE_BLOCK / 64];
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += PLACE_BLOCK) before += bsum[b];
    const int base = place_block_sum(before, s_part);
    const int i = blockIdx.x * PLACE_BLOCK + threadIdx.x;
    const int f = i < nf ? forder[i] : 0;
    const int c = i < nf ? runs[f].count : 0;
    int incl = c;
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += y;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wbase += s_wave[w];
    if (i < nf) {
        new_start[f] = base + wbase + incl - c;
        flip[f] = (uint8_t)(i & 1);                   // alternate fusions in opposite directions
    }
}

// pair p of fusion f (one run) goes to sweep position new_start[f] + its rank inside the fusion (mirrored for every other
// fusion); identity: the caller's order.  The bound travels in the padding bytes of the device copy.  The DP cells of the
// slice (dsa_timing.cells: 2 matrices of (Lref + 1) x (Lread + 1)) are summed here, on every path — with or without a
// sweep order, whatever the runs of the fusions look like.
__global__ __launch_bounds__(256) void k_plan_permute(const dsa_pair* __restrict__ pairs, int64_t n, const dsa_fusion* __restrict__ fusions,
                               const PlanRun* __restrict__ runs,
                               const int32_t* __restrict__ new_start, const uint8_t* __restrict__ flip, const int32_t* __restrict__ rank,
                               const uint16_t* __restrict__ bound, PlanGlobals* __restrict__ glob, int force_identity,
                               dsa_pair* __restrict__ sweep, int32_t* __restrict__ orig)
{
    __shared__ unsigned long long s_cells[4];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cells = 0;
    if (p < n) {
        dsa_pair pr = pairs[p];
        const int f = pr.fusion_idx;
        cells = (unsigned long long)(pr.read_len + 1) * (unsigned long long)(fusions[f].ref0_len + 1 + fusions[f].ref1_len + 1);
        int64_t q = p;
        uint32_t b = 0;
        if (!force_identity && glob->identity == 0) {
            const int r = rank[p], c = runs[f].count;
            q = (int64_t)new_start[f] + (flip[f] ? c - 1 - r : r);
            b = bound[p];
        }
        pr.pad_[0] = (uint8_t)(b & 0xFF);
        pr.pad_[1] = (uint8_t)(b >> 8);
        sweep[q] = pr;
        orig[q] = (int32_t)p;
    }
    for (int d = 32; d >= 1; d >>= 1) cells += __shfl_xor(cells, d, 64);
    if ((threadIdx.x & 63) == 0) s_cells[threadIdx.x >> 6] = cells;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&glob->cells, s_cells[0] + s_cells[1] + s_cells[2] + s_cells[3]);
}

}  // namespace dsa
