// dsa_diag.hpp — everything diagnostic about the split-read kernels in one place; a product build (no -D flag) compiles none
// of it.  Builds with these flags go to build_var/ (profiles/microbench/build_variant.sh) and are named by DEFUSE_DSA_LIB.
//
//   -DDSA_PRUNE_STATS   row-group, stage-cycle and task statistics of a run, printed on stderr by the host (diag_dump_slice)
//   -DDSA_NO_PRUNE      the fill sweeps every row of every tile (no exact pruning)
//   -DDSA_NO_GAP_SKIP   ... sweeps through dead row groups inside a tile instead of skipping to the next live boundary
//   -DDSA_NARROW        a last tile of at most 16 columns is swept by a 16-column instantiation of the loop (measured: slower)
//   -DDSA_ABLATE_TAIL   the fill kernel ends after its sweeps (timing only: no records)
//   -DDSA_ABLATE_REPLAY ... after the combine step (timing only)
#pragma once
#include <hip/hip_runtime.h>

namespace dsa {

#ifdef DSA_NO_PRUNE
constexpr bool DIAG_PRUNE = false;
#else
constexpr bool DIAG_PRUNE = true;
#endif
#ifdef DSA_NO_GAP_SKIP
constexpr bool DIAG_GAP_SKIP = false;
#else
constexpr bool DIAG_GAP_SKIP = true;
#endif
#ifdef DSA_NARROW
constexpr bool DIAG_NARROW = true;
#else
constexpr bool DIAG_NARROW = false;
#endif
#ifdef DSA_ABLATE_TAIL
constexpr bool DIAG_TAIL = false;
#else
constexpr bool DIAG_TAIL = true;
#endif
#ifdef DSA_ABLATE_REPLAY
constexpr bool DIAG_REPLAY = false;
#else
constexpr bool DIAG_REPLAY = true;
#endif

// statistics slots (Geom::stats, 16 x u64 per lane of the pipeline)
enum DiagSlot {
    DS_GROUPS_SKIPPED = 0, DS_GROUPS = 1, DS_UNUSED2 = 2, DS_WAVE_CYCLES = 3, DS_BARRIER = 4, DS_TABLES = 5, DS_TAIL = 6, DS_ROWMAX = 7,
    DS_COMBINE = 8, DS_REPLAY = 9, DS_GREPLAY_SETUP = 10, DS_GREPLAY_SWEEP = 11, DS_GREPLAY_WAVES = 12, DS_GREPLAY_STEPS = 13,
    DS_GAP_GROUPS = 14, DS_SLOWEST_LISTED = 15
};

#ifdef DSA_PRUNE_STATS
struct DiagClock {                                   // wave-level cycle stamps (s_memtime)
    unsigned long long t;
    __device__ DiagClock() : t(__builtin_readcyclecounter()) {}
    __device__ unsigned long long lap()
    {
        const unsigned long long n = __builtin_readcyclecounter(), d = n - t;
        t = n;
        return d;
    }
};
#define DSA_STAT_ADD(g, slot, v) atomicAdd(&(g).stats[slot], (unsigned long long)(v))
#define DSA_STAT_MAX(g, slot, v) atomicMax(&(g).stats[slot], (unsigned long long)(v))
#else
struct DiagClock {
    __device__ unsigned long long lap() { return 0; }
};
#define DSA_STAT_ADD(g, slot, v) ((void)0)
#define DSA_STAT_MAX(g, slot, v) ((void)0)
#endif

}  // namespace dsa
