// dsa_diag_host.hpp — host side of the -DDSA_PRUNE_STATS builds (dsa_diag.hpp): after a slice, the statistics the kernels
// gathered and the left-over replay's tasks by last row and sidedness, on stderr.  Not part of a product build.
#pragma once
#ifdef DSA_PRUNE_STATS
#include <cstdio>
#include <vector>

#include "dsa_kernels.hpp"

namespace dsa {

inline void diag_dump_slice(const unsigned long long* d_stats, const uint2* d_gtasks, size_t ng, const ReplayTask* d_tasks, size_t nt)
{
    std::vector<uint2> gt(ng);
    std::vector<ReplayTask> tk(nt);
    if (ng) (void)hipMemcpy(gt.data(), d_gtasks, ng * sizeof(uint2), hipMemcpyDeviceToHost);
    if (nt) (void)hipMemcpy(tk.data(), d_tasks, nt * sizeof(ReplayTask), hipMemcpyDeviceToHost);
    long hist[3][10] = {};
    for (size_t i = 0; i < ng; ++i) {
        const ReplayTask& t = tk[gt[i].x & ~GTASK_OWNER];
        const int kind = t.chunk0 != NO_CHUNK && t.chunk1 != NO_CHUNK ? 2 : t.chunk0 != NO_CHUNK ? 0 : 1;
        hist[kind][std::min(9, (t.last_row & TASK_ROW) / 8)]++;
    }
    for (int k = 0; k < 3; ++k) {
        fprintf(stderr, "[stats] left-over replay tasks, %s, by last row / 8:", k == 0 ? "M1 side only" : k == 1 ? "M2 side only" : "both sides");
        for (int b = 0; b < 10; ++b) fprintf(stderr, " %ld", hist[k][b]);
        fprintf(stderr, "\n");
    }
    unsigned long long h[16];
    (void)hipMemcpy(h, d_stats, sizeof h, hipMemcpyDeviceToHost);
    const double wc = h[DS_WAVE_CYCLES] ? (double)h[DS_WAVE_CYCLES] : 1.0;
    fprintf(stderr,
            "[stats] fill: row groups not swept %llu of %llu (of which skipped between live parts %llu); wave cycles %.4g: tile barriers %.1f %%, "
            "table builds %.1f %%, tail %.1f %% (row maxima %.1f, combine %.1f, replay %.1f), sweeps %.1f %%\n",
            h[DS_GROUPS_SKIPPED], h[DS_GROUPS], h[DS_GAP_GROUPS], wc, 100.0 * h[DS_BARRIER] / wc, 100.0 * h[DS_TABLES] / wc, 100.0 * h[DS_TAIL] / wc,
            100.0 * h[DS_ROWMAX] / wc, 100.0 * h[DS_COMBINE] / wc, 100.0 * h[DS_REPLAY] / wc,
            100.0 * (wc - h[DS_BARRIER] - h[DS_TABLES] - h[DS_TAIL]) / wc);
    fprintf(stderr, "[stats] left-over replay: %llu waves, set-up %llu, sweep %llu cycles, %llu steps (lane 0); slowest lane of the listed count: %llu cycles, "
            "%llu kept rows, %llu tasks\n", h[DS_GREPLAY_WAVES], h[DS_GREPLAY_SETUP], h[DS_GREPLAY_SWEEP], h[DS_GREPLAY_STEPS], h[DS_SLOWEST_LISTED] >> 24,
            (h[DS_SLOWEST_LISTED] >> 8) & 0xFFFF, h[DS_SLOWEST_LISTED] & 0xFF);
}

inline void diag_dump_plan()
{
    unsigned long long h[8] = {}, zero[8] = {};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_plan_phase), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_plan_phase), zero, sizeof zero);
    double tot = 0;
    for (int k = 0; k < 6; ++k) tot += (double)h[k];
    if (tot == 0) return;
    const char* names[6] = {"prologue + first read", "windows packed", "11-mers hashed", "diagonals + vote", "bounds + keys", "ranks"};
    fprintf(stderr, "[stats] k_plan_fusion wave cycles %.4g:", tot);
    for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.1f %%", names[k], 100.0 * h[k] / tot);
    fprintf(stderr, "\n");
}

}  // namespace dsa
#endif
