/*
 * defuse_mpe.h — C ABI of the MI355X mate-pair EM clustering ("mpe") used by the drop-in
 * `clustermatepairs` tool.
 *
 * Replaces MatePairEM::DoClustering (tools/MatePairEM.cpp:540-636) for a whole batch of bin pairs:
 * for every problem (one bin pair after the filters of tools/clustermatepairs.cpp:478-545) a mixture
 * model over the mate pairs' break coordinates is fitted by EM for K = 1..min(10,N) (k-means
 * initialisation with KKZ seeding + AS 136, tools/asa136.C), K is chosen by BIC, and every component
 * reports the mate pairs whose pair probability exceeds the precision threshold.  Everything is FP64,
 * evaluated in the reference's operation order (serial prefix sums, serial reductions): every problem runs in
 * one wave whose lanes share the elementwise work and run the independent serial chains of its K fits side by
 * side (DEFUSE_MPE_WAVE_MIN=n sends problems of fewer than n mate pairs through the one-lane-per-fit
 * transcription the wave version is checked against); problems are independent.
 *
 * Plain C types, host pointers.  Returns 0 on success, negative on failure (codes of defuse_dsa.h).
 */
#ifndef DEFUSE_MPE_H_
#define DEFUSE_MPE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_KMAX 10            /* mKMax, tools/MatePairEM.cpp:57 */

typedef struct mpe_params {
    double  fragment_mean;     /* -u */
    double  fragment_stddev;   /* -s */
    double  min_probability;   /* mMinProbability: normalpdf(-s*PhiInv((1-precision)/2), 0, s), computed by the caller */
    int32_t min_cluster_size;  /* -m */
    int32_t pad_;
} mpe_params;

typedef struct mpe_timing {
    float   kernel_ms;
    int32_t n_problems;
    int64_t n_mate_pairs;
    int64_t em_iterations;     /* total EM iterations over all problems and all K */
    int32_t n_failed;          /* problems that hit one of the reference's DebugCheck exits */
    int32_t n_wave_problems;   /* problems fitted by a wave of their own (all, unless DEFUSE_MPE_WAVE_MIN is set) */
} mpe_timing;

/* Problem p owns mate pairs [prob_off[p], prob_off[p+1]).  Per mate pair: x, y = strand-remapped
 * alignment ends of the two sides, u = fragment_mean - len1 - len2; to_xo / to_yo = rank of the mate
 * pair inside its problem when sorted by x (resp. y) descending, ties by index ascending.
 * Outputs: n_clusters[p] = number of emitted clusters (<= MPE_KMAX) of problem p; member[i] bit j set
 * iff mate pair i belongs to emitted cluster j of its problem; status[p] = 0 ok, 1 = the reference
 * would have exited through a DebugCheck (the tool then exits 1 as the reference does). */
int mpe_cluster_batch(int device, const mpe_params* params, const int64_t* prob_off, int32_t n_problems,
                      const double* x, const double* y, const double* u, const int32_t* to_xo, const int32_t* to_yo,
                      int32_t* n_clusters, uint16_t* member, int32_t* status, mpe_timing* timing);
/* The same over several GPUs of the node (bin pairs are independent): the problems are cut into n_devices
 * contiguous shares of about equal mate pair count, share k runs on HIP device devices[k] from a host thread of its own, and
 * the results land in the caller's arrays exactly as one call would have left them (cluster numbering is the caller's
 * prefix sum over n_clusters, so it does not depend on the shares).  A device may be named more than once.  timing->kernel_ms
 * is the longest share's. */
int mpe_cluster_batch_sharded(const int* devices, int32_t n_devices, const mpe_params* params, const int64_t* prob_off,
                              int32_t n_problems, const double* x, const double* y, const double* u, const int32_t* to_xo,
                              const int32_t* to_yo, int32_t* n_clusters, uint16_t* member, int32_t* status, mpe_timing* timing);
const char* mpe_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
