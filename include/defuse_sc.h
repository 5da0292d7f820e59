/*
 * defuse_sc.h — C ABI of the MI355X greedy set cover ("sc") used by the drop-in `setcover` tool.
 *
 * Replaces SetCover(clusters, solution) of the reference (tools/setcover.cpp:30-110): repeatedly
 * take the cluster with the most still-unassigned fragments, assign them to it, and decrement every
 * cluster that contains them.  Among clusters of equal size the one that arrived at that size most
 * recently wins (boost::bimap multiset_of semantics); initial arrival order is
 * ascending cluster index.  The greedy decomposes over the connected components of the
 * cluster/fragment graph, which is what the device exploits: one lane (small components) or one wave
 * (large ones) per component, no global rounds.
 *
 * Plain C types, host pointers.  Returns 0 on success, negative on failure (same codes as defuse_dsa.h).
 */
#ifndef DEFUSE_SC_H_
#define DEFUSE_SC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sc_timing {
    float   build_ms;        /* fragment -> clusters index (radix sort) */
    float   components_ms;   /* connected components (label propagation) */
    float   greedy_ms;       /* per-component greedy */
    float   total_ms;
    int32_t n_components;
    int32_t n_large;         /* components handled by a whole wave */
    int32_t cc_iterations;
    int32_t pad_;
} sc_timing;

/* clusters are given in CSR form: cluster c owns elements[cluster_off[c] .. cluster_off[c+1]) in file
 * order (duplicates allowed, as in the reference).  owner[e] receives the cluster that fragment e was
 * assigned to, or -1 if e occurs in no cluster; owner must have room for max_element+1 entries. */
int sc_cover(int device, const int64_t* cluster_off, const int32_t* elements, int32_t n_clusters,
             int32_t max_element, int32_t* owner, sc_timing* timing);
const char* sc_last_error(void);
/* Optional: brings the runtime and the device's context up (a few tenths of a second in a fresh process), so that a caller
 * can let that happen on a thread of its own while it parses its input.  Thread-safe; 0, or -2 without a usable device
 * (sc_cover reports that again). */
int sc_prepare(int device);

#ifdef __cplusplus
}
#endif
#endif
