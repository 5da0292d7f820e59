/*
 * defuse_hc.h — C ABI of the MI355X average-linkage clusterer ("hc"), a batch version of
 * HierarchicalClusterer::DoClustering (tools/HierarchicalClusterer.cpp:46-140).
 *
 * The reference class is compiled into clustermatepairs (tools/makefile) but no tool calls it; this
 * ABI sits where a caller of DoClustering(distances, threshold) / GetClusters() would sit, one call
 * for the distance tables of many gene pairs at once (one workgroup per table on the device).
 *
 * Per table: every item starts as its own cluster (index 0..n-1); while the smallest distance between
 * two live clusters is < threshold, they are merged into a new cluster with index n + (merges so far),
 * members = members of the smaller index followed by members of the larger one, and its distance to every
 * other live cluster c is (size1*d(1,c) + size2*d(2,c)) / (size1+size2) (:105-108).  Equal smallest
 * distances: the pair whose distance entered the table first wins (multiset_of<double>::begin(); entry
 * order = (i,j) row-major for the input, then, per merge, the live clusters in list order).  The result
 * lists the live clusters in the order of the reference's index list: untouched items by index, then the
 * merged clusters by creation (:124-139).
 *
 * Only distances[i][j] with j > i are read, as in the reference (:63-66).  NaN distances are not supported.
 * Plain C types, host pointers.  Returns 0 on success, negative on failure (codes of defuse_dsa.h).
 */
#ifndef DEFUSE_HC_H_
#define DEFUSE_HC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HC_MAX_ITEMS 16384      /* per table: stamps are 32-bit (3 n^2 < 2^32) */

typedef struct hc_timing {
    float   upload_ms;
    float   kernel_ms;
    float   total_ms;
    int32_t n_merges;        /* over all tables */
} hc_timing;

/* n_items[p]: items of table p; dist_off[p]: index into `distances` of its n×n row-major doubles;
 * item_off[p] = n_items[0] + ... + n_items[p-1] is where table p's output starts (computed by the callee).
 * members[item_off[p] + k]: the k-th listed item (result clusters back to back, members in the reference's order);
 * cluster_of[item_off[p] + k]: index of the result cluster that member belongs to (0 .. n_clusters[p]-1, non-decreasing). */
int hc_cluster_batch(int device, int32_t n_tables, const int32_t* n_items, const int64_t* dist_off,
                     const double* distances, const double* thresholds,
                     int32_t* members, int32_t* cluster_of, int32_t* n_clusters, hc_timing* timing);
const char* hc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
