/*
 * defuse_cov.h — C ABI of the MI355X sampling kernel behind the drop-in `calccov` tool.
 *
 * Replaces the per-fragment loop of the reference's calccov (tools/calccov.cpp:155-215): for every concordant
 * fragment on a sampled transcript, the sample positions of that transcript that fall into the unsequenced span of the
 * fragment give a (sample, fragment length) pair, and those inside the anchored part of each of its two reads give a
 * (sample, split position) and a (sample, split minimum) pair (CalculateSplitPos / CalculateSplitMin, :236-250).
 * Fragments are independent.  Output order is the reference's with its unordered_set walked in ascending sample index
 * (canonical order): fragment by fragment; the length samples of a fragment by sample index; its split
 * samples read 0 then read 1, each by sample index.
 *
 * Plain C types, host pointers.  Returns 0 on success, negative on failure (codes of defuse_dsa.h).
 */
#ifndef DEFUSE_COV_H_
#define DEFUSE_COV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cov_fragment {
    int32_t ref;                 /* index of the sampled transcript (into ref_sample_off)          */
    int32_t start[2], end[2];    /* regions of the fragment's two alignments, in file order        */
} cov_fragment;

typedef struct cov_timing {
    float   kernel_ms;
    int32_t pad_;
    int64_t n_length_samples, n_split_samples;
} cov_timing;

/* Transcript r owns the samples [ref_sample_off[r], ref_sample_off[r+1]) (sample index = position in sample_pos:
 * samples are generated transcript by transcript, tools/calccov.cpp:128-150).
 * Outputs (caller-allocated, capacities in entries): length_idx/length_val (sample index, fragment length) and
 * split_idx/split_pos/split_min.  *n_length / *n_split receive the required counts; when a capacity is too small the
 * call returns DSA_E_CAPACITY (-1) after setting them, and writes nothing. */
int cov_sample_batch(int device, const int64_t* ref_sample_off, int32_t n_refs, const int32_t* sample_pos,
                     const cov_fragment* fragments, int64_t n_fragments, int32_t trim_length, int32_t split_min_anchor,
                     int32_t* length_idx, int32_t* length_val, int64_t length_cap, int64_t* n_length,
                     int32_t* split_idx, double* split_pos, double* split_min, int64_t split_cap, int64_t* n_split,
                     cov_timing* timing);
const char* cov_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
