/*
 * defuse_la.h — C ABI of the MI355X batched "local realignment" scorer ("la") used by the drop-in
 * `localalign` tool.
 *
 * Replaces SimpleAligner::Align (tools/SimpleAligner.cpp:24-64) for a whole batch of
 * (reference, sequence) pairs: H(i,0) = 0, H(0,j) = j*gap, else
 *     H(i,j) = max(H(i-1,j-1) + (reference[i-1] == sequence[j-1] ? match : mismatch),
 *                  H(i-1,j) + gap, H(i,j-1) + gap)
 * with an exact byte comparison (case-sensitive, as the reference), and the score of a pair is
 * max(0, max over i >= 1, j >= 1 of H(i,j)).  Same recurrence as the split-read DP of defuse_dsa.h with
 * run-time scores and a whole-matrix maximum instead of row maxima.
 *
 * Pairs are independent.  The device packs two pairs per lane in 16-bit fields when the scores allow
 * (gap <= mismatch <= 0 and the value range fits), else computes one pair per lane in int32.
 *
 * Plain C types, host pointers.  Returns 0 on success, negative on failure (codes of defuse_dsa.h).
 */
#ifndef DEFUSE_LA_H_
#define DEFUSE_LA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct la_item {
    int64_t ref_off;     /* byte offset of the reference in pool */
    int64_t seq_off;     /* byte offset of the sequence in pool  */
    int32_t ref_len;
    int32_t seq_len;
} la_item;

typedef struct la_timing {
    float   pack_ms;
    float   kernel_ms;
    float   total_ms;
    int32_t n_packed16;  /* pairs scored by the packed 16-bit kernel */
    int32_t n_int32;     /* pairs scored by the int32 kernel         */
    int32_t pad_;
    int64_t cells;       /* sum of (ref_len+1)*(seq_len+1)           */
} la_timing;

/* scores[k] receives SimpleAligner::Align(reference_k, sequence_k) for the given scoring
 * (tools/localalign.cpp:84: `aligner.Align(reference, sequence)`). */
int la_align_batch(int device, int32_t match, int32_t mismatch, int32_t gap, const uint8_t* pool, int64_t pool_len,
                   const la_item* items, int64_t n_items, int32_t* scores, la_timing* timing);
/* The same with a minimum score per pair (NULL: none): scores[k] is exact when it is >= min_score[k]; a pair
 * that scores less only gets some value below min_score[k] (and >= 0).  This is what the tool's threshold
 * needs (tools/localalign.cpp:89-92 drops the lines below it), and it lets the device leave every tile as
 * soon as nothing in it can still reach the minimum. */
int la_align_batch_min(int device, int32_t match, int32_t mismatch, int32_t gap, const uint8_t* pool, int64_t pool_len,
                       const la_item* items, int64_t n_items, const int32_t* min_score, int32_t* scores, la_timing* timing);
const char* la_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
