/*
 * defuse_cmp.h — C ABI of the MI355X bin-pair builder ("cmp" = clustermatepairs) of the drop-in `clustermatepairs` tool.
 *
 * Replaces the front half of the reference tool's main loop for all fragments of the input at once
 * (tools/clustermatepairs.cpp:453-476): per fragment
 *
 *     CheckConcordant(alignments, minFusionRange)            tools/clustermatepairs.cpp:211-244
 *       -> Binning::GetBins with length = extend = minFusionRange   :146-176   (C++ int division)
 *     AddBinPairs(alignments, Binning(1 << 15, minFusionRange), binPairs)      :246-290
 *       -> PackAlignment (relative positions in 16 bits, DebugChecks)          :178-192
 *       -> RefBinPacked (18 bits reference, 1 bit strand, 13 bits bin)         :28-65
 *
 * and the unordered_map of bin pairs those calls fill.  On the device: one thread per fragment decides concordance and
 * counts, a scan gives every fragment its place, a second pass writes (bin-pair key, packed alignment) entries for the
 * `first` and the `second` list of every bin pair, and a stable radix sort by key — entries arrive in file order, so a
 * stable sort leaves every list in the order a serial reader would have appended it — turns them into the bin pairs in
 * ascending key order, the canonical visiting order of SURVEY.md 8(c).
 *
 * Plain C types, host pointers, caller-owned buffers.  Returns 0 on success, negative on failure (codes of defuse_dsa.h).
 */
#ifndef DEFUSE_CMP_H_
#define DEFUSE_CMP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One line of the compact alignment input (tools/AlignmentStream.cpp:156-199), reference names replaced by their index in
 * order of first appearance.  Records of one fragment are consecutive (FragmentAlignmentStream, :201-221). */
typedef struct cmp_record {
    int32_t  fragment;          /* readID.fragmentIndex */
    int32_t  start, end;        /* region */
    uint32_t meta;              /* reference index (bits 0-27) | strand << 28 | read end << 29 */
} cmp_record;
#define CMP_META(ref, strand, read_end) ((uint32_t)(ref) | ((uint32_t)(strand) << 28) | ((uint32_t)(read_end) << 29))

/* AlignmentPacked (tools/clustermatepairs.cpp:67-78): what a bin pair's lists hold */
typedef struct cmp_packed {
    int32_t  fragment, read_end;
    uint16_t rel_start, rel_end;
} cmp_packed;

typedef struct cmp_stats {
    int64_t n_fragments, n_concordant;
    int64_t n_keys;             /* bin pairs */
    int64_t n_first, n_second;  /* entries of all `first` / all `second` lists */
    /* the first alignment (in file order) on which the reference would have stopped, -1 if none:
     * kind 1 = DebugCheck of PackAlignment (relative position outside 16 bits), 2 = "Packing failed, too many reference
     * sequences" (value = the reference index), 3 = "Packing failed, chromosome too large" (value = the bin) */
    int64_t err_record;
    int32_t err_kind, err_value;
    float   device_ms;          /* kernels and sorts, HIP events */
    float   pad_;
} cmp_stats;

typedef struct cmp_binner cmp_binner;
int  cmp_bin_create(cmp_binner** out, int device);          /* fails without a GPU: there is no CPU path */
void cmp_bin_destroy(cmp_binner* b);
/* room for the whole input on the device; then the records and the fragment starts may be uploaded in pieces (from several
 * host threads if the caller parsed in pieces): records [at, at + n), and frag_start values — the index of every fragment's
 * first record, the caller adds one more entry = n_records at the end */
int  cmp_bin_reserve(cmp_binner* b, int64_t n_records, int64_t n_fragments);
int  cmp_bin_upload_records(cmp_binner* b, const cmp_record* recs, int64_t n, int64_t at);
int  cmp_bin_upload_fragments(cmp_binner* b, const uint32_t* frag_start, int64_t n, int64_t at);
/* everything on the device; the counts tell the caller how much room cmp_bin_fetch needs */
int  cmp_bin_run(cmp_binner* b, int32_t min_fusion_range, cmp_stats* stats);
/* bin pair k: key = (first.id << 32) | second.id (RefBinPacked ids), ascending; its lists are
 * first[off_first[k] .. off_first[k+1]) and second[off_second[k] .. off_second[k+1]) (n_keys + 1 offsets each) */
int  cmp_bin_fetch(cmp_binner* b, uint64_t* keys, int64_t* off_first, int64_t* off_second, cmp_packed* first, cmp_packed* second);
/* entries [from, from + n) of all `first` (which = 0) or all `second` (1) lists: a caller with several host threads passes
 * NULL for the lists above and lets every thread copy its own share into memory it touches for the first time itself */
int  cmp_bin_fetch_part(cmp_binner* b, int which, int64_t from, int64_t n, cmp_packed* out);
const char* cmp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
