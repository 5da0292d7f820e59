/*
 * defuse_dsa.h — C ABI of the MI355X split-read alignment path ("dsa" = defuse split align).
 *
 * The reference (amcpherson/defuse) has no FFI: its boundary is the process boundary of the
 * `dosplitalign` tool.  Inside that tool the hot path is one C++ call,
 *
 *     SplitAlignmentTask::Align(aligner, readInfo, readSeq)        tools/SplitAlignment.cpp:371-444
 *       -> SplitReadAligner::Align(read, ref1, ref2)               tools/SplitReadAligner.cpp:77-89
 *            -> FillMatrix x2                                      tools/SplitReadAligner.cpp:24-75
 *       -> SplitReadAligner::GetAlignments(minScore,true,false,false)  tools/SplitReadAligner.cpp:156-298
 *       -> refSplit de-duplication, score=min(score1,score2)       tools/SplitAlignment.cpp:381-400
 *
 * executed once per candidate (fusion, read, revComp) inside SplitReadRealigner::DoAlignment
 * (tools/SplitAlignment.cpp:266-303).  dsa_align_batch() below replaces exactly that call for a
 * whole batch of candidates; the entries of dsa_record are the nine integer columns that
 * SplitAlignment::WriteAlignment prints (tools/SplitAlignment.cpp:305-317).
 *
 * Plain C types only: no torch, no HIP types.  All pointers given to the *_batch entry points are
 * HOST pointers unless the name says "_dev".  Every function returns 0 on success and a negative
 * DSA_E* code on failure; nothing throws across this boundary.  A ctx is bound to one device and
 * must not be used from two threads at once (one ctx per thread is fine).
 */
#ifndef DEFUSE_DSA_H_
#define DEFUSE_DSA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSA_OK            0
#define DSA_E_CAPACITY  (-1)   /* out_cap too small; *out_n holds the required count            */
#define DSA_E_DEVICE    (-2)   /* HIP error (no device, launch failure, out of memory)          */
#define DSA_E_ARG       (-3)   /* inconsistent arguments (offsets out of range, negative sizes) */
#define DSA_E_LIMIT     (-4)   /* a length exceeds what the kernels support (see dsa_limits)    */
#define DSA_E_BUSY      (-5)   /* dsa_stream_submit: every slot is in flight, collect first     */

/* Scoring is fixed by the reference (tools/SplitAlignment.cpp:25-29, :234):
 * match +2, mismatch -1, gap -2, minSplitScore = minAnchor*match = 8, endGaps=false. */
#define DSA_MATCH       2
#define DSA_MISMATCH  (-1)
#define DSA_GAP       (-2)
#define DSA_MIN_SPLIT   8

typedef struct dsa_ctx dsa_ctx;

/* One candidate fusion = SplitAlignmentTask: two reference windows, mSplitAlignSeq[0] and [1]
 * (tools/SplitAlignment.h:69), given as offsets into one concatenated byte buffer.  Bytes are
 * compared raw (tools/SplitReadAligner.cpp:51), so 'N'=='N' matches and case matters. */
typedef struct dsa_fusion {
    int32_t fusion_id;            /* mFusionID, copied to the records                            */
    int32_t ref0_off, ref0_len;   /* mSplitAlignSeq[0] (as fetched, forward)                     */
    int32_t ref1_off, ref1_len;   /* mSplitAlignSeq[1] (as fetched; the kernel reverses it)      */
} dsa_fusion;

/* One candidate (fusion, read) = one call of SplitAlignmentTask::Align.  The read bytes are already
 * oriented (reverse-complemented by the caller iff revcomp, tools/SplitAlignment.cpp:286-290). */
typedef struct dsa_pair {
    int32_t fusion_idx;           /* index into fusions[]                                        */
    int32_t read_off, read_len;   /* into read_bytes                                             */
    int32_t frag;                 /* ReadID.fragmentIndex                                        */
    uint8_t read_end;             /* ReadID.readEnd (0/1)                                        */
    uint8_t revcomp;              /* SplitReadInfo.revComp                                       */
    uint8_t pad_[2];
} dsa_pair;

/* One output line of dosplitalign (tools/SplitAlignment.cpp:305-317), plus the index of the
 * candidate it belongs to, so that a caller that reordered its candidates (e.g. grouped them by
 * fusion) can restore its own output order. */
typedef struct dsa_record {
    int32_t fusion_id, frag, read_end, revcomp;
    int32_t ref_first, ref_second;      /* refSplit  */
    int32_t read_first, read_second;    /* readSplit */
    int32_t score;                      /* min(score1, score2) */
    int32_t pair_idx;                   /* index into pairs[] of the batch (n_pairs < 2^31) */
} dsa_record;

typedef struct dsa_limits {
    int32_t max_read_len;         /* longest read the DP kernels accept                          */
    int32_t max_ref_len;          /* longest reference window                                    */
    int32_t tile_cols;            /* DP tile width (reference columns per register tile)         */
} dsa_limits;

/* Timings of the most recent dsa_run / dsa_align_batch, measured with HIP events on the ctx stream. */
typedef struct dsa_timing {
    float   pack_ms;              /* reference bytes -> codes (rows are packed inside the fill)  */
    float   fill_ms;              /* the DP fill kernel(s), with combine and table replay in their tail */
    float   finish_ms;            /* left-over tile replay + emit kernels                        */
    float   total_ms;             /* elapsed host time of dsa_run (stage times overlap between slices) */
    int32_t fill_launches;        /* number of DP fill launches in fill_ms                       */
    int32_t n_generic_tasks;      /* of n_replay_tasks: tiles re-run by the generic replay kernel   */
    int64_t cells;                /* DP cells of the batch: sum over pairs of (Lref0+1 + Lref1+1)*(Lread+1), exact on every path */
    int64_t n_records;
    int64_t n_replay_tasks;       /* tiles re-run to enumerate tied columns                      */
    float   plan_ms;              /* the sweep planning that preceded this run (dsa_upload's or dsa_plan's) */
    float   pad_;
} dsa_timing;

/* ---- context ---------------------------------------------------------------------------- */
int  dsa_create(dsa_ctx** out, int device);     /* device = HIP ordinal; fails (DSA_E_DEVICE) without a GPU */
void dsa_destroy(dsa_ctx* ctx);
int  dsa_get_limits(const dsa_ctx* ctx, dsa_limits* out);
const char* dsa_last_error(const dsa_ctx* ctx); /* human-readable text for the last failure      */
const char* dsa_version(void);                 /* "... src <hash of the sources and of the flags actually used>" */
/* "sched=<iterative-ilp|default> <compiler flags>": the instruction scheduler the split-read kernels were built with (a
 * compiler that lacks the flag still builds the library, with a slower fill kernel and another source hash). */
const char* dsa_build_flags(void);
/* HIP devices this process sees (0 without a GPU), and the ordinal a tool should use: DEFUSE_GPU if set, else
 * pid mod device count — the pipeline starts up to --parallel independent tool processes (scripts/defuse_run.pl:33,285;
 * SURVEY 8(b)), which spreads them over the GPUs of a node without any of them assuming it owns one. */
int dsa_device_count(void);
int dsa_pick_device(void);
/* The rule behind dsa_pick_device for n devices: counting from pid mod n, the first device whose lock file
 * ($DEFUSE_GPU_LOCK_DIR or /tmp)/defuse_gpu.<d>.lock this process can take with flock (kept until the process ends);
 * pid mod n when all are taken.  Up to n concurrent tool processes therefore use n different GPUs. */
int dsa_pick_device_among(int n_devices);
/* The sweep planning (order of the fusions and of a fusion's pairs, per-pair score bounds) is a speed heuristic: any order
 * and any true lower bound give the same records.  Its parts can be switched off per context — for measurements of what
 * each is worth and for tests; a new context starts from the environment (DEFUSE_DSA_NO_REORDER / _NO_RANK / _NO_TIGHTEN /
 * _NO_LPT = 1).  Takes effect with the next dsa_upload / dsa_plan. */
#define DSA_PLAN_NO_REORDER  1u   /* the caller's pair order, no bounds                              */
#define DSA_PLAN_NO_RANK     2u   /* no ordering of a fusion's pairs by their estimated read split   */
#define DSA_PLAN_NO_TIGHTEN  4u   /* no per-pair score bound: pruning against minScore only          */
#define DSA_PLAN_NO_LPT      8u   /* no cost ranking of the fusions inside a size class              */
int dsa_set_plan_options(dsa_ctx* ctx, unsigned flags);
/* Upper bound of the per-slice scratch planes of one pipeline lane (default 16 GiB, or DEFUSE_DSA_SCRATCH_MB at
 * dsa_create): a caller that keeps several uploads resident side by side (one ctx each) sizes them with this. */
int dsa_set_scratch_budget(dsa_ctx* ctx, int64_t bytes);
/* ctx gives up its own pipeline lanes (streams, events, scratch planes — and a stream given with dsa_set_stream: lane 0's
 * stream is one per set of lanes, whoever of the sharers sets it) and uses donor's from now on; both must be on
 * the same device and must not run at the same time (dsa_run is synchronous, so a caller that runs its resident
 * uploads one after the other may let all of them share one set).  The lanes live until the last sharer is destroyed. */
int dsa_share_scratch(dsa_ctx* ctx, dsa_ctx* donor);

/* ---- one-shot: host buffers in, host records out ----------------------------------------- */
/* Records are ordered by pair index, then in the reference's emission order (read split a
 * ascending, ref1 column ascending, ref2 column ascending), after refSplit de-duplication. */
int dsa_align_batch(dsa_ctx* ctx,
                    const uint8_t* ref_bytes, int64_t ref_bytes_len,
                    const dsa_fusion* fusions, int32_t n_fusions,
                    const uint8_t* read_bytes, int64_t read_bytes_len,
                    const dsa_pair* pairs, int64_t n_pairs,
                    dsa_record* out, int64_t out_cap, int64_t* out_n);

/* ---- staged: keep the batch resident in HBM, run it repeatedly (bench, multi-pass callers) - */
/* pairs must be grouped by fusion_idx (all pairs of one fusion contiguous) for best throughput;
 * correctness does not depend on it. */
int dsa_upload(dsa_ctx* ctx,
               const uint8_t* ref_bytes, int64_t ref_bytes_len,
               const dsa_fusion* fusions, int32_t n_fusions,
               const uint8_t* read_bytes, int64_t read_bytes_len,
               const dsa_pair* pairs, int64_t n_pairs);
/* Plans the sweep of the resident upload once more — everything dsa_upload does per candidate after its copies (sweep
 * order of the fusions and of the pairs inside a fusion, per-pair score bounds).  A caller that times the path per
 * batch calls dsa_plan + dsa_run per step, so that all work the reference does per candidate inside the loop of
 * SplitReadRealigner::DoAlignment (tools/SplitAlignment.cpp:266-303) is inside its clock (bench.py does). */
int dsa_plan(dsa_ctx* ctx);
int dsa_run(dsa_ctx* ctx, int64_t* out_n);                       /* all kernels, records stay on device */
int dsa_download(dsa_ctx* ctx, dsa_record* out, int64_t out_cap, int64_t* out_n);
/* Same as dsa_download, but `out_device` is device memory of the ctx's GPU (room for out_cap records): the
 * records never visit the host.  This is what the multi-GPU gather sends over RCCL. */
int dsa_copy_records_device(dsa_ctx* ctx, void* out_device, int64_t out_cap, int64_t* out_n);
int dsa_get_timing(const dsa_ctx* ctx, dsa_timing* out);
/* Use an existing HIP stream (e.g. torch's current stream) instead of the ctx's own; pass the
 * hipStream_t as an opaque pointer, NULL restores the private stream. */
int dsa_set_stream(dsa_ctx* ctx, void* hip_stream);
int dsa_synchronize(dsa_ctx* ctx);

/* ---- streaming: what a tool that aligns one batch after the other calls -------------------------------------- */
/* SplitReadRealigner::DoAlignment (tools/SplitAlignment.cpp:266-303) walks its candidates once; a binding that cuts them
 * into batches keeps up to `depth` of them in flight: while batch k is planned and aligned, the host buffers of batch k+1
 * are copied in and the records of batch k-1 are copied out (three HIP streams, one worker thread).  Batches are collected
 * in the order of their submission.  The host buffers given to dsa_stream_submit (inputs AND `out`) must stay valid and
 * untouched until the matching dsa_stream_collect returned; buffers from dsa_host_alloc (pinned) make the copies
 * asynchronous, any other host memory works but is staged by the runtime. */
typedef struct dsa_stream dsa_stream;
int  dsa_stream_create(dsa_stream** out, int device, int depth);          /* depth 1..8 batches in flight */
void dsa_stream_destroy(dsa_stream* s);                                   /* finishes what was submitted  */
int  dsa_stream_submit(dsa_stream* s,
                       const uint8_t* ref_bytes, int64_t ref_bytes_len,
                       const dsa_fusion* fusions, int32_t n_fusions,
                       const uint8_t* read_bytes, int64_t read_bytes_len,
                       const dsa_pair* pairs, int64_t n_pairs,
                       dsa_record* out, int64_t out_cap);                 /* DSA_E_BUSY: depth batches in flight */
/* Blocks until the oldest batch is done and its records are in the `out` of its submit; *out_n = their number.
 * DSA_E_CAPACITY: they did not fit (*out_n = the number) — the batch stays the oldest until dsa_stream_recollect copied
 * them into a larger buffer. */
int  dsa_stream_collect(dsa_stream* s, int64_t* out_n);
int  dsa_stream_recollect(dsa_stream* s, dsa_record* out, int64_t out_cap, int64_t* out_n);
const char* dsa_stream_last_error(const dsa_stream* s);
/* pinned host memory for the buffers of a stream (NULL when it cannot be had) */
void* dsa_host_alloc(size_t bytes);
void  dsa_host_free(void* p);
/* pins memory the caller already owns (e.g. a mapping it shares with another process) for the same purpose; pinning costs
 * about 0.2 ms per MiB on an MI355X host, a copy from unpinned memory about 20 % more time than from pinned memory
 * (profiles/r04/tools/fixed_costs.txt) — it pays for buffers that are reused many times */
int   dsa_host_register(void* p, size_t bytes);
int   dsa_host_unregister(void* p);

#ifdef __cplusplus
}
#endif
#endif /* DEFUSE_DSA_H_ */
