/*
 * dsa_abi_double.c — TEST DOUBLE of the streaming entry points of include/defuse_dsa.h.  TEST INFRASTRUCTURE ONLY.
 *
 * The host logic of bin/dosplitalign (parsing, de-duplication, batch cutting, the counting sort by fusion, the worker process and
 * its slots, formatting, writing) has to be testable where there is no GPU: the `-m "not gpu"` tests and the sanitizer builds
 * point DEFUSE_DSA_LIB at this library, which answers dsa_stream_* synchronously with the CPU oracle's records
 * (oracle/dsa_oracle.c, compiled in).  It is built by tests/test_tool_host.py into tests/shim/_build/ and nothing outside tests/
 * knows it exists; the product library is defuse_amd/libdefuse_dsa.so, which has no CPU path at all.
 */
#include <stdlib.h>
#include <string.h>

#include "../../oracle/dsa_oracle.c"

#define MAXDEPTH 8
struct dsa_stream {
    int depth;
    unsigned long n_submitted, n_collected;
    struct { dsa_record* full; int64_t n; dsa_record* out; int64_t out_cap; } job[MAXDEPTH];
};

int dsa_pick_device(void) { return 0; }

int dsa_stream_create(dsa_stream** out, int device, int depth)
{
    (void)device;
    if (!out || depth < 1 || depth > MAXDEPTH) return DSA_E_ARG;
    if (getenv("DSA_DOUBLE_NO_DEVICE")) return DSA_E_DEVICE;
    dsa_stream* s = (dsa_stream*)calloc(1, sizeof *s);
    s->depth = depth;
    *out = s;
    return DSA_OK;
}

void dsa_stream_destroy(dsa_stream* s) { free(s); }

int dsa_stream_submit(dsa_stream* s, const uint8_t* ref_bytes, int64_t ref_bytes_len, const dsa_fusion* fusions, int32_t n_fusions,
                      const uint8_t* read_bytes, int64_t read_bytes_len, const dsa_pair* pairs, int64_t n_pairs, dsa_record* out, int64_t out_cap)
{
    if (s->n_submitted - s->n_collected >= (unsigned long)s->depth) return DSA_E_BUSY;
    const int k = (int)(s->n_submitted % (unsigned long)s->depth);
    for (int64_t p = 0; p < n_pairs; p++) {             /* what the real library validates */
        if (pairs[p].fusion_idx < 0 || pairs[p].fusion_idx >= n_fusions) return DSA_E_ARG;
        if (pairs[p].read_off < 0 || (int64_t)pairs[p].read_off + pairs[p].read_len > read_bytes_len) return DSA_E_ARG;
    }
    for (int32_t f = 0; f < n_fusions; f++)
        if ((int64_t)fusions[f].ref0_off + fusions[f].ref0_len > ref_bytes_len || (int64_t)fusions[f].ref1_off + fusions[f].ref1_len > ref_bytes_len) return DSA_E_ARG;
    int64_t n = 0;
    s->job[k].full = NULL;
    s->job[k].out = out;
    s->job[k].out_cap = out_cap;
    if (getenv("DSA_DOUBLE_FAKE")) {
        /* host-side profiling without a GPU (profiles/microbench/tool_host_profile.sh): no alignment at all, every pair gets one
         * made-up record (two for every third pair) so that formatting and writing have their usual amount of work */
        for (int64_t p = 0; p < n_pairs; p++)
            for (int r = 0; r < (p % 3 == 0 ? 2 : 1); r++) {
                if (n < out_cap) {
                    dsa_record* o = &out[n];
                    o->fusion_id = fusions[pairs[p].fusion_idx].fusion_id; o->frag = pairs[p].frag; o->read_end = pairs[p].read_end; o->revcomp = pairs[p].revcomp;
                    o->ref_first = 200 + r; o->ref_second = 150; o->read_first = 37; o->read_second = 38; o->score = 70; o->pair_idx = (int32_t)p;
                }
                n++;
            }
        s->job[k].n = n < out_cap ? n : out_cap;
        s->n_submitted++;
        return DSA_OK;
    }
    int rc = ora_align_batch(ref_bytes, ref_bytes_len, fusions, n_fusions, read_bytes, read_bytes_len, pairs, n_pairs, out, out_cap, &n);
    if (rc == DSA_E_CAPACITY) {
        s->job[k].full = (dsa_record*)malloc(sizeof(dsa_record) * (size_t)n);
        ora_align_batch(ref_bytes, ref_bytes_len, fusions, n_fusions, read_bytes, read_bytes_len, pairs, n_pairs, s->job[k].full, n, &n);
    }
    s->job[k].n = n;
    s->n_submitted++;
    return DSA_OK;
}

int dsa_stream_collect(dsa_stream* s, int64_t* out_n)
{
    if (s->n_collected >= s->n_submitted) return DSA_E_ARG;
    const int k = (int)(s->n_collected % (unsigned long)s->depth);
    if (out_n) *out_n = s->job[k].n;
    if (s->job[k].full) return DSA_E_CAPACITY;
    s->n_collected++;
    return DSA_OK;
}

int dsa_stream_recollect(dsa_stream* s, dsa_record* out, int64_t out_cap, int64_t* out_n)
{
    if (s->n_collected >= s->n_submitted) return DSA_E_ARG;
    const int k = (int)(s->n_collected % (unsigned long)s->depth);
    if (out_n) *out_n = s->job[k].n;
    if (s->job[k].n > out_cap) return DSA_E_CAPACITY;
    memcpy(out, s->job[k].full, sizeof(dsa_record) * (size_t)s->job[k].n);
    free(s->job[k].full);
    s->job[k].full = NULL;
    s->n_collected++;
    return DSA_OK;
}

const char* dsa_stream_last_error(const dsa_stream* s) { (void)s; return "test double"; }
int dsa_host_register(void* p, size_t bytes) { (void)p; (void)bytes; return DSA_OK; }
