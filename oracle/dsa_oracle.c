/*
 * dsa_oracle.c — CPU restatement of deFuse's split-read alignment.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (defuse_amd/csrc, the tool binaries) never links, imports or executes it.
 *
 * It follows the reference's algorithm literally — full (Lref+1)x(Lread+1) matrices, the same
 * loop order and the same scan order — so that it can serve as the checker for the HIP path:
 *
 *   ora_fill_matrix      tools/SplitReadAligner.cpp:24-75   (FillMatrix; storage tools/Matrix.h:63-66)
 *   ora_find_max_row     tools/SplitReadAligner.cpp:91-122  (FindMaxRowEntry, both overloads)
 *   ora_get_alignments   tools/SplitReadAligner.cpp:156-298 (GetAlignments, forceSplit=true,
 *                                                            firstOnly=false, backTrace=false)
 *   ora_task_align       tools/SplitAlignment.cpp:371-400   (minScore, refSplit dedup, min score)
 *   ora_align_batch      tools/SplitAlignment.cpp:266-303   (the per-candidate loop body only)
 *
 * Parity status: UNPINNED by the rules of this build.  The one known-answer vector it reproduces
 * (tests/golden/smoke, recorded in SURVEY.md Appendix A) was produced at survey time by the reference's sources compiled
 * against stand-in Boost headers, which does not count as the reference; the reference holds no test or golden vector for
 * this path and cannot be built in this image (Boost headers absent).  What stands behind it is the line-by-line reading
 * against the cited ranges and the agreement of two independent formulations (this file and the HIP kernels) on every test.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/defuse_dsa.h"

typedef struct ora_split {          /* SplitReadAlignment, tools/SplitReadAligner.h:21-30 */
    int ref_first, ref_second;
    int read_first, read_second;
    int score, score1, score2;
} ora_split;

/* Matrix<int> element (i,j) lives at j*length + i, tools/Matrix.h:63-66 */
#define M(mat, len, i, j) ((mat)[(size_t)(j) * (size_t)(len) + (size_t)(i)])

/* tools/SplitReadAligner.cpp:24-75 with mEndGaps=false; the backtrace matrix is not restated
 * (the pipeline never reads it: GetAlignments is called with backTrace=false). */
void ora_fill_matrix(const uint8_t* seq1, int n1, const uint8_t* seq2, int n2, int* matrix)
{
    const int length = n1 + 1, height = n2 + 1;
    for (int i = 0; i < length; i++) {
        for (int j = 0; j < height; j++) {
            if (j == 0) {
                M(matrix, length, i, j) = 0;
            } else if (i == 0) {
                M(matrix, length, i, j) = M(matrix, length, i, j - 1) + DSA_GAP;
            } else {
                int match = M(matrix, length, i - 1, j - 1) + ((seq1[i - 1] == seq2[j - 1]) ? DSA_MATCH : DSA_MISMATCH);
                int gap_ref = M(matrix, length, i - 1, j) + DSA_GAP;
                int gap_read = M(matrix, length, i, j - 1) + DSA_GAP;
                int mx = match;
                if (gap_ref > mx) mx = gap_ref;
                if (gap_read > mx) mx = gap_read;
                M(matrix, length, i, j) = mx;
            }
        }
    }
}

/* tools/SplitReadAligner.cpp:104-122; cols may be NULL (first overload, :91-102). Returns #cols. */
static int ora_find_max_row(const int* matrix, int length, int j, int min_accepted, int* max_out, int* cols)
{
    int mx = 0, n = 0;
    for (int i = 0; i < length; i++) {
        int v = M(matrix, length, i, j);
        if (v >= min_accepted && v > mx) {
            mx = v;
            n = 0;
            if (cols) cols[n] = i;
            n++;
        } else if (v >= min_accepted && v == mx) {
            if (cols) cols[n] = i;
            n++;
        }
    }
    *max_out = mx;
    return n;
}

/* tools/SplitReadAligner.cpp:77-89 + :156-298.  Returns the number of alignments the reference
 * would push (may exceed cap; only the first cap are stored). */
long ora_get_alignments(const uint8_t* read, int lq, const uint8_t* ref1, int l1,
                        const uint8_t* ref2, int l2, int min_score, ora_split* out, long cap)
{
    uint8_t* ref2r = (uint8_t*)malloc((size_t)l2 + 1);
    uint8_t* readr = (uint8_t*)malloc((size_t)lq + 1);
    for (int i = 0; i < l2; i++) ref2r[i] = ref2[l2 - 1 - i];
    for (int j = 0; j < lq; j++) readr[j] = read[lq - 1 - j];

    int* m1 = (int*)malloc(sizeof(int) * (size_t)(l1 + 1) * (size_t)(lq + 1));
    int* m2 = (int*)malloc(sizeof(int) * (size_t)(l2 + 1) * (size_t)(lq + 1));
    ora_fill_matrix(ref1, l1, read, lq, m1);
    ora_fill_matrix(ref2r, l2, readr, lq, m2);

    int max_score = 0;
    int* a_max = (int*)malloc(sizeof(int) * (size_t)(lq + 2));
    int n_a = 0;
    for (int a = 0; a <= lq; a++) {
        int b = lq - a;
        int r1 = 0, r2 = 0;
        ora_find_max_row(m1, l1 + 1, a, DSA_MIN_SPLIT, &r1, NULL);
        ora_find_max_row(m2, l2 + 1, b, DSA_MIN_SPLIT, &r2, NULL);
        int s = r1 + r2;
        if (s >= min_score && s > max_score) {
            max_score = s;
            n_a = 0;
            a_max[n_a++] = a;
        } else if (s >= min_score && s == max_score) {
            a_max[n_a++] = a;
        }
    }

    long n_out = 0;
    if (max_score != 0) {
        int* c1 = (int*)malloc(sizeof(int) * (size_t)(l1 + 1));
        int* c2 = (int*)malloc(sizeof(int) * (size_t)(l2 + 1));
        for (int k = 0; k < n_a; k++) {
            int a = a_max[k], b = lq - a;
            int r1 = 0, r2 = 0;
            int n1 = ora_find_max_row(m1, l1 + 1, a, DSA_MIN_SPLIT, &r1, c1);
            int n2 = ora_find_max_row(m2, l2 + 1, b, DSA_MIN_SPLIT, &r2, c2);
            for (int x = 0; x < n1; x++) {
                for (int y = 0; y < n2; y++) {
                    if (n_out < cap) {
                        ora_split* s = &out[n_out];
                        s->ref_first = c1[x];
                        s->ref_second = l2 - c2[y] - 1;
                        s->read_first = a;
                        s->read_second = b;
                        s->score = max_score;
                        s->score1 = M(m1, l1 + 1, c1[x], a);
                        s->score2 = M(m2, l2 + 1, c2[y], b);
                    }
                    n_out++;
                }
            }
        }
        free(c1);
        free(c2);
    }
    free(a_max);
    free(m1);
    free(m2);
    free(ref2r);
    free(readr);
    return n_out;
}

/* minScore exactly as written at tools/SplitAlignment.cpp:379:
 *   (int)((float)readSeq.length() * (float)matchScore * 0.90)
 * float*float is rounded to float, then promoted to double for the multiplication by 0.90. */
int ora_min_score(int lq)
{
    float f = (float)lq * (float)DSA_MATCH;
    return (int)((double)f * 0.90);
}

/* tools/SplitAlignment.cpp:371-400: Align + GetAlignments + de-duplicate on refSplit (first kept)
 * + score=min(score1,score2).  Returns the number of records (may exceed cap). */
long ora_task_align(const uint8_t* read, int lq, const uint8_t* ref1, int l1,
                    const uint8_t* ref2, int l2, ora_split* out, long cap)
{
    long raw_cap = 1024;
    ora_split* raw = (ora_split*)malloc(sizeof(ora_split) * (size_t)raw_cap);
    long n_raw = ora_get_alignments(read, lq, ref1, l1, ref2, l2, ora_min_score(lq), raw, raw_cap);
    if (n_raw > raw_cap) {
        raw_cap = n_raw;
        raw = (ora_split*)realloc(raw, sizeof(ora_split) * (size_t)raw_cap);
        n_raw = ora_get_alignments(read, lq, ref1, l1, ref2, l2, ora_min_score(lq), raw, raw_cap);
    }
    long n_out = 0;
    for (long k = 0; k < n_raw; k++) {
        int dup = 0;
        for (long p = 0; p < k && !dup; p++)
            dup = (raw[p].ref_first == raw[k].ref_first && raw[p].ref_second == raw[k].ref_second);
        if (dup) continue;
        if (n_out < cap) {
            out[n_out] = raw[k];
            out[n_out].score = raw[k].score1 < raw[k].score2 ? raw[k].score1 : raw[k].score2;
        }
        n_out++;
    }
    free(raw);
    return n_out;
}

/* Same contract as dsa_align_batch (include/defuse_dsa.h) minus the context. */
int ora_align_batch(const uint8_t* ref_bytes, int64_t ref_bytes_len,
                    const dsa_fusion* fusions, int32_t n_fusions,
                    const uint8_t* read_bytes, int64_t read_bytes_len,
                    const dsa_pair* pairs, int64_t n_pairs,
                    dsa_record* out, int64_t out_cap, int64_t* out_n)
{
    (void)ref_bytes_len; (void)read_bytes_len; (void)n_fusions;
    int64_t n = 0;
    long cap = 4096;
    ora_split* tmp = (ora_split*)malloc(sizeof(ora_split) * (size_t)cap);
    for (int64_t p = 0; p < n_pairs; p++) {
        const dsa_pair* pr = &pairs[p];
        const dsa_fusion* fu = &fusions[pr->fusion_idx];
        long k = ora_task_align(read_bytes + pr->read_off, pr->read_len,
                                ref_bytes + fu->ref0_off, fu->ref0_len,
                                ref_bytes + fu->ref1_off, fu->ref1_len, tmp, cap);
        if (k > cap) {
            cap = k;
            tmp = (ora_split*)realloc(tmp, sizeof(ora_split) * (size_t)cap);
            k = ora_task_align(read_bytes + pr->read_off, pr->read_len,
                               ref_bytes + fu->ref0_off, fu->ref0_len,
                               ref_bytes + fu->ref1_off, fu->ref1_len, tmp, cap);
        }
        for (long q = 0; q < k; q++) {
            if (n < out_cap) {
                dsa_record* r = &out[n];
                r->fusion_id = fu->fusion_id;
                r->frag = pr->frag;
                r->read_end = pr->read_end;
                r->revcomp = pr->revcomp;
                r->ref_first = tmp[q].ref_first;
                r->ref_second = tmp[q].ref_second;
                r->read_first = tmp[q].read_first;
                r->read_second = tmp[q].read_second;
                r->score = tmp[q].score;
                r->pair_idx = (int32_t)p;
            }
            n++;
        }
    }
    free(tmp);
    *out_n = n;
    return n > out_cap ? DSA_E_CAPACITY : DSA_OK;
}

/* Row maxima of one matrix as the kernel tests want them: for each read-prefix length j the
 * plain maximum over all columns (no min-accepted filter).  out has lq+1 entries. */
void ora_row_maxima(const uint8_t* ref, int lr, const uint8_t* read, int lq, int* out)
{
    int* m = (int*)malloc(sizeof(int) * (size_t)(lr + 1) * (size_t)(lq + 1));
    ora_fill_matrix(ref, lr, read, lq, m);
    for (int j = 0; j <= lq; j++) {
        int mx = M(m, lr + 1, 0, j);
        for (int i = 1; i <= lr; i++)
            if (M(m, lr + 1, i, j) > mx) mx = M(m, lr + 1, i, j);
        out[j] = mx;
    }
    free(m);
}

/* ------------------------------------------------------------------------------------------------
 * localalign.  Restates SimpleAligner::Align, tools/SimpleAligner.cpp:24-64: the
 * same full-matrix sweep (i outer over the reference, j inner over the sequence), row 0 (j == 0) all
 * zero, column 0 (i == 0) j*gap, exact byte comparison, the running maximum taken over the cells with
 * i >= 1 and j >= 1 only and started at 0.  Parity unpinned: the reference holds no test or golden
 * output for this tool and its sources need Boost headers (Common.h) that this image lacks; the tests
 * cross-check this restatement against an independently written recursion instead.
 * ---------------------------------------------------------------------------------------------- */
int ora_simple_align(int match, int mismatch, int gap, const uint8_t* reference, int lr, const uint8_t* sequence, int ls)
{
    const int length = lr + 1, height = ls + 1;
    int* m = (int*)malloc(sizeof(int) * (size_t)length * (size_t)height);
    int overall = 0;
    for (int i = 0; i < length; i++) {
        for (int j = 0; j < height; j++) {
            if (j == 0) {
                M(m, length, i, j) = 0;
            } else if (i == 0) {
                M(m, length, i, j) = M(m, length, i, j - 1) + gap;
            } else {
                int diag = M(m, length, i - 1, j - 1) + (reference[i - 1] == sequence[j - 1] ? match : mismatch);
                int gap_ref = M(m, length, i - 1, j) + gap;
                int gap_read = M(m, length, i, j - 1) + gap;
                int best = gap_ref > gap_read ? gap_ref : gap_read;
                if (diag > best) best = diag;
                if (best > overall) overall = best;
                M(m, length, i, j) = best;
            }
        }
    }
    free(m);
    return overall;
}
