/*
 * mpe_oracle.c — CPU restatement (C, glibc libm as the reference uses) of deFuse's mate-pair EM
 * clustering.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench-side
 * baselines may load this library; the product never links, imports or executes it.
 *
 *   ora_kmns / optra / qtran     tools/asa136.C:13-758 (AS 136, Hartigan-Wong; column-major a[i+j*m])
 *   ora_cdf_inverse              tools/asa241.C:424-563 (AS 241, PPND16)
 *   ora_normalpdf                tools/Common.cpp:61-69
 *   log_likelihood               tools/MatePairEM.cpp:96-137
 *   update_responsibilities      tools/MatePairEM.cpp:139-181
 *   max_likelihood               tools/MatePairEM.cpp:192-325
 *   select_kkz                   tools/MatePairEM.cpp:327-386
 *   expectation_maximization     tools/MatePairEM.cpp:388-494
 *   do_clustering                tools/MatePairEM.cpp:540-636 (sort ranks are inputs: key desc, index asc)
 *
 * Same problem layout as include/defuse_mpe.h so that a test can hand identical arrays to both.
 * It is the second, faster restatement beside oracle/clustermatepairs_oracle.py (tests compare the two);
 * when oracle/_ref/libasa_ref.so exists (the reference's own asa136.C/asa241.C compiled as they lie, see
 * oracle/Makefile) tests/test_asa_ref.py pins ora_kmns and ora_cdf_inverse against it.
 *
 * Parity status of the EM itself: UNPINNED (MatePairEM.cpp includes Common.h, which needs Boost; the
 * reference holds no vector for it).
 *
 * Besides the results it reports how close a run came to the decisions a last-ulp difference of
 * exp/log could flip (ora_mpe_diag): the GPU uses ocml's exp/log, the reference glibc's.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KMAX 10
#define R8_HUGE 1.0e30

typedef struct ora_mpe_params {      /* = mpe_params of include/defuse_mpe.h */
    double  fragment_mean, fragment_stddev, min_probability;
    int32_t min_cluster_size, pad_;
} ora_mpe_params;

typedef struct ora_mpe_diag {
    double  min_prob_margin;      /* min |prob - minProb| / minProb over all membership tests (MatePairEM.cpp:622-627) */
    double  min_tol_margin;       /* min ||dLL| - tol| / tol over all convergence tests (:473) */
    double  min_bic_gap;          /* min (BIC_K - BIC_best) / |BIC_best| over the K that lost (:599-606) */
    double  min_deriv_margin;     /* min nonzero |partial| / (|RXYU/var| + |NK (cx+cy)/var| + lambda |cs|) over the breakpoints visited (:281-295), M steps after a fit's first */
    double  min_merge_margin;     /* min |SX-SY| / max(SX,SY) over the prefix-sum comparisons that were NOT exactly equal (:224), M steps after a fit's first */
    int64_t nk_zero;              /* MaxLikelihood returned false (NK == 0, :277) */
    int64_t nk_zero_first_iter;   /* ... in the first iteration of a fit: the only way stale mA/mB could be read */
    int64_t ll_underflow;         /* LogLikelihood returned -DBL_MAX (:127-131) */
    int64_t kkz_fail;             /* SelectKKZ returned false (:375-378) */
    int64_t em_iterations;
    int64_t merge_equal;          /* prefix-sum comparisons that were exactly equal */
    int64_t all_k_failed;         /* problems where every K failed and the refit ran at K = 1 */
    double  min_deriv_margin_first;   /* the same two figures over the FIRST M step of every fit and all M steps of K = 1 fits: their responsibilities are 0/1 (k-means) or 1/K, no exp/log has */
    double  min_merge_margin_first;   /* been evaluated yet, so every implementation computes these decisions from bit-identical operands */
    int64_t deriv_zero;           /* breakpoints where the derivative estimate was exactly 0 (identical in any IEEE evaluation of the same operands) */
    int64_t iters_by_k[12];       /* [K] = EM iterations of the fit with K components (K = 1..10), [11] = of the refit of the chosen K, [0] = the chosen K */
    double  ll_by_k[12];          /* [K] = log-likelihood the fit with K components ended with (0 when it gave none) — per problem only, not merged */
} ora_mpe_diag;

/* ------------------------------------------------------------------------------------------------
 * AS 241 and normalpdf
 * ---------------------------------------------------------------------------------------------- */
static double poly8(const double* a, double x)          /* r8poly_value, asa241.C:565-600: Horner from the top */
{
    double v = 0.0;
    for (int i = 7; i >= 0; i--) v = v * x + a[i];
    return v;
}

double ora_cdf_inverse(double p)
{
    static const double a[8] = {3.3871328727963666080, 1.3314166789178437745e+2, 1.9715909503065514427e+3, 1.3731693765509461125e+4,
                                4.5921953931549871457e+4, 6.7265770927008700853e+4, 3.3430575583588128105e+4, 2.5090809287301226727e+3};
    static const double b[8] = {1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3,
                                2.1213794301586595867e+4, 3.9307895800092710610e+4, 2.8729085735721942674e+4, 5.2264952788528545610e+3};
    static const double c[8] = {1.42343711074968357734, 4.63033784615654529590, 5.76949722146069140550, 3.64784832476320460504,
                                1.27045825245236838258, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4};
    static const double d[8] = {1.0, 2.05319162663775882187, 1.67638483018380384940, 6.89767334985100004550e-1,
                                1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9};
    static const double e[8] = {6.65790464350110377720, 5.46378491116411436990, 1.78482653991729133580, 2.96560571828504891230e-1,
                                2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7};
    static const double f[8] = {1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
                                7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15};
    if (p <= 0.0) return -R8_HUGE;
    if (1.0 <= p) return R8_HUGE;
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        return q * poly8(a, r) / poly8(b, r);
    }
    double r = (q < 0.0) ? p : 1.0 - p;
    if (r <= 0.0) exit(1);
    r = sqrt(-log(r));
    double value;
    if (r <= 5.0) {
        r = r - 1.6;
        value = poly8(c, r) / poly8(d, r);
    } else {
        r = r - 5.0;
        value = poly8(e, r) / poly8(f, r);
    }
    return (q < 0.0) ? -value : value;
}

double ora_normalpdf(double x, double mu, double sigma)
{
    const double coeff = 1.0 / (sigma * sqrt(2 * M_PI));
    const double dist = (x - mu) / sigma;
    return coeff * exp(-0.5 * dist * dist);
}

/* mMinProbability, tools/MatePairEM.cpp:49-50 */
double ora_min_probability(double stddev, double precision)
{
    const double x = -stddev * ora_cdf_inverse((1 - precision) / 2);
    return ora_normalpdf(x, 0, stddev);
}

/* ------------------------------------------------------------------------------------------------
 * AS 136.  Indices i, l are kept 1-based where the algorithm compares them with live[] / ncp[].
 * ---------------------------------------------------------------------------------------------- */
#define A_(i, j) a[(i) - 1 + ((j) - 1) * m]
#define C_(l, j) c[(l) - 1 + ((j) - 1) * k]

static double sqdist(const double* a, int m, int n, const double* c, int k, int i, int l)
{
    double s = 0.0;
    for (int j = 1; j <= n; j++) {
        const double df = A_(i, j) - C_(l, j);
        s = s + df * df;
    }
    return s;
}

static void transfer(const double* a, int m, int n, double* c, int k, int* ic1, int* ic2, int* nc, double* an1, double* an2,
                     int i, int l1, int l2)
{
    const double al1 = (double)nc[l1 - 1], alw = al1 - 1.0, al2 = (double)nc[l2 - 1], alt = al2 + 1.0;
    for (int j = 1; j <= n; j++) {
        C_(l1, j) = (C_(l1, j) * al1 - A_(i, j)) / alw;
        C_(l2, j) = (C_(l2, j) * al2 + A_(i, j)) / alt;
    }
    nc[l1 - 1] -= 1;
    nc[l2 - 1] += 1;
    an2[l1 - 1] = alw / al1;
    an1[l1 - 1] = (1.0 < alw) ? alw / (alw - 1.0) : R8_HUGE;
    an1[l2 - 1] = alt / al2;
    an2[l2 - 1] = alt / (alt + 1.0);
    ic1[i - 1] = l2;
    ic2[i - 1] = l1;
}

static void optra(const double* a, int m, int n, double* c, int k, int* ic1, int* ic2, int* nc, double* an1, double* an2,
                  int* ncp, double* d, int* itran, int* live, int* indx)
{
    for (int l = 1; l <= k; l++)
        if (itran[l - 1] == 1) live[l - 1] = m + 1;
    for (int i = 1; i <= m; i++) {
        *indx = *indx + 1;
        const int l1 = ic1[i - 1];
        int l2 = ic2[i - 1];
        const int ll = l2;
        if (1 < nc[l1 - 1]) {
            if (ncp[l1 - 1] != 0) d[i - 1] = sqdist(a, m, n, c, k, i, l1) * an1[l1 - 1];
            double r2 = sqdist(a, m, n, c, k, i, l2) * an2[l2 - 1];
            for (int l = 1; l <= k; l++) {
                if ((i < live[l1 - 1] || i < live[l2 - 1]) && l != l1 && l != ll) {
                    const double rr = r2 / an2[l - 1];
                    const double dc = sqdist(a, m, n, c, k, i, l);
                    if (dc < rr) {
                        r2 = dc * an2[l - 1];
                        l2 = l;
                    }
                }
            }
            if (d[i - 1] <= r2) {
                ic2[i - 1] = l2;
            } else {
                *indx = 0;
                live[l1 - 1] = m + i;
                live[l2 - 1] = m + i;
                ncp[l1 - 1] = i;
                ncp[l2 - 1] = i;
                transfer(a, m, n, c, k, ic1, ic2, nc, an1, an2, i, l1, l2);
            }
        }
        if (*indx == m) return;
    }
    for (int l = 1; l <= k; l++) {
        itran[l - 1] = 0;
        live[l - 1] = live[l - 1] - m;
    }
}

static void qtran(const double* a, int m, int n, double* c, int k, int* ic1, int* ic2, int* nc, double* an1, double* an2,
                  int* ncp, double* d, int* itran, int* indx)
{
    int icoun = 0, istep = 0;
    for (;;) {
        for (int i = 1; i <= m; i++) {
            icoun = icoun + 1;
            istep = istep + 1;
            const int l1 = ic1[i - 1], l2 = ic2[i - 1];
            if (1 < nc[l1 - 1]) {
                if (istep <= ncp[l1 - 1]) d[i - 1] = sqdist(a, m, n, c, k, i, l1) * an1[l1 - 1];
                if (istep < ncp[l1 - 1] || istep < ncp[l2 - 1]) {
                    const double r2 = d[i - 1] / an2[l2 - 1];
                    const double dd = sqdist(a, m, n, c, k, i, l2);
                    if (dd < r2) {
                        icoun = 0;
                        *indx = 0;
                        itran[l1 - 1] = 1;
                        itran[l2 - 1] = 1;
                        ncp[l1 - 1] = istep + m;
                        ncp[l2 - 1] = istep + m;
                        transfer(a, m, n, c, k, ic1, ic2, nc, an1, an2, i, l1, l2);
                    }
                }
            }
            if (icoun == m) return;
        }
    }
}

/* Same argument list as the reference's kmns (asa136.H); wss is left untouched on ifault 1 and 3 as there. */
void ora_kmns(const double* a, int m, int n, double* c, int k, int* ic1, int* nc, int iter, double* wss, int* ifault)
{
    *ifault = 0;
    if (k <= 1 || m <= k) {
        *ifault = 3;
        return;
    }
    int* ic2 = (int*)malloc(sizeof(int) * (size_t)m);
    double* d = (double*)calloc((size_t)m, sizeof(double));
    double an1[64], an2[64];
    int ncp[64], itran[64], live[64];
    double* an1p = an1; double* an2p = an2; int* ncpp = ncp; int* itranp = itran; int* livep = live;
    if (k > 64) {
        an1p = (double*)malloc(sizeof(double) * (size_t)k); an2p = (double*)malloc(sizeof(double) * (size_t)k);
        ncpp = (int*)malloc(sizeof(int) * (size_t)k); itranp = (int*)malloc(sizeof(int) * (size_t)k); livep = (int*)malloc(sizeof(int) * (size_t)k);
    }
    for (int l = 0; l < k; l++) livep[l] = 0;
    for (int i = 1; i <= m; i++) {
        double dt[2];
        ic1[i - 1] = 1;
        ic2[i - 1] = 2;
        for (int il = 1; il <= 2; il++) dt[il - 1] = sqdist(a, m, n, c, k, i, il);
        if (dt[1] < dt[0]) {
            ic1[i - 1] = 2;
            ic2[i - 1] = 1;
            const double t = dt[0]; dt[0] = dt[1]; dt[1] = t;
        }
        for (int l = 3; l <= k; l++) {
            const double db = sqdist(a, m, n, c, k, i, l);
            if (db < dt[1]) {
                if (dt[0] <= db) {
                    dt[1] = db;
                    ic2[i - 1] = l;
                } else {
                    dt[1] = dt[0];
                    ic2[i - 1] = ic1[i - 1];
                    dt[0] = db;
                    ic1[i - 1] = l;
                }
            }
        }
    }
    for (int l = 1; l <= k; l++) {
        nc[l - 1] = 0;
        for (int j = 1; j <= n; j++) C_(l, j) = 0.0;
    }
    for (int i = 1; i <= m; i++) {
        const int l = ic1[i - 1];
        nc[l - 1] += 1;
        for (int j = 1; j <= n; j++) C_(l, j) = C_(l, j) + A_(i, j);
    }
    int empty = 0;
    for (int l = 1; l <= k; l++) if (nc[l - 1] == 0) empty = 1;
    if (empty) {
        *ifault = 1;
        goto done;
    }
    for (int l = 1; l <= k; l++) {
        const double aa = (double)nc[l - 1];
        for (int j = 1; j <= n; j++) C_(l, j) = C_(l, j) / aa;
        an2p[l - 1] = aa / (aa + 1.0);
        an1p[l - 1] = (1.0 < aa) ? aa / (aa - 1.0) : R8_HUGE;
        itranp[l - 1] = 1;
        ncpp[l - 1] = -1;
    }
    int indx = 0;
    *ifault = 2;
    for (int ij = 1; ij <= iter; ij++) {
        optra(a, m, n, c, k, ic1, ic2, nc, an1p, an2p, ncpp, d, itranp, livep, &indx);
        if (indx == m) {
            *ifault = 0;
            break;
        }
        qtran(a, m, n, c, k, ic1, ic2, nc, an1p, an2p, ncpp, d, itranp, &indx);
        if (k == 2) {
            *ifault = 0;
            break;
        }
        for (int l = 1; l <= k; l++) ncpp[l - 1] = 0;
    }
    for (int l = 1; l <= k; l++) {
        wss[l - 1] = 0.0;
        for (int j = 1; j <= n; j++) C_(l, j) = 0.0;
    }
    for (int i = 1; i <= m; i++) {
        const int ii = ic1[i - 1];
        for (int j = 1; j <= n; j++) C_(ii, j) = C_(ii, j) + A_(i, j);
    }
    for (int j = 1; j <= n; j++) {
        for (int l = 1; l <= k; l++) C_(l, j) = C_(l, j) / (double)nc[l - 1];
        for (int i = 1; i <= m; i++) {
            const int ii = ic1[i - 1];
            const double da = A_(i, j) - C_(ii, j);
            wss[ii - 1] = wss[ii - 1] + da * da;
        }
    }
done:
    free(ic2);
    free(d);
    if (k > 64) { free(an1p); free(an2p); free(ncpp); free(itranp); free(livep); }
}

/* ------------------------------------------------------------------------------------------------
 * MatePairEM
 * ---------------------------------------------------------------------------------------------- */
typedef struct em_state {
    int N, K;
    double mean, sd, lambda, tol, min_prob, min_size;   /* mMinClusterSize is a double, MatePairEM.h:41 */
    const double *X, *Y, *U;
    const int32_t *ToXO, *ToYO;
    double *XO, *YO;
    double W[KMAX], A[KMAX], B[KMAX];
    double *R, *RXO, *RYO;          /* [K][N] */
    double *ex;                     /* [K][N] exponents */
    double *SX, *SY, *CX, *CY, *CS; /* CX.. up to 4N+1 entries */
    double *ka; int *ic1;           /* k-means scratch */
    ora_mpe_diag* dg;
    int failed;                     /* a DebugCheck of the reference fired */
    int in_refit;                   /* the EM run is the refit of the chosen K (diagnostics only) */
} em_state;

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a < b ? b : a; }   /* std::max(a,b): returns a unless a<b */

static void exponents(em_state* s)
{
    const int N = s->N, K = s->K;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < K; j++) {
            const double t = (s->A[j] + s->B[j] - s->X[i] - s->Y[i] - s->U[i]) / s->sd;
            s->ex[(size_t)j * N + i] = -0.5 * (t * t) - s->lambda * dmax(0.0, s->X[i] - s->A[j]) - s->lambda * dmax(0.0, s->Y[i] - s->B[j]);
        }
}

static double log_likelihood(em_state* s)
{
    const int N = s->N, K = s->K;
    exponents(s);
    double LL = 0.0;
    for (int i = 0; i < N; i++) {
        double maxexp = s->ex[i];
        for (int j = 1; j < K; j++) maxexp = dmax(maxexp, s->ex[(size_t)j * N + i]);
        double sum = 0.0;
        for (int j = 0; j < K; j++) sum += s->W[j] * exp(s->ex[(size_t)j * N + i] - maxexp);
        if (sum == 0.0) {
            LL = -DBL_MAX;
            break;
        }
        LL = LL + log(sum) + maxexp;
    }
    return LL;
}

static void update_responsibilities(em_state* s)
{
    const int N = s->N, K = s->K;
    exponents(s);
    for (int i = 0; i < N; i++) {
        const int ixo = s->ToXO[i], iyo = s->ToYO[i];
        double maxexp = s->ex[i];
        for (int j = 1; j < K; j++) maxexp = dmax(maxexp, s->ex[(size_t)j * N + i]);
        double norm = 0.0;
        for (int j = 0; j < K; j++) norm += s->W[j] * exp(s->ex[(size_t)j * N + i] - maxexp);
        if (norm == 0.0) {
            s->failed = __LINE__;
            return;
        }
        for (int j = 0; j < K; j++) {
            const double r = s->W[j] * exp(s->ex[(size_t)j * N + i] - maxexp) / norm;
            s->R[(size_t)j * N + i] = r;
            s->RXO[(size_t)j * N + ixo] = r;
            s->RYO[(size_t)j * N + iyo] = r;
        }
    }
}

static int max_likelihood(em_state* s, const double* R, const double* RXO, const double* RYO, double* a_out, double* b_out, int first_iter)
{
    const int N = s->N;
    const double *XO = s->XO, *YO = s->YO;
    double *SX = s->SX, *SY = s->SY, *CX = s->CX, *CY = s->CY, *CS = s->CS;
    double acc = 0.0;
    for (int i = 0; i < N; i++) { acc = (i == 0) ? RXO[0] : acc + RXO[i]; SX[i] = acc; }
    for (int i = 0; i < N; i++) { acc = (i == 0) ? RYO[0] : acc + RYO[i]; SY[i] = acc; }
    int i = 0, j = 0, nc = 0;
    CX[nc] = XO[0]; CY[nc] = YO[0]; CS[nc] = 0.0; nc++;
    while (i < N && j < N) {
        if (i + 1 < N && XO[i] == XO[i + 1]) { i++; continue; }
        if (j + 1 < N && YO[j] == YO[j + 1]) { j++; continue; }
        if (SX[i] == SY[j]) {
            s->dg->merge_equal++;
            CX[nc] = XO[i]; CY[nc] = YO[j]; CS[nc] = SX[i]; nc++;
            if (i + 1 < N && j + 1 < N) { CX[nc] = XO[i + 1]; CY[nc] = YO[j + 1]; CS[nc] = SX[i]; nc++; }
            i++; j++;
        } else {
            const double big = dmax(fabs(SX[i]), fabs(SY[j]));
            if (big > 0.0) {
                double* m = first_iter ? &s->dg->min_merge_margin_first : &s->dg->min_merge_margin;
                *m = dmin(*m, fabs(SX[i] - SY[j]) / big);
            }
            if (SX[i] < SY[j]) {
                CX[nc] = XO[i]; CY[nc] = YO[j]; CS[nc] = SX[i]; nc++;
                if (i + 1 < N) { CX[nc] = XO[i + 1]; CY[nc] = YO[j]; CS[nc] = SX[i]; nc++; }
                i++;
            } else {
                CX[nc] = XO[i]; CY[nc] = YO[j]; CS[nc] = SY[j]; nc++;
                if (j + 1 < N) { CX[nc] = XO[i]; CY[nc] = YO[j + 1]; CS[nc] = SY[j]; nc++; }
                j++;
            }
        }
    }
    double NK = 0.0;
    for (int t = 0; t < N; t++) NK += R[t];
    if (NK == 0.0) {
        s->dg->nk_zero++;
        if (first_iter) s->dg->nk_zero_first_iter++;
        return 0;
    }
    double RXYU = 0.0;
    for (int t = 0; t < N; t++) RXYU += R[t] * (s->X[t] + s->Y[t] + s->U[t]);
    const double var = s->sd * s->sd;      /* pow(sd, 2.0) */
    int minindex = 0;
    while (minindex < nc) {
        const double partial = (RXYU - NK * (CX[minindex] + CY[minindex])) / var + s->lambda * CS[minindex];
        const double scale = fabs(RXYU / var) + fabs(NK * (CX[minindex] + CY[minindex]) / var) + fabs(s->lambda * CS[minindex]);
        if (partial == 0.0) s->dg->deriv_zero++;
        else if (scale > 0.0) {
            double* m = first_iter ? &s->dg->min_deriv_margin_first : &s->dg->min_deriv_margin;
            *m = dmin(*m, fabs(partial) / scale);
        }
        if (partial > 0) break;
        minindex++;
    }
    if (minindex >= nc) {       /* the reference reads CS[size] here: undefined behaviour (SURVEY a-10) */
        s->failed = __LINE__;
        return 0;
    }
    const double aplusb = (RXYU + var * s->lambda * CS[minindex]) / NK;
#ifdef MPE_TRACE
    fprintf(stderr, "  ML: NK %.17g RXYU %.17g minindex %d nc %d CS %.17g aplusb %.17g\n", NK, RXYU, minindex, nc, CS[minindex], aplusb);
#endif
    double a, b;
    if (minindex == 0) {
        const double min_a = CX[0], max_a = aplusb - CY[0];
        a = 0.5 * (min_a + max_a);
        b = aplusb - a;
    } else if (CS[minindex] != CS[minindex - 1]) {
        a = CX[minindex];
        b = CY[minindex];
    } else {
        const double min_a = dmax(CX[minindex], aplusb - CY[minindex - 1]);
        const double max_a = dmin(CX[minindex - 1], aplusb - CY[minindex]);
        a = 0.5 * (min_a + max_a);
        b = aplusb - a;
    }
    *a_out = a;
    *b_out = b;
    return 1;
}

static int select_kkz(em_state* s, int k, double* A, double* B, double* dist_min)
{
    const int N = s->N;
    const double *X = s->X, *Y = s->Y;
    double l2max = X[0] * Y[0];
    int imax = 0;
    for (int i = 1; i < N; i++) {
        const double l2 = X[i] * Y[i];
        if (l2 > l2max) { imax = i; l2max = l2; }
    }
    int na = 0;
    A[na] = X[imax]; B[na] = Y[imax]; na++;
    while (na < k) {
        for (int i = 0; i < N; i++) {
            double md = (X[i] - A[0]) * (X[i] - A[0]) + (Y[i] - B[0]) * (Y[i] - B[0]);
            for (int j = 1; j < na; j++) {
                const double dj = (X[i] - A[j]) * (X[i] - A[j]) + (Y[i] - B[j]) * (Y[i] - B[j]);
                md = dmin(md, dj);     /* std::min(md, dj) */
            }
            dist_min[i] = md;
        }
        double dmaxv = dist_min[0];
        int idx = 0;
        for (int i = 0; i < N; i++)
            if (dist_min[i] > dmaxv) { dmaxv = dist_min[i]; idx = i; }
        if (dmaxv == 0.0) return 0;
        A[na] = X[idx]; B[na] = Y[idx]; na++;
    }
    return 1;
}

/* returns 1 and *ll on success, 0 when the reference's function returns false; s->failed on a DebugCheck */
static int expectation_maximization(em_state* s, double* ll)
{
    const int N = s->N, K = s->K;
    /* mW/mA/mB.resize(K) keep their old leading values; s->W/A/B are fixed arrays that persist likewise */
    if (K == 1 || K == N) {
        for (int j = 0; j < K; j++)
            for (int i = 0; i < N; i++) {
                s->R[(size_t)j * N + i] = 1.0 / K;
                s->RXO[(size_t)j * N + i] = 1.0 / K;
                s->RYO[(size_t)j * N + i] = 1.0 / K;
            }
    } else {
        double px[KMAX], py[KMAX], c[2 * KMAX], wss[KMAX];
        int nc[KMAX], ifault;
        if (!select_kkz(s, K, px, py, s->SX)) {
            s->dg->kkz_fail++;
            return 0;
        }
        /* both inserts go to begin(): a = [Y..., X...], c = [py..., px...] (MatePairEM.cpp:423-429) */
        memcpy(s->ka, s->Y, sizeof(double) * (size_t)N);
        memcpy(s->ka + N, s->X, sizeof(double) * (size_t)N);
        for (int j = 0; j < K; j++) { c[j] = py[j]; c[K + j] = px[j]; }
        ora_kmns(s->ka, N, 2, c, K, s->ic1, nc, 1000, wss, &ifault);
        if (ifault == 1 || ifault == 3) {
            s->failed = __LINE__;
            return 0;
        }
        for (int i = 0; i < N; i++) {
            const int ixo = s->ToXO[i], iyo = s->ToYO[i];
            for (int j = 0; j < K; j++) {
                const double v = (j == s->ic1[i] - 1) ? 1 : 0;
                s->R[(size_t)j * N + i] = v;
                s->RXO[(size_t)j * N + ixo] = v;
                s->RYO[(size_t)j * N + iyo] = v;
            }
        }
    }
    double last = 0.0;
    int valid = 0, first = 1;
    for (;;) {
        for (int j = 0; j < K; j++) {
            double a, b;
            if (max_likelihood(s, s->R + (size_t)j * N, s->RXO + (size_t)j * N, s->RYO + (size_t)j * N, &a, &b, first || K == 1)) {   /* K = 1: R is exactly 1 in every iteration, as exp-free as a first M step */
                s->A[j] = a;
                s->B[j] = b;
            }
            if (s->failed) return 0;
        }
        first = 0;
        for (int j = 0; j < K; j++) {
            double nk = 0.0;
            for (int i = 0; i < N; i++) nk += s->R[(size_t)j * N + i];
            s->W[j] = nk / N;
        }
        const double likelihood = log_likelihood(s);
        s->dg->em_iterations++;
        s->dg->iters_by_k[s->in_refit ? 11 : K]++;
#ifdef MPE_TRACE
        fprintf(stderr, "K %d ll %.17g last %.17g valid %d A0 %.17g B0 %.17g W0 %.17g\n", K, likelihood, last, valid, s->A[0], s->B[0], s->W[0]);
#endif
        if (valid) s->dg->min_tol_margin = dmin(s->dg->min_tol_margin, fabs(fabs(likelihood - last) - s->tol) / s->tol);
        if (valid && fabs(likelihood - last) < s->tol) break;
        if (valid && likelihood == -DBL_MAX) {
            s->dg->ll_underflow++;
            return 0;
        }
        if (!(!valid || (likelihood / last < 1.0000001))) {
            s->failed = __LINE__;
            return 0;
        }
        last = likelihood;
        valid = 1;
        update_responsibilities(s);
#ifdef MPE_TRACE
        for (int j = 0; j < K; j++) { fprintf(stderr, "R[%d]:", j); for (int i = 0; i < N; i++) fprintf(stderr, " %.6g", s->R[(size_t)j * N + i]); fprintf(stderr, " | RXO:"); for (int i = 0; i < N; i++) fprintf(stderr, " %.6g", s->RXO[(size_t)j * N + i]); fprintf(stderr, "\n"); }
#endif
        if (s->failed) return 0;
    }
    *ll = last;
    return 1;
}

/* One problem (MatePairEM.cpp:540-636).  member bit j of mate pair i = membership in emitted cluster j.
 * Returns the number of emitted clusters, -1 when a DebugCheck of the reference would have ended the process. */
static int do_clustering(em_state* s, uint16_t* member)
{
    const int N = s->N;
    for (int i = 0; i < N; i++) member[i] = 0;
    if ((double)N < s->min_size) return 0;
    for (int i = 0; i < N; i++) {
        s->XO[s->ToXO[i]] = s->X[i];
        s->YO[s->ToYO[i]] = s->Y[i];
    }
    /* canonical start: W/A/B begin at 0 for every problem; they are never read before they
     * are written unless nk_zero_first_iter counts an event */
    for (int j = 0; j < KMAX; j++) s->W[j] = s->A[j] = s->B[j] = 0.0;
    double min_bic = 0.0, bics[KMAX + 1];
    int min_valid = 0, kmin = 1, bic_ok[KMAX + 1] = {0};
    const int kend = N < KMAX ? N : KMAX;
    for (int K = 1; K <= kend; K++) {
        s->K = K;
        double ll;
        if (!expectation_maximization(s, &ll)) {
            if (s->failed) return -1;
            continue;
        }
        const double bic = -2.0 * ll + K * 2.0 * log((double)N);
        s->dg->ll_by_k[K] = ll;
        bics[K] = bic;
        bic_ok[K] = 1;
        if (!min_valid || bic < min_bic) {
            min_bic = bic;
            kmin = K;
            min_valid = 1;
        }
    }
    if (!min_valid) s->dg->all_k_failed++;
    for (int K = 1; K <= kend; K++)
        if (bic_ok[K] && K != kmin && min_bic != 0.0)
            s->dg->min_bic_gap = dmin(s->dg->min_bic_gap, (bics[K] - min_bic) / fabs(min_bic));
    s->K = kmin;
    s->dg->iters_by_k[0] = kmin;
    s->in_refit = 1;
    double ll;
    const int refit_ok = expectation_maximization(s, &ll);
    s->in_refit = 0;
    if (!refit_ok) {
        if (s->failed) return -1;
        return 0;                      /* "Error: No clusters" */
    }
    int emitted = 0;
    for (int j = 0; j < s->K; j++) {
        int cnt = 0;
        for (int i = 0; i < N; i++) {
            const double x = s->X[i], y = s->Y[i], u = s->U[i], a = s->A[j], b = s->B[j];
            const double prob = ora_normalpdf(a + b - x - y, u, s->sd) * exp(-s->lambda * dmax(0.0, x - a) - s->lambda * dmax(0.0, y - b));
            s->dg->min_prob_margin = dmin(s->dg->min_prob_margin, fabs(prob - s->min_prob) / s->min_prob);
            if (prob > s->min_prob) s->ic1[cnt++] = i;
        }
        if ((double)cnt >= s->min_size) {
            for (int t = 0; t < cnt; t++) member[s->ic1[t]] |= (uint16_t)(1u << emitted);
            emitted++;
        }
    }
    return emitted;
}

static void diag_init(ora_mpe_diag* d)
{
    memset(d, 0, sizeof(*d));
    d->min_prob_margin = d->min_tol_margin = d->min_bic_gap = d->min_deriv_margin = d->min_merge_margin = DBL_MAX;
    d->min_deriv_margin_first = d->min_merge_margin_first = DBL_MAX;
}

static void diag_merge(ora_mpe_diag* into, const ora_mpe_diag* d)
{
    into->min_prob_margin = dmin(into->min_prob_margin, d->min_prob_margin);
    into->min_tol_margin = dmin(into->min_tol_margin, d->min_tol_margin);
    into->min_bic_gap = dmin(into->min_bic_gap, d->min_bic_gap);
    into->min_deriv_margin = dmin(into->min_deriv_margin, d->min_deriv_margin);
    into->min_merge_margin = dmin(into->min_merge_margin, d->min_merge_margin);
    into->min_deriv_margin_first = dmin(into->min_deriv_margin_first, d->min_deriv_margin_first);
    into->min_merge_margin_first = dmin(into->min_merge_margin_first, d->min_merge_margin_first);
    into->nk_zero += d->nk_zero;
    into->nk_zero_first_iter += d->nk_zero_first_iter;
    into->ll_underflow += d->ll_underflow;
    into->kkz_fail += d->kkz_fail;
    into->em_iterations += d->em_iterations;
    into->merge_equal += d->merge_equal;
    into->all_k_failed += d->all_k_failed;
    into->deriv_zero += d->deriv_zero;
    for (int k = 0; k < 12; k++) into->iters_by_k[k] += d->iters_by_k[k];
}

/* Same arrays as mpe_cluster_batch (include/defuse_mpe.h).  diag and prob_diag may be NULL; prob_diag[p] receives
 * problem p's own figures (to find the problems that come closest to a knife edge).  Problems run on OpenMP threads. */
int ora_mpe_cluster_batch(const ora_mpe_params* params, const int64_t* prob_off, int32_t n_problems,
                          const double* x, const double* y, const double* u, const int32_t* to_xo, const int32_t* to_yo,
                          int32_t* n_clusters, uint16_t* member, int32_t* status, ora_mpe_diag* diag, ora_mpe_diag* prob_diag)
{
    ora_mpe_diag total;
    diag_init(&total);
    int rc = 0;
#pragma omp parallel
    {
        ora_mpe_diag mine;
        diag_init(&mine);
        size_t cap = 0;
        double* buf = NULL;
        int* ibuf = NULL;
#pragma omp for schedule(dynamic, 1)
        for (int32_t p = 0; p < n_problems; p++) {
            const int64_t o = prob_off[p];
            const int N = (int)(prob_off[p + 1] - o);
            ora_mpe_diag dp;
            diag_init(&dp);
            if (N <= 0) {
                n_clusters[p] = 0;
                status[p] = 0;
                if (prob_diag) prob_diag[p] = dp;
                continue;
            }
            if ((size_t)N > cap) {
                free(buf);
                free(ibuf);
                cap = (size_t)N * 2;
                buf = (double*)malloc(sizeof(double) * cap * (2 + 4 * KMAX + 2 + 3 * 4 + 2) + 64);
                ibuf = (int*)malloc(sizeof(int) * cap);
            }
            em_state s;
            memset(&s, 0, sizeof(s));
            s.N = N;
            s.mean = params->fragment_mean;
            s.sd = params->fragment_stddev;
            s.lambda = 0.1;
            s.tol = 0.001;
            s.min_prob = params->min_probability;
            s.min_size = (double)params->min_cluster_size;
            s.X = x + o; s.Y = y + o; s.U = u + o; s.ToXO = to_xo + o; s.ToYO = to_yo + o;
            double* q = buf;
            s.XO = q; q += N; s.YO = q; q += N;
            s.R = q; q += (size_t)KMAX * N; s.RXO = q; q += (size_t)KMAX * N; s.RYO = q; q += (size_t)KMAX * N; s.ex = q; q += (size_t)KMAX * N;
            s.SX = q; q += N; s.SY = q; q += N;
            s.CX = q; q += 4 * (size_t)N + 2; s.CY = q; q += 4 * (size_t)N + 2; s.CS = q; q += 4 * (size_t)N + 2;   /* <= 2 entries per merge step, <= 2N steps */
            s.ka = q; q += 2 * (size_t)N;
            s.ic1 = ibuf;
            s.dg = &dp;
            const int e = do_clustering(&s, member + o);
            n_clusters[p] = e < 0 ? 0 : e;
            status[p] = e < 0 ? s.failed : 0;
            if (prob_diag) prob_diag[p] = dp;
            diag_merge(&mine, &dp);
        }
        free(buf);
        free(ibuf);
#pragma omp critical
        diag_merge(&total, &mine);
    }
    if (diag) *diag = total;
    return rc;
}
