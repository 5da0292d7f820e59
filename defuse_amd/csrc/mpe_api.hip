// mpe_api.hip — mate-pair EM clustering on gfx950 (include/defuse_mpe.h), replacing
// MatePairEM::DoClustering (tools/MatePairEM.cpp:540-636) and everything it calls:
//   MaxLikelihood :192-325, LogLikelihood :96-137, UpdateResponsibilities :139-181,
//   UpdateMixWeights :183-190, SelectKKZ :327-386, ExpectationMaximization :388-494,
//   kmns / optra / qtran (AS 136, tools/asa136.C).
//
// FP64 throughout, no contraction into FMA, and every sum is taken in the reference's serial order
// (MaxLikelihood compares two prefix sums for exact equality, so a tree reduction would change
// results).  The model selection loop of DoClustering (K = 1..min(10,N), :599-606) runs its K fits
// independently, so k_mpe_fit gives every (problem, K) its own lane and workspace slice; lanes are
// ordered K-major over problems sorted by size, so a wave holds fits of one K and of similar N.
// k_mpe_final then picks the K of minimal BIC per problem (first minimum in K order, as the
// reference's strict '<'), refits it and derives the memberships, one problem per lane.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include "hip_raii.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/defuse_mpe.h"

namespace {

thread_local std::string g_mpe_err;     // per host thread: the sharded call runs one per device
std::string g_mpe_err_sharded;

#define MPE_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[256];                                                                         \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_mpe_err = b_;                                                                       \
            return -2;                                                                            \
        }                                                                                         \
    } while (0)

constexpr double R8_HUGE = 1.0e30;          // tools/asa136.C r8_huge
constexpr double DBL_MAX_ = 1.7976931348623157e308;
constexpr double LAMBDA = 0.1, TOLERANCE = 0.001;   // tools/MatePairEM.cpp:55-56
constexpr int KMEANS_ITER = 1000;

// per-problem workspace, carved out of one global allocation: doubles then ints
struct Work {
    int N;
    const double *X, *Y, *U;
    const int *ToXO, *ToYO;
    double *XO, *YO;
    double *R, *RXO, *RYO, *EX;     // [K][N]
    double *SX, *SY;                // [N]
    double *CX, *CY, *CS;           // [4N+4]
    double *ka, *kd, *dist;         // kmns: a [2N], d [N]; KKZ: DistMin [N]
    const double* XYU;              // wave version: X + Y + U per mate pair (constant over a fit)
    const int* TX;                  // wave version: ToXO
    int* XfromY;                    // wave version: rank in x order of the mate pair with rank s in y order
    int *ic1, *ic2;                 // [N]
    double W[MPE_KMAX], A[MPE_KMAX], B[MPE_KMAX];
    double sd;
    long long iters;
    int fail;
};

// workspace of one fit with K components
__host__ __device__ inline size_t work_doubles(int n, int K) { return (size_t)n * (2 + 4 * K + 2 + 12 + 4) + 16; }
__host__ __device__ inline size_t work_ints(int n) { return (size_t)n * 2 + 8; }
// what the wave kernel uses of a fit's slot: RXO (K*n), SX, SY, kd; behind the largest fit's own arrays the ones all fits
// of the problem share (XO, YO, X+Y+U, the k-means point array 2n, XfromY)
__host__ __device__ inline size_t wave_work_doubles(int n, int K, bool largest) { return (size_t)n * (K + 3 + (largest ? 6 : 0)) + 16; }

__device__ double dist2(const double* a, int m, const double* c, int k, int i, int l)   // n = 2
{
    double s = 0.0;
    for (int j = 1; j <= 2; ++j) {
        const double df = a[i - 1 + (j - 1) * m] - c[l - 1 + (j - 1) * k];
        s = s + df * df;
    }
    return s;
}

__device__ void transfer(const double* a, int m, double* c, int k, int* nc, double* an1, double* an2, int* ic1, int* ic2, int i,
                         int l1, int l2)
{
    const double al1 = (double)nc[l1 - 1], alw = al1 - 1.0, al2 = (double)nc[l2 - 1], alt = al2 + 1.0;
    for (int j = 1; j <= 2; ++j) {
        c[l1 - 1 + (j - 1) * k] = (c[l1 - 1 + (j - 1) * k] * al1 - a[i - 1 + (j - 1) * m]) / alw;
        c[l2 - 1 + (j - 1) * k] = (c[l2 - 1 + (j - 1) * k] * al2 + a[i - 1 + (j - 1) * m]) / alt;
    }
    nc[l1 - 1] -= 1;
    nc[l2 - 1] += 1;
    an2[l1 - 1] = alw / al1;
    an1[l1 - 1] = 1.0 < alw ? alw / (alw - 1.0) : R8_HUGE;
    an1[l2 - 1] = alt / al2;
    an2[l2 - 1] = alt / (alt + 1.0);
    ic1[i - 1] = l2;
    ic2[i - 1] = l1;
}

// AS 136 with n = 2 (tools/asa136.C:13-336 kmns, :339-566 optra, :569-758 qtran); returns ifault
__device__ int kmns(const double* a, int m, double* c, int k, int* ic1, int* ic2, double* d, int iter)
{
    if (k <= 1 || m <= k) return 3;
    int nc[MPE_KMAX], ncp[MPE_KMAX], itran[MPE_KMAX], live[MPE_KMAX];
    double an1[MPE_KMAX], an2[MPE_KMAX];
    for (int i = 1; i <= m; ++i) {
        ic1[i - 1] = 1;
        ic2[i - 1] = 2;
        double dt[2];
        for (int il = 1; il <= 2; ++il) dt[il - 1] = dist2(a, m, c, k, i, il);
        if (dt[1] < dt[0]) {
            ic1[i - 1] = 2;
            ic2[i - 1] = 1;
            const double t = dt[0];
            dt[0] = dt[1];
            dt[1] = t;
        }
        for (int l = 3; l <= k; ++l) {
            const double db = dist2(a, m, c, k, i, l);
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
    for (int l = 1; l <= k; ++l) {
        nc[l - 1] = 0;
        for (int j = 1; j <= 2; ++j) c[l - 1 + (j - 1) * k] = 0.0;
    }
    for (int i = 1; i <= m; ++i) {
        const int l = ic1[i - 1];
        nc[l - 1] += 1;
        for (int j = 1; j <= 2; ++j) c[l - 1 + (j - 1) * k] = c[l - 1 + (j - 1) * k] + a[i - 1 + (j - 1) * m];
    }
    for (int l = 1; l <= k; ++l)
        if (nc[l - 1] == 0) return 1;
    for (int l = 1; l <= k; ++l) {
        const double aa = (double)nc[l - 1];
        for (int j = 1; j <= 2; ++j) c[l - 1 + (j - 1) * k] = c[l - 1 + (j - 1) * k] / aa;
        an2[l - 1] = aa / (aa + 1.0);
        an1[l - 1] = 1.0 < aa ? aa / (aa - 1.0) : R8_HUGE;
        itran[l - 1] = 1;
        ncp[l - 1] = -1;
    }
    int indx = 0, ifault = 2;
    for (int ij = 1; ij <= iter; ++ij) {
        // ---- optra
        {
            for (int l = 1; l <= k; ++l)
                if (itran[l - 1] == 1) live[l - 1] = m + 1;
            bool early = false;
            for (int i = 1; i <= m; ++i) {
                indx += 1;
                const int l1 = ic1[i - 1];
                int l2 = ic2[i - 1];
                const int ll = l2;
                if (1 < nc[l1 - 1]) {
                    if (ncp[l1 - 1] != 0) d[i - 1] = dist2(a, m, c, k, i, l1) * an1[l1 - 1];
                    double r2 = dist2(a, m, c, k, i, l2) * an2[l2 - 1];
                    for (int l = 1; l <= k; ++l) {
                        if ((i < live[l1 - 1] || i < live[l2 - 1]) && l != l1 && l != ll) {
                            const double rr = r2 / an2[l - 1];
                            const double dc = dist2(a, m, c, k, i, l);
                            if (dc < rr) {
                                r2 = dc * an2[l - 1];
                                l2 = l;
                            }
                        }
                    }
                    if (d[i - 1] <= r2) {
                        ic2[i - 1] = l2;
                    } else {
                        indx = 0;
                        live[l1 - 1] = m + i;
                        live[l2 - 1] = m + i;
                        ncp[l1 - 1] = i;
                        ncp[l2 - 1] = i;
                        transfer(a, m, c, k, nc, an1, an2, ic1, ic2, i, l1, l2);
                    }
                }
                if (indx == m) { early = true; break; }
            }
            if (!early)
                for (int l = 1; l <= k; ++l) {
                    itran[l - 1] = 0;
                    live[l - 1] = live[l - 1] - m;
                }
        }
        if (indx == m) { ifault = 0; break; }
        // ---- qtran
        {
            int icoun = 0, istep = 0;
            bool done = false;
            while (!done) {
                for (int i = 1; i <= m; ++i) {
                    icoun += 1;
                    istep += 1;
                    const int l1 = ic1[i - 1], l2 = ic2[i - 1];
                    if (1 < nc[l1 - 1]) {
                        if (istep <= ncp[l1 - 1]) d[i - 1] = dist2(a, m, c, k, i, l1) * an1[l1 - 1];
                        if (istep < ncp[l1 - 1] || istep < ncp[l2 - 1]) {
                            const double r2 = d[i - 1] / an2[l2 - 1];
                            const double dd = dist2(a, m, c, k, i, l2);
                            if (dd < r2) {
                                icoun = 0;
                                indx = 0;
                                itran[l1 - 1] = 1;
                                itran[l2 - 1] = 1;
                                ncp[l1 - 1] = istep + m;
                                ncp[l2 - 1] = istep + m;
                                transfer(a, m, c, k, nc, an1, an2, ic1, ic2, i, l1, l2);
                            }
                        }
                    }
                    if (icoun == m) { done = true; break; }
                }
            }
        }
        if (k == 2) { ifault = 0; break; }
        for (int l = 1; l <= k; ++l) ncp[l - 1] = 0;
    }
    // the final recomputation of centres and wss (asa136.C:262-300) does not change ic1: omitted
    return ifault;
}

__device__ void exponents(Work& w, int K)
{
    for (int i = 0; i < w.N; ++i)
        for (int j = 0; j < K; ++j) {
            const double t = (w.A[j] + w.B[j] - w.X[i] - w.Y[i] - w.U[i]) / w.sd;
            w.EX[(size_t)j * w.N + i] = -0.5 * (t * t) - LAMBDA * fmax(0.0, w.X[i] - w.A[j]) - LAMBDA * fmax(0.0, w.Y[i] - w.B[j]);
        }
}

__device__ double log_likelihood(Work& w, int K)   // tools/MatePairEM.cpp:96-137
{
    exponents(w, K);
    double LL = 0.0;
    for (int i = 0; i < w.N; ++i) {
        double maxexp = w.EX[i];
        for (int j = 1; j < K; ++j) maxexp = fmax(maxexp, w.EX[(size_t)j * w.N + i]);
        double sum = 0.0;
        for (int j = 0; j < K; ++j) sum += w.W[j] * exp(w.EX[(size_t)j * w.N + i] - maxexp);
        if (sum == 0.0) return -DBL_MAX_;
        LL = LL + log(sum) + maxexp;
    }
    return LL;
}

__device__ bool update_responsibilities(Work& w, int K)   // :139-181
{
    exponents(w, K);
    for (int i = 0; i < w.N; ++i) {
        const int ixo = w.ToXO[i], iyo = w.ToYO[i];
        double maxexp = w.EX[i];
        for (int j = 1; j < K; ++j) maxexp = fmax(maxexp, w.EX[(size_t)j * w.N + i]);
        double norm = 0.0;
        for (int j = 0; j < K; ++j) norm += w.W[j] * exp(w.EX[(size_t)j * w.N + i] - maxexp);
        if (norm == 0.0) return false;                     // DebugCheck(norm != 0.0)
        for (int j = 0; j < K; ++j) {
            const double r = w.W[j] * exp(w.EX[(size_t)j * w.N + i] - maxexp) / norm;
            w.R[(size_t)j * w.N + i] = r;
            w.RXO[(size_t)j * w.N + ixo] = r;
            w.RYO[(size_t)j * w.N + iyo] = r;
        }
    }
    return true;
}

// :192-325; returns 0 = no update (NK == 0), 1 = ok, -1 = the reference would read past the end
__device__ int max_likelihood(Work& w, const double* R, const double* RXO, const double* RYO, double& a, double& b)
{
    const int N = w.N;
    double acc = 0.0;
    for (int i = 0; i < N; ++i) { acc = i == 0 ? RXO[0] : acc + RXO[i]; w.SX[i] = acc; }
    for (int i = 0; i < N; ++i) { acc = i == 0 ? RYO[0] : acc + RYO[i]; w.SY[i] = acc; }
    int i = 0, j = 0, n = 0;
    auto push = [&](double cx, double cy, double cs) { w.CX[n] = cx; w.CY[n] = cy; w.CS[n] = cs; ++n; };
    push(w.XO[0], w.YO[0], 0.0);
    while (i < N && j < N) {
        if (i + 1 < N && w.XO[i] == w.XO[i + 1]) { ++i; continue; }
        if (j + 1 < N && w.YO[j] == w.YO[j + 1]) { ++j; continue; }
        if (w.SX[i] == w.SY[j]) {
            push(w.XO[i], w.YO[j], w.SX[i]);
            if (i + 1 < N && j + 1 < N) push(w.XO[i + 1], w.YO[j + 1], w.SX[i]);
            ++i;
            ++j;
        } else if (w.SX[i] < w.SY[j]) {
            push(w.XO[i], w.YO[j], w.SX[i]);
            if (i + 1 < N) push(w.XO[i + 1], w.YO[j], w.SX[i]);
            ++i;
        } else {
            push(w.XO[i], w.YO[j], w.SY[j]);
            if (j + 1 < N) push(w.XO[i], w.YO[j + 1], w.SY[j]);
            ++j;
        }
    }
    double NK = 0.0;
    for (int t = 0; t < N; ++t) NK += R[t];
    if (NK == 0.0) return 0;
    double RXYU = 0.0;
    for (int t = 0; t < N; ++t) RXYU += R[t] * (w.X[t] + w.Y[t] + w.U[t]);
    const double var = w.sd * w.sd;
    int mi = 0;
    while (mi < n) {
        if ((RXYU - NK * (w.CX[mi] + w.CY[mi])) / var + LAMBDA * w.CS[mi] > 0) break;
        ++mi;
    }
    if (mi >= n) return -1;
    const double aplusb = (RXYU + var * LAMBDA * w.CS[mi]) / NK;
    if (mi == 0) {
        const double min_a = w.CX[0], max_a = aplusb - w.CY[0];
        a = 0.5 * (min_a + max_a);
        b = aplusb - a;
    } else if (w.CS[mi] != w.CS[mi - 1]) {
        a = w.CX[mi];
        b = w.CY[mi];
    } else {
        const double min_a = fmax(w.CX[mi], aplusb - w.CY[mi - 1]);
        const double max_a = fmin(w.CX[mi - 1], aplusb - w.CY[mi]);
        a = 0.5 * (min_a + max_a);
        b = aplusb - a;
    }
    return 1;
}

__device__ bool select_kkz(Work& w, int k, double* A, double* B)   // :327-386
{
    const int N = w.N;
    double l2max = w.X[0] * w.Y[0];
    int imax = 0;
    for (int i = 1; i < N; ++i) {
        const double l2 = w.X[i] * w.Y[i];
        if (l2 > l2max) { imax = i; l2max = l2; }
    }
    int na = 1;
    A[0] = w.X[imax];
    B[0] = w.Y[imax];
    while (na < k) {
        for (int i = 0; i < N; ++i) {
            double md = (w.X[i] - A[0]) * (w.X[i] - A[0]) + (w.Y[i] - B[0]) * (w.Y[i] - B[0]);
            for (int j = 1; j < na; ++j) {
                const double dj = (w.X[i] - A[j]) * (w.X[i] - A[j]) + (w.Y[i] - B[j]) * (w.Y[i] - B[j]);
                md = fmin(md, dj);
            }
            w.dist[i] = md;
        }
        double dmax = w.dist[0];
        int idx = 0;
        for (int i = 0; i < N; ++i)
            if (w.dist[i] > dmax) { dmax = w.dist[i]; idx = i; }
        if (dmax == 0.0) return false;
        A[na] = w.X[idx];
        B[na] = w.Y[idx];
        ++na;
    }
    return true;
}

// :388-494; returns true and sets ll on success
__device__ bool expectation_maximization(Work& w, int K, double& ll)
{
    const int N = w.N;
    if (K == 1 || K == N) {
        const double v = 1.0 / K;
        for (int j = 0; j < K; ++j)
            for (int i = 0; i < N; ++i) {
                w.R[(size_t)j * N + i] = v;
                w.RXO[(size_t)j * N + i] = v;
                w.RYO[(size_t)j * N + i] = v;
            }
    } else {
        double px[MPE_KMAX], py[MPE_KMAX], c[2 * MPE_KMAX];
        if (!select_kkz(w, K, px, py)) return false;
        for (int i = 0; i < N; ++i) {          // both inserts are at begin(): a = [Y..., X...], c = [py..., px...]
            w.ka[i] = w.Y[i];
            w.ka[N + i] = w.X[i];
        }
        for (int j = 0; j < K; ++j) {
            c[j] = py[j];
            c[K + j] = px[j];
        }
        const int ifault = kmns(w.ka, N, c, K, w.ic1, w.ic2, w.kd, KMEANS_ITER);
        if (ifault == 1 || ifault == 3) { w.fail = 1; return false; }     // DebugCheck(ifault != 1 / != 3)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < K; ++j) {
                const double v = (j == w.ic1[i] - 1) ? 1.0 : 0.0;
                w.R[(size_t)j * N + i] = v;
                w.RXO[(size_t)j * N + w.ToXO[i]] = v;
                w.RYO[(size_t)j * N + w.ToYO[i]] = v;
            }
    }
    double last = 0.0;
    bool valid = false;
    for (;;) {
        for (int j = 0; j < K; ++j) {
            double a, b;
            const int rc = max_likelihood(w, w.R + (size_t)j * N, w.RXO + (size_t)j * N, w.RYO + (size_t)j * N, a, b);
            if (rc < 0) { w.fail = 1; return false; }
            if (rc > 0) { w.A[j] = a; w.B[j] = b; }
        }
        for (int j = 0; j < K; ++j) {              // UpdateMixWeights :183-190
            double nk = 0.0;
            for (int i = 0; i < N; ++i) nk += w.R[(size_t)j * N + i];
            w.W[j] = nk / N;
        }
        const double like = log_likelihood(w, K);
        w.iters += 1;
        if (valid && fabs(like - last) < TOLERANCE) break;
        if (valid && like == -DBL_MAX_) return false;
        if (valid && !(like / last < 1.0000001)) { w.fail = 1; return false; }   // DebugCheck
        last = like;
        valid = true;
        if (!update_responsibilities(w, K)) { w.fail = 1; return false; }
    }
    ll = last;
    return true;
}

// ------------------------------------------------------------------------------------------------------
// Wave versions.  A lane that works through a whole fit by itself spends N*K exp/log/divide sequences per EM iteration one
// after the other and reads its private arrays with 64 different addresses per wave instruction; the few largest bin pairs
// of a run then decide the kernel time.  In the wave kernel (k_mpe_problem_wave, below)
//   * everything elementwise over the mate pairs (exponents, exp, log, responsibilities, KKZ distances, memberships) runs
//     lane-strided over i, coalesced;
//   * every sum the reference takes serially stays one serial chain of the same additions in the same order, but
//     independent chains run side by side in different lanes;
//   * the AS 136 k-means is sequential by construction (every transfer changes the centres the next point sees): one lane
//     per fit.
// Results are those of the lane version operation for operation: same expressions, same order, same ocml calls.
constexpr int WV = 64;
#ifndef MPE_CHAIN
#define MPE_CHAIN 4     // elements the serial sums of the M step load ahead of their additions
#endif
#ifndef MPE_WPE
#define MPE_WPE 3       // waves per SIMD the wave kernel is compiled for (profiles/microbench/mpe_occ.sh)
#endif

// MaxLikelihood (:192-325) for one component in one lane, streaming: the two prefix sums advance with the
// walk (SX[i] = SX[i-1] + RXO[i]) and the first breakpoint with a positive derivative ends it, so neither the
// prefix arrays nor the breakpoint list are stored.  nk receives the component's sum of responsibilities
// (UpdateMixWeights :183-190 takes the same sum in the same order).  Return codes as max_likelihood.
#ifdef MPE_PHASE_STATS
__device__ unsigned long long g_mpe_walk[4];      // wave cycles in the M step's serial sums / in its breakpoint walk, walk steps (longest lane), calls
#endif
__device__ int max_likelihood_stream(const Work& w, const double* RXO_base, int stride, double& a, double& b, double& nk)
{
#ifdef MPE_PHASE_STATS
    const unsigned long long tw0 = __builtin_readcyclecounter();
    unsigned long long n_steps = 0;
#endif
    // the component's responsibilities in x order: element r at RXO_base[r * stride] (the fits keep them component-minor,
    // so that the lanes of a fit, which own its components, touch neighbouring words)
    auto RXO = [&](int r) { return RXO_base[(size_t)r * stride]; };
    const int N = w.N;
    const int* TX = w.TX;
    const int* XfromY = w.XfromY;
    double NK = 0.0, RXYU = 0.0;
    int t = 0;
    // Loads in batches so that only the additions are serial, and the batches overlap: a batch's responsibilities are gathered
    // through indices that were fetched while the batch before it was summed (one round trip to memory per batch, not two).
    int ixn[MPE_CHAIN];
    double qn[MPE_CHAIN];
    if (N >= MPE_CHAIN) {
#pragma unroll
        for (int v = 0; v < MPE_CHAIN; ++v) { ixn[v] = TX[v]; qn[v] = w.XYU[v]; }
    }
    for (; t + MPE_CHAIN <= N; t += MPE_CHAIN) {
        double r[MPE_CHAIN], q[MPE_CHAIN];
#pragma unroll
        for (int v = 0; v < MPE_CHAIN; ++v) { r[v] = RXO(ixn[v]); q[v] = qn[v]; }
        if (t + 2 * MPE_CHAIN <= N) {
#pragma unroll
            for (int v = 0; v < MPE_CHAIN; ++v) { ixn[v] = TX[t + MPE_CHAIN + v]; qn[v] = w.XYU[t + MPE_CHAIN + v]; }
        }
#pragma unroll
        for (int v = 0; v < MPE_CHAIN; ++v) { NK += r[v]; RXYU += r[v] * q[v]; }
    }
    for (; t < N; ++t) {
        const double r = RXO(TX[t]);
        NK += r;
        RXYU += r * w.XYU[t];
    }
    nk = NK;
#ifdef MPE_PHASE_STATS
    const unsigned long long tw1 = __builtin_readcyclecounter();
#endif
    if (NK == 0.0) return 0;
    const double var = w.sd * w.sd;
    double pcx = 0.0, pcy = 0.0, pcs = 0.0, ccx = 0.0, ccy = 0.0, ccs = 0.0;
    int mi = 0;
    bool found = false;
    // the sign of (RXYU - NK*(cx+cy))/var + LAMBDA*cs without the division wherever it is not in doubt: the product with
    // the rounded reciprocal is within a few ulp of the quotient, so an estimate clear of zero by 1e-9 of its terms has the
    // sign of the exact expression; otherwise the expression itself is evaluated
    const double inv_var = 1.0 / var;
    auto push = [&](double cx, double cy, double cs) {
        if (found) return;
        const double q = RXYU - NK * (cx + cy), lc = LAMBDA * cs;
        const double est = q * inv_var + lc;
        bool pos;
        if (fabs(est) > 1e-9 * (fabs(q * inv_var) + fabs(lc)) && fabs(est) > 1e-290 && fabs(est) < 1e290) pos = est > 0;
        else pos = q / var + lc > 0;
        if (pos) { found = true; ccx = cx; ccy = cy; ccs = cs; }
        else { pcx = cx; pcy = cy; pcs = cs; ++mi; }
    };
    // the walk keeps XO[i], XO[i+1], RXO[i+1] (and the same for j) in registers: one load per advance, off the
    // critical compare; every kind of step advances through the same code so that the lanes of a wave stay together
    int i = 0, j = 0;
    double sx = RXO(0), sy = RXO(XfromY[0]);
    double xi = w.XO[0], yj = w.YO[0];
    double xn = 0.0, yn = 0.0, rxn = 0.0, ryn = 0.0;
    if (N > 1) { xn = w.XO[1]; yn = w.YO[1]; rxn = RXO(1); ryn = RXO(XfromY[1]); }
    // the y side reaches its responsibilities through the rank map (RXO(XfromY[j])): the index of the element after the next is
    // fetched one advance early, so that an advance waits for one round trip to memory and not for two in a row
    int jx2 = N > 2 ? XfromY[2] : 0;
    push(xi, yj, 0.0);
    while (!found && i < N && j < N) {
#ifdef MPE_PHASE_STATS
        ++n_steps;
#endif
        const bool hi = i + 1 < N, hj = j + 1 < N;
        bool adv_i, adv_j;
        if (hi && xi == xn) { adv_i = true; adv_j = false; }
        else if (hj && yj == yn) { adv_i = false; adv_j = true; }
        else {
            const bool eq = sx == sy, lt = sx < sy;
            const double cs = (eq || lt) ? sx : sy;
            push(xi, yj, cs);
            const bool second = eq ? (hi && hj) : (lt ? hi : hj);
            if (second) push((eq || lt) ? xn : xi, (eq || !lt) ? yn : yj, cs);
            adv_i = eq || lt;
            adv_j = eq || !lt;
        }
        if (adv_i) {
            ++i;
            xi = xn;
            if (i < N) sx = sx + rxn;
            if (i + 1 < N) { xn = w.XO[i + 1]; rxn = RXO(i + 1); }
        }
        if (adv_j) {
            ++j;
            yj = yn;
            if (j < N) sy = sy + ryn;
            if (j + 1 < N) { yn = w.YO[j + 1]; ryn = RXO(jx2); }
            if (j + 2 < N) jx2 = XfromY[j + 2];
        }
    }
#ifdef MPE_PHASE_STATS
    {
        const unsigned long long tw2 = __builtin_readcyclecounter();
        unsigned long long ms = n_steps;
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(ms, off); ms = o > ms ? o : ms; }   // (only the lanes in here take part)
        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) {
            atomicAdd(&g_mpe_walk[0], tw1 - tw0);
            atomicAdd(&g_mpe_walk[1], tw2 - tw1);
            atomicAdd(&g_mpe_walk[2], ms);
            atomicAdd(&g_mpe_walk[3], 1ull);
        }
    }
#endif
    if (!found) return -1;
    const double aplusb = (RXYU + var * LAMBDA * ccs) / NK;
    if (mi == 0) {
        const double min_a = ccx, max_a = aplusb - ccy;
        a = 0.5 * (min_a + max_a);
        b = aplusb - a;
    } else if (ccs != pcs) {
        a = ccx;
        b = ccy;
    } else {
        const double min_a = fmax(ccx, aplusb - pcy);
        const double max_a = fmin(pcx, aplusb - ccy);
        a = 0.5 * (min_a + max_a);
        b = aplusb - a;
    }
    return 1;
}

// index of the first maximum of v over the wave's candidates (the serial scans keep the first with strict '>')
__device__ int wave_first_argmax(double v, int idx, double& vmax)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    vmax = v;
    return idx;
}

// carve the workspace of a fit with up to K components and set up the sorted copies
// The wave version keeps the responsibilities only in x order (R[t] = RXO[ToXO[t]], RYO[s] = RXO[XfromY[s]]) and its
// arrays in the global workspace.  Keeping them in LDS was tried (profiles/r01/clustermatepairs/mpe_lds_share.txt): a fit's chains get
// about three times faster, but 160 KiB per CU hold a quarter of the fits the wave slots do and the kernel as a whole got
// slower at every share of fits moved to LDS; with many fits in flight per CU the L2 hit latency is hidden well enough.
__device__ void init_work(Work& w, int N, int K, int64_t b, const double* x, const double* y, const double* u, const int32_t* to_xo,
                          const int32_t* to_yo, double* d, int* ip, double sd, int first = 0, int step = 1)
{
    w.N = N;
    w.X = x + b; w.Y = y + b; w.U = u + b;
    w.ToXO = to_xo + b; w.ToYO = to_yo + b;
    w.XO = d; d += N;
    w.YO = d; d += N;
    w.R = d; d += (size_t)K * N;
    w.RXO = d; d += (size_t)K * N;
    w.RYO = d; d += (size_t)K * N;
    w.EX = d; d += (size_t)K * N;
    w.SX = d; d += N;
    w.SY = d; d += N;
    w.CX = d; d += 4 * (size_t)N + 4;
    w.CY = d; d += 4 * (size_t)N + 4;
    w.CS = d; d += 4 * (size_t)N + 4;
    w.ka = d; d += 2 * (size_t)N;
    w.kd = d; d += N;
    w.dist = d; d += N;
    w.ic1 = ip;
    w.ic2 = ip + N;
    double* xyu = w.CX;                       // the breakpoint list is not stored by the wave version
    w.TX = w.ToXO;
    w.XfromY = (int*)w.CY;
    w.XYU = xyu;
    w.sd = sd;
    w.iters = 0;
    w.fail = 0;
    for (int j = 0; j < MPE_KMAX; ++j) w.W[j] = w.A[j] = w.B[j] = 0.0;
    for (int i = first; i < N; i += step) {
        w.XO[w.ToXO[i]] = w.X[i];
        w.YO[w.ToYO[i]] = w.Y[i];
        if (step > 1) {
            xyu[i] = w.X[i] + w.Y[i] + w.U[i];
            w.XfromY[w.ToYO[i]] = w.ToXO[i];
        }
    }
}

// fit_state[(p - p0) * MPE_KMAX + K - 1]: 0 = the fit gave no likelihood (the reference `continue`s),
// 1 = bic valid, 2 = the reference would have exited through a DebugCheck
__global__ void k_mpe_fit(mpe_params prm, const int64_t* __restrict__ prob_off, int p0, int first, int n_chunk,
                          const int32_t* __restrict__ order, const double* __restrict__ x, const double* __restrict__ y,
                          const double* __restrict__ u, const int32_t* __restrict__ to_xo, const int32_t* __restrict__ to_yo,
                          const int64_t* __restrict__ wd_off, const int64_t* __restrict__ wi_off, double* __restrict__ wdoubles,
                          int* __restrict__ wints, double* __restrict__ bic, int32_t* __restrict__ fit_state,
                          unsigned long long* __restrict__ iters)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_chunk * MPE_KMAX) return;
    const int K = (int)(t / n_chunk) + 1;                 // K-major: a wave fits one K
    const int q = order[first + t % n_chunk];             // chunk-relative problem, largest first
    const int p = p0 + q;
    const int64_t b = prob_off[p];
    const int N = (int)(prob_off[p + 1] - b);
    const int slot = q * MPE_KMAX + K - 1;
    fit_state[slot] = 0;
    if ((double)N < (double)prm.min_cluster_size || N == 0) return;       // :542-545
    if (K > (N < MPE_KMAX ? N : MPE_KMAX)) return;
    Work w;
    init_work(w, N, K, b, x, y, u, to_xo, to_yo, wdoubles + wd_off[slot], wints + wi_off[slot], prm.fragment_stddev);
    double ll;
    if (expectation_maximization(w, K, ll)) {
        bic[slot] = -2.0 * ll + K * 2.0 * log((double)N);
        fit_state[slot] = 1;
    }
    if (w.fail) fit_state[slot] = 2;
    atomicAdd(iters, (unsigned long long)w.iters);
}

__global__ void k_mpe_final(mpe_params prm, const int64_t* __restrict__ prob_off, int p0, int first, int n_chunk,
                            const int32_t* __restrict__ order, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ u,
                            const int32_t* __restrict__ to_xo, const int32_t* __restrict__ to_yo,
                            const int64_t* __restrict__ wd_off, const int64_t* __restrict__ wi_off, double* __restrict__ wdoubles,
                            int* __restrict__ wints, const double* __restrict__ bic, const int32_t* __restrict__ fit_state,
                            int32_t* __restrict__ n_clusters, uint16_t* __restrict__ member, int32_t* __restrict__ status,
                            unsigned long long* __restrict__ iters)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_chunk) return;
    const int q = order[first + qi];
    const int p = p0 + q;
    const int64_t b = prob_off[p];
    const int N = (int)(prob_off[p + 1] - b);
    n_clusters[p] = 0;
    status[p] = 0;
    for (int i = 0; i < N; ++i) member[b + i] = 0;
    if ((double)N < (double)prm.min_cluster_size || N == 0) return;       // :542-545
    const int kmax = N < MPE_KMAX ? N : MPE_KMAX;
    // model selection :599-606; a fit that tripped a DebugCheck ends the reference's run there
    double min_bic = 0.0;
    bool have = false, failed = false;
    int k_min = 1;
    for (int K = 1; K <= kmax; ++K) {
        const int st = fit_state[q * MPE_KMAX + K - 1];
        if (st == 2) { failed = true; break; }
        if (st != 1) continue;
        const double v = bic[q * MPE_KMAX + K - 1];
        if (!have || v < min_bic) { min_bic = v; k_min = K; have = true; }
    }
    if (failed) { status[p] = 1; return; }
    // the refit reuses the workspace of the problem's largest fit
    const int slot = q * MPE_KMAX + kmax - 1;
    Work w;
    init_work(w, N, kmax, b, x, y, u, to_xo, to_yo, wdoubles + wd_off[slot], wints + wi_off[slot], prm.fragment_stddev);
    double ll;
    if (expectation_maximization(w, k_min, ll)) {
        const double coeff = 1.0 / (w.sd * sqrt(2 * M_PI));                // normalpdf, tools/Common.cpp:61-69
        int emitted = 0;
        for (int j = 0; j < k_min; ++j) {
            int count = 0;
            for (int i = 0; i < N; ++i) {
                const double dist = ((w.A[j] + w.B[j] - w.X[i] - w.Y[i]) - w.U[i]) / w.sd;
                const double prob = coeff * exp(-0.5 * dist * dist) *
                                    exp(-LAMBDA * fmax(0.0, w.X[i] - w.A[j]) - LAMBDA * fmax(0.0, w.Y[i] - w.B[j]));
                const bool in = prob > prm.min_probability;
                w.ic1[i] = in ? 1 : 0;
                count += in ? 1 : 0;
            }
            if ((double)count >= (double)prm.min_cluster_size) {
                for (int i = 0; i < N; ++i)
                    if (w.ic1[i]) member[b + i] |= (uint16_t)(1u << emitted);
                ++emitted;
            }
        }
        n_clusters[p] = emitted;
    }
    status[p] = w.fail;
    atomicAdd(iters, (unsigned long long)w.iters);
}

// ------------------------------------------------------------------------------------------------------
// One wave per problem: the K = 1..kmax fits of the model selection side by side.  A wave per fit (the first wave version,
// profiles/r01/clustermatepairs/) was bound by VALU issue with few lanes at work — its M step kept K of 64 lanes busy, its
// log-likelihood chain one — and five such waves per SIMD already filled the issue slots.  Here lane l = K(K-1)/2 + j owns component j of the fit with K components:
// the M steps of all ten fits of a problem are one pass of up to 55 lanes, their log-likelihood chains and their k-means
// start-ups run in ten lanes at once, and the elementwise E steps go through the fits one after the other with all lanes.
// A fit that has converged drops out; the wave lasts as long as the problem's slowest fit, not the sum of its fits.  The
// KKZ seeds of a fit with K centres are the first K of the sequence for kmax (each seed depends on the earlier ones only),
// the data-dependent arrays (XO, YO, X+Y+U, the k-means point array, the rank maps) are shared by the fits, and the refit of
// the chosen K that the reference runs after the model selection is that fit's state, which is still there.
constexpr int MPE_NCOMP = MPE_KMAX * (MPE_KMAX + 1) / 2;     // 55 component lanes

// -DMPE_PHASE_STATS (diagnostic builds, profiles/microbench/build_variant.sh): wave cycles per phase of k_mpe_problem_wave,
// summed over the problems and printed by mpe_cluster_batch — set-up, KKZ seeds, k-means start-ups, M steps, E steps,
// log-likelihood chains + loop control, model selection + memberships
#ifdef MPE_PHASE_STATS
__device__ unsigned long long g_mpe_phase[8];
#define MPE_STAMP(k)                                                   \
    do {                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        ph_[k] += now_ - t_;                                           \
        t_ = now_;                                                     \
    } while (0)
#else
#define MPE_STAMP(k)
#endif

struct ProblemShared {
    double W[MPE_NCOMP], A[MPE_NCOMP], B[MPE_NCOMP];
    double px[MPE_KMAX], py[MPE_KMAX];
    double c[MPE_KMAX + 1][2 * MPE_KMAX];
    double like[MPE_KMAX + 1], last[MPE_KMAX + 1];
    int active[MPE_KMAX + 1], valid[MPE_KMAX + 1], state[MPE_KMAX + 1], zero[MPE_KMAX + 1], ifault[MPE_KMAX + 1];
    int n_seeds, any_active;
};

struct FitArrays { double *RXO, *SX, *SY, *kd; int *ic1, *ic2; };

__device__ __forceinline__ FitArrays fit_arrays(int N, int K, double* d, int* ip)
{
    FitArrays f;
    f.RXO = d; d += (size_t)K * N;
    f.SX = d; d += N;
    f.SY = d; d += N;
    f.kd = d;
    f.ic1 = ip;
    f.ic2 = ip + N;
    return f;
}

__global__ __launch_bounds__(WV) __attribute__((amdgpu_waves_per_eu(MPE_WPE, MPE_WPE))) void k_mpe_problem_wave(
    mpe_params prm, const int64_t* __restrict__ prob_off, int p0, const int32_t* __restrict__ order, const double* __restrict__ x,
    const double* __restrict__ y, const double* __restrict__ u, const int32_t* __restrict__ to_xo, const int32_t* __restrict__ to_yo,
    const int64_t* __restrict__ wd_off, const int64_t* __restrict__ wi_off, double* __restrict__ wdoubles, int* __restrict__ wints,
    int32_t* __restrict__ n_clusters, uint16_t* __restrict__ member, int32_t* __restrict__ status, unsigned long long* __restrict__ iters,
    long long* __restrict__ iters_by_k, double* __restrict__ ll_by_k)
{
    __shared__ ProblemShared s;
#ifdef MPE_PHASE_STATS
    unsigned long long ph_[8] = {}, t_ = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x;
    const int q = order[blockIdx.x];
    const int p = p0 + q;
    const int64_t b = prob_off[p];
    const int N = (int)(prob_off[p + 1] - b);
    if (lane == 0) { n_clusters[p] = 0; status[p] = 0; }
    for (int i = lane; i < N; i += WV) member[b + i] = 0;
    if ((double)N < (double)prm.min_cluster_size || N == 0) return;       // :542-545
    const int kmax = N < MPE_KMAX ? N : MPE_KMAX;
    // component lane -> (fit, component)
    int myK = 0, myJ = 0;
    for (int K = 1, l = 0; K <= kmax; ++K)
        for (int j = 0; j < K; ++j, ++l)
            if (l == lane) { myK = K; myJ = j; }
    const int myL = myK ? myK * (myK - 1) / 2 + myJ : 0;

    // arrays shared by the fits live behind the largest fit's own
    const int slot_max = q * MPE_KMAX + kmax - 1;
    double* shared_d = wdoubles + wd_off[slot_max] + (size_t)kmax * N + 3 * (size_t)N;
    Work w;
    w.N = N;
    w.X = x + b; w.Y = y + b; w.U = u + b;
    w.ToXO = to_xo + b; w.ToYO = to_yo + b;
    w.TX = w.ToXO;
    w.XO = shared_d; shared_d += N;
    w.YO = shared_d; shared_d += N;
    double* xyu = shared_d; shared_d += N;
    w.XYU = xyu;
    w.ka = shared_d; shared_d += 2 * (size_t)N;
    w.XfromY = (int*)shared_d;
    w.sd = prm.fragment_stddev;
    for (int i = lane; i < N; i += WV) {
        w.XO[w.ToXO[i]] = w.X[i];
        w.YO[w.ToYO[i]] = w.Y[i];
        xyu[i] = w.X[i] + w.Y[i] + w.U[i];
        w.XfromY[w.ToYO[i]] = w.ToXO[i];
        w.ka[i] = w.Y[i];                      // both inserts are at begin(): a = [Y..., X...]
        w.ka[N + i] = w.X[i];
    }
    if (lane <= MPE_KMAX) { s.active[lane] = 0; s.valid[lane] = 0; s.state[lane] = 0; s.zero[lane] = 0; s.ifault[lane] = 0; s.like[lane] = 0.0; s.last[lane] = 0.0; }
    if (lane < MPE_NCOMP) { s.W[lane] = 0.0; s.A[lane] = 0.0; s.B[lane] = 0.0; }
    __syncthreads();

    MPE_STAMP(0);
    // ---- KKZ seeds (:327-386), once for kmax; a fit needs seeds unless K == 1 or K == N
    {
        double best = 0.0;
        int bi = -1;
        for (int i = lane; i < N; i += WV) {
            const double l2 = w.X[i] * w.Y[i];
            if (bi < 0 || l2 > best) { best = l2; bi = i; }
        }
        double vm;
        const int imax = wave_first_argmax(best, bi, vm);
        if (lane == 0) { s.px[0] = w.X[imax]; s.py[0] = w.Y[imax]; s.n_seeds = 1; }
        __syncthreads();
        for (int na = 1; na < kmax; ++na) {
            best = 0.0;
            bi = -1;
            for (int i = lane; i < N; i += WV) {
                double md = (w.X[i] - s.px[0]) * (w.X[i] - s.px[0]) + (w.Y[i] - s.py[0]) * (w.Y[i] - s.py[0]);
                for (int j = 1; j < na; ++j) {
                    const double dj = (w.X[i] - s.px[j]) * (w.X[i] - s.px[j]) + (w.Y[i] - s.py[j]) * (w.Y[i] - s.py[j]);
                    md = fmin(md, dj);
                }
                if (bi < 0 || md > best) { best = md; bi = i; }
            }
            const int idx = wave_first_argmax(best, bi, vm);
            if (vm == 0.0) break;                              // SelectKKZ fails for every K > na
            __syncthreads();
            if (lane == 0) { s.px[na] = w.X[idx]; s.py[na] = w.Y[idx]; s.n_seeds = na + 1; }
            __syncthreads();
        }
        __syncthreads();
    }
    MPE_STAMP(1);
    // ---- start-up of every fit (:388-450): uniform responsibilities, or k-means from the seeds (one fit per lane)
    if (lane >= 1 && lane <= kmax) {
        const int K = lane;
        if (K == 1 || K == N) s.active[K] = 1;
        else if (s.n_seeds >= K) {
            FitArrays f = fit_arrays(N, K, wdoubles + wd_off[q * MPE_KMAX + K - 1], wints + wi_off[q * MPE_KMAX + K - 1]);
            for (int j = 0; j < K; ++j) { s.c[K][j] = s.py[j]; s.c[K][K + j] = s.px[j]; }
            const int ifault = kmns(w.ka, N, s.c[K], K, f.ic1, f.ic2, f.kd, KMEANS_ITER);
            s.ifault[K] = ifault;
            if (ifault == 1 || ifault == 3) s.state[K] = 2;    // DebugCheck(ifault != 1 / != 3)
            else s.active[K] = 1;
        }
    }
    __syncthreads();
    for (int K = 1; K <= kmax; ++K) {
        if (!s.active[K]) continue;
        FitArrays f = fit_arrays(N, K, wdoubles + wd_off[q * MPE_KMAX + K - 1], wints + wi_off[q * MPE_KMAX + K - 1]);
        if (K == 1 || K == N) {
            const double v = 1.0 / K;
            for (size_t t = lane; t < (size_t)K * N; t += WV) f.RXO[t] = v;
        } else {
            for (int i = lane; i < N; i += WV) {
                const int own = f.ic1[i] - 1, ixo = w.TX[i];
                for (int j = 0; j < K; ++j) f.RXO[(size_t)ixo * K + j] = (j == own) ? 1.0 : 0.0;
            }
        }
    }
    __syncthreads();

    MPE_STAMP(2);
    // ---- EM of all fits
    long long my_iters = 0;
    for (;;) {
        // M step: every component of every running fit
        if (myK && s.active[myK]) {
            double a = 0.0, bb = 0.0, nk = 0.0;
            FitArrays f = fit_arrays(N, myK, wdoubles + wd_off[q * MPE_KMAX + myK - 1], wints + wi_off[q * MPE_KMAX + myK - 1]);
            const int rc = max_likelihood_stream(w, f.RXO + myJ, myK, a, bb, nk);
            if (rc < 0) s.state[myK] = 2;                      // the reference would read past the end: DebugCheck
            if (rc > 0) { s.A[myL] = a; s.B[myL] = bb; }
            s.W[myL] = nk / N;
        }
        __syncthreads();
        if (lane >= 1 && lane <= kmax && s.state[lane] == 2) s.active[lane] = 0;
        __syncthreads();
        MPE_STAMP(3);
        // E step, fit after fit: exponents, exp, mixture sum, log — and, from the same registers, the responsibilities
        // W_j e_j / sum of UpdateResponsibilities (:139-181).  The reference updates them after the convergence test; nothing
        // reads them between here and the next M step, and a fit that stops in this iteration never reads them again, so
        // writing them now is the same — without the K*N array of exponentials and the second pass over it.
        for (int K = 1; K <= kmax; ++K) {
            if (!s.active[K]) continue;
            FitArrays f = fit_arrays(N, K, wdoubles + wd_off[q * MPE_KMAX + K - 1], wints + wi_off[q * MPE_KMAX + K - 1]);
            const int l0 = K * (K - 1) / 2;
            bool zero = false;
            for (int i = lane; i < N; i += WV) {
                const double xi = w.X[i], yi = w.Y[i], ui = w.U[i];
                double ex[MPE_KMAX];
#pragma unroll
                for (int j = 0; j < MPE_KMAX; ++j)
                    if (j < K) {
                        const double t = (s.A[l0 + j] + s.B[l0 + j] - xi - yi - ui) / w.sd;
                        ex[j] = -0.5 * (t * t) - LAMBDA * fmax(0.0, xi - s.A[l0 + j]) - LAMBDA * fmax(0.0, yi - s.B[l0 + j]);
                    }
                double maxexp = ex[0];
#pragma unroll
                for (int j = 1; j < MPE_KMAX; ++j)
                    if (j < K) maxexp = fmax(maxexp, ex[j]);
                double sum = 0.0;
#pragma unroll
                for (int j = 0; j < MPE_KMAX; ++j)
                    if (j < K) {
                        ex[j] = exp(ex[j] - maxexp);
                        sum += s.W[l0 + j] * ex[j];
                    }
                if (sum == 0.0) zero = true;
                f.SX[i] = log(sum);
                f.SY[i] = maxexp;
                const int ixo = w.TX[i];
#pragma unroll
                for (int j = 0; j < MPE_KMAX; ++j)
                    if (j < K) f.RXO[(size_t)ixo * K + j] = s.W[l0 + j] * ex[j] / sum;
            }
            if (zero) s.zero[K] = 1;
        }
        __syncthreads();
        MPE_STAMP(4);
        // log-likelihood chains (:96-137) and the loop control of ExpectationMaximization (:455-492), one fit per lane
        if (lane >= 1 && lane <= kmax && s.active[lane]) {
            const int K = lane;
            FitArrays f = fit_arrays(N, K, wdoubles + wd_off[q * MPE_KMAX + K - 1], wints + wi_off[q * MPE_KMAX + K - 1]);
            double LL = 0.0;
            if (s.zero[K]) LL = -DBL_MAX_;
            else {
                int i = 0;
                for (; i + 4 <= N; i += 4) {
                    double l1[4], l2[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) { l1[v] = f.SX[i + v]; l2[v] = f.SY[i + v]; }
#pragma unroll
                    for (int v = 0; v < 4; ++v) LL = LL + l1[v] + l2[v];
                }
                for (; i < N; ++i) LL = LL + f.SX[i] + f.SY[i];
            }
            my_iters += 1;
            const double like = LL, last = s.last[K];
            const bool valid = s.valid[K] != 0;
            if (valid && fabs(like - last) < TOLERANCE) { s.active[K] = 0; s.state[K] = 1; s.like[K] = last; }      // converged: ll = last
            else if (valid && like == -DBL_MAX_) { s.active[K] = 0; }                                              // no likelihood, no failure
            else if (valid && !(like / last < 1.0000001)) { s.active[K] = 0; s.state[K] = 2; }                     // DebugCheck
            else {
                s.last[K] = like;
                s.valid[K] = 1;
                if (s.zero[K]) { s.active[K] = 0; s.state[K] = 2; }                                                // DebugCheck(norm != 0.0), :172
            }
        }
        __syncthreads();
        if (lane == 0) {
            int any = 0;
            for (int K = 1; K <= kmax; ++K) any |= s.active[K];
            s.any_active = any;
        }
        __syncthreads();
        MPE_STAMP(5);
        if (!s.any_active) break;
    }
    const long long fit_iters = my_iters;                  // lane K: iterations of the fit with K components
    if (iters_by_k && lane >= 1 && lane <= MPE_KMAX) {                                                      // diagnostics (DEFUSE_MPE_DUMP_ITERS)
        iters_by_k[(int64_t)p * 12 + lane] = fit_iters;
        ll_by_k[(int64_t)p * 12 + lane] = (lane <= kmax && s.state[lane] == 1) ? s.like[lane] : 0.0;
    }
    for (int off = 32; off > 0; off >>= 1) my_iters += __shfl_xor(my_iters, off);

    // ---- model selection (:599-606); a fit that tripped a DebugCheck ends the reference's run there
    double min_bic = 0.0;
    bool have = false, failed = false;
    int k_min = 1;
    for (int K = 1; K <= kmax; ++K) {
        const int st = s.state[K];
        if (st == 2) { failed = true; break; }
        if (st != 1) continue;
        const double v = -2.0 * s.like[K] + K * 2.0 * log((double)N);
        if (!have || v < min_bic) { min_bic = v; k_min = K; have = true; }
    }
    if (failed) {
        if (lane == 0) { status[p] = 1; atomicAdd(iters, (unsigned long long)my_iters); }
        return;
    }
    // the refit of k_min (:608-615) repeats that fit exactly: its parameters are still in s.A / s.B (its iterations are
    // counted again, as the refit's would be); a fit without a likelihood leaves no clusters.
    const long long refit_iters = __shfl(fit_iters, k_min);
    my_iters += refit_iters;
    if (iters_by_k && lane == 0) {
        iters_by_k[(int64_t)p * 12] = k_min;
        iters_by_k[(int64_t)p * 12 + 11] = refit_iters;
    }
    if (s.state[k_min] == 1) {
        const int l0 = k_min * (k_min - 1) / 2;
        int* flags = wints + wi_off[q * MPE_KMAX + kmax - 1];
        const double coeff = 1.0 / (w.sd * sqrt(2 * M_PI));                // normalpdf, tools/Common.cpp:61-69
        int emitted = 0;
        for (int j = 0; j < k_min; ++j) {
            int count = 0;
            for (int i = lane; i < N; i += WV) {
                const double dist = ((s.A[l0 + j] + s.B[l0 + j] - w.X[i] - w.Y[i]) - w.U[i]) / w.sd;
                const double prob = coeff * exp(-0.5 * dist * dist) *
                                    exp(-LAMBDA * fmax(0.0, w.X[i] - s.A[l0 + j]) - LAMBDA * fmax(0.0, w.Y[i] - s.B[l0 + j]));
                const bool in = prob > prm.min_probability;
                flags[i] = in ? 1 : 0;
                count += in ? 1 : 0;
            }
            for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
            if ((double)count >= (double)prm.min_cluster_size) {
                for (int i = lane; i < N; i += WV)
                    if (flags[i]) member[b + i] |= (uint16_t)(1u << emitted);
                ++emitted;
            }
        }
        if (lane == 0) n_clusters[p] = emitted;
    }
    MPE_STAMP(6);
#ifdef MPE_PHASE_STATS
    if (lane == 0)
        for (int k = 0; k < 7; ++k) atomicAdd(&g_mpe_phase[k], ph_[k]);
#endif
    if (lane == 0) atomicAdd(iters, (unsigned long long)my_iters);
}

template <typename T>
struct DBuf {
    T* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

}  // namespace

extern "C" const char* mpe_last_error(void) { return g_mpe_err.empty() ? g_mpe_err_sharded.c_str() : g_mpe_err.c_str(); }

extern "C" int mpe_cluster_batch(int device, const mpe_params* params, const int64_t* prob_off, int32_t n_problems,
                                 const double* x, const double* y, const double* u, const int32_t* to_xo,
                                 const int32_t* to_yo, int32_t* n_clusters, uint16_t* member, int32_t* status,
                                 mpe_timing* timing)
{
    mpe_timing t{};
    if (!params || n_problems < 0 || (n_problems && !prob_off)) { g_mpe_err = "bad arguments"; return -3; }
    const int64_t n_mp = n_problems ? prob_off[n_problems] : 0;
    t.n_problems = n_problems;
    t.n_mate_pairs = n_mp;
    if (n_problems == 0) { if (timing) *timing = t; return 0; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_mpe_err = "no usable HIP device"; return -2; }
    MPE_HIP(hipSetDevice(device));
    for (int p = 0; p < n_problems; ++p) {
        const int64_t n = prob_off[p + 1] - prob_off[p];
        if (n < 0 || n > 0x7FFFFFF) { g_mpe_err = "problem too large"; return -4; }
    }
    DBuf<int64_t> d_off;
    DBuf<double> d_x, d_y, d_u;
    DBuf<int32_t> d_txo, d_tyo, d_nc, d_status;
    DBuf<uint16_t> d_member;
    DBuf<unsigned long long> d_iters;
    MPE_HIP(d_off.alloc(n_problems + 1));
    MPE_HIP(d_x.alloc(n_mp)); MPE_HIP(d_y.alloc(n_mp)); MPE_HIP(d_u.alloc(n_mp)); MPE_HIP(d_txo.alloc(n_mp)); MPE_HIP(d_tyo.alloc(n_mp));
    MPE_HIP(d_nc.alloc(n_problems)); MPE_HIP(d_status.alloc(n_problems)); MPE_HIP(d_member.alloc(n_mp)); MPE_HIP(d_iters.alloc(1));
    MPE_HIP(hipMemcpy(d_off.p, prob_off, (n_problems + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (n_mp) {
        MPE_HIP(hipMemcpy(d_x.p, x, n_mp * sizeof(double), hipMemcpyHostToDevice));
        MPE_HIP(hipMemcpy(d_y.p, y, n_mp * sizeof(double), hipMemcpyHostToDevice));
        MPE_HIP(hipMemcpy(d_u.p, u, n_mp * sizeof(double), hipMemcpyHostToDevice));
        MPE_HIP(hipMemcpy(d_txo.p, to_xo, n_mp * sizeof(int32_t), hipMemcpyHostToDevice));
        MPE_HIP(hipMemcpy(d_tyo.p, to_yo, n_mp * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    MPE_HIP(hipMemset(d_iters.p, 0, sizeof(unsigned long long)));
    // DEFUSE_MPE_DUMP_ITERS=<file>: per problem twelve int64 — [0] the chosen K, [K] the EM iterations of the fit with K
    // components, [11] those of the refit — for the problems a wave fits (tests compare them with the restatement's)
    const char* dump_iters = getenv("DEFUSE_MPE_DUMP_ITERS");
    DBuf<long long> d_by_k;
    DBuf<double> d_ll_by_k;                  // ... followed in the file by twelve doubles per problem: [K] the log-likelihood fit K ended with
    if (dump_iters) {
        MPE_HIP(d_by_k.alloc((size_t)n_problems * 12));
        MPE_HIP(hipMemset(d_by_k.p, 0, (size_t)n_problems * 12 * sizeof(long long)));
        MPE_HIP(d_ll_by_k.alloc((size_t)n_problems * 12));
        MPE_HIP(hipMemset(d_ll_by_k.p, 0, (size_t)n_problems * 12 * sizeof(double)));
    }
    hipraii::Event e0, e1, e2;               // destroyed on every return, the early ones of MPE_HIP included
    MPE_HIP(e0.create());
    MPE_HIP(e1.create());
    MPE_HIP(e2.create());
    hipraii::Stream s_wave;
    MPE_HIP(s_wave.create(hipStreamNonBlocking));
    // problems with at least this many mate pairs get a wave per fit (DEFUSE_MPE_WAVE_MIN; 0 = all, large = none).  The wave
    // version is the faster one at every size (profiles/microbench/mpe_sweep.sh); the lane version stays as the literal
    // transcription it is checked against
    int64_t wave_min = 0;
    if (const char* e = getenv("DEFUSE_MPE_WAVE_MIN")) wave_min = atoll(e);

    // problems are taken in chunks whose fit workspaces (one per problem and K) fit the budget
    size_t budget = (size_t)32 << 30;       // of 288 GB HBM: one chunk for tens of millions of fragments
    if (const char* e = getenv("DEFUSE_MPE_SCRATCH_MB")) budget = std::max<size_t>(1, (size_t)atoll(e)) << 20;
    auto slot_doubles = [&](int p, int K) -> size_t {
        const int n = (int)(prob_off[p + 1] - prob_off[p]);
        const int kmax = n < MPE_KMAX ? n : MPE_KMAX;
        if (!(n >= params->min_cluster_size && n > 0 && K <= kmax)) return 0;
        return n >= wave_min ? wave_work_doubles(n, K, K == kmax) : work_doubles(n, K);      // the lane version keeps far more per fit
    };
    int p0 = 0;
    while (p0 < n_problems) {
        int p1 = p0;
        size_t bytes = 0;
        while (p1 < n_problems) {
            size_t add = 0;
            for (int K = 1; K <= MPE_KMAX; ++K) add += slot_doubles(p1, K) * sizeof(double) + (slot_doubles(p1, K) ? work_ints((int)(prob_off[p1 + 1] - prob_off[p1])) * sizeof(int) : 0);
            if (p1 > p0 && bytes + add > budget) break;
            bytes += add;
            ++p1;
        }
        const int n_chunk = p1 - p0;
        std::vector<int64_t> wd((size_t)n_chunk * MPE_KMAX + 1, 0), wi((size_t)n_chunk * MPE_KMAX + 1, 0);
        for (int q = 0; q < n_chunk; ++q)
            for (int K = 1; K <= MPE_KMAX; ++K) {
                const size_t sl = (size_t)q * MPE_KMAX + K - 1;
                const size_t nd = slot_doubles(p0 + q, K);
                wd[sl + 1] = wd[sl] + (int64_t)nd;
                wi[sl + 1] = wi[sl] + (int64_t)(nd ? work_ints((int)(prob_off[p0 + q + 1] - prob_off[p0 + q])) : 0);
            }
        std::vector<int32_t> order((size_t)n_chunk);
        for (int q = 0; q < n_chunk; ++q) order[q] = q;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            return prob_off[p0 + a + 1] - prob_off[p0 + a] > prob_off[p0 + b + 1] - prob_off[p0 + b];
        });
        DBuf<int64_t> d_wd, d_wi;
        DBuf<int32_t> d_order, d_state;
        DBuf<double> d_work, d_bic;
        DBuf<int> d_iwork;
        MPE_HIP(d_wd.alloc(wd.size())); MPE_HIP(d_wi.alloc(wi.size())); MPE_HIP(d_order.alloc(order.size()));
        MPE_HIP(d_state.alloc((size_t)n_chunk * MPE_KMAX)); MPE_HIP(d_bic.alloc((size_t)n_chunk * MPE_KMAX));
        MPE_HIP(d_work.alloc((size_t)wd.back())); MPE_HIP(d_iwork.alloc((size_t)wi.back()));
        MPE_HIP(hipMemcpy(d_wd.p, wd.data(), wd.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        MPE_HIP(hipMemcpy(d_wi.p, wi.data(), wi.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        MPE_HIP(hipMemcpy(d_order.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        // the largest problems get a wave per fit, the rest a lane per fit; the two groups are independent and
        // run side by side on two streams
        int n_large = 0;
        while (n_large < n_chunk && prob_off[p0 + order[n_large] + 1] - prob_off[p0 + order[n_large]] >= wave_min) ++n_large;
        const int n_small = n_chunk - n_large;
        t.n_wave_problems += n_large;
        MPE_HIP(hipEventRecord(e0, 0));
        MPE_HIP(hipStreamWaitEvent(s_wave, e0, 0));
        if (n_large)                 // order[0 .. n_large): the problems with a wave of their own, largest first
            hipLaunchKernelGGL(k_mpe_problem_wave, dim3((unsigned)n_large), dim3(WV), 0, s_wave, *params, d_off.p, p0, d_order.p, d_x.p,
                               d_y.p, d_u.p, d_txo.p, d_tyo.p, d_wd.p, d_wi.p, d_work.p, d_iwork.p, d_nc.p, d_member.p, d_status.p,
                               d_iters.p, d_by_k.p, d_ll_by_k.p);
        if (n_small) {
            const int64_t n_fit = (int64_t)n_small * MPE_KMAX;
            hipLaunchKernelGGL(k_mpe_fit, dim3((unsigned)((n_fit + 63) / 64)), dim3(64), 0, 0, *params, d_off.p, p0, n_large, n_small,
                               d_order.p, d_x.p, d_y.p, d_u.p, d_txo.p, d_tyo.p, d_wd.p, d_wi.p, d_work.p, d_iwork.p, d_bic.p, d_state.p,
                               d_iters.p);
            hipLaunchKernelGGL(k_mpe_final, dim3((unsigned)((n_small + 63) / 64)), dim3(64), 0, 0, *params, d_off.p, p0, n_large, n_small,
                               d_order.p, d_x.p, d_y.p, d_u.p, d_txo.p, d_tyo.p, d_wd.p, d_wi.p, d_work.p, d_iwork.p, d_bic.p, d_state.p,
                               d_nc.p, d_member.p, d_status.p, d_iters.p);
        }
        MPE_HIP(hipEventRecord(e2, s_wave));
        MPE_HIP(hipStreamWaitEvent(0, e2, 0));
        MPE_HIP(hipEventRecord(e1, 0));
        MPE_HIP(hipDeviceSynchronize());
        MPE_HIP(hipGetLastError());
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        t.kernel_ms += ms;
        p0 = p1;
    }
    MPE_HIP(hipMemcpy(n_clusters, d_nc.p, n_problems * sizeof(int32_t), hipMemcpyDeviceToHost));
    MPE_HIP(hipMemcpy(status, d_status.p, n_problems * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_mp) MPE_HIP(hipMemcpy(member, d_member.p, n_mp * sizeof(uint16_t), hipMemcpyDeviceToHost));
    unsigned long long it = 0;
    MPE_HIP(hipMemcpy(&it, d_iters.p, sizeof it, hipMemcpyDeviceToHost));

    t.em_iterations = (int64_t)it;
#ifdef MPE_PHASE_STATS
    {
        unsigned long long ph[8] = {};
        MPE_HIP(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_mpe_phase), sizeof ph));
        unsigned long long tot = 0;
        for (int k = 0; k < 7; ++k) tot += ph[k];
        const char* names[7] = {"set-up", "KKZ seeds", "k-means start-ups", "M steps", "E steps", "likelihood chains + control", "selection + memberships"};
        fprintf(stderr, "[mpe phases] wave cycles");
        for (int k = 0; k < 7; ++k) fprintf(stderr, " | %s %.1f %%", names[k], tot ? 100.0 * ph[k] / tot : 0.0);
        fprintf(stderr, " | total %.3g cycles, kernel %.1f ms\n", (double)tot, t.kernel_ms);
        unsigned long long wk[4] = {};
        MPE_HIP(hipMemcpyFromSymbol(wk, HIP_SYMBOL(g_mpe_walk), sizeof wk));
        fprintf(stderr, "[mpe M step] serial sums %.3g cycles, breakpoint walk %.3g cycles, walk steps of the longest lane %.4g per M step (%llu M steps), "
                "%.0f cycles per walk step\n", (double)wk[0], (double)wk[1], wk[3] ? (double)wk[2] / wk[3] : 0.0, wk[3], wk[2] ? (double)wk[1] / wk[2] : 0.0);
        unsigned long long zero[8] = {};
        MPE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mpe_phase), zero, sizeof zero));
        MPE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mpe_walk), zero, sizeof wk));
    }
#endif
    if (dump_iters) {
        std::vector<long long> h((size_t)n_problems * 12);
        MPE_HIP(hipMemcpy(h.data(), d_by_k.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        std::vector<double> hl((size_t)n_problems * 12);
        MPE_HIP(hipMemcpy(hl.data(), d_ll_by_k.p, hl.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(dump_iters, "wb")) {
            fwrite(h.data(), sizeof(long long), h.size(), f);
            fwrite(hl.data(), sizeof(double), hl.size(), f);
            fclose(f);
        }
    }
    for (int p = 0; p < n_problems; ++p) t.n_failed += status[p] != 0;
    if (timing) *timing = t;
    return 0;
}

extern "C" int mpe_cluster_batch_sharded(const int* devices, int32_t n_devices, const mpe_params* params, const int64_t* prob_off,
                                         int32_t n_problems, const double* x, const double* y, const double* u,
                                         const int32_t* to_xo, const int32_t* to_yo, int32_t* n_clusters, uint16_t* member,
                                         int32_t* status, mpe_timing* timing)
{
    if (!devices || n_devices < 1 || !params || n_problems < 0 || (n_problems && !prob_off)) { g_mpe_err = "bad arguments"; return -3; }
    if (n_devices == 1 || n_problems == 0)
        return mpe_cluster_batch(devices[0], params, prob_off, n_problems, x, y, u, to_xo, to_yo, n_clusters, member, status, timing);
    // contiguous shares of about equal mate pair count
    std::vector<int32_t> cut((size_t)n_devices + 1, n_problems);
    cut[0] = 0;
    const int64_t total = prob_off[n_problems];
    for (int k = 1; k < n_devices; ++k) {
        const int64_t want = total / n_devices * k;
        int32_t at = (int32_t)(std::lower_bound(prob_off, prob_off + n_problems + 1, want) - prob_off);
        cut[k] = std::min(std::max(at, cut[k - 1]), n_problems);
    }
    std::vector<mpe_timing> tm((size_t)n_devices);
    std::vector<int> rc((size_t)n_devices, 0);
    std::vector<std::string> err((size_t)n_devices);
    std::vector<std::thread> th;
    for (int k = 0; k < n_devices; ++k)
        th.emplace_back([&, k]() {
            const int32_t p0 = cut[k], p1 = cut[k + 1];
            tm[k] = mpe_timing{};
            if (p1 == p0) return;
            const int64_t base = prob_off[p0];
            std::vector<int64_t> off((size_t)(p1 - p0) + 1);
            for (int32_t p = p0; p <= p1; ++p) off[(size_t)(p - p0)] = prob_off[p] - base;
            rc[k] = mpe_cluster_batch(devices[k], params, off.data(), p1 - p0, x + base, y + base, u + base, to_xo + base, to_yo + base,
                                      n_clusters + p0, member + base, status + p0, &tm[k]);
            if (rc[k]) err[k] = g_mpe_err;
        });
    for (std::thread& t : th) t.join();
    mpe_timing t{};
    t.n_problems = n_problems;
    t.n_mate_pairs = total;
    for (int k = 0; k < n_devices; ++k) {
        if (rc[k]) { g_mpe_err.clear(); g_mpe_err_sharded = "share " + std::to_string(k) + " on device " + std::to_string(devices[k]) + ": " + err[k]; return rc[k]; }
        t.kernel_ms = std::max(t.kernel_ms, tm[k].kernel_ms);
        t.em_iterations += tm[k].em_iterations;
        t.n_failed += tm[k].n_failed;
        t.n_wave_problems += tm[k].n_wave_problems;
    }
    if (timing) *timing = t;
    return 0;
}
