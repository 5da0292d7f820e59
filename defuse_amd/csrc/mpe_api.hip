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
    // wave version: runs of equal XO / YO ("groups": the breakpoint walk of MaxLikelihood compares prefix sums at their ends only)
    int Gx, Gy;                     // number of groups
    const int *gx, *gy;             // group of the element with rank r in x order / rank s in y order
    const double *XG, *YG;          // the groups' coordinates
    int *ic1, *ic2;                 // [N]
    double W[MPE_KMAX], A[MPE_KMAX], B[MPE_KMAX];
    double sd;
    long long iters;
    int fail;
};

// workspace of one fit with K components
__host__ __device__ inline size_t work_doubles(int n, int K) { return (size_t)n * (2 + 4 * K + 2 + 12 + 4) + 16; }
__host__ __device__ inline size_t work_ints(int n) { return (size_t)n * 2 + 8; }
// what the wave kernel uses of a fit's slot: RXO (K*n), SX, SY, kd, the prefix sums at the group ends in x and in y order
// (K*n each); behind the largest fit's own arrays the ones all fits of the problem share (XO, YO, X+Y+U, the k-means point
// array 2n, XfromY, the group maps gx/gy, the group coordinates XG, YG)
__host__ __device__ inline size_t wave_fit_doubles(int n, int K) { return (size_t)n * (3 * K + 3); }
__host__ __device__ inline size_t wave_work_doubles(int n, int K, bool largest) { return wave_fit_doubles(n, K) + (size_t)n * (largest ? 9 : 0) + 16; }

// squared distance of a point to centre l (n = 2; asa136.C's inner loops over j), the point's coordinates in registers
__device__ __forceinline__ double dist2p(double ax, double ay, const double* c, int k, int l)
{
    double s = 0.0;
    const double d1 = ax - c[l - 1];
    s = s + d1 * d1;
    const double d2 = ay - c[l - 1 + k];
    s = s + d2 * d2;
    return s;
}

template <int S>
__device__ void transfer(double ax, double ay, double* c, int k, int* nc, double* an1, double* an2, int* ic1, int* ic2, int i,
                         int l1, int l2)
{
    const double al1 = (double)nc[l1 - 1], alw = al1 - 1.0, al2 = (double)nc[l2 - 1], alt = al2 + 1.0;
    for (int j = 1; j <= 2; ++j) {
        const double aj = j == 1 ? ax : ay;
        c[l1 - 1 + (j - 1) * k] = (c[l1 - 1 + (j - 1) * k] * al1 - aj) / alw;
        c[l2 - 1 + (j - 1) * k] = (c[l2 - 1 + (j - 1) * k] * al2 + aj) / alt;
    }
    nc[l1 - 1] -= 1;
    nc[l2 - 1] += 1;
    an2[l1 - 1] = alw / al1;
    an1[l1 - 1] = 1.0 < alw ? alw / (alw - 1.0) : R8_HUGE;
    an1[l2 - 1] = alt / al2;
    an2[l2 - 1] = alt / (alt + 1.0);
    ic1[(size_t)(i - 1) * S] = l2;
    ic2[(size_t)(i - 1) * S] = l1;
}

// AS 136 with n = 2 (tools/asa136.C:13-336 kmns, :339-566 optra, :569-758 qtran); returns ifault
// S: the stride of the per-point arrays (x / y coordinates a and ay, d, ic1, ic2).  1: a fit's own arrays.  64 (k_mpe_kmeans): the
// arrays of the 64 fits of a wave interleaved point by point — element i of lane l at [i * 64 + l] — so that a wave's load of "its
// fits' point i" is one stretch of memory instead of 64 cache lines (the L1 serves a line per cycle: the one-lane-per-fit kernel
// was bound by exactly that, profiles/r04/clustermatepairs/).
// The per-cluster state (counts, the two factors, the live sets) is the caller's: k entries each — private arrays of a lane end
// up in scratch memory, the wave kernel hands in LDS.
struct KmState { int nc[MPE_KMAX], ncp[MPE_KMAX], itran[MPE_KMAX], live[MPE_KMAX]; double an1[MPE_KMAX], an2[MPE_KMAX]; };

template <int S>
__device__ int kmns(const double* a, const double* ay_, int m, double* c, int k, int* ic1, int* ic2, double* d, int iter, KmState& st)
{
    auto P = [](int i) { return (size_t)i * S; };          // element i of a per-point array
    if (k <= 1 || m <= k) return 3;
    int *nc = st.nc, *ncp = st.ncp, *itran = st.itran, *live = st.live;
    double *an1 = st.an1, *an2 = st.an2;
    for (int i = 1; i <= m; ++i) {
        ic1[P(i - 1)] = 1;
        ic2[P(i - 1)] = 2;
        double dt[2];
        const double ax = a[P(i - 1)], ay = ay_[P(i - 1)];
        for (int il = 1; il <= 2; ++il) dt[il - 1] = dist2p(ax, ay, c, k, il);
        if (dt[1] < dt[0]) {
            ic1[P(i - 1)] = 2;
            ic2[P(i - 1)] = 1;
            const double t = dt[0];
            dt[0] = dt[1];
            dt[1] = t;
        }
        for (int l = 3; l <= k; ++l) {
            const double db = dist2p(ax, ay, c, k, l);
            if (db < dt[1]) {
                if (dt[0] <= db) {
                    dt[1] = db;
                    ic2[P(i - 1)] = l;
                } else {
                    dt[1] = dt[0];
                    ic2[P(i - 1)] = ic1[P(i - 1)];
                    dt[0] = db;
                    ic1[P(i - 1)] = l;
                }
            }
        }
    }
    for (int l = 1; l <= k; ++l) {
        nc[l - 1] = 0;
        for (int j = 1; j <= 2; ++j) c[l - 1 + (j - 1) * k] = 0.0;
    }
    for (int i = 1; i <= m; ++i) {
        const int l = ic1[P(i - 1)];
        nc[l - 1] += 1;
        for (int j = 1; j <= 2; ++j) c[l - 1 + (j - 1) * k] = c[l - 1 + (j - 1) * k] + (j == 1 ? a[P(i - 1)] : ay_[P(i - 1)]);
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
            // the point's coordinates, clusters and distance are fetched one point ahead: nothing that happens to point i
            // touches those of point i + 1 (a transfer writes ic1 / ic2 of its own point only), and the point array is constant
            double axn = a[0], ayn = ay_[0], dn = d[0];
            int l1n = ic1[0], l2n = ic2[0];
            double cx[MPE_KMAX], cy[MPE_KMAX], a2[MPE_KMAX];     // centres and the factor an2, in registers for the scan below
            auto reload = [&]() {
#pragma unroll
                for (int l = 0; l < MPE_KMAX; ++l)
                    if (l < k) { cx[l] = c[l]; cy[l] = c[l + k]; a2[l] = an2[l]; }
            };
            reload();
            for (int i = 1; i <= m; ++i) {
                indx += 1;
                const double ax = axn, ay = ayn;
                double di = dn;
                const int l1 = l1n;
                int l2 = l2n;
                if (i < m) { axn = a[P(i)]; ayn = ay_[P(i)]; dn = d[P(i)]; l1n = ic1[P(i)]; l2n = ic2[P(i)]; }
                const int ll = l2;
                if (1 < nc[l1 - 1]) {
                    if (ncp[l1 - 1] != 0) { di = dist2p(ax, ay, c, k, l1) * an1[l1 - 1]; d[P(i - 1)] = di; }
                    double r2 = dist2p(ax, ay, c, k, l2) * an2[l2 - 1];
                    // the scan over the centres from registers (centres and factors change with a transfer only; the live
                    // set of l2 when l2 does): the same quotients, distances and comparisons in the same order
                    const int live1 = live[l1 - 1];
                    int live2 = live[l2 - 1];
#pragma unroll
                    for (int l = 1; l <= MPE_KMAX; ++l) {
                        if (l <= k && (i < live1 || i < live2) && l != l1 && l != ll) {
                            double dc = 0.0;
                            const double d1 = ax - cx[l - 1];
                            dc = dc + d1 * d1;
                            const double d2 = ay - cy[l - 1];
                            dc = dc + d2 * d2;
                            // the reference compares dc with the quotient rr = r2 / an2[l]: an FP64 division per centre and point,
                            // most of this loop's time.  dc < fl(r2 / a2) is decided by the product wherever it is not within
                            // rounding of r2 (fl(dc * a2) = dc a2 (1 + e2), fl(r2 / a2) = (r2 / a2)(1 + e1), |e| <= 2^-53, a2 > 0: the
                            // two sides differ by a factor inside 1 +- 2^-51); only then is the quotient itself taken
                            const double t = dc * a2[l - 1], margin = r2 * 1e-15;
                            const bool closer = t < r2 - margin ? true : t > r2 + margin ? false : dc < r2 / a2[l - 1];
                            if (closer) {
                                r2 = dc * a2[l - 1];
                                l2 = l;
                                live2 = live[l - 1];
                            }
                        }
                    }
                    if (di <= r2) {
                        ic2[P(i - 1)] = l2;
                    } else {
                        indx = 0;
                        live[l1 - 1] = m + i;
                        live[l2 - 1] = m + i;
                        ncp[l1 - 1] = i;
                        ncp[l2 - 1] = i;
                        transfer<S>(ax, ay, c, k, nc, an1, an2, ic1, ic2, i, l1, l2);
                        reload();
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
                double axn = a[0], ayn = ay_[0], dn = d[0];          // one point ahead, as in optra
                int l1n = ic1[0], l2n = ic2[0];
                for (int i = 1; i <= m; ++i) {
                    icoun += 1;
                    istep += 1;
                    const double ax = axn, ay = ayn;
                    double di = dn;
                    const int l1 = l1n, l2 = l2n;
                    if (i < m) { axn = a[P(i)]; ayn = ay_[P(i)]; dn = d[P(i)]; l1n = ic1[P(i)]; l2n = ic2[P(i)]; }
                    if (1 < nc[l1 - 1]) {
                        if (istep <= ncp[l1 - 1]) { di = dist2p(ax, ay, c, k, l1) * an1[l1 - 1]; d[P(i - 1)] = di; }
                        if (istep < ncp[l1 - 1] || istep < ncp[l2 - 1]) {
                            const double r2 = di / an2[l2 - 1];
                            const double dd = dist2p(ax, ay, c, k, l2);
                            if (dd < r2) {
                                icoun = 0;
                                indx = 0;
                                itran[l1 - 1] = 1;
                                itran[l2 - 1] = 1;
                                ncp[l1 - 1] = istep + m;
                                ncp[l2 - 1] = istep + m;
                                transfer<S>(ax, ay, c, k, nc, an1, an2, ic1, ic2, i, l1, l2);
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
        KmState kst;
        const int ifault = kmns<1>(w.ka, w.ka + N, N, c, K, w.ic1, w.ic2, w.kd, KMEANS_ITER, kst);
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
#define MPE_WPE 2       // waves per SIMD the wave kernel is compiled for (profiles/microbench/em_variants.sh: 2 and 3 within 4 %, 4 and 1 slower)
#endif

// MaxLikelihood (:192-325) for one component in one lane.  The reference builds the whole list of breakpoints — a merge of
// the prefix sums of the responsibilities in x order and in y order, compared at the ends of the runs of equal coordinates —
// and takes the first one with a positive derivative.  Along the list cx + cy never grows (both orders descend) and cs never
// shrinks (responsibilities are not negative), and every operation of the derivative is monotone in floating point as well:
// once positive it stays positive.  So the list is never built:
//   1. one pass of serial sums: NK and RXYU in the caller's order (UpdateMixWeights :183-190 takes the same NK in the same
//      order), and the two prefix sums, each the reference's chain of additions, kept at the group ends only (GA, GB);
//   2. a binary search over the x groups for the last state "x has just arrived at group g" whose breakpoint is not positive.
//      Where y stands at that moment follows from the merge rule: it has passed every group with a smaller sum and, of a run
//      of sums equal to x's, as many groups as x has passed of its own run of that sum (equal sums advance both sides);
//   3. the reference's walk from that state, group by group, until the first positive breakpoint.
// Any state the search accepts is a state the walk goes through with nothing positive before it, so the result is the
// reference's whatever the search does (a run of ties too long to count is simply not accepted); g_mpe_no_jump switches the
// search off (DEFUSE_MPE_NO_JUMP, for tests).  nk receives NK.  Returns 0 = no update (NK == 0), 1 = ok, -1 = the reference
// would read past the end of its list.
#ifdef MPE_PHASE_STATS
__device__ unsigned long long g_mpe_walk[4];      // wave cycles in the M step's serial sums / in search + walk, walk steps (longest lane), calls
#define MPE_WALK_ARG , unsigned long long (&wk_)[4]
#else
#define MPE_WALK_ARG
#endif
__device__ int g_mpe_no_jump;
#ifdef MPE_PATH_STATS
__device__ unsigned long long g_mpe_path[10];      // lanes: no hint, hint accepted, hint positive, windows undecided; waves: with a bisection, all; lanes: hint in front of group 2
#define MPE_PATH(k) atomicAdd(&g_mpe_path[k], 1ull)
#else
#define MPE_PATH(k)
#endif
constexpr int MPE_TIE_SCAN = 4;                   // ties counted one by one up to here, by bisection beyond

__device__ int max_likelihood_groups(const Work& w, const double* RXO_base, double* GA_base, double* GB_base, int stride, int& hint_g,
                                     int& hint_h, double& a, double& b, double& nk MPE_WALK_ARG)
{
#ifdef MPE_PHASE_STATS
    const unsigned long long tw0 = __builtin_readcyclecounter();
    unsigned long long n_steps = 0;
#endif
    // the component's arrays are component-minor: element r at base[r * stride], so that the lanes of a fit, which own its
    // components, touch neighbouring words
    auto RXO = [&](int r) { return RXO_base[(size_t)r * stride]; };
    auto GA = [&](int g) { return GA_base[(size_t)g * stride]; };
    auto GB = [&](int h) { return GB_base[(size_t)h * stride]; };
    const int N = w.N;
    const int* TX = w.TX;
    const int* XfromY = w.XfromY;
    double NK = 0.0, RXYU = 0.0, px = 0.0, py = 0.0;
    int t = 0;
    // Loads in batches so that only the additions are serial, and the batches overlap: a batch's responsibilities are gathered
    // through indices that were fetched while the batch before it was summed (one round trip to memory per batch, not two).
    int ixn[MPE_CHAIN], iyn[MPE_CHAIN];
    if (N >= MPE_CHAIN) {
#pragma unroll
        for (int v = 0; v < MPE_CHAIN; ++v) { ixn[v] = TX[v]; iyn[v] = XfromY[v]; }
    }
    for (; t + MPE_CHAIN <= N; t += MPE_CHAIN) {
        double r[MPE_CHAIN], q[MPE_CHAIN], rx[MPE_CHAIN], ry[MPE_CHAIN];
        int gxc[MPE_CHAIN], gyc[MPE_CHAIN];
#pragma unroll
        for (int v = 0; v < MPE_CHAIN; ++v) {
            r[v] = RXO(ixn[v]); q[v] = w.XYU[t + v];
            rx[v] = RXO(t + v); ry[v] = RXO(iyn[v]);
            gxc[v] = w.gx[t + v]; gyc[v] = w.gy[t + v];
        }
        if (t + 2 * MPE_CHAIN <= N) {
#pragma unroll
            for (int v = 0; v < MPE_CHAIN; ++v) { ixn[v] = TX[t + MPE_CHAIN + v]; iyn[v] = XfromY[t + MPE_CHAIN + v]; }
        }
#pragma unroll
        for (int v = 0; v < MPE_CHAIN; ++v) {
            NK += r[v]; RXYU += r[v] * q[v];
            px += rx[v]; py += ry[v];                        // 0.0 + r == r: the chains start as the reference's do
            GA_base[(size_t)gxc[v] * stride] = px;           // the last element of a group writes last
            GB_base[(size_t)gyc[v] * stride] = py;
        }
    }
    for (; t < N; ++t) {
        const double r = RXO(TX[t]);
        NK += r;
        RXYU += r * w.XYU[t];
        px += RXO(t);
        py += RXO(XfromY[t]);
        GA_base[(size_t)w.gx[t] * stride] = px;
        GB_base[(size_t)w.gy[t] * stride] = py;
    }
    nk = NK;
#ifdef MPE_PHASE_STATS
    const unsigned long long tw1 = __builtin_readcyclecounter();
#endif
    if (NK == 0.0) return 0;
    const double var = w.sd * w.sd;
    const double inv_var = 1.0 / var;
    // the sign of (RXYU - NK*(cx+cy))/var + LAMBDA*cs without the division wherever it is not in doubt: the product with
    // the rounded reciprocal is within a few ulp of the quotient, so an estimate clear of zero by 1e-9 of its terms has the
    // sign of the exact expression; otherwise the expression itself is evaluated
    auto positive = [&](double cx, double cy, double cs) {
        const double q = RXYU - NK * (cx + cy), lc = LAMBDA * cs;
        const double est = q * inv_var + lc;
        if (fabs(est) > 1e-9 * (fabs(q * inv_var) + fabs(lc)) && fabs(est) > 1e-290 && fabs(est) < 1e290) return est > 0;
        return q / var + lc > 0;
    };
    const int Gx = w.Gx, Gy = w.Gy;
    const double* XG = w.XG;
    const double* YG = w.YG;
    double pcx = 0.0, pcy = 0.0, pcs = 0.0, ccx = 0.0, ccy = 0.0, ccs = 0.0;
    int mi = 0;
    bool found = false;
    auto push = [&](double cx, double cy, double cs) {
        if (found) return;
        if (positive(cx, cy, cs)) { found = true; ccx = cx; ccy = cy; ccs = cs; }
        else { pcx = cx; pcy = cy; pcs = cs; ++mi; }
    };
    int g = 0, h = 0;
    if (Gx > 2 && !g_mpe_no_jump) {
        int lo = 0, lo_h = 0, lo_l = 0, hi = Gx;             // lo: accepted arrival (0 = the start), hi: not accepted
        bool searched = false;
        // (a) where the last M step of this component ended: responsibilities move little between EM iterations, so the last
        // arrival but one of its walk is tried first, with everything it needs fetched at once — the four sums in front
        // of x, a window of eight sums around where y stood then.  Accepted: the walk starts there.  Positive: the
        // search below looks in front of it.  Anything the windows do not decide: the search below, unrestricted.
        if (hint_g == 0) searched = true;                    // it ended in front of the second arrival: the walk from the start is short
        else if (hint_g >= 1) {
            const int gm = hint_g < Gx ? hint_g : Gx - 1;
            const int ws = hint_h - 4 > 0 ? (hint_h - 4 < Gy - 8 ? hint_h - 4 : (Gy - 8 > 0 ? Gy - 8 : 0)) : 0;
            double aw[4], bw[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) aw[k] = gm - 1 - k >= 0 ? GA(gm - 1 - k) : -1.0;       // sums are never negative
#pragma unroll
            for (int k = 0; k < 8; ++k) bw[k] = ws + k < Gy ? GB(ws + k) : DBL_MAX_;
            const double xg = XG[gm];
            const double v = aw[0];
            int n_lt = 0, n_eq = 0, nx = 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) { n_lt += bw[k] < v ? 1 : 0; n_eq += bw[k] == v ? 1 : 0; }
#pragma unroll
            for (int k = 1; k < 4; ++k) nx += (nx == k && aw[k] == v) ? 1 : 0;
            bool decided = (ws == 0 || bw[0] < v) && (n_lt + n_eq < 8 || ws + 8 >= Gy);    // the window holds the whole story of v
            if (n_eq > nx && nx == 4 && gm - 5 >= 0) decided = false;                       // x's run of v may be longer than seen
            if (decided) {
                const int l = ws + n_lt;
                const int hm = l + (n_eq < nx ? n_eq : nx);
                bool ok = hm < Gy;
                if (ok) ok = !positive(xg, YG[hm], v);
                if (ok) { lo = gm; lo_h = hm; lo_l = l; searched = true; MPE_PATH(1); }
                else { hi = gm; MPE_PATH(2); }
            } else {
                MPE_PATH(3);
#ifdef MPE_PATH_STATS
                if (!(ws == 0 || bw[0] < v)) MPE_PATH(7);                 // the first sum not below v lies in front of the window
                else if (!(n_lt + n_eq < 8 || ws + 8 >= Gy) && n_eq == 0) atomicAdd(&g_mpe_path[8], 1ull);   // behind it
#endif
            }
        }
        if (hint_g == 0) MPE_PATH(6);
        if (hint_g < 0) MPE_PATH(0);
#ifdef MPE_PATH_STATS
        {
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true), bis = __builtin_amdgcn_ballot_w64(!searched && hi - lo > 1);
            if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) { MPE_PATH(5); if (bis) MPE_PATH(4); }
        }
#endif
        // (b) bisection over the x groups; per candidate the first y group whose sum is not below v by bisection as well
        while (!searched && hi - lo > 1) {
            const int gm = (lo + hi) >> 1;                   // x arrives at group gm >= 1 with cs = the sum at the end of gm - 1
            const double v = GA(gm - 1);
            int l = lo_l, r = Gy;                            // (sums never shrink: not in front of the one found for the
            while (l < r) {                                  // accepted arrival)
                const int m = (l + r) >> 1;
                if (GB(m) < v) l = m + 1;
                else r = m;
            }
            int hm = l;
            if (l < Gy && GB(l) == v) {                      // equal sums advance both sides: y passes as many of them as x has
                int nx = 1, gg = gm - 2;                     // x groups with this sum up to gm - 1: one by one, then by bisection
                while (gg >= 0 && nx <= MPE_TIE_SCAN && GA(gg) == v) { ++nx; --gg; }
                if (nx > MPE_TIE_SCAN) {
                    int a0 = 0, a1 = gg + 1;                 // first group in [0, gg + 1] with the sum v (gg + 1 has it)
                    while (a0 < a1) {
                        const int m = (a0 + a1) >> 1;
                        if (GA(m) < v) a0 = m + 1;
                        else a1 = m;
                    }
                    nx = gm - a0;
                }
                int nb = 1;                                  // y groups with this sum from l on, as far as they matter
                while (nb < nx && nb <= MPE_TIE_SCAN && l + nb < Gy && GB(l + nb) == v) ++nb;
                if (nb < nx && nb > MPE_TIE_SCAN) {
                    int b0 = l + nb, b1 = Gy;                // first group behind l + nb - 1 with a larger sum
                    while (b0 < b1) {
                        const int m = (b0 + b1) >> 1;
                        if (GB(m) > v) b1 = m;
                        else b0 = m + 1;
                    }
                    nb = b0 - l < nx ? b0 - l : nx;
                }
                hm = l + nb;
            }
            bool ok = hm < Gy;                               // otherwise y ran out before: the walk ends there
            if (ok) ok = !positive(XG[gm], YG[hm], v);
            if (ok) { lo = gm; lo_h = hm; lo_l = l; }
            else hi = gm;
        }
        if (lo > 0) {                                        // the last breakpoint before the state, as the walk would have left it
            g = lo; h = lo_h;
            pcx = XG[g]; pcy = YG[h]; pcs = GA(g - 1);
            mi = 1;
        }
    }
    // the walk, group by group, the next two groups of either side on their way; every kind of step advances through the same
    // code so that the lanes of a wave stay together
    double sx = GA(g), sy = GB(h);
    double xi = XG[g], yj = YG[h];
    double xn = 0.0, yn = 0.0, sxn = 0.0, syn = 0.0, xn2 = 0.0, yn2 = 0.0, sxn2 = 0.0, syn2 = 0.0;
    if (g + 1 < Gx) { xn = XG[g + 1]; sxn = GA(g + 1); }
    if (h + 1 < Gy) { yn = YG[h + 1]; syn = GB(h + 1); }
    if (g + 2 < Gx) { xn2 = XG[g + 2]; sxn2 = GA(g + 2); }
    if (h + 2 < Gy) { yn2 = YG[h + 2]; syn2 = GB(h + 2); }
    if (g == 0) push(xi, yj, 0.0);
    // the next M step of this component tries the last arrival but one this walk has seen (the state it started from counts)
    int ga0 = -1, ha0 = 0, ga1 = g, ha1 = h;
    hint_g = found ? 0 : -1;
    hint_h = 0;
    while (!found && g < Gx && h < Gy) {
#ifdef MPE_PHASE_STATS
        ++n_steps;
#endif
        const bool hi = g + 1 < Gx, hj = h + 1 < Gy;
        const bool eq = sx == sy, lt = sx < sy;
        const double cs = (eq || lt) ? sx : sy;
        push(xi, yj, cs);
        const bool second = eq ? (hi && hj) : (lt ? hi : hj);
        if (second) push((eq || lt) ? xn : xi, (eq || !lt) ? yn : yj, cs);
        if (found) { hint_g = ga0 >= 0 ? ga0 : ga1; hint_h = ga0 >= 0 ? ha0 : ha1; }
        if (eq || lt) {
            ga0 = ga1; ha0 = ha1;
            ga1 = g + 1; ha1 = eq ? h + 1 : h;
            ++g;
            xi = xn; sx = sxn; xn = xn2; sxn = sxn2;
            if (g + 2 < Gx) { xn2 = XG[g + 2]; sxn2 = GA(g + 2); }
        }
        if (eq || !lt) {
            ++h;
            yj = yn; sy = syn; yn = yn2; syn = syn2;
            if (h + 2 < Gy) { yn2 = YG[h + 2]; syn2 = GB(h + 2); }
        }
    }
#ifdef MPE_PHASE_STATS
    {   // per lane; the lanes of the fit that runs longest have seen every M step of the wave (the kernel takes the largest)
        const unsigned long long tw2 = __builtin_readcyclecounter();
        unsigned long long ms = n_steps;
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(ms, off); ms = o > ms ? o : ms; }   // (only the lanes in here take part)
        wk_[0] += tw1 - tw0;
        wk_[1] += tw2 - tw1;
        wk_[2] += ms;
        wk_[3] += 1;
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
    double like[MPE_KMAX + 1], last[MPE_KMAX + 1];
    int active[MPE_KMAX + 1], valid[MPE_KMAX + 1], state[MPE_KMAX + 1], zero[MPE_KMAX + 1], ifault[MPE_KMAX + 1];
    int n_seeds, any_active;
};

// What the three kernels of a wave problem hand to each other (round 4: the k-means start-ups left the wave kernel).
//   k_mpe_seed         one wave per problem: the arrays all fits share, the runs of equal coordinates, the KKZ seeds
//   k_mpe_kmeans       ONE LANE PER FIT, 64 fits of similar size per wave: AS 136 from the seeds.  Inside the wave kernel the
//                      start-ups ran in at most nine of 64 lanes while the others waited (18 % of its wave cycles)
//   k_mpe_problem_wave one wave per problem: responsibilities from the k-means assignments, EM of all fits, model selection
struct ProblemSeeds {
    double px[MPE_KMAX], py[MPE_KMAX];
    int n_seeds;            // -1: the problem is not fitted at all (fewer mate pairs than a cluster needs)
    int Gx, Gy, pad_;
};
constexpr int KM_NOT_RUN = -1;          // ifault slot of a fit that needs no k-means (K = 1, K = N) or has no seeds

// the arrays shared by the fits of a problem live behind the largest fit's own
struct SharedArrays { double *XO, *YO, *XYU, *ka, *XG, *YG; int *XfromY, *gmap; };
__device__ __forceinline__ SharedArrays shared_arrays(double* shared_d, int N)
{
    SharedArrays a;
    a.XO = shared_d; shared_d += N;
    a.YO = shared_d; shared_d += N;
    a.XYU = shared_d; shared_d += N;
    a.ka = shared_d; shared_d += 2 * (size_t)N;
    a.XfromY = (int*)shared_d; shared_d += N;
    a.gmap = (int*)shared_d; shared_d += N;             // gx [N], gy [N]
    a.XG = shared_d; shared_d += N;
    a.YG = shared_d;
    return a;
}

struct FitArrays { double *RXO, *SX, *SY, *kd, *GA, *GB; int *ic1, *ic2; };

__device__ __forceinline__ FitArrays fit_arrays(int N, int K, double* d, int* ip)
{
    FitArrays f;
    f.RXO = d; d += (size_t)K * N;
    f.SX = d; d += N;
    f.SY = d; d += N;
    f.kd = d; d += N;
    f.GA = d; d += (size_t)K * N;
    f.GB = d;
    f.ic1 = ip;
    f.ic2 = ip + N;
    return f;
}

// ---- k_mpe_seed: set-up of the shared arrays and the KKZ seeds (:327-386), one wave per problem
__global__ __launch_bounds__(WV) void k_mpe_seed(mpe_params prm, const int64_t* __restrict__ prob_off, int p0, const int32_t* __restrict__ order,
                                                const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ u,
                                                const int32_t* __restrict__ to_xo, const int32_t* __restrict__ to_yo, const int64_t* __restrict__ wd_off,
                                                double* __restrict__ wdoubles, ProblemSeeds* __restrict__ seeds)
{
    __shared__ double s_px[MPE_KMAX], s_py[MPE_KMAX];
    __shared__ int s_n_seeds;
    const int lane = threadIdx.x;
    const int q = order[blockIdx.x];
    const int p = p0 + q;
    const int64_t b = prob_off[p];
    const int N = (int)(prob_off[p + 1] - b);
    ProblemSeeds& out = seeds[blockIdx.x];
    if ((double)N < (double)prm.min_cluster_size || N == 0) {       // :542-545
        if (lane == 0) out.n_seeds = -1;
        return;
    }
    const int kmax = N < MPE_KMAX ? N : MPE_KMAX;
    const SharedArrays a = shared_arrays(wdoubles + wd_off[q * MPE_KMAX + kmax - 1] + wave_fit_doubles(N, kmax), N);
    const double *X = x + b, *Y = y + b, *U = u + b;
    const int32_t *TX = to_xo + b, *TY = to_yo + b;
    for (int i = lane; i < N; i += WV) {
        a.XO[TX[i]] = X[i];
        a.YO[TY[i]] = Y[i];
        a.XYU[i] = X[i] + Y[i] + U[i];
        a.XfromY[TY[i]] = TX[i];
    }
    __syncthreads();
    // the runs of equal coordinates in x order and in y order, once per problem: group of every rank, coordinate of every group
    auto runs = [&](const double* O, int* grp, double* coord) {
        int cnt = 0;
        for (int base = 0; base < N; base += WV) {
            const int r = base + lane;
            const bool first = r < N && (r == 0 || O[r] != O[r - 1]);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(first);
            const int gi = cnt + __popcll(m & ((1ull << lane) - 1ull)) + (first ? 1 : 0) - 1;
            if (r < N) {
                grp[r] = gi;
                if (first) coord[gi] = O[r];
            }
            cnt += __popcll(m);
        }
        return cnt;
    };
    const int Gx = runs(a.XO, a.gmap, a.XG);
    const int Gy = runs(a.YO, a.gmap + N, a.YG);
    // KKZ seeds, once for kmax; a fit needs seeds unless K == 1 or K == N
    double best = 0.0;
    int bi = -1;
    for (int i = lane; i < N; i += WV) {
        const double l2 = X[i] * Y[i];
        if (bi < 0 || l2 > best) { best = l2; bi = i; }
    }
    double vm;
    const int imax = wave_first_argmax(best, bi, vm);
    if (lane == 0) { s_px[0] = X[imax]; s_py[0] = Y[imax]; s_n_seeds = 1; }
    __syncthreads();
    for (int na = 1; na < kmax; ++na) {
        best = 0.0;
        bi = -1;
        for (int i = lane; i < N; i += WV) {
            double md = (X[i] - s_px[0]) * (X[i] - s_px[0]) + (Y[i] - s_py[0]) * (Y[i] - s_py[0]);
            for (int j = 1; j < na; ++j) {
                const double dj = (X[i] - s_px[j]) * (X[i] - s_px[j]) + (Y[i] - s_py[j]) * (Y[i] - s_py[j]);
                md = fmin(md, dj);
            }
            if (bi < 0 || md > best) { best = md; bi = i; }
        }
        const int idx = wave_first_argmax(best, bi, vm);
        if (vm == 0.0) break;                              // SelectKKZ fails for every K > na
        __syncthreads();
        if (lane == 0) { s_px[na] = X[idx]; s_py[na] = Y[idx]; s_n_seeds = na + 1; }
        __syncthreads();
    }
    __syncthreads();
    if (lane < MPE_KMAX) { out.px[lane] = lane < s_n_seeds ? s_px[lane] : 0.0; out.py[lane] = lane < s_n_seeds ? s_py[lane] : 0.0; }
    if (lane == 0) { out.n_seeds = s_n_seeds; out.Gx = Gx; out.Gy = Gy; out.pad_ = 0; }
}

// ---- k_mpe_kmeans: the AS 136 start-up of every fit that needs one (:388-450), one lane per fit.  Block (Kidx, gi): the fits with
// K = 2 + Kidx centres of the 64 problems of size rank gi * 64 .. gi * 64 + 63 (the problems are sorted by size): as alike as 64
// independent sequential k-means runs get.  Per-lane cluster state in LDS (private arrays would live in scratch memory), the lane
// stride an odd number of 8-byte words.  The per-point arrays of the block's 64 fits — coordinates, d, ic1, ic2 — are INTERLEAVED
// in a scratch slab of the block (kmns<64>): every lane walks its own points, and point i of all lanes is one 512-byte stretch.
struct KmLane { KmState st; double c[2 * MPE_KMAX]; double pad_; };
static_assert(sizeof(KmLane) % 16 == 8, "lane stride: an odd number of 8-byte words");
constexpr int KM_ROW_DOUBLES = WV * 4;      // per point and block: 64 x (ax, ay, d: 8 B each; ic1, ic2: 4 B each) = 256 doubles
__global__ __launch_bounds__(WV) void k_mpe_kmeans(const int64_t* __restrict__ prob_off, int p0, int n_large, int n_groups, const int32_t* __restrict__ order,
                                                  const double* __restrict__ x, const double* __restrict__ y, const int64_t* __restrict__ wi_off,
                                                  int* __restrict__ wints, const ProblemSeeds* __restrict__ seeds, const int64_t* __restrict__ km_off,
                                                  int64_t km_total, double* __restrict__ km_scratch, int* __restrict__ ifault_out)
{
    __shared__ KmLane s_lane[WV];
    const int lane = threadIdx.x;
    const int Kidx = MPE_KMAX - 2 - (int)blockIdx.x / n_groups, gi = blockIdx.x % n_groups;      // the longest runs (most centres) first
    const int K = 2 + Kidx, r = gi * WV + lane;
    const bool in_range = r < n_large;
    const int q = order[in_range ? r : n_large - 1];
    const int p = p0 + q;
    const int64_t b = prob_off[p];
    const int N = in_range ? (int)(prob_off[p + 1] - b) : 0;
    int nmax = N;
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    const ProblemSeeds& sd = seeds[in_range ? r : n_large - 1];
    const int kmax = N < MPE_KMAX ? N : MPE_KMAX;
    const bool run = in_range && sd.n_seeds >= K && K <= kmax && K != N;
    int result = KM_NOT_RUN;
    if (run) {
        double* slab = km_scratch + ((size_t)Kidx * (size_t)km_total + (size_t)km_off[gi]) * KM_ROW_DOUBLES;
        double* ax = slab + lane;                                   // kmns' a = [Y..., X...] (both inserts at begin(), :402-410)
        double* ay = slab + (size_t)nmax * WV + lane;
        double* d = slab + 2 * (size_t)nmax * WV + lane;
        int* ic1 = reinterpret_cast<int*>(slab + 3 * (size_t)nmax * WV) + lane;
        int* ic2 = ic1 + (size_t)nmax * WV;
        const double *X = x + b, *Y = y + b;
        for (int i = 0; i < N; ++i) {
            ax[(size_t)i * WV] = Y[i];
            ay[(size_t)i * WV] = X[i];
        }
        KmLane& L = s_lane[lane];
        for (int j = 0; j < K; ++j) { L.c[j] = sd.py[j]; L.c[K + j] = sd.px[j]; }
        result = kmns<WV>(ax, ay, N, L.c, K, ic1, ic2, d, KMEANS_ITER, L.st);
        int* out = wints + wi_off[q * MPE_KMAX + K - 1];            // the fit's ic1, where the wave kernel reads the assignments
        for (int i = 0; i < N; ++i) out[i] = ic1[(size_t)i * WV];
    }
    if (in_range) ifault_out[(int64_t)r * (MPE_KMAX + 1) + K] = result;
}

__global__ __launch_bounds__(WV) __attribute__((amdgpu_waves_per_eu(MPE_WPE, MPE_WPE))) void k_mpe_problem_wave(
    mpe_params prm, const int64_t* __restrict__ prob_off, int p0, const int32_t* __restrict__ order, const double* __restrict__ x,
    const double* __restrict__ y, const double* __restrict__ u, const int32_t* __restrict__ to_xo, const int32_t* __restrict__ to_yo,
    const int64_t* __restrict__ wd_off, const int64_t* __restrict__ wi_off, double* __restrict__ wdoubles, int* __restrict__ wints,
    int32_t* __restrict__ n_clusters, uint16_t* __restrict__ member, int32_t* __restrict__ status, unsigned long long* __restrict__ iters,
    long long* __restrict__ iters_by_k, double* __restrict__ ll_by_k, const ProblemSeeds* __restrict__ seeds, const int* __restrict__ km_ifault)
{
    __shared__ ProblemShared s;
#ifdef MPE_PHASE_STATS
    unsigned long long ph_[8] = {}, t_ = __builtin_readcyclecounter(), wk_[4] = {};
#define MPE_WALK_PASS , wk_
#else
#define MPE_WALK_PASS
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

    // arrays shared by the fits live behind the largest fit's own; k_mpe_seed has filled them
    const int slot_max = q * MPE_KMAX + kmax - 1;
    const SharedArrays sa = shared_arrays(wdoubles + wd_off[slot_max] + wave_fit_doubles(N, kmax), N);
    const ProblemSeeds& sd = seeds[blockIdx.x];
    Work w;
    w.N = N;
    w.X = x + b; w.Y = y + b; w.U = u + b;
    w.ToXO = to_xo + b; w.ToYO = to_yo + b;
    w.TX = w.ToXO;
    w.XO = sa.XO;
    w.YO = sa.YO;
    w.XYU = sa.XYU;
    w.ka = sa.ka;
    w.XfromY = sa.XfromY;
    w.gx = sa.gmap; w.gy = sa.gmap + N; w.XG = sa.XG; w.YG = sa.YG;
    w.Gx = sd.Gx;
    w.Gy = sd.Gy;
    w.sd = prm.fragment_stddev;
    if (lane <= MPE_KMAX) { s.active[lane] = 0; s.valid[lane] = 0; s.state[lane] = 0; s.zero[lane] = 0; s.ifault[lane] = 0; s.like[lane] = 0.0; s.last[lane] = 0.0; }
    if (lane < MPE_NCOMP) { s.W[lane] = 0.0; s.A[lane] = 0.0; s.B[lane] = 0.0; }
    if (lane == 0) s.n_seeds = sd.n_seeds;
    __syncthreads();
    MPE_STAMP(0);
    MPE_STAMP(1);
    // ---- start-up of every fit (:388-450): uniform responsibilities, or the k-means assignments k_mpe_kmeans left
    if (lane >= 1 && lane <= kmax) {
        const int K = lane;
        if (K == 1 || K == N) s.active[K] = 1;
        else if (s.n_seeds >= K) {
            const int ifault = km_ifault[(int64_t)blockIdx.x * (MPE_KMAX + 1) + K];
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
    int hint_g = -1, hint_h = 0;               // where this lane's component ended its last M step (the next one starts its search there)
    for (;;) {
        // M step: every component of every running fit
        if (myK && s.active[myK]) {
            double a = 0.0, bb = 0.0, nk = 0.0;
            FitArrays f = fit_arrays(N, myK, wdoubles + wd_off[q * MPE_KMAX + myK - 1], wints + wi_off[q * MPE_KMAX + myK - 1]);
            const int rc = max_likelihood_groups(w, f.RXO + myJ, f.GA + myJ, f.GB + myJ, myK, hint_g, hint_h, a, bb, nk MPE_WALK_PASS);
            if (rc < 0) s.state[myK] = 2;                      // the reference would read past the end: DebugCheck
            if (rc > 0) { s.A[myL] = a; s.B[myL] = bb; }
            s.W[myL] = nk / N;
        }
        __syncthreads();
        if (lane >= 1 && lane <= kmax && s.state[lane] == 2) s.active[lane] = 0;
        __syncthreads();
        MPE_STAMP(3);
        // E step: exponents, exp, mixture sum, log — and, from the same registers, the responsibilities
        // W_j e_j / sum of UpdateResponsibilities (:139-181).  The reference updates them after the convergence test; nothing
        // reads them between here and the next M step, and a fit that stops in this iteration never reads them again, so
        // writing them now is the same — without the K*N array of exponentials and the second pass over it.
        // (mate pairs in the outer loop: X, Y, U and the rank are fetched once for all fits)
        {
            unsigned zero_mask = 0;
            for (int i = lane; i < N; i += WV) {
                const double xi = w.X[i], yi = w.Y[i], ui = w.U[i];
                const int ixo = w.TX[i];
                for (int K = 1; K <= kmax; ++K) {
                    if (!s.active[K]) continue;
                    FitArrays f = fit_arrays(N, K, wdoubles + wd_off[q * MPE_KMAX + K - 1], wints + wi_off[q * MPE_KMAX + K - 1]);
                    const int l0 = K * (K - 1) / 2;
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
                    if (sum == 0.0) zero_mask |= 1u << K;
                    f.SX[i] = log(sum);
                    f.SY[i] = maxexp;
#pragma unroll
                    for (int j = 0; j < MPE_KMAX; ++j)
                        if (j < K) f.RXO[(size_t)ixo * K + j] = s.W[l0 + j] * ex[j] / sum;
                }
            }
            for (int K = 1; K <= kmax; ++K)
                if (zero_mask >> K & 1u) s.zero[K] = 1;
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
                for (; i + 8 <= N; i += 8) {
                    double l1[8], l2[8];
#pragma unroll
                    for (int v = 0; v < 8; ++v) { l1[v] = f.SX[i + v]; l2[v] = f.SY[i + v]; }
#pragma unroll
                    for (int v = 0; v < 8; ++v) LL = LL + l1[v] + l2[v];
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
    for (int k = 0; k < 4; ++k) {
        unsigned long long m = wk_[k];
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(m, off); m = o > m ? o : m; }
        if (lane == 0) atomicAdd(&g_mpe_walk[k], m);
    }
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
    constexpr size_t MAX_SHARES = 4;
    hipraii::Stream s_wave, s_share[MAX_SHARES - 1];
    hipraii::Event e_seed, e_share[MAX_SHARES - 1];
    MPE_HIP(s_wave.create(hipStreamNonBlocking));
    MPE_HIP(e_seed.create());
    for (size_t k = 0; k + 1 < MAX_SHARES; ++k) {
        MPE_HIP(s_share[k].create(hipStreamNonBlocking));
        MPE_HIP(e_share[k].create());
    }
    // problems with at least this many mate pairs get a wave per fit (DEFUSE_MPE_WAVE_MIN; 0 = all, large = none).  The wave
    // version is the faster one at every size (profiles/microbench/mpe_sweep.sh); the lane version stays as the literal
    // transcription it is checked against
    int64_t wave_min = 0;
    if (const char* e = getenv("DEFUSE_MPE_WAVE_MIN")) wave_min = atoll(e);
    {   // DEFUSE_MPE_NO_JUMP=1: the M step walks its breakpoints from the first one (tests compare the two)
        const char* e = getenv("DEFUSE_MPE_NO_JUMP");
        const int no_jump = e && *e && *e != '0';
        MPE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mpe_no_jump), &no_jump, sizeof no_jump));
    }

    // problems are taken in chunks whose fit workspaces (one per problem and K) fit the budget
    size_t budget = (size_t)64 << 30;       // of 288 GB HBM: one chunk for tens of millions of fragments
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
        DBuf<ProblemSeeds> d_seeds;
        DBuf<int> d_km_ifault;
        DBuf<int64_t> d_km_off;
        DBuf<double> d_km_scratch;
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
        if (n_large) {               // order[0 .. n_large): the problems with a wave of their own, largest first
            MPE_HIP(d_seeds.alloc((size_t)n_large));
            MPE_HIP(d_km_ifault.alloc((size_t)n_large * (MPE_KMAX + 1)));
            // scratch slabs of the k-means blocks: group gi (64 problems of neighbouring size ranks) needs its largest problem's
            // number of points, for each of the nine K
            const int n_groups = (n_large + WV - 1) / WV;
            std::vector<int64_t> km_off((size_t)n_groups + 1, 0);
            for (int gi = 0; gi < n_groups; ++gi) {
                const int first = order[(size_t)gi * WV];              // sorted by size: the group's largest
                km_off[(size_t)gi + 1] = km_off[(size_t)gi] + (prob_off[p0 + first + 1] - prob_off[p0 + first]);
            }
            MPE_HIP(d_km_off.alloc(km_off.size()));
            MPE_HIP(hipMemcpy(d_km_off.p, km_off.data(), km_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            MPE_HIP(d_km_scratch.alloc((size_t)(MPE_KMAX - 1) * (size_t)km_off.back() * KM_ROW_DOUBLES));
            hipLaunchKernelGGL(k_mpe_seed, dim3((unsigned)n_large), dim3(WV), 0, s_wave, *params, d_off.p, p0, d_order.p, d_x.p, d_y.p, d_u.p, d_txo.p,
                               d_tyo.p, d_wd.p, d_work.p, d_seeds.p);
            // The k-means kernel ends in a long tail: its longest waves (64 start-ups of the largest problems in lockstep)
            // run for tens of milliseconds with most of the card idle.  So the problems go in shares on streams of their own
            // (k-means, then EM, each): the groups of the largest problems in the first, all smaller ones in the last, whose
            // short k-means is over at once and whose EM waves fill the card under the first share's tail.
            std::vector<int> cuts{0};                                  // group boundaries of the shares
            if (n_groups >= 8) {
                std::string spec = "0.3";
                if (const char* e = getenv("DEFUSE_MPE_SHARES")) spec = e;          // cut points as fractions of the groups, comma separated
                for (size_t at = 0; at < spec.size() && cuts.size() < MAX_SHARES;) {
                    const size_t comma = std::min(spec.find(',', at), spec.size());
                    const double f = atof(spec.substr(at, comma - at).c_str());
                    const int g = (int)(n_groups * std::min(1.0, std::max(0.0, f)));
                    if (g > cuts.back() && g < n_groups) cuts.push_back(g);
                    at = comma + 1;
                }
            }
            cuts.push_back(n_groups);
            auto share = [&](hipStream_t st, int g0, int g1) {         // groups [g0, g1) of the sorted problems
                if (g0 >= g1) return;
                const int r0 = g0 * WV, n = std::min(n_large, g1 * WV) - r0;
                hipLaunchKernelGGL(k_mpe_kmeans, dim3((unsigned)((g1 - g0) * (MPE_KMAX - 1))), dim3(WV), 0, st, d_off.p, p0, n, g1 - g0, d_order.p + r0, d_x.p,
                                   d_y.p, d_wi.p, d_iwork.p, d_seeds.p + r0, d_km_off.p + g0, (int64_t)km_off.back(), d_km_scratch.p,
                                   d_km_ifault.p + (size_t)r0 * (MPE_KMAX + 1));
                hipLaunchKernelGGL(k_mpe_problem_wave, dim3((unsigned)n), dim3(WV), 0, st, *params, d_off.p, p0, d_order.p + r0, d_x.p,
                                   d_y.p, d_u.p, d_txo.p, d_tyo.p, d_wd.p, d_wi.p, d_work.p, d_iwork.p, d_nc.p, d_member.p, d_status.p,
                                   d_iters.p, d_by_k.p, d_ll_by_k.p, d_seeds.p + r0, d_km_ifault.p + (size_t)r0 * (MPE_KMAX + 1));
            };
            const int n_shares = (int)cuts.size() - 1;                 // the last share on s_wave itself, the others on streams of their own
            if (n_shares > 1) MPE_HIP(hipEventRecord(e_seed, s_wave));
            for (int k = 0; k + 1 < n_shares; ++k) {
                MPE_HIP(hipStreamWaitEvent(s_share[k], e_seed, 0));
                share(s_share[k], cuts[(size_t)k], cuts[(size_t)k + 1]);
            }
            share(s_wave, cuts[(size_t)n_shares - 1], n_groups);
            for (int k = 0; k + 1 < n_shares; ++k) {
                MPE_HIP(hipEventRecord(e_share[k], s_share[k]));
                MPE_HIP(hipStreamWaitEvent(s_wave, e_share[k], 0));
            }
        }
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
        fprintf(stderr, "[mpe M step] serial sums %.3g cycles, breakpoint search + walk %.3g cycles, walk steps of the longest lane %.4g per M step (%llu M steps), "
                "%.0f cycles per walk step\n", (double)wk[0], (double)wk[1], wk[3] ? (double)wk[2] / wk[3] : 0.0, wk[3], wk[2] ? (double)wk[1] / wk[2] : 0.0);
        unsigned long long zero[10] = {};
        MPE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mpe_phase), zero, sizeof ph));
        MPE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mpe_walk), zero, sizeof wk));
#ifdef MPE_PATH_STATS
        unsigned long long pa[10] = {};
        MPE_HIP(hipMemcpyFromSymbol(pa, HIP_SYMBOL(g_mpe_path), sizeof pa));
        fprintf(stderr, "[mpe M step] lanes: no hint %llu, hint in front of group 2 %llu, hint accepted %llu, hint positive %llu, windows undecided %llu (in front %llu, behind %llu); waves with a bisection %llu of %llu\n",
                pa[0], pa[6], pa[1], pa[2], pa[3], pa[7], pa[8], pa[4], pa[5]);
        MPE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mpe_path), zero, sizeof pa));
#endif
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
