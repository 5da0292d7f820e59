// hc_api.hip — batched average-linkage clustering on gfx950 (include/defuse_hc.h), replacing
// HierarchicalClusterer::DoClustering of tools/HierarchicalClusterer.cpp:46-140.
//
// One workgroup per distance table (tables of different gene pairs are independent).  The table lives in
// HBM/L2 as a full symmetric matrix D plus a matrix S of entry stamps, which stand in for the position of a
// distance inside the reference's multiset_of<double> (equal keys keep their order of entry).  Per merge:
// a workgroup-wide arg-min over (distance, stamp) of the live pairs, one strided pass that rewrites the row
// and column of the merged cluster, and O(1) bookkeeping (member lists are linked, so concatenation is a
// pointer move).  FP64, no contraction: the merged distance is (s1*d1 + s2*d2) / (s1+s2) as the reference
// computes it.  Latency/HBM bound integer-and-compare work; nothing here is a contraction.
#include <hip/hip_runtime.h>

#include "hip_raii.hpp"

#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/defuse_hc.h"

#pragma clang fp contract(off)

namespace {

std::string g_err;

#define HC_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[256];                                                                         \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_err = b_;                                                                           \
            return -2;                                                                            \
        }                                                                                         \
    } while (0)

struct HcTable {
    int64_t in_off;      // into the caller's distances
    int64_t mat_off;     // into D and S (n*n entries)
    int64_t slot_off;    // into the slot arrays (6 n int32)
    int64_t item_off;    // into members / cluster_of
    double  threshold;
    int32_t n;
    int32_t pad_;
};

struct Best {
    double   d;
    uint32_t s;
    int32_t  i, j;
};

__device__ __forceinline__ bool better(double d, uint32_t s, double bd, uint32_t bs)
{
    return d < bd || (d == bd && s < bs);
}

constexpr int HC_THREADS = 256;

__global__ __launch_bounds__(HC_THREADS) void k_hc(const HcTable* __restrict__ tabs, const double* __restrict__ dist_in,
                                                   double* __restrict__ D_all, uint32_t* __restrict__ S_all,
                                                   int32_t* __restrict__ slots_all, int32_t* __restrict__ members,
                                                   int32_t* __restrict__ cluster_of, int32_t* __restrict__ n_clusters,
                                                   int32_t* __restrict__ n_merges)
{
    const HcTable t = tabs[blockIdx.x];
    const int n = t.n;
    const int tid = threadIdx.x;
    if (n == 0) {
        if (tid == 0) { n_clusters[blockIdx.x] = 0; n_merges[blockIdx.x] = 0; }
        return;
    }
    const double* in = dist_in + t.in_off;
    double*   D = D_all + t.mat_off;
    uint32_t* S = S_all + t.mat_off;
    int32_t* ord  = slots_all + t.slot_off;      // the reference's cluster index of the slot
    int32_t* size = ord + n;
    int32_t* head = size + n;
    int32_t* tail = head + n;
    int32_t* next = tail + n;
    int32_t* act  = next + n;                    // live slots, any order (ties are decided by stamps)

    __shared__ Best s_best[HC_THREADS / 64];
    __shared__ Best s_win;
    __shared__ int s_m, s_pos;

    // distances[i][j], j > i, mirrored; stamp = position in the reference's insertion order (:63-66)
    const uint32_t nn = (uint32_t)n * (uint32_t)n;
    for (uint32_t idx = tid; idx < nn; idx += HC_THREADS) {
        const uint32_t i = idx / n, j = idx - i * n;
        const uint32_t a = i < j ? i : j, b = i < j ? j : i;
        D[idx] = (i == j) ? 0.0 : in[(size_t)a * n + b];
        S[idx] = a * n + b;
    }
    for (int i = tid; i < n; i += HC_THREADS) {
        ord[i] = i; size[i] = 1; head[i] = i; tail[i] = i; next[i] = -1; act[i] = i;
    }
    if (tid == 0) s_m = n;
    __syncthreads();

    int merges = 0;
    for (;;) {
        const int m = s_m;
        if (m < 2) break;
        // arg-min over the live pairs
        Best b{0.0, 0xffffffffu, -1, -1};
        const uint32_t mm = (uint32_t)m * (uint32_t)m;
        for (uint32_t idx = tid; idx < mm; idx += HC_THREADS) {
            const uint32_t p = idx / m, q = idx - p * m;
            if (p >= q) continue;
            const int i = act[p], j = act[q];
            const double d = D[(size_t)i * n + j];
            const uint32_t s = S[(size_t)i * n + j];
            if (b.i < 0 || better(d, s, b.d, b.s)) b = Best{d, s, i, j};
        }
        for (int off = 32; off > 0; off >>= 1) {
            Best o;
            o.d = __shfl_xor(b.d, off);
            o.s = __shfl_xor(b.s, off);
            o.i = __shfl_xor(b.i, off);
            o.j = __shfl_xor(b.j, off);
            if (o.i >= 0 && (b.i < 0 || better(o.d, o.s, b.d, b.s))) b = o;
        }
        if ((tid & 63) == 0) s_best[tid >> 6] = b;
        __syncthreads();
        if (tid == 0) {
            Best w = s_best[0];
            for (int k = 1; k < HC_THREADS / 64; ++k) {
                const Best o = s_best[k];
                if (o.i >= 0 && (w.i < 0 || better(o.d, o.s, w.d, w.s))) w = o;
            }
            s_win = w;
        }
        __syncthreads();
        const Best w = s_win;
        if (!(w.d < t.threshold)) break;                           // :79
        const int first  = ord[w.i] < ord[w.j] ? w.i : w.j;         // SortedPair: smaller cluster index first
        const int second = ord[w.i] < ord[w.j] ? w.j : w.i;
        const double sf = (double)size[first], ss = (double)size[second], sm = sf + ss;
        const uint32_t stamp0 = nn + (uint32_t)merges * 2u * (uint32_t)n;
        for (int p = tid; p < m; p += HC_THREADS) {
            const int c = act[p];
            if (c == second) { s_pos = p; continue; }
            if (c == first) continue;
            const double d1 = D[(size_t)first * n + c], d2 = D[(size_t)second * n + c];
            const double dn = (sf * d1 + ss * d2) / sm;            // :107
            const uint32_t st = stamp0 + (uint32_t)ord[c];         // entry order = list order of the live clusters
            D[(size_t)first * n + c] = dn; D[(size_t)c * n + first] = dn;
            S[(size_t)first * n + c] = st; S[(size_t)c * n + first] = st;
        }
        __syncthreads();
        if (tid == 0) {
            next[tail[first]] = head[second];                       // members of first, then of second (:89-91)
            tail[first] = tail[second];
            size[first] += size[second];
            ord[first] = n + merges;                                // index of the merged cluster (:88)
            act[s_pos] = act[m - 1];
            s_m = m - 1;
        }
        ++merges;
        __syncthreads();
    }

    // result order: the reference's index list = live clusters by ascending index (:124-139)
    const int m = s_m;
    for (int p = tid; p < m; p += HC_THREADS) {
        const int c = act[p];
        int rank = 0, start = 0;
        for (int q = 0; q < m; ++q) {
            const int o = act[q];
            if (ord[o] < ord[c]) { ++rank; start += size[o]; }
        }
        int k = 0;
        for (int e = head[c]; e >= 0; e = next[e], ++k) {
            members[t.item_off + start + k] = e;
            cluster_of[t.item_off + start + k] = rank;
        }
    }
    if (tid == 0) { n_clusters[blockIdx.x] = m; n_merges[blockIdx.x] = merges; }
}

template <class T>
struct DevMem {
    T* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" const char* hc_last_error(void) { return g_err.c_str(); }

extern "C" int hc_cluster_batch(int device, int32_t n_tables, const int32_t* n_items, const int64_t* dist_off,
                                const double* distances, const double* thresholds,
                                int32_t* members, int32_t* cluster_of, int32_t* n_clusters, hc_timing* timing)
{
    const double t_begin = now_ms();
    if (timing) *timing = hc_timing{0, 0, 0, 0};
    if (n_tables < 0 || (n_tables > 0 && (!n_items || !dist_off || !thresholds || !n_clusters))) {
        g_err = "hc_cluster_batch: null argument";
        return -1;
    }
    if (n_tables == 0) return 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        g_err = "hc_cluster_batch: no such HIP device";
        return -2;
    }
    HC_HIP(hipSetDevice(device));

    std::vector<HcTable> tabs(n_tables);
    int64_t mat = 0, slots = 0, items = 0, in_end = 0;
    for (int32_t p = 0; p < n_tables; ++p) {
        const int64_t n = n_items[p];
        if (n < 0 || n > HC_MAX_ITEMS || dist_off[p] < 0) {
            g_err = "hc_cluster_batch: table size out of range";
            return -1;
        }
        tabs[p] = HcTable{dist_off[p], mat, slots, items, thresholds[p], (int32_t)n, 0};
        mat += n * n; slots += 6 * n; items += n;
        if (dist_off[p] + n * n > in_end) in_end = dist_off[p] + n * n;
    }
    if ((items > 0 && (!members || !cluster_of)) || (in_end > 0 && !distances)) {
        g_err = "hc_cluster_batch: null argument";
        return -1;
    }

    DevMem<HcTable> d_tabs; DevMem<double> d_in, d_D; DevMem<uint32_t> d_S;
    DevMem<int32_t> d_slots, d_members, d_cluster_of, d_ncl, d_nmerge;
    HC_HIP(d_tabs.alloc(n_tables)); HC_HIP(d_in.alloc(in_end)); HC_HIP(d_D.alloc(mat)); HC_HIP(d_S.alloc(mat));
    HC_HIP(d_slots.alloc(slots)); HC_HIP(d_members.alloc(items)); HC_HIP(d_cluster_of.alloc(items));
    HC_HIP(d_ncl.alloc(n_tables)); HC_HIP(d_nmerge.alloc(n_tables));
    HC_HIP(hipMemcpy(d_tabs.p, tabs.data(), sizeof(HcTable) * n_tables, hipMemcpyHostToDevice));
    if (in_end) HC_HIP(hipMemcpy(d_in.p, distances, sizeof(double) * in_end, hipMemcpyHostToDevice));
    const double t_up = now_ms();

    hipraii::Event e0, e1;                 // destroyed on every return
    HC_HIP(e0.create()); HC_HIP(e1.create());
    HC_HIP(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_hc, dim3(n_tables), dim3(HC_THREADS), 0, nullptr, d_tabs.p, d_in.p, d_D.p, d_S.p, d_slots.p,
                       d_members.p, d_cluster_of.p, d_ncl.p, d_nmerge.p);
    HC_HIP(hipGetLastError());
    HC_HIP(hipEventRecord(e1, nullptr));
    HC_HIP(hipEventSynchronize(e1));
    float kms = 0;
    HC_HIP(hipEventElapsedTime(&kms, e0, e1));

    std::vector<int32_t> nmerge(n_tables);
    HC_HIP(hipMemcpy(n_clusters, d_ncl.p, sizeof(int32_t) * n_tables, hipMemcpyDeviceToHost));
    HC_HIP(hipMemcpy(nmerge.data(), d_nmerge.p, sizeof(int32_t) * n_tables, hipMemcpyDeviceToHost));
    if (items) {
        HC_HIP(hipMemcpy(members, d_members.p, sizeof(int32_t) * items, hipMemcpyDeviceToHost));
        HC_HIP(hipMemcpy(cluster_of, d_cluster_of.p, sizeof(int32_t) * items, hipMemcpyDeviceToHost));
    }
    if (timing) {
        int64_t tot = 0;
        for (int32_t v : nmerge) tot += v;
        timing->upload_ms = (float)(t_up - t_begin);
        timing->kernel_ms = kms;
        timing->total_ms = (float)(now_ms() - t_begin);
        timing->n_merges = (int32_t)tot;
    }
    return 0;
}
