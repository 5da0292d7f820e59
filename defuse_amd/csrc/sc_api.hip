// sc_api.hip — greedy set cover on gfx950 (include/defuse_sc.h), replacing SetCover() of
// tools/setcover.cpp:30-110.  Integer/byte work: radix sort + label propagation are HBM-bound, the
// per-component greedy is latency-bound; nothing here is shaped into a GEMM.
#include <hip/hip_runtime.h>

#include "hip_raii.hpp"
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/defuse_sc.h"

namespace {

std::string g_err;
constexpr int SMALL_MAX = 32;   // components with more clusters than this get a whole wave

#define SC_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[256];                                                                         \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_err = b_;                                                                           \
            return -2;                                                                            \
        }                                                                                         \
    } while (0)

// occurrence k of the input -> (element key, owning cluster)
__global__ void k_occurrences(const int64_t* __restrict__ cluster_off, int32_t n_clusters, int32_t* __restrict__ occ_cluster)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    for (int64_t k = cluster_off[c]; k < cluster_off[c + 1]; ++k) occ_cluster[k] = c;
}

// e2c_off[e] = first position of key e in the sorted keys (lower bound); e2c_off[max_element+1] = n
__global__ void k_lower_bounds(const int32_t* __restrict__ keys, int64_t n, int32_t max_element, int64_t* __restrict__ off)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > (int64_t)max_element + 1) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < e) lo = mid + 1; else hi = mid;
    }
    off[e] = lo;
}

__global__ void k_iota(int32_t* __restrict__ a, int32_t n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}

// hook: every fragment pulls the labels of its clusters down to their minimum
__global__ void k_hook(const int64_t* __restrict__ e2c_off, const int32_t* __restrict__ e2c, int32_t max_element,
                       int32_t* __restrict__ label, int* __restrict__ changed)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > max_element) return;
    const int64_t b = e2c_off[e], en = e2c_off[e + 1];
    if (en - b < 2) return;
    int m = 0x7FFFFFFF;
    for (int64_t k = b; k < en; ++k) m = min(m, label[e2c[k]]);
    for (int64_t k = b; k < en; ++k) {
        const int c = e2c[k];
        if (label[c] > m) {
            atomicMin(&label[c], m);
            *changed = 1;
        }
    }
}

// compress: label[c] = label[label[c]] until it is a root
__global__ void k_compress(int32_t* __restrict__ label, int32_t n, int* __restrict__ changed)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int l = label[c];
    int r = l;
    while (label[r] != r) r = label[r];
    if (r != l) {
        label[c] = r;
        *changed = 1;
    }
}

// segment heads of the (label-sorted) cluster list -> component table
__global__ void k_mark_heads(const int32_t* __restrict__ sorted_label, int32_t n, int32_t* __restrict__ head_flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head_flag[i] = (i == 0 || sorted_label[i] != sorted_label[i - 1]) ? 1 : 0;
}
__global__ void k_scatter_heads(const int32_t* __restrict__ head_flag, const int32_t* __restrict__ head_rank, int32_t n,
                                int32_t* __restrict__ comp_begin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head_flag[i]) comp_begin[head_rank[i]] = i;
    if (i == n - 1) comp_begin[head_rank[i] + head_flag[i]] = n;   // end sentinel
}

struct Greedy {
    const int64_t* cluster_off;
    const int32_t* elements;
    const int64_t* e2c_off;
    const int32_t* e2c;
    const int32_t* comp_clusters;   // clusters sorted by (component, index)
    const int32_t* comp_begin;
    int32_t n_components;
    int32_t* size;
    int32_t* seq;
    int32_t* owner;
};

// assignment step of tools/setcover.cpp:76-106 for the chosen cluster
__device__ __forceinline__ void take_cluster(const Greedy& g, int best, int& counter)
{
    for (int64_t k = g.cluster_off[best]; k < g.cluster_off[best + 1]; ++k) {
        const int e = g.elements[k];
        if (g.owner[e] >= 0) continue;
        g.owner[e] = best;
        for (int64_t q = g.e2c_off[e]; q < g.e2c_off[e + 1]; ++q) {     // ascending cluster index, duplicates included
            const int c2 = g.e2c[q];
            g.size[c2] -= 1;
            g.seq[c2] = ++counter;
        }
    }
}

// one lane per small component
__global__ void k_greedy_small(Greedy g)
{
    const int comp = blockIdx.x * blockDim.x + threadIdx.x;
    if (comp >= g.n_components) return;
    const int b = g.comp_begin[comp], e = g.comp_begin[comp + 1];
    if (e - b > SMALL_MAX) return;
    int counter = 0;
    for (int i = b; i < e; ++i) {
        const int c = g.comp_clusters[i];
        g.size[c] = (int)(g.cluster_off[c + 1] - g.cluster_off[c]);
        g.seq[c] = ++counter;                        // initial arrival order: ascending cluster index
    }
    for (;;) {
        int best = -1, bs = -1, bq = -1;
        for (int i = b; i < e; ++i) {
            const int c = g.comp_clusters[i];
            const int s = g.size[c], q = g.seq[c];
            if (s > bs || (s == bs && q > bq)) { best = c; bs = s; bq = q; }
        }
        if (bs <= 0) break;
        take_cluster(g, best, counter);
    }
}

// one wave per large component: the arg-max over (size, arrival) is a wave reduction, the assignment
// step stays sequential (its order defines the arrival stamps)
__global__ void k_greedy_large(Greedy g, const int32_t* __restrict__ large_list, int32_t n_large)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n_large) return;
    const int comp = large_list[wave];
    const int b = g.comp_begin[comp], e = g.comp_begin[comp + 1];
    for (int i = b + lane; i < e; i += 64) {
        const int c = g.comp_clusters[i];
        g.size[c] = (int)(g.cluster_off[c + 1] - g.cluster_off[c]);
        g.seq[c] = i - b + 1;
    }
    int counter = e - b;
    __threadfence_block();
    for (;;) {
        long long key = -1;                           // (size << 32 | seq), cluster carried alongside
        int best = -1;
        for (int i = b + lane; i < e; i += 64) {
            const int c = g.comp_clusters[i];
            const long long k = ((long long)g.size[c] << 32) | (unsigned)g.seq[c];
            if (k > key) { key = k; best = c; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const long long ok = __shfl_xor(key, d, 64);
            const int ob = __shfl_xor(best, d, 64);
            if (ok > key) { key = ok; best = ob; }
        }
        if ((key >> 32) <= 0) break;                  // wave-uniform after the butterfly
        if (lane == 0) take_cluster(g, best, counter);
        __threadfence_block();
        counter = __shfl(counter, 0, 64);
    }
}

__global__ void k_collect_large(const int32_t* __restrict__ comp_begin, int32_t n_components, int32_t* __restrict__ list,
                                int32_t* __restrict__ n_large)
{
    const int comp = blockIdx.x * blockDim.x + threadIdx.x;
    if (comp >= n_components) return;
    if (comp_begin[comp + 1] - comp_begin[comp] > SMALL_MAX) list[atomicAdd(n_large, 1)] = comp;
}

template <typename T>
struct Buf {
    T* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

}  // namespace

extern "C" const char* sc_last_error(void) { return g_err.c_str(); }

extern "C" int sc_prepare(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -2;
    if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) return -2;
    return 0;
}

extern "C" int sc_cover(int device, const int64_t* cluster_off, const int32_t* elements, int32_t n_clusters,
                        int32_t max_element, int32_t* owner, sc_timing* timing)
{
    sc_timing t{};
    if (n_clusters < 0 || max_element < -1 || (n_clusters && !cluster_off)) { g_err = "bad arguments"; return -3; }
    const int64_t n_occ = n_clusters ? cluster_off[n_clusters] : 0;
    for (int64_t k = 0; k < n_occ; ++k)
        if (elements[k] < 0 || elements[k] > max_element) { g_err = "element out of range"; return -3; }
    if (n_occ >= ((int64_t)1 << 31)) { g_err = "more than 2^31-1 cluster lines"; return -4; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_err = "no usable HIP device"; return -2; }
    SC_HIP(hipSetDevice(device));
    for (int64_t e = 0; e <= max_element; ++e) owner[e] = -1;
    if (n_clusters == 0 || n_occ == 0) { if (timing) *timing = t; return 0; }

    hipraii::Event ev[4];                 // destroyed on every return
    for (auto& e : ev) SC_HIP(e.create());
    Buf<int64_t> d_coff, d_e2c_off;
    Buf<int32_t> d_el, d_occ_cluster, d_keys_sorted, d_e2c, d_label, d_label_sorted, d_idx, d_comp_clusters, d_flag, d_rank,
        d_comp_begin, d_size, d_seq, d_owner, d_large, d_nlarge;
    Buf<int> d_changed;
    Buf<uint8_t> d_tmp;
    const int nel = max_element + 1;
    SC_HIP(d_coff.alloc(n_clusters + 1)); SC_HIP(d_el.alloc(n_occ)); SC_HIP(d_occ_cluster.alloc(n_occ));
    SC_HIP(d_keys_sorted.alloc(n_occ)); SC_HIP(d_e2c.alloc(n_occ)); SC_HIP(d_e2c_off.alloc(nel + 2));
    SC_HIP(d_label.alloc(n_clusters)); SC_HIP(d_label_sorted.alloc(n_clusters)); SC_HIP(d_idx.alloc(n_clusters));
    SC_HIP(d_comp_clusters.alloc(n_clusters)); SC_HIP(d_flag.alloc(n_clusters)); SC_HIP(d_rank.alloc(n_clusters));
    SC_HIP(d_comp_begin.alloc(n_clusters + 2)); SC_HIP(d_size.alloc(n_clusters)); SC_HIP(d_seq.alloc(n_clusters));
    SC_HIP(d_owner.alloc(nel)); SC_HIP(d_large.alloc(n_clusters)); SC_HIP(d_nlarge.alloc(1)); SC_HIP(d_changed.alloc(1));
    SC_HIP(hipMemcpy(d_coff.p, cluster_off, (n_clusters + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    SC_HIP(hipMemcpy(d_el.p, elements, n_occ * sizeof(int32_t), hipMemcpyHostToDevice));
    SC_HIP(hipMemset(d_owner.p, 0xFF, nel * sizeof(int32_t)));
    const int B = 256;
    auto grid = [&](int64_t n) { return dim3((unsigned)((n + B - 1) / B)); };

    // 1. fragment -> clusters index: stable radix sort of the occurrences by fragment keeps each list in
    //    ascending cluster order (the order tools/setcover.cpp:47-60 builds it in)
    SC_HIP(hipEventRecord(ev[0]));
    hipLaunchKernelGGL(k_occurrences, grid(n_clusters), dim3(B), 0, 0, d_coff.p, n_clusters, d_occ_cluster.p);
    size_t tmp_bytes = 0, need = 0;
    int bits = 1;
    while ((1ll << bits) <= max_element) ++bits;
    SC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_el.p, d_keys_sorted.p, d_occ_cluster.p, d_e2c.p, (int)n_occ, 0, bits));
    tmp_bytes = need;
    int cbits = 1;
    while ((1ll << cbits) < n_clusters) ++cbits;
    SC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_label.p, d_label_sorted.p, d_idx.p, d_comp_clusters.p, n_clusters, 0, cbits));
    tmp_bytes = std::max(tmp_bytes, need);
    SC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_flag.p, d_rank.p, n_clusters));
    tmp_bytes = std::max(tmp_bytes, need);
    SC_HIP(d_tmp.alloc(tmp_bytes));
    need = tmp_bytes;
    SC_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, need, d_el.p, d_keys_sorted.p, d_occ_cluster.p, d_e2c.p, (int)n_occ, 0, bits));
    hipLaunchKernelGGL(k_lower_bounds, grid(nel + 1), dim3(B), 0, 0, d_keys_sorted.p, n_occ, max_element, d_e2c_off.p);
    SC_HIP(hipEventRecord(ev[1]));

    // 2. connected components of the cluster/fragment graph
    hipLaunchKernelGGL(k_iota, grid(n_clusters), dim3(B), 0, 0, d_label.p, n_clusters);
    int iterations = 0;
    for (;;) {
        int changed = 0;
        SC_HIP(hipMemset(d_changed.p, 0, sizeof(int)));
        hipLaunchKernelGGL(k_hook, grid(nel), dim3(B), 0, 0, d_e2c_off.p, d_e2c.p, max_element, d_label.p, d_changed.p);
        hipLaunchKernelGGL(k_compress, grid(n_clusters), dim3(B), 0, 0, d_label.p, n_clusters, d_changed.p);
        SC_HIP(hipMemcpy(&changed, d_changed.p, sizeof(int), hipMemcpyDeviceToHost));
        ++iterations;
        if (!changed) break;
        if (iterations > 10000) { g_err = "connected components did not converge"; return -2; }
    }
    // clusters grouped by component, ascending index inside a component (stable sort by label)
    hipLaunchKernelGGL(k_iota, grid(n_clusters), dim3(B), 0, 0, d_idx.p, n_clusters);
    need = tmp_bytes;
    SC_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, need, d_label.p, d_label_sorted.p, d_idx.p, d_comp_clusters.p, n_clusters, 0, cbits));
    hipLaunchKernelGGL(k_mark_heads, grid(n_clusters), dim3(B), 0, 0, d_label_sorted.p, n_clusters, d_flag.p);
    need = tmp_bytes;
    SC_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, need, d_flag.p, d_rank.p, n_clusters));
    hipLaunchKernelGGL(k_scatter_heads, grid(n_clusters), dim3(B), 0, 0, d_flag.p, d_rank.p, n_clusters, d_comp_begin.p);
    int last_rank = 0, last_flag = 0;
    SC_HIP(hipMemcpy(&last_rank, d_rank.p + (n_clusters - 1), sizeof(int), hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(&last_flag, d_flag.p + (n_clusters - 1), sizeof(int), hipMemcpyDeviceToHost));
    const int n_components = last_rank + last_flag;
    SC_HIP(hipEventRecord(ev[2]));

    // 3. greedy, independently per component
    Greedy g{d_coff.p, d_el.p, d_e2c_off.p, d_e2c.p, d_comp_clusters.p, d_comp_begin.p, n_components, d_size.p, d_seq.p, d_owner.p};
    SC_HIP(hipMemset(d_nlarge.p, 0, sizeof(int)));
    hipLaunchKernelGGL(k_collect_large, grid(n_components), dim3(B), 0, 0, d_comp_begin.p, n_components, d_large.p, d_nlarge.p);
    hipLaunchKernelGGL(k_greedy_small, grid(n_components), dim3(B), 0, 0, g);
    int n_large = 0;
    SC_HIP(hipMemcpy(&n_large, d_nlarge.p, sizeof(int), hipMemcpyDeviceToHost));
    if (n_large > 0)
        hipLaunchKernelGGL(k_greedy_large, grid((int64_t)n_large * 64), dim3(B), 0, 0, g, d_large.p, n_large);
    SC_HIP(hipEventRecord(ev[3]));
    SC_HIP(hipDeviceSynchronize());
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpy(owner, d_owner.p, nel * sizeof(int32_t), hipMemcpyDeviceToHost));
    (void)hipEventElapsedTime(&t.build_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&t.components_ms, ev[1], ev[2]);
    (void)hipEventElapsedTime(&t.greedy_ms, ev[2], ev[3]);
    (void)hipEventElapsedTime(&t.total_ms, ev[0], ev[3]);
    t.n_components = n_components;
    t.n_large = n_large;
    t.cc_iterations = iterations;
    if (timing) *timing = t;
    return 0;
}
