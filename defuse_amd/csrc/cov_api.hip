// cov_api.hip — the sampling kernel of calccov (include/defuse_cov.h) for gfx950.
//
// One lane per fragment.  A transcript holds length x density samples (a few dozen at the pipeline's 0.01 per base,
// scripts/config.txt:105), so a lane walks its transcript's samples in index order — which is the output order — and
// tests the three ranges of tools/calccov.cpp:171-213.  Pass 1 counts, an exclusive scan places every fragment's
// output, pass 2 writes.  Integer compares plus two FP64 divisions per split sample (IEEE division and floor, the same
// values the reference computes); HBM-bound and small next to the text parsing around it.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/defuse_cov.h"
#include "../../include/defuse_dsa.h"

namespace {

thread_local std::string g_cov_err;

int cov_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_cov_err = buf;
    return code;
}

template <typename T>
struct DBuf {
    T* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};
struct Stream {
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Stream()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
    }
};

#define COV_HIP(call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return cov_fail(DSA_E_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Ranges { int us, ue, as[2], ae[2]; };
__device__ __forceinline__ Ranges ranges_of(const cov_fragment& f, int trim, int anchor)
{
    Ranges r;
    r.us = min(f.start[0] + trim, f.start[1] + trim);                 // unseqStart / unseqEnd, tools/calccov.cpp:171-172
    r.ue = max(f.end[0] - trim, f.end[1] - trim);
    for (int e = 0; e < 2; ++e) {
        r.as[e] = f.start[e] + anchor;                                  // anchoredStart / anchoredEnd, :190-191
        r.ae[e] = f.end[e] - anchor + 1;
    }
    return r;
}

__global__ void k_cov_count(const cov_fragment* __restrict__ frags, int64_t n, const int64_t* __restrict__ off,
                            const int32_t* __restrict__ pos, int trim, int anchor, int64_t* __restrict__ n_len,
                            int64_t* __restrict__ n_split)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cov_fragment f = frags[i];
    const Ranges r = ranges_of(f, trim, anchor);
    int64_t cl = 0, cs = 0;
    for (int64_t k = off[f.ref]; k < off[f.ref + 1]; ++k) {
        const int p = pos[k];
        cl += (p >= r.us && p <= r.ue) ? 1 : 0;
        cs += (p >= r.as[0] && p <= r.ae[0]) ? 1 : 0;
        cs += (p >= r.as[1] && p <= r.ae[1]) ? 1 : 0;
    }
    n_len[i] = cl;
    n_split[i] = cs;
}

__global__ void k_cov_write(const cov_fragment* __restrict__ frags, int64_t n, const int64_t* __restrict__ off,
                            const int32_t* __restrict__ pos, int trim, int anchor, const int64_t* __restrict__ at_len,
                            const int64_t* __restrict__ at_split, int32_t* __restrict__ length_idx, int32_t* __restrict__ length_val,
                            int32_t* __restrict__ split_idx, double* __restrict__ split_pos, double* __restrict__ split_min)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cov_fragment f = frags[i];
    const Ranges r = ranges_of(f, trim, anchor);
    const int fragment_length = max(f.end[0], f.end[1]) - min(f.start[0], f.start[1]);     // :174
    int64_t wl = at_len[i], ws = at_split[i];
    const int64_t lo = off[f.ref], hi = off[f.ref + 1];
    for (int64_t k = lo; k < hi; ++k) {
        const int p = pos[k];
        if (p >= r.us && p <= r.ue) {
            length_idx[wl] = (int32_t)k;
            length_val[wl] = fragment_length;
            ++wl;
        }
    }
    for (int e = 0; e < 2; ++e) {
        const int start = f.start[e], end = f.end[e];
        const double pos_range = end - start + 1.0 - 2.0 * anchor;                               // CalculateSplitPos :236-242
        const double min_range = floor(0.5 * (end - start + 1.0 - 2.0 * anchor));                // CalculateSplitMin :244-250
        for (int64_t k = lo; k < hi; ++k) {
            const int p = pos[k];
            if (p >= r.as[e] && p <= r.ae[e]) {
                const double pos_value = fmax(0.0, (double)(p - start - anchor));
                const double min_value = fmax(0.0, (double)min(p - start - anchor, end + 1 - p - anchor));
                split_idx[ws] = (int32_t)k;
                split_pos[ws] = pos_value / pos_range;
                split_min[ws] = min_value / min_range;
                ++ws;
            }
        }
    }
}

}  // namespace

extern "C" {

const char* cov_last_error(void) { return g_cov_err.c_str(); }

int cov_sample_batch(int device, const int64_t* ref_sample_off, int32_t n_refs, const int32_t* sample_pos,
                     const cov_fragment* fragments, int64_t n_fragments, int32_t trim_length, int32_t split_min_anchor,
                     int32_t* length_idx, int32_t* length_val, int64_t length_cap, int64_t* n_length,
                     int32_t* split_idx, double* split_pos, double* split_min, int64_t split_cap, int64_t* n_split,
                     cov_timing* timing)
{
    if (n_refs < 0 || n_fragments < 0 || !n_length || !n_split) return cov_fail(DSA_E_ARG, "bad arguments");
    *n_length = *n_split = 0;
    if (timing) *timing = cov_timing{};
    if (n_fragments == 0) return DSA_OK;
    if (!ref_sample_off || !fragments) return cov_fail(DSA_E_ARG, "null pointer with non-zero size");
    const int64_t n_samples = ref_sample_off[n_refs];
    for (int32_t r = 0; r < n_refs; ++r)
        if (ref_sample_off[r] > ref_sample_off[r + 1]) return cov_fail(DSA_E_ARG, "ref_sample_off is not ascending");
    if (n_samples >= ((int64_t)1 << 31)) return cov_fail(DSA_E_LIMIT, "more than 2^31-1 samples");
    for (int64_t i = 0; i < n_fragments; ++i)
        if (fragments[i].ref < 0 || fragments[i].ref >= n_refs) return cov_fail(DSA_E_ARG, "fragment %lld: bad transcript index", (long long)i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return cov_fail(DSA_E_DEVICE, "no HIP device %d", device);
    COV_HIP(hipSetDevice(device));
    Stream st;
    COV_HIP(hipStreamCreate(&st.s));
    COV_HIP(hipEventCreate(&st.e0));
    COV_HIP(hipEventCreate(&st.e1));
    DBuf<int64_t> d_off, d_nl, d_ns, d_al, d_as;
    DBuf<int32_t> d_pos;
    DBuf<cov_fragment> d_fr;
    DBuf<uint8_t> d_tmp;
    COV_HIP(d_off.alloc((size_t)n_refs + 1));
    COV_HIP(d_pos.alloc((size_t)n_samples));
    COV_HIP(d_fr.alloc((size_t)n_fragments));
    COV_HIP(d_nl.alloc((size_t)n_fragments + 1));
    COV_HIP(d_ns.alloc((size_t)n_fragments + 1));
    COV_HIP(d_al.alloc((size_t)n_fragments + 1));
    COV_HIP(d_as.alloc((size_t)n_fragments + 1));
    COV_HIP(hipMemcpyAsync(d_off.p, ref_sample_off, ((size_t)n_refs + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st.s));
    if (n_samples) COV_HIP(hipMemcpyAsync(d_pos.p, sample_pos, (size_t)n_samples * sizeof(int32_t), hipMemcpyHostToDevice, st.s));
    COV_HIP(hipMemcpyAsync(d_fr.p, fragments, (size_t)n_fragments * sizeof(cov_fragment), hipMemcpyHostToDevice, st.s));
    COV_HIP(hipMemsetAsync(d_nl.p + n_fragments, 0, sizeof(int64_t), st.s));
    COV_HIP(hipMemsetAsync(d_ns.p + n_fragments, 0, sizeof(int64_t), st.s));
    const unsigned grid = (unsigned)((n_fragments + 255) / 256);
    COV_HIP(hipEventRecord(st.e0, st.s));
    hipLaunchKernelGGL(k_cov_count, dim3(grid), dim3(256), 0, st.s, d_fr.p, n_fragments, d_off.p, d_pos.p, trim_length, split_min_anchor, d_nl.p, d_ns.p);
    size_t tmp = 0;
    COV_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, d_nl.p, d_al.p, (int)(n_fragments + 1), st.s));
    COV_HIP(d_tmp.alloc(tmp));
    COV_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp, d_nl.p, d_al.p, (int)(n_fragments + 1), st.s));
    COV_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp, d_ns.p, d_as.p, (int)(n_fragments + 1), st.s));
    int64_t totals[2] = {0, 0};
    COV_HIP(hipMemcpyAsync(&totals[0], d_al.p + n_fragments, sizeof(int64_t), hipMemcpyDeviceToHost, st.s));
    COV_HIP(hipMemcpyAsync(&totals[1], d_as.p + n_fragments, sizeof(int64_t), hipMemcpyDeviceToHost, st.s));
    COV_HIP(hipStreamSynchronize(st.s));
    COV_HIP(hipGetLastError());
    *n_length = totals[0];
    *n_split = totals[1];
    if (totals[0] > length_cap || totals[1] > split_cap)
        return cov_fail(DSA_E_CAPACITY, "need room for %lld length and %lld split samples", (long long)totals[0], (long long)totals[1]);
    if ((totals[0] && (!length_idx || !length_val)) || (totals[1] && (!split_idx || !split_pos || !split_min)))
        return cov_fail(DSA_E_ARG, "null output");
    DBuf<int32_t> d_li, d_lv, d_si;
    DBuf<double> d_sp, d_sm;
    COV_HIP(d_li.alloc((size_t)totals[0]));
    COV_HIP(d_lv.alloc((size_t)totals[0]));
    COV_HIP(d_si.alloc((size_t)totals[1]));
    COV_HIP(d_sp.alloc((size_t)totals[1]));
    COV_HIP(d_sm.alloc((size_t)totals[1]));
    hipLaunchKernelGGL(k_cov_write, dim3(grid), dim3(256), 0, st.s, d_fr.p, n_fragments, d_off.p, d_pos.p, trim_length, split_min_anchor,
                       d_al.p, d_as.p, d_li.p, d_lv.p, d_si.p, d_sp.p, d_sm.p);
    COV_HIP(hipEventRecord(st.e1, st.s));
    if (totals[0]) {
        COV_HIP(hipMemcpyAsync(length_idx, d_li.p, (size_t)totals[0] * sizeof(int32_t), hipMemcpyDeviceToHost, st.s));
        COV_HIP(hipMemcpyAsync(length_val, d_lv.p, (size_t)totals[0] * sizeof(int32_t), hipMemcpyDeviceToHost, st.s));
    }
    if (totals[1]) {
        COV_HIP(hipMemcpyAsync(split_idx, d_si.p, (size_t)totals[1] * sizeof(int32_t), hipMemcpyDeviceToHost, st.s));
        COV_HIP(hipMemcpyAsync(split_pos, d_sp.p, (size_t)totals[1] * sizeof(double), hipMemcpyDeviceToHost, st.s));
        COV_HIP(hipMemcpyAsync(split_min, d_sm.p, (size_t)totals[1] * sizeof(double), hipMemcpyDeviceToHost, st.s));
    }
    COV_HIP(hipStreamSynchronize(st.s));
    COV_HIP(hipGetLastError());
    if (timing) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, st.e0, st.e1);
        timing->kernel_ms = ms;
        timing->n_length_samples = totals[0];
        timing->n_split_samples = totals[1];
    }
    return DSA_OK;
}

}  // extern "C"
