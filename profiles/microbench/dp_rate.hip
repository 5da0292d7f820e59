// Issue rate of the fast fill's row pattern in isolation (gfx950): 64 register-resident columns,
// per column  add (diagonal term) + pk_maximum3 (cell) + sub (drift) + 1/2 pk_maximum3 (row maximum).
//   MODE 0: score operand from a register        (pure VALU ceiling of the pattern)
//   MODE 1: score operand from LDS tables, one ds_read_b128 per 4 columns, 2 reads in flight (as k_fill_fast)
//   MODE 2: as 1, plus one dwordx4 load and two dwordx4 stores per 4 rows (the kernel's HBM streams)
// Prints nominal-2.4-GHz cycles per (wave, column, row) at 4 (then 3, 2, 1) waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o dp_rate dp_rate.hip && ./dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t max3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(v2h, a), __builtin_bit_cast(v2h, b)),
                                                                      __builtin_bit_cast(v2h, c)));
}
constexpr int W = 64, TROW = 68, ROWS = 77 * 7 * 8;
constexpr uint32_t BIAS2 = 0x04000400u, TWO2 = 0x00020002u;

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_dp(uint32_t* out, const uint4* in, uint4* st, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint32_t T[4 * 25 * TROW];
    for (int e = threadIdx.x; e < 4 * 25 * TROW; e += 256) T[e] = 0x00030003u + ((e * seed) & 0x00030000u);
    __syncthreads();
    uint32_t X[W];
#pragma unroll
    for (int i = 0; i < W; ++i) X[i] = BIAS2 + i * TWO2;
    const int lane = threadIdx.x & 63;
    uint32_t cls = (threadIdx.x * 7 + seed) % 16;
    uint32_t bprev = BIAS2, acc_out = 0;
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
    uint4 ld = make_uint4(BIAS2, BIAS2, BIAS2, BIAS2);
    for (int j4 = 0; j4 < ROWS / 4; ++j4) {
        if (MODE == 2) ld = in[base + (size_t)(j4 & 63) * gridDim.x * 256];
        uint32_t cm[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t bcur = MODE == 2 ? (s == 0 ? ld.x : s == 1 ? ld.y : s == 2 ? ld.z : ld.w) : BIAS2;
            const uint4* trow = reinterpret_cast<const uint4*>(T + ((cls + s) & 15) * TROW);
            uint4 vq[3];
            const uint4 vreg = make_uint4(0x00030003u + seed, 0x00060006u, 0x00030003u, 0x00060006u);
#pragma unroll
            for (int k = 0; k < 3; ++k) vq[k] = MODE ? trow[k] : vreg;
            uint32_t a = bprev + vq[0].x;
            uint32_t up = bcur - TWO2;
#pragma unroll
            for (int q = 0; q < W / 4; ++q) {
                const uint4 v = vq[0];
                vq[0] = vq[1];
                vq[1] = vq[2];
                if (q + 3 < W / 4) vq[2] = MODE ? trow[q + 3] : vreg;
                const uint4 vn = vq[0];
                uint32_t an;
                an = X[4 * q + 0] + v.y; X[4 * q + 0] = max3(a, X[4 * q + 0], up); a = an;
                an = X[4 * q + 1] + v.z; X[4 * q + 1] = max3(a, X[4 * q + 1], X[4 * q + 0]); a = an;
                an = X[4 * q + 2] + v.w; X[4 * q + 2] = max3(a, X[4 * q + 2], X[4 * q + 1]); a = an;
                an = X[4 * q + 3] + vn.x; X[4 * q + 3] = max3(a, X[4 * q + 3], X[4 * q + 2]); a = an;
                up = X[4 * q + 3];
            }
            uint32_t a0 = BIAS2, a1 = BIAS2;
#pragma unroll
            for (int i = 0; i < W; i += 4) {
                a0 = max3(a0, X[i] - i * TWO2, X[i + 1] - (i + 1) * TWO2);
                a1 = max3(a1, X[i + 2] - (i + 2) * TWO2, X[i + 3] - (i + 3) * TWO2);
            }
            cm[s] = max3(a0, a1, BIAS2);
            bprev = bcur;
            cls += a0 & 1;
        }
        if (MODE == 2) {
            st[base + (size_t)(j4 & 63) * gridDim.x * 256] = make_uint4(cm[0], cm[1], cm[2], cm[3]);
            st[base + (size_t)(64 + (j4 & 63)) * gridDim.x * 256] = make_uint4(X[63], cm[1], X[62], cm[3]);
        } else
            acc_out ^= cm[0] ^ cm[1] ^ cm[2] ^ cm[3];
        // keep X in range: values would otherwise grow without bound
#pragma unroll
        for (int i = 0; i < W; i += 16) X[i] = (X[i] & 0x03FF03FFu) | BIAS2;
    }
    out[base] = acc_out ^ X[5] ^ lane;
}

template <int MODE>
void run(const char* name, int wg_per_cu = 4)
{
    const int grid = 256 * wg_per_cu * 4;   // 4 rounds of wg_per_cu workgroups per CU
    // dynamic LDS pads the workgroup so that only wg_per_cu of them fit a CU (160 KB)
    const size_t pad = wg_per_cu >= 4 ? 0 : (size_t)(160 * 1024 / wg_per_cu) - 27200 - 2048;
    uint32_t* out;
    uint4 *in, *st;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&in, (size_t)grid * 256 * 16 * 64);
    hipMalloc(&st, (size_t)grid * 256 * 16 * 128);
    hipMemset(in, 0x04, (size_t)grid * 256 * 16 * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_dp<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    k_dp<MODE><<<grid, 256, pad>>>(out, in, st, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_dp<MODE><<<grid, 256, pad>>>(out, in, st, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double colsteps_per_simd = (double)grid * 4 / 1024.0 * ROWS * W;
    printf("%-24s %d waves/SIMD %8.3f ms  %6.2f nominal cycles per column step\n", name, wg_per_cu, ms, ms * 1e-3 * 2.4e9 / colsteps_per_simd);
    hipFree(out);
    hipFree(in);
    hipFree(st);
}

int main()
{
    run<0>("registers only");
    run<1>("+ LDS score tables");
    run<2>("+ HBM row streams");
    for (int n = 3; n >= 1; --n) run<2>("+ HBM row streams", n);
    return 0;
}
