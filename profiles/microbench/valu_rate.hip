// Microbenchmark: issue rate of the VALU ops the DP kernel is made of (gfx950).
// Each kernel runs ITER iterations of 16 independent instructions of one kind per wave;
// launched with enough waves to fill every SIMD at 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 4096
#define OP16(OPSTR)                                                                                    \
    asm volatile(OPSTR " %0, %0, %16\n" OPSTR " %1, %1, %16\n" OPSTR " %2, %2, %16\n" OPSTR " %3, %3, %16\n"  \
                 OPSTR " %4, %4, %16\n" OPSTR " %5, %5, %16\n" OPSTR " %6, %6, %16\n" OPSTR " %7, %7, %16\n"  \
                 OPSTR " %8, %8, %16\n" OPSTR " %9, %9, %16\n" OPSTR " %10, %10, %16\n" OPSTR " %11, %11, %16\n" \
                 OPSTR " %12, %12, %16\n" OPSTR " %13, %13, %16\n" OPSTR " %14, %14, %16\n" OPSTR " %15, %15, %16\n" \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),    \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                 : "v"(b))

#define KERNEL(NAME, OPSTR)                                                 \
    __global__ void NAME(unsigned* out, unsigned seed)                      \
    {                                                                       \
        unsigned a[16];                                                     \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 17 + i + seed;    \
        unsigned b = seed | 1;                                              \
        for (int it = 0; it < ITER; ++it) { OP16(OPSTR); }                  \
        unsigned s = 0;                                                     \
        for (int i = 0; i < 16; ++i) s ^= a[i];                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                     \
    }

KERNEL(k_pk_add_u16, "v_pk_add_u16")
KERNEL(k_pk_max_i16, "v_pk_max_i16")
KERNEL(k_add_u32, "v_add_u32")
KERNEL(k_max_i32, "v_max_i32")
KERNEL(k_xor_b32, "v_xor_b32")
KERNEL(k_pk_add_f16, "v_pk_add_f16")

// dependent chain of v_pk_max_i16 (one accumulator): latency incl. the wait state the compiler inserts
__global__ void k_chain_pk(unsigned* out, unsigned seed)
{
    unsigned a = threadIdx.x + seed, b = seed | 1;
    for (int it = 0; it < ITER; ++it) {
        asm volatile("v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     "v_pk_max_i16 %0, %0, %1\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n s_nop 0\n"
                     : "+v"(a) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <typename K>
double run(K kern, const char* name, int waves_per_simd, unsigned* d_out, int ops_per_iter)
{
    int blocks = 256 * waves_per_simd;   // 256 CUs, block = 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * ITER * ops_per_iter;
    double per_simd_per_s = wave_instr / 1024.0 / (ms * 1e-3);
    printf("%-16s waves/SIMD=%d  %.3f ms  %.3e wave-instr/s/SIMD  => %.2f cycles/instr @2.4GHz\n", name, waves_per_simd, ms,
           per_simd_per_s, 2.4e9 / per_simd_per_s);
    return ms;
}

int main()
{
    unsigned* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned));
    for (int w : {1, 2, 4, 8}) {
        run(k_pk_add_u16, "v_pk_add_u16", w, d_out, 16);
        run(k_pk_max_i16, "v_pk_max_i16", w, d_out, 16);
        run(k_add_u32, "v_add_u32", w, d_out, 16);
        run(k_max_i32, "v_max_i32", w, d_out, 16);
        run(k_xor_b32, "v_xor_b32", w, d_out, 16);
        run(k_pk_add_f16, "v_pk_add_f16", w, d_out, 16);
        run(k_chain_pk, "chain pk+nop", w, d_out, 16);
    }
    return 0;
}
