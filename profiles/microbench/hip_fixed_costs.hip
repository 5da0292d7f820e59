// What a short-lived tool process pays around its GPU work on this box: runtime start, first kernel of a code object,
// device and pinned allocations of the sizes dosplitalign uses, and (measured by the caller: wall time of the process minus
// the "main ends" stamp printed last) the teardown the system does at exit.
//   hipcc --offload-arch=gfx950 -O2 -o hip_fixed_costs hip_fixed_costs.hip && ./hip_fixed_costs [device MiB] [pinned MiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/time.h>
#include <unistd.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double wall()
{
    timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

__global__ void k_touch(unsigned* p, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = (unsigned)i;
}

int main(int argc, char** argv)
{
    const size_t dev_mib = argc > 1 ? (size_t)atoll(argv[1]) : 4096, pin_mib = argc > 2 ? (size_t)atoll(argv[2]) : 64;
    const bool skip_free = argc > 3;
    if (const char* t0 = getenv("PROBE_T0")) printf("process start -> main: %.1f ms\n", 1e3 * (wall() - atof(t0)));
    double t = now();
    int n = 0;
    hipGetDeviceCount(&n);
    printf("hipGetDeviceCount (runtime start): %.1f ms, %d devices\n", 1e3 * (now() - t), n);
    t = now();
    hipSetDevice(0);
    hipFree(nullptr);
    printf("hipSetDevice + context: %.1f ms\n", 1e3 * (now() - t));
    t = now();
    hipStream_t st[4];
    for (auto& s : st) hipStreamCreate(&s);
    printf("4 streams: %.1f ms\n", 1e3 * (now() - t));
    t = now();
    unsigned* small = nullptr;
    hipMalloc((void**)&small, 1 << 20);
    printf("hipMalloc 1 MiB: %.2f ms\n", 1e3 * (now() - t));
    t = now();
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, st[0], small, (size_t)1 << 18);
    hipStreamSynchronize(st[0]);
    printf("first kernel (code object load + launch + sync): %.1f ms\n", 1e3 * (now() - t));
    t = now();
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, st[0], small, (size_t)1 << 18);
    hipStreamSynchronize(st[0]);
    printf("second kernel: %.3f ms\n", 1e3 * (now() - t));
    for (size_t mib : {(size_t)64, (size_t)512, dev_mib}) {
        t = now();
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, mib << 20);
        const double a = now() - t;
        t = now();
        hipLaunchKernelGGL(k_touch, dim3((unsigned)((mib << 18) / 256)), dim3(256), 0, st[0], (unsigned*)p, mib << 18);
        hipStreamSynchronize(st[0]);
        const double k = now() - t;
        t = now();
        if (!(skip_free && mib == dev_mib)) hipFree(p);
        printf("hipMalloc %zu MiB: %.2f ms (%s), first touch of all of it %.2f ms, hipFree %.2f ms\n", mib, 1e3 * a, hipGetErrorString(e), 1e3 * k, 1e3 * (now() - t));
    }
    for (size_t mib : {(size_t)16, pin_mib, 4 * pin_mib}) {
        t = now();
        void* p = nullptr;
        hipError_t e = hipHostMalloc(&p, mib << 20, hipHostMallocDefault);
        const double a = now() - t;
        t = now();
        void* d = nullptr;
        hipMalloc(&d, mib << 20);
        hipMemcpyAsync(d, p, mib << 20, hipMemcpyHostToDevice, st[1]);
        hipStreamSynchronize(st[1]);
        const double c = now() - t;
        t = now();
        hipMemcpyAsync(d, p, mib << 20, hipMemcpyHostToDevice, st[1]);
        hipStreamSynchronize(st[1]);
        const double c2 = now() - t;
        void* pageable = malloc(mib << 20);
        for (size_t i = 0; i < (mib << 20); i += 4096) ((char*)pageable)[i] = 1;
        t = now();
        hipMemcpyAsync(d, pageable, mib << 20, hipMemcpyHostToDevice, st[1]);
        hipStreamSynchronize(st[1]);
        const double c3 = now() - t;
        t = now();
        hipMemcpyAsync(pageable, d, mib << 20, hipMemcpyDeviceToHost, st[1]);
        hipStreamSynchronize(st[1]);
        const double c4 = now() - t;
        t = now();
        if (!skip_free) { hipHostFree(p); hipFree(d); }
        printf("hipHostMalloc %zu MiB: %.2f ms (%s); H2D from it, first %.2f ms, again %.2f ms; from / to pageable memory %.2f / %.2f ms; free %.2f ms\n", mib, 1e3 * a,
               hipGetErrorString(e), 1e3 * c, 1e3 * c2, 1e3 * c3, 1e3 * c4, 1e3 * (now() - t));
        free(pageable);
    }
    printf("main ends at wall %.6f\n", wall());
    fflush(stdout);
    if (getenv("PROBE_UNDERSCORE_EXIT")) _exit(0);
    return 0;
}
