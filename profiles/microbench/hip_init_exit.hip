// Start and end of a GPU process on this box as a function of how many HIP streams (hardware queues) it makes:
// hipInit, every hipStreamCreate by itself, the first kernel on every stream, and (caller) main's last line -> process gone.
//   ./hip_init_exit <n streams> [reset]      with GPU_MAX_HW_QUEUES unset / 1 / 2 in the environment
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
__global__ void k_touch(unsigned* p) { p[threadIdx.x] = threadIdx.x; }

int main(int argc, char** argv)
{
    const int ns = argc > 1 ? atoi(argv[1]) : 4;
    const double t_main = now();
    double t = now();
    hipInit(0);
    printf("hipInit %.1f", 1e3 * (now() - t));
    t = now();
    int n = 0;
    hipGetDeviceCount(&n);
    hipSetDevice(0);
    unsigned* d = nullptr;
    hipMalloc((void**)&d, 4096);
    printf(" | device+first malloc %.1f | streams", 1e3 * (now() - t));
    hipStream_t st[16];
    for (int k = 0; k < ns; ++k) {
        t = now();
        hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
        printf(" %.1f", 1e3 * (now() - t));
    }
    printf(" | first kernel per stream");
    for (int k = 0; k < ns; ++k) {
        t = now();
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[k], d);
        hipStreamSynchronize(st[k]);
        printf(" %.1f", 1e3 * (now() - t));
    }
    t = now();
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    printf(" | null stream %.1f", 1e3 * (now() - t));
    if (argc > 2) {
        t = now();
        for (int k = 0; k < ns; ++k) hipStreamDestroy(st[k]);
        printf(" | destroy streams %.1f", 1e3 * (now() - t));
        t = now();
        hipDeviceReset();
        printf(" | hipDeviceReset %.1f", 1e3 * (now() - t));
    }
    printf(" | main %.1f ms\nmain ends at wall %.6f\n", 1e3 * (now() - t_main), wall());
    fflush(stdout);
    _exit(0);
}
