// Where the host time of a text-in / text-out tool goes on the GPU box: a cluster file of the given number of lines is written,
// then read back (a) through mmap + MAP_POPULATE, (b) by threaded pread into anonymous memory (with and without transparent
// huge pages), each followed by a threaded newline scan and a threaded parse of the first three integer fields.
//   g++ -O2 -pthread -o /tmp/host_io_probe profiles/microbench/host_io_probe.cpp && /tmp/host_io_probe /tmp/probe.txt 100000000 16
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static void par(int T, F f)
{
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { f(t); });
    for (auto& x : th) x.join();
}

int main(int argc, char** argv)
{
    const char* path = argv[1];
    const long lines = atol(argv[2]);
    const int T = atoi(argv[3]);
    {
        double t0 = now();
        FILE* f = fopen(path, "w");
        std::string buf;
        for (long i = 0; i < lines; i += 2) {
            char tmp[160];
            const long c = i / 60, fr = (i * 2654435761u) % 50000000;
            int k = snprintf(tmp, sizeof tmp, "%ld\t0\t%ld\t1\tchr6\t+\t%ld\t%ld\n%ld\t1\t%ld\t0\tchr5\t-\t%ld\t%ld\n", c, fr, 17696 + i % 1000,
                             17745 + i % 1000, c, fr, 188438574 + i % 1000, 188438623 + i % 1000);
            buf.append(tmp, k);
            if (buf.size() > (8u << 20)) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
        }
        fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
        printf("wrote %ld lines in %.2f s\n", lines, now() - t0);
    }
    int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t n = st.st_size;
    printf("file %.2f GB, %d threads\n", n / 1e9, T);
    auto scan_and_parse = [&](const char* p, const char* what) {
        std::vector<long> cnt(T), sum(T);
        double t0 = now();
        par(T, [&](int t) {
            size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1);
            long c = 0;
            for (size_t i = lo; i < hi;) {
                const char* nl = (const char*)memchr(p + i, '\n', hi - i);
                if (!nl) break;
                ++c;
                i = nl - p + 1;
            }
            cnt[t] = c;
        });
        double t1 = now();
        par(T, [&](int t) {
            size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1);
            if (lo) { const char* nl = (const char*)memchr(p + lo, '\n', hi - lo); lo = nl ? nl - p + 1 : hi; }
            long s = 0;
            for (size_t i = lo; i < hi;) {
                long v[3] = {0, 0, 0};
                size_t k = i;
                for (int f = 0; f < 3; ++f) {
                    long x = 0;
                    while (k < hi && (unsigned)(p[k] - '0') < 10u) x = x * 10 + (p[k++] - '0');
                    v[f] = x;
                    ++k;
                }
                s += v[0] + v[1] + v[2];
                const char* nl = (const char*)memchr(p + k, '\n', hi - k);
                if (!nl) break;
                i = nl - p + 1;
            }
            sum[t] = s;
        });
        double t2 = now();
        long c = 0, s = 0;
        for (int t = 0; t < T; ++t) { c += cnt[t]; s += sum[t]; }
        printf("  %s: newline scan %.3f s (%ld lines), parse of three fields %.3f s (checksum %ld)\n", what, t1 - t0, c, t2 - t1, s);
    };
    {
        double t0 = now();
        char* p = (char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        printf("mmap + MAP_POPULATE: %.3f s\n", now() - t0);
        scan_and_parse(p, "mapped file, first pass");
        scan_and_parse(p, "mapped file, second pass");
        munmap(p, n);
    }
    for (int huge = 0; huge < 2; ++huge) {
        double t0 = now();
        const size_t cap = (n + (2u << 20)) & ~((size_t)(2u << 20) - 1);
        char* p = (char*)mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (huge) madvise(p, cap, MADV_HUGEPAGE);
        par(T, [&](int t) {
            size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1);
            while (lo < hi) {
                ssize_t g = pread(fd, p + lo, std::min<size_t>(hi - lo, 8u << 20), lo);
                if (g <= 0) break;
                lo += g;
            }
        });
        printf("pread by %d threads into anonymous memory%s: %.3f s (%.1f GB/s)\n", T, huge ? " (MADV_HUGEPAGE)" : "", now() - t0, n / (now() - t0) / 1e9);
        scan_and_parse(p, "anonymous copy");
        munmap(p, cap);
    }
    unlink(path);
    return 0;
}
