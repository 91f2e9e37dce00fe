// Host memory bandwidth seen by N threads on the GPU box (what bounds the file reader's parsers and consume_batch's packers):
// every thread copies its slice of a 320 MB source into a destination, best of 5.   g++ -O2 -pthread -o host_bw host_bw.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main() {
    const size_t n = (size_t)320 << 20;
    char *src = (char *)malloc(n), *dst = (char *)malloc(n);
    memset(src, 1, n); memset(dst, 2, n);
    for (int nt : {1, 2, 4, 8, 16, 32, 64}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { const size_t a = n * t / nt, b = n * (t + 1) / nt; memcpy(dst + a, src + a, b - a); });
            for (auto &x : th) x.join();
            best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%2d threads: %.1f ms, %.1f GB/s copied (read + write: twice that), %.2f GB/s per thread\n", nt, best * 1e3, n / best / 1e9, n / best / 1e9 / nt);
    }
    return 0;
}
