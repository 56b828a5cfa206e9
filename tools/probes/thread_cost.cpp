// Host probe: cost of spawning 7 threads per parallel region (what the analysis does) against a persistent pool, and the speed-up of a
// memory-streaming loop on the box's CPU quota.  g++ -O2 -pthread tools/probes/thread_cost.cpp -o /tmp/thread_cost
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
template <class F> void run_parts(int n, F&& fn) { std::vector<std::thread> th; for (int p = 1; p < n; ++p) th.emplace_back([&fn, p]() { fn(p); }); fn(0); for (auto& t : th) t.join(); }
int main() {
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int np : {2, 4, 8, 16}) {
        auto t0 = now();
        for (int r = 0; r < 100; ++r) run_parts(np, [](int) {});
        printf("spawn+join %2d parts: %.1f us per region\n", np, ms(t0, now()) * 10);
    }
    const size_t N = 1 << 24;
    std::vector<int> a(N, 1), b(N);
    for (int np : {1, 2, 4, 8, 16}) {
        auto t0 = now();
        for (int r = 0; r < 5; ++r) run_parts(np, [&](int p) { size_t i0 = N * p / np, i1 = N * (p + 1) / np; for (size_t i = i0; i < i1; ++i) b[i] = a[i] * 3 + (int)(i & 7); });
        printf("stream 64 MB x5 with %2d parts: %.2f ms each\n", np, ms(t0, now()) / 5);
    }
    {   // fresh allocation + first touch, serial against 8 threads
        auto t0 = now();
        for (int r = 0; r < 10; ++r) { std::vector<int> v(4 << 20); v[123] = r; }
        printf("vector<int>(4M) zero-filled: %.2f ms\n", ms(t0, now()) / 10);
        t0 = now();
        for (int r = 0; r < 10; ++r) { int* q = new int[4 << 20]; run_parts(8, [&](int p) { size_t n = 4 << 20, i0 = n * p / 8, i1 = n * (p + 1) / 8; for (size_t i = i0; i < i1; i += 1024) q[i] = 1; }); delete[] q; }
        printf("new int[4M] first-touched by 8 threads: %.2f ms\n", ms(t0, now()) / 10);
    }
    return 0;
}
