#include "rt_bvh.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <x86intrin.h>
namespace rt { extern unsigned long long P[16]; }
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 128;
    std::vector<double> c, k;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            auto vx = [&](int a, int b, double* o) { o[0] = -10.0 + 5.0 * a / n; o[1] = -5.0 + 5.0 * b / n; o[2] = -6.0 + 0.1 * std::sin(0.37 * a) * std::cos(0.21 * b); };
            double v[4][3];
            vx(i, j, v[0]); vx(i + 1, j, v[1]); vx(i, j + 1, v[2]); vx(i + 1, j + 1, v[3]);
            int t[2][3] = {{0, 1, 2}, {1, 3, 2}};
            for (auto& tt : t) { for (int q : tt) for (int d = 0; d < 3; ++d) c.push_back(v[q][d]); for (int d = 0; d < 3; ++d) k.push_back(0.5); }
        }
    if (argc > 2) {  // "soup": the same number of small triangles at random places, in random order (no spatial runs)
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (x >> 11) * (1.0 / 9007199254740992.0); };
        for (size_t t = 0; t < c.size() / 9; ++t) {
            const double o[3] = {rnd() * 5.0, rnd() * 5.0, -6.0 + rnd()};
            for (int v = 0; v < 3; ++v)
                for (int d = 0; d < 3; ++d) c[t * 9 + v * 3 + d] = o[d] + 0.05 * rnd();
        }
    }
    int64_t E = c.size() / 9;
    const double* cp = c.data(); const double* kp = k.data();
    double best = 1e9;
    unsigned long long tsc0 = 0, tsc1 = 0;
    for (int r = 0; r < 30; ++r) {
        if (r == 10) { for (auto& p : rt::P) p = 0; tsc0 = __rdtsc(); }
        auto t0 = std::chrono::steady_clock::now();
        rt::Scene* s = rt::build_scene(1, &cp, &kp, &E, 0, 1);
        double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
        if (ms < best) best = ms;
        delete s;
    }
    tsc1 = __rdtsc();
    double tot = double(tsc1 - tsc0);
    printf("best %.2f ms total %.0fk | ", best, tot / 20000);
    const char* nm[] = {"cent bounds", "binning", "sah eval", "sides", "partition", "sides_ready", "splits", "rec init", "tree", "flatten"};
    for (int i = 0; i < 10; ++i) printf("%s %lluk | ", nm[i], rt::P[i] / 20000);
}
