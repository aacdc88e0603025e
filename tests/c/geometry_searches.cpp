// Host check of remora_amd/csrc/rmr_geometry.h (compiled by tests/test_host_cpu.py with g++ and a two-line stand-in for
// <hip/hip_runtime.h> that defines __host__ / __device__ away): the searches that start from a hint (used by rmr_call_read
// on the host) return what plain bisection returns (used by geometry_kernel on the device) on random non-decreasing arrays
// with runs of equal values, for hints anywhere including outside the array; and whole geometry rows agree.
#include "rmr_geometry.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rmr;
int main() {
    srand(7);
    long bad = 0, n_checks = 0;
    for (int trial = 0; trial < 3000; ++trial) {
        const int n = 1 + rand() % 400;
        std::vector<int64_t> a(n);
        int64_t v = rand() % 5;
        for (int i = 0; i < n; ++i) { v += (rand() % 4 == 0) ? 0 : rand() % 12; a[i] = v; }
        for (int q = 0; q < 200; ++q) {
            const int64_t x = (rand() % (int)(v + 20)) - 10, hint = (rand() % (n + 6)) - 3;
            bad += ub_right(a.data(), n, x) != ub_right_near(a.data(), n, x, hint);
            bad += lb_left(a.data(), n, x) != lb_left_near(a.data(), n, x, hint);
            n_checks += 2;
        }
        // whole rows
        int64_t g0[6], g1[6];
        const int nb = n - 1;
        if (nb >= 1) for (int q = 0; q < 50; ++q) {
            const int64_t f = rand() % nb; const int bsj = rand() % 2, off = rand() % 5 - 2;
            const int64_t s0 = chunk_geometry_row(a.data(), nb, a[nb], f, bsj, off, 50, 50, g0, false), s1 = chunk_geometry_row(a.data(), nb, a[nb], f, bsj, off, 50, 50, g1, true);
            bad += s0 != s1; for (int k = 0; k < 6; ++k) bad += g0[k] != g1[k];
            ++n_checks;
        }
    }
    printf("%ld checks, %ld mismatches\n", n_checks, bad);
    return bad != 0;
}
