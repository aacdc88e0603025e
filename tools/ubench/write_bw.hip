// Micro-benchmark (measurement tool, not product code): what does a pure store stream reach on this part?  3.6 GB of
// 16-byte stores (the size of one encode launch), plain and non-temporal, one wave per 14.4 KB run like the encode
// kernel, and as one flat grid-stride stream.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/write_bw tools/ubench/write_bw.hip && /tmp/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT, bool PER_WAVE>
__global__ __launch_bounds__(256) void k(f32x4 *out, long long n_runs, int run16) {
    const int lane = threadIdx.x & 63;
    const f32x4 v = {1.f, 0.f, 0.f, 1.f};
    if (PER_WAVE) {
        const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
        for (long long c = wave; c < n_runs; c += n_waves) {
            f32x4 *dst = out + c * run16;
            for (int f = lane; f < run16; f += 64) {
                if (NT) __builtin_nontemporal_store(v, dst + f);
                else dst[f] = v;
            }
        }
    } else {
        const long long total = n_runs * run16, stride = (long long)gridDim.x * 256;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
            if (NT) __builtin_nontemporal_store(v, out + i);
            else out[i] = v;
        }
    }
}

template <bool NT, bool PW>
void run(const char *name, f32x4 *d, long long n_runs, int run16, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<NT, PW>), dim3(grid), dim3(256), 0, 0, d, n_runs, run16);
    (void)hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<NT, PW>), dim3(grid), dim3(256), 0, 0, d, n_runs, run16);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n_runs * run16 * 16.0;
    printf("  %-44s grid %6d: %.3f ms per launch, %.0f GB/s\n", name, grid, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const long long n_runs = 250000; const int run16 = 900;  // 250 k x 14,400 B
    f32x4 *d; (void)hipMalloc(&d, (size_t)n_runs * run16 * 16);
    for (int g : {cus * 8, cus * 16, cus * 32}) {
        run<true, true>("non-temporal, one wave per 14.4 KB run", d, n_runs, run16, g);
        run<false, true>("plain, one wave per 14.4 KB run", d, n_runs, run16, g);
        run<true, false>("non-temporal, flat grid-stride", d, n_runs, run16, g);
        run<false, false>("plain, flat grid-stride", d, n_runs, run16, g);
    }
    return 0;
}
