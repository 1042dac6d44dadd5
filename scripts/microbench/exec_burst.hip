// Microbenchmark: does a short DENSE burst (all 64 lanes, B instructions) every P sparse instructions (1 lane) keep a
// wavefront out of the slow sparse-EXEC mode?   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o exec_burst exec_burst.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int B>
__global__ void burst_kernel(double *out, int n_periods, int period_iters, double seed, int k_lanes)
{
    const int lane = threadIdx.x & 63;
    double x = seed + blockIdx.x * 1e-9 + threadIdx.x * 1e-7, y = 1.000000001, z = x + 0.5;
    for (int p = 0; p < n_periods; ++p) {
        if (lane < k_lanes) {
            for (int i = 0; i < period_iters; ++i) {
#pragma unroll
                for (int k = 0; k < 16; ++k) z = __builtin_fma(z, y, 1e-9);
            }
        }
#pragma unroll
        for (int k = 0; k < B; ++k) x = __builtin_fma(x, y, 1e-9);   // all lanes
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + z;
}

template <typename F>
static float timeit(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int B>
static void row(double *d_out, int period)
{
    const int total = 131072;
    for (int k : {1, 64}) {
        printf("{\"sparse_instructions_per_period\": %d, \"dense_burst\": %d, \"lanes_in_sparse_part\": %d, \"ms\": {", period, B, k);
        bool first = true;
        for (int wg : {1024, 3072}) {
            float ms = timeit([&] { hipLaunchKernelGGL(burst_kernel<B>, dim3(wg), dim3(64), 0, 0, d_out, total / period, period / 16, 0.7, k); });
            printf("%s\"%dx64\": %.3f", first ? "" : ", ", wg, ms);
            first = false;
        }
        printf("}}\n");
        fflush(stdout);
    }
}

int main()
{
    double *d_out; (void)hipMalloc(&d_out, sizeof(double) * 64 * 8192);
    for (int period : {1024, 8192}) {
        row<0>(d_out, period); row<1>(d_out, period); row<4>(d_out, period); row<16>(d_out, period); row<64>(d_out, period);
    }
    return 0;
}
