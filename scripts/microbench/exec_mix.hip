// Microbenchmark: a wave alternates DENSE segments (all 64 lanes) with SPARSE segments (k lanes inside a divergent
// branch).  Is the cost of the sparse part additive (sparse instructions x ~4.6), and does it depend on what the other
// waves of the CU are doing?   hipcc --offload-arch=gfx950 -O3 -o exec_mix exec_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int DENSE, int SPARSE>
__global__ void mix_kernel(double *out, int iters, double seed, int k_lanes, int odd_waves_dense_only)
{
    const int lane = threadIdx.x & 63;
    double x = seed + blockIdx.x * 1e-9 + threadIdx.x * 1e-7, y = 1.000000001, z = x + 0.5;
    const bool sparse_here = lane < k_lanes && !(odd_waves_dense_only && (blockIdx.x & 1));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < DENSE; ++k) x = __builtin_fma(x, y, 1e-9);
        if (sparse_here) {
#pragma unroll
            for (int k = 0; k < SPARSE; ++k) z = __builtin_fma(z, y, x);   // depends on x: cannot be hoisted / merged
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + z;
}

template <int DENSE, int SPARSE>
static float run(int n_wg, int iters, int k_lanes, int odd, double *d_out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mix_kernel<DENSE, SPARSE>), dim3(n_wg), dim3(64), 0, 0, d_out, iters, 0.7, k_lanes, odd);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    double *d_out; (void)hipMalloc(&d_out, sizeof(double) * 64 * 8192);
    const int iters = 2000;
    const int ks[] = {0, 1, 4, 8, 12, 15, 16, 32, 64};
    const int wgs[] = {256, 1024, 3072};
    for (int odd = 0; odd < 2; ++odd)
        for (int k : ks) {
            printf("{\"dense_per_iter\": 32, \"sparse_per_iter\": 8, \"lanes_in_sparse_branch\": %d, \"only_even_waves_have_the_branch\": %d, \"ms\": {", k, odd);
            for (int s = 0; s < 3; ++s) printf("%s\"%dx64\": %.3f", s ? ", " : "", wgs[s], run<32, 8>(wgs[s], iters, k, odd, d_out));
            printf("}}\n");
            fflush(stdout);
        }
    for (int k : ks) {
        printf("{\"dense_per_iter\": 8, \"sparse_per_iter\": 32, \"lanes_in_sparse_branch\": %d, \"ms\": {", k);
        for (int s = 0; s < 3; ++s) printf("%s\"%dx64\": %.3f", s ? ", " : "", wgs[s], run<8, 32>(wgs[s], iters, k, 0, d_out));
        printf("}}\n");
        fflush(stdout);
    }
    return 0;
}
