// Microbenchmark: cost of a dependent FP64 FMA chain as a function of the number (and position) of ACTIVE lanes in
// the wavefront, with 1 / 4 / 16 wavefronts per CU.
//   hipcc --offload-arch=gfx950 -O3 -o exec_density exec_density.hip && ./exec_density
#include <hip/hip_runtime.h>
#include <stdio.h>

// mode 0: lanes outside the mask EXIT (thin wave); mode 1: all 64 lanes stay alive, the masked-out ones only skip the
// chain (a divergent branch: EXEC is sparse while the chain runs); mode 2: as 1, and the masked-out lanes then run
// the chain themselves afterwards (the two sides of a divergent if / else)
__global__ void chain_kernel(double *out, int iters, double seed, unsigned long long mask, int f32, int mode)
{
    const int lane = threadIdx.x & 63;
    const bool in_mask = (mask >> lane) & 1ull;
    if (mode == 0 && !in_mask) return;
    if (mode >= 1) {
        double x = seed + blockIdx.x * 1e-9 + threadIdx.x * 1e-7, y = 1.000000001;
        if (in_mask) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int k = 0; k < 32; ++k) x = __builtin_fma(x, y, 1e-9);
            }
        } else if (mode == 2) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int k = 0; k < 32; ++k) x = __builtin_fma(x, y, 2e-9);
            }
        }
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x;
        return;
    }
    double x = seed + blockIdx.x * 1e-9 + threadIdx.x * 1e-7, y = 1.000000001;
    float xf = (float)x, yf = 1.0000001f;
    if (f32) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k) xf = __builtin_fmaf(xf, yf, 1e-9f);
        }
        x = xf;
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k) x = __builtin_fma(x, y, 1e-9);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x;
}

static float run(int n_wg, int block, int iters, unsigned long long mask, int f32, double *d_out, int mode = 0)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel, dim3(n_wg), dim3(block), 0, 0, d_out, iters, 0.7, mask, f32, mode);
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
    const int iters = 4000;
    struct { const char *name; unsigned long long mask; } masks[] = {
        {"1", 1ull}, {"2", 3ull}, {"4", 0xfull}, {"8", 0xffull}, {"16", 0xffffull}, {"17", 0x1ffffull},
        {"32", 0xffffffffull}, {"33", 0x1ffffffffull}, {"48", 0xffffffffffffull}, {"64", ~0ull},
        {"16 spread (every 4th)", 0x1111111111111111ull}, {"2: lanes 0 and 32", 0x100000001ull}, {"1: lane 40", 1ull << 40}};
    const int shapes[][2] = {{256, 64}, {1024, 64}, {4096, 64}};
    for (int mode = 1; mode <= 2; ++mode)
        for (auto &m : masks) {
            printf("{\"type\": \"f64, all lanes alive, %s\", \"lanes_in_branch\": \"%s\", \"ms\": {", mode == 1 ? "others skip" : "others run the else side afterwards", m.name);
            for (int s = 0; s < 3; ++s)
                printf("%s\"%dx%d\": %.3f", s ? ", " : "", shapes[s][0], shapes[s][1], run(shapes[s][0], shapes[s][1], iters, m.mask, 0, d_out, mode));
            printf("}}\n");
            fflush(stdout);
        }
    for (int f32 = 0; f32 < 1; ++f32)
        for (auto &m : masks) {
            printf("{\"type\": \"%s\", \"active_lanes\": \"%s\", \"instructions_per_wave\": %d, \"ms\": {", f32 ? "f32" : "f64", m.name, iters * 32);
            for (int s = 0; s < 3; ++s)
                printf("%s\"%dx%d\": %.3f", s ? ", " : "", shapes[s][0], shapes[s][1], run(shapes[s][0], shapes[s][1], iters, m.mask, f32, d_out));
            printf("}}\n");
            fflush(stdout);
        }
    return 0;
}
