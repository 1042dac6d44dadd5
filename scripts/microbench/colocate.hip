// Microbenchmark: does a one-lane-active wavefront run slower when other such wavefronts share its CU?
// Each wave runs a dependent chain of `iters` x 32 instructions of one kind; grid = n_wg workgroups of `block` threads.
//   hipcc --offload-arch=gfx950 -O3 -o colocate colocate.hip && ./colocate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int KIND>
__global__ void chain_kernel(double *out, int iters, double seed)
{
    __shared__ double lds[256];
    lds[threadIdx.x] = seed + threadIdx.x;
    __syncthreads();
    if (KIND < 6 && (threadIdx.x & 63) != 0) return;  // one active lane per wave (kinds 6-8: all 64 lanes)
    double x = seed + blockIdx.x * 1e-9 + threadIdx.x * 1e-7, y = 1.000000001;
    double x1 = x + 0.1, x2 = x + 0.2, x3 = x + 0.3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (KIND == 0) x = __builtin_fma(x, y, 1e-9);                 // dependent f64 FMA
            else if (KIND == 1) x = 1.0 / (x + 1.5);                      // IEEE f64 division (rcp + refinement)
            else if (KIND == 2) x = sqrt(x * x + 2.0);                    // IEEE f64 sqrt
            else if (KIND == 3) x = __builtin_fma(x, lds[(k * 8 + (int)threadIdx.x) & 255], 1e-9);  // + LDS read
            else if (KIND == 4) { if (x > 0.5) x = __builtin_fma(x, y, -0.3); else x = __builtin_fma(x, y, 0.4); }  // + branch
            else if (KIND == 5) { float f = (float)x; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); x = f; }  // f32 + cvt
            else if (KIND == 6) x = __builtin_fma(x, y, 1e-9);            // dependent f64 FMA, 64 lanes active
            else if (KIND == 7 || KIND == 8) {                            // four independent chains (ILP 4): 4 FMAs per k
                x = __builtin_fma(x, y, 1e-9); x1 = __builtin_fma(x1, y, 1e-9);
                x2 = __builtin_fma(x2, y, 1e-9); x3 = __builtin_fma(x3, y, 1e-9);
            }
        }
    }
    if (KIND == 8 && (threadIdx.x & 63) != 0) return;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)] = x + x1 + x2 + x3;
}

template <int KIND>
float run(int n_wg, int block, int iters, double *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel<KIND>, dim3(n_wg), dim3(block), 0, 0, d_out, iters, 0.7);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    double *d_out; hipMalloc(&d_out, sizeof(double) * 4 * 64 * 8192);
    const int iters = 4000;
    const char *names[] = {"fma_f64", "div_f64", "sqrt_f64", "fma_f64+lds", "fma_f64+branch", "fma_f32+cvt", "fma_f64 64 lanes", "4 independent fma_f64 chains, 64 lanes (4x the instructions)",
                           "4 independent chains, 64 lanes"};
    const int shapes[][2] = {{1, 64}, {64, 64}, {256, 64}, {512, 64}, {1024, 64}, {2048, 64}, {4096, 64}, {64, 256}, {256, 256}, {512, 256}, {1024, 256}};
    for (int kind = 0; kind < 8; ++kind) {
        if (kind >= 1 && kind <= 5 && kind != 4) continue;
        printf("{\"kind\": \"%s\", \"instructions_per_wave\": %d, \"ms\": {", names[kind], iters * 32);
        for (size_t s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
            float ms = 0;
            switch (kind) {
            case 0: ms = run<0>(shapes[s][0], shapes[s][1], iters, d_out); break;
            case 1: ms = run<1>(shapes[s][0], shapes[s][1], iters, d_out); break;
            case 2: ms = run<2>(shapes[s][0], shapes[s][1], iters, d_out); break;
            case 3: ms = run<3>(shapes[s][0], shapes[s][1], iters, d_out); break;
            case 4: ms = run<4>(shapes[s][0], shapes[s][1], iters, d_out); break;
            case 5: ms = run<5>(shapes[s][0], shapes[s][1], iters, d_out); break;
            case 6: ms = run<6>(shapes[s][0], shapes[s][1], iters, d_out); break;
            default: ms = run<7>(shapes[s][0], shapes[s][1], iters, d_out); break;
            }
            printf("%s\"%dx%d\": %.3f", s ? ", " : "", shapes[s][0], shapes[s][1], ms);
        }
        printf("}}\n");
        fflush(stdout);
    }
    return 0;
}
