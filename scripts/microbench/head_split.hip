// Microbenchmark: where does the head / antenna kernel's time go?  Same frames, same launch shape as
// seqik_head_kernel (256 threads, grid-stride), variants:
//   0 full        per-lane AoS loads (48 B per lane and array, stride 48 B), arithmetic, 7 SoA stores
//   1 copy        the same loads and stores, no arithmetic (memory side alone)
//   2 math        the arithmetic on frames that stay in cache (arithmetic side alone)
//   3 staged      the two AoS inputs come in as 16-byte fully coalesced loads staged through LDS per wavefront
//   4 staged copy
//   5 staged, non-temporal loads and stores            6 the same, copy only
//   7 per-lane AoS loads as 0, non-temporal loads (16 B) and stores
//   8 pair        TWO consecutive frames per lane (128 per wavefront and iteration): 12 staged non-temporal 16-byte loads, the
//                 seven rows leave as 16-byte non-temporal stores (1 KiB contiguous per instruction)      9 the same, copy only
//   10 staged as 5, the 64 x 7 results transposed through the wavefront's LDS block and stored 16 bytes per lane (two rows
//                 per instruction: lanes 0-31 row j, lanes 32-63 row j + 1)                                 11 the same, copy only
//   12 pair with the NEXT iteration's 12 loads issued before this iteration's arithmetic (register double buffer)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o head_split head_split.hip && ./head_split [n_frames]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../sequential-inverse-kinematics_amd/csrc/seqik_head.hpp"

using seqik::HeadArgs;
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__global__ void __launch_bounds__(256) head_variant(HeadArgs a)
{
    constexpr bool STAGED = MODE >= 3 && MODE <= 6, COPY = (MODE == 1 || MODE == 4 || MODE == 6), NT = MODE >= 5;
    __shared__ d2 s_stage[STAGED ? 4 * 384 : 1];  // per wave: 2 arrays x 3072 B = 384 x 16 B
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = a.n_frames;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < n; t0 += stride) {
        const int64_t t = t0 + threadIdx.x;
        double out[7];
        bool have = false;
        if (MODE == 2) {
            const int64_t tt = t & 4095;
            seqik::head_angles_compute(a.r_head + tt * 6, a.l_head + tt * 6, a.neck, a.rest_head_pitch, a.rest_antenna_pitch, true, out);
            if (out[0] + out[1] + out[2] + out[3] + out[4] + out[5] + out[6] == 123.456) a.angles[tt] = 1.0;  // never
            continue;
        }
        if (!STAGED) {
            if (t < n) {
                const double *rb = a.r_head + t * 6, *lb = a.l_head + t * 6;
                double rr[6], ll[6];
                if (NT) {
                    const d2 *r2 = reinterpret_cast<const d2 *>(rb), *l2 = reinterpret_cast<const d2 *>(lb);
                    for (int j = 0; j < 3; ++j) {
                        const d2 x = __builtin_nontemporal_load(r2 + j), y = __builtin_nontemporal_load(l2 + j);
                        rr[2 * j] = x.x; rr[2 * j + 1] = x.y; ll[2 * j] = y.x; ll[2 * j + 1] = y.y;
                    }
                    rb = rr; lb = ll;
                }
                if (COPY) { for (int j = 0; j < 6; ++j) out[j] = rb[j] + lb[j]; out[6] = rb[0] - lb[5]; }
                else seqik::head_angles_compute(rb, lb, a.neck, a.rest_head_pitch, a.rest_antenna_pitch, true, out);
                have = true;
            }
        } else {
            d2 *st = s_stage + wave * 384;
            const int64_t w0 = t0 + wave * 64;  // first frame of this wave
            if (w0 + 64 <= n) {
                // 16-byte chunks: 192 per array, lane takes chunks lane, 64 + lane, 128 + lane (1 KiB contiguous per load)
                const d2 *gr = reinterpret_cast<const d2 *>(a.r_head + w0 * 6);
                const d2 *gl = reinterpret_cast<const d2 *>(a.l_head + w0 * 6);
                d2 r0, r1, r2, l0, l1, l2;
                if (NT) {
                    r0 = __builtin_nontemporal_load(gr + lane); r1 = __builtin_nontemporal_load(gr + 64 + lane);
                    r2 = __builtin_nontemporal_load(gr + 128 + lane);
                    l0 = __builtin_nontemporal_load(gl + lane); l1 = __builtin_nontemporal_load(gl + 64 + lane);
                    l2 = __builtin_nontemporal_load(gl + 128 + lane);
                } else {
                    r0 = gr[lane]; r1 = gr[64 + lane]; r2 = gr[128 + lane];
                    l0 = gl[lane]; l1 = gl[64 + lane]; l2 = gl[128 + lane];
                }
                st[lane] = r0; st[64 + lane] = r1; st[128 + lane] = r2;
                st[192 + lane] = l0; st[256 + lane] = l1; st[320 + lane] = l2;
                wave_lds_fence();
                const double *sr = reinterpret_cast<const double *>(st) + lane * 6;
                const double *sl = reinterpret_cast<const double *>(st + 192) + lane * 6;
                if (COPY) { for (int j = 0; j < 6; ++j) out[j] = sr[j] + sl[j]; out[6] = sr[0] - sl[5]; }
                else seqik::head_angles_compute(sr, sl, a.neck, a.rest_head_pitch, a.rest_antenna_pitch, true, out);
                have = true;
                wave_lds_fence();  // the next iteration's writes stay behind these reads
            } else if (t < n) {
                seqik::head_angles_compute(a.r_head + t * 6, a.l_head + t * 6, a.neck, a.rest_head_pitch, a.rest_antenna_pitch, true, out);
                have = true;
            }
        }
        if (have) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (NT) __builtin_nontemporal_store(out[j], a.angles + j * n + t);
                else a.angles[j * n + t] = out[j];
            }
        }
    }
}


// modes 8 / 9 / 12: two consecutive frames per lane
template <int MODE>
__global__ void __launch_bounds__(256) head_pair(HeadArgs a)
{
    constexpr bool COPY = MODE == 9, PREFETCH = MODE == 12;
    __shared__ d2 s_stage[4 * 768];  // per wave: 2 arrays x 6144 B = 768 x 16 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = a.n_frames, n_pair_waves = n / 128;      // whole 128-frame groups; the tail is left to mode 5's path
    const int64_t wstride = (int64_t)gridDim.x * 4;
    d2 *st = s_stage + wave * 768;
    d2 buf[12];
    int64_t g = (int64_t)blockIdx.x * 4 + wave;
    auto issue = [&](int64_t grp) {
        const d2 *gr = reinterpret_cast<const d2 *>(a.r_head + grp * 128 * 6);
        const d2 *gl = reinterpret_cast<const d2 *>(a.l_head + grp * 128 * 6);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            buf[k] = __builtin_nontemporal_load(gr + k * 64 + lane);
            buf[6 + k] = __builtin_nontemporal_load(gl + k * 64 + lane);
        }
    };
    if (PREFETCH && g < n_pair_waves) issue(g);
    for (; g < n_pair_waves; g += wstride) {
        if (!PREFETCH) issue(g);
#pragma unroll
        for (int k = 0; k < 6; ++k) { st[k * 64 + lane] = buf[k]; st[384 + k * 64 + lane] = buf[6 + k]; }
        wave_lds_fence();
        if (PREFETCH && g + wstride < n_pair_waves) issue(g + wstride);
        d2 o[7];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const double *sr = reinterpret_cast<const double *>(st) + (2 * lane + f) * 6;
            const double *sl = reinterpret_cast<const double *>(st + 384) + (2 * lane + f) * 6;
            double out[7];
            if (COPY) { for (int j = 0; j < 6; ++j) out[j] = sr[j] + sl[j]; out[6] = sr[0] - sl[5]; }
            else seqik::head_angles_compute(sr, sl, a.neck, a.rest_head_pitch, a.rest_antenna_pitch, true, out);
#pragma unroll
            for (int j = 0; j < 7; ++j) { if (f == 0) o[j].x = out[j]; else o[j].y = out[j]; }
        }
        wave_lds_fence();
        const int64_t t = g * 128 + 2 * lane;
#pragma unroll
        for (int j = 0; j < 7; ++j) __builtin_nontemporal_store(o[j], reinterpret_cast<d2 *>(a.angles + j * n + t));
    }
}

// modes 10 / 11: one frame per lane, results transposed through LDS, 16-byte stores (two rows per instruction)
template <int MODE>
__global__ void __launch_bounds__(256) head_tstore(HeadArgs a)
{
    constexpr bool COPY = MODE == 11;
    __shared__ d2 s_stage[4 * 384];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, n = a.n_frames;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 + 256 <= n; t0 += stride) {
        const int64_t w0 = t0 + wave * 64;
        d2 *st = s_stage + wave * 384;
        const d2 *gr = reinterpret_cast<const d2 *>(a.r_head + w0 * 6);
        const d2 *gl = reinterpret_cast<const d2 *>(a.l_head + w0 * 6);
        const d2 r0 = __builtin_nontemporal_load(gr + lane), r1 = __builtin_nontemporal_load(gr + 64 + lane),
                 r2 = __builtin_nontemporal_load(gr + 128 + lane);
        const d2 l0 = __builtin_nontemporal_load(gl + lane), l1 = __builtin_nontemporal_load(gl + 64 + lane),
                 l2 = __builtin_nontemporal_load(gl + 128 + lane);
        st[lane] = r0; st[64 + lane] = r1; st[128 + lane] = r2;
        st[192 + lane] = l0; st[256 + lane] = l1; st[320 + lane] = l2;
        wave_lds_fence();
        const double *sr = reinterpret_cast<const double *>(st) + lane * 6;
        const double *sl = reinterpret_cast<const double *>(st + 192) + lane * 6;
        double out[7];
        if (COPY) { for (int j = 0; j < 6; ++j) out[j] = sr[j] + sl[j]; out[6] = sr[0] - sl[5]; }
        else seqik::head_angles_compute(sr, sl, a.neck, a.rest_head_pitch, a.rest_antenna_pitch, true, out);
        wave_lds_fence();
        double *so = reinterpret_cast<double *>(st);           // [8][64] doubles = 4 KiB of the wave's 6 KiB block
#pragma unroll
        for (int j = 0; j < 7; ++j) so[j * 64 + lane] = out[j];
        wave_lds_fence();
        const int half = lane >> 5, l32 = lane & 31;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 2 * jj + half;                        // lanes 0-31: row 2 jj, lanes 32-63: row 2 jj + 1
            const d2 v = *reinterpret_cast<const d2 *>(so + j * 64 + 2 * l32);
            if (j < 7) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(a.angles + j * n + w0 + 2 * l32));
        }
        wave_lds_fence();
    }
}

// calibration of the box: (13) a plain 16-byte-per-lane copy of the same byte volume (76 n bytes each way, one stream in, one
// out -- the guide's "float4 copy"), (14) the kernel's read : write mix as contiguous planes (12 x 16 B read, 7 x 16 B written
// per lane and iteration), (15 / 16) the same two with non-temporal accesses
template <int MODE>
__global__ void __launch_bounds__(256) cal_copy(const d2 *__restrict__ in, d2 *__restrict__ out, int64_t n16_in, int64_t n16_out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr bool NT = MODE >= 15;
    if (MODE == 13 || MODE == 15) {
        for (int64_t i = i0; i < n16_out; i += stride) {
            const d2 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        }
    } else {
        const int64_t plane_in = n16_in / 12, plane_out = n16_out / 7;   // equal by construction (n / 2 each)
        for (int64_t i = i0; i < plane_out; i += stride) {
            d2 acc = {0.0, 0.0};
#pragma unroll
            for (int p = 0; p < 12; ++p) { const d2 v = NT ? __builtin_nontemporal_load(in + p * plane_in + i) : in[p * plane_in + i]; acc += v; }
#pragma unroll
            for (int p = 0; p < 7; ++p) { if (NT) __builtin_nontemporal_store(acc, out + p * plane_out + i); else out[p * plane_out + i] = acc; }
        }
    }
}

template <int MODE>
static float run_cal(const double *in, double *out, int64_t n, int reps, int per_cu)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // 13 / 15: 76 n bytes each way (in: the first 76 n bytes of the 96 n-byte input); 14 / 16: 96 n in, 56 n out
    const int64_t n16_in = (MODE == 13 || MODE == 15) ? n * 76 / 16 : n * 96 / 16;
    const int64_t n16_out = (MODE == 13 || MODE == 15) ? n * 76 / 16 : n * 56 / 16;
    const int64_t work = (MODE == 13 || MODE == 15) ? n16_out : n16_out / 7;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256 * (int64_t)per_cu) blocks = 256 * (int64_t)per_cu;
    auto launch = [&] { hipLaunchKernelGGL(cal_copy<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, reinterpret_cast<const d2 *>(in), reinterpret_cast<d2 *>(out), n16_in, n16_out); };
    for (int i = 0; i < 2; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

template <int MODE>
static float run_new(const HeadArgs &a, int reps, int per_cu)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int64_t per_block = (MODE == 8 || MODE == 9 || MODE == 12) ? 512 : 256;
    int64_t blocks = (a.n_frames + per_block - 1) / per_block;
    if (blocks > 256 * (int64_t)per_cu) blocks = 256 * (int64_t)per_cu;
    auto launch = [&] {
        if (MODE == 8 || MODE == 9 || MODE == 12) hipLaunchKernelGGL(head_pair<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, a);
        else hipLaunchKernelGGL(head_tstore<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, a);
    };
    for (int i = 0; i < 2; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

static int g_per_cu = 8;

template <int MODE>
static float run(const HeadArgs &a, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int64_t blocks = (a.n_frames + 255) / 256;
    if (blocks > 256 * g_per_cu) blocks = 256 * g_per_cu;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(head_variant<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, a);
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(head_variant<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int64_t n = argc > 1 ? atoll(argv[1]) : 64000000;
    if (getenv("BLOCKS_PER_CU")) g_per_cu = atoi(getenv("BLOCKS_PER_CU"));
    double *r, *l, *neck, *out;
    (void)hipMalloc(&r, n * 48); (void)hipMalloc(&l, n * 48); (void)hipMalloc(&neck, 24); (void)hipMalloc(&out, n * 56);
    {   // plausible key points: a 6000-frame pattern repeated
        double *h = (double *)malloc(6000 * 48 * 2);
        for (int i = 0; i < 6000; ++i) {
            const double ph = i * 0.01;
            double *rb = h + i * 6, *lb = h + 6000 * 6 + i * 6;
            rb[0] = 0.3 + 0.01 * sin(ph); rb[1] = -0.12 + 0.01 * cos(ph); rb[2] = 0.35 + 0.005 * sin(2 * ph);
            rb[3] = 0.45 + 0.02 * sin(ph + 1); rb[4] = -0.2 + 0.02 * cos(ph + 2); rb[5] = 0.3 + 0.02 * sin(ph + 3);
            lb[0] = 0.3 + 0.01 * cos(ph); lb[1] = 0.12 + 0.01 * sin(ph); lb[2] = 0.35 + 0.005 * cos(2 * ph);
            lb[3] = 0.45 + 0.02 * cos(ph + 1); lb[4] = 0.2 + 0.02 * sin(ph + 2); lb[5] = 0.3 + 0.02 * cos(ph + 3);
        }
        for (int64_t o = 0; o < n; o += 6000) {
            const int64_t m = (n - o) < 6000 ? (n - o) : 6000;
            (void)hipMemcpy(r + o * 6, h, m * 48, hipMemcpyHostToDevice);
            (void)hipMemcpy(l + o * 6, h + 6000 * 6, m * 48, hipMemcpyHostToDevice);
        }
        const double nk[3] = {0.0, 0.0, 0.2};
        (void)hipMemcpy(neck, nk, 24, hipMemcpyHostToDevice);
        free(h);
    }
    HeadArgs a = {};
    a.rec = 6; a.roll_in = nullptr;
    a.r_head = r; a.l_head = l; a.neck = neck; a.neck_stride = 0; a.rest_head_pitch = 0.1; a.rest_antenna_pitch = 0.2;
    a.angles = out; a.n_frames = n; a.compute_ant = 1;
    // accuracy of the device arithmetic (hardware seeds + Newton steps) against the same formulas evaluated on the host
    // with IEEE division and square root: first 6000 frames, staged variant
    double max_diff = 0.0;
    {
        hipLaunchKernelGGL(head_variant<5>, dim3(2048), dim3(256), 0, 0, a);
        (void)hipDeviceSynchronize();
        const int m = 6000;
        double *hr = (double *)malloc(m * 48), *hl = (double *)malloc(m * 48), *ho = (double *)malloc(sizeof(double) * 7 * m),
               *dev = (double *)malloc(sizeof(double) * 7 * m);
        (void)hipMemcpy(hr, r, m * 48, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hl, l, m * 48, hipMemcpyDeviceToHost);
        for (int j = 0; j < 7; ++j) (void)hipMemcpy(dev + j * m, out + j * n, m * 8, hipMemcpyDeviceToHost);
        const double nk[3] = {0.0, 0.0, 0.2};
        HeadArgs h = a;
        h.r_head = hr; h.l_head = hl; h.neck = nk; h.angles = ho; h.n_frames = m;
        for (int t = 0; t < m; ++t) seqik::head_angles_frame(h, t);
        for (int i = 0; i < 7 * m; ++i) { const double d = fabs(ho[i] - dev[i]); if (d > max_diff) max_diff = d; }
    }
    const double gb = 152.0 * n / 1e9;
    // best of 6 rounds of 10 launches each, variants interleaved (back-to-back runs of one variant differ by ~10 %)
    float t[8] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    for (int round = 0; round < 6; ++round) {
        t[0] = fminf(t[0], run<0>(a, 10)); t[1] = fminf(t[1], run<1>(a, 10)); t[2] = fminf(t[2], run<2>(a, 10));
        t[3] = fminf(t[3], run<3>(a, 10)); t[4] = fminf(t[4], run<4>(a, 10)); t[5] = fminf(t[5], run<5>(a, 10));
        t[6] = fminf(t[6], run<6>(a, 10)); t[7] = fminf(t[7], run<7>(a, 10));
    }
    printf("{\"frames\": %lld, \"blocks_per_cu\": %d, \"newton_steps\": %d, \"max_abs_diff_device_vs_host_formulas\": %.3e, "
           "\"full_ms\": %.3f, \"copy_ms\": %.3f, \"math_only_ms\": %.3f, \"staged_ms\": %.3f, \"staged_copy_ms\": %.3f, "
           "\"staged_nt_ms\": %.3f, \"staged_nt_copy_ms\": %.3f, \"full_nt_ms\": %.3f, \"TBps\": {\"full\": %.2f, \"copy\": %.2f, \"staged\": %.2f, "
           "\"staged_copy\": %.2f, \"staged_nt\": %.2f, \"staged_nt_copy\": %.2f}}\n",
           (long long)n, g_per_cu, SEQIK_HEAD_NEWTON, max_diff, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7],
           gb / t[0], gb / t[1], gb / t[3], gb / t[4], gb / t[5], gb / t[6]);
    // ---- round 5: 16-byte stores.  Results of every new variant against variant 5's (same arithmetic: must be equal) ----
    {
        const int64_t m = (n / 512) * 512;  // the new variants cover whole groups only
        double *ref = (double *)malloc(sizeof(double) * 7 * 4096), *got = (double *)malloc(sizeof(double) * 7 * 4096);
        hipLaunchKernelGGL(head_variant<5>, dim3(2048), dim3(256), 0, 0, a);
        (void)hipDeviceSynchronize();
        for (int j = 0; j < 7; ++j) (void)hipMemcpy(ref + j * 4096, out + j * n + (m - 4096), 4096 * 8, hipMemcpyDeviceToHost);
        int same[3] = {1, 1, 1};
        for (int v = 0; v < 3; ++v) {
            (void)hipMemset(out, 0, n * 56);
            if (v == 0) hipLaunchKernelGGL(head_pair<8>, dim3(2048), dim3(256), 0, 0, a);
            else if (v == 1) hipLaunchKernelGGL(head_tstore<10>, dim3(2048), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL(head_pair<12>, dim3(2048), dim3(256), 0, 0, a);
            (void)hipDeviceSynchronize();
            for (int j = 0; j < 7; ++j) (void)hipMemcpy(got + j * 4096, out + j * n + (m - 4096), 4096 * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < 7 * 4096; ++i) if (got[i] != ref[i]) same[v] = 0;
        }
        const int per_cus[5] = {3, 4, 6, 8, 16};
        for (int pc = 0; pc < 5; ++pc) {
            float u[6] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
            g_per_cu = per_cus[pc];
            for (int round = 0; round < 5; ++round) {
                u[0] = fminf(u[0], run<5>(a, 10)); u[1] = fminf(u[1], run_new<8>(a, 10, per_cus[pc])); u[2] = fminf(u[2], run_new<9>(a, 10, per_cus[pc]));
                u[3] = fminf(u[3], run_new<10>(a, 10, per_cus[pc])); u[4] = fminf(u[4], run_new<11>(a, 10, per_cus[pc]));
                u[5] = fminf(u[5], run_new<12>(a, 10, per_cus[pc]));
            }
            printf("{\"blocks_per_cu\": %d, \"equal_to_variant_5\": [%d, %d, %d], \"staged_nt_ms\": %.3f, \"pair_ms\": %.3f, \"pair_copy_ms\": %.3f, "
                   "\"tstore_ms\": %.3f, \"tstore_copy_ms\": %.3f, \"pair_prefetch_ms\": %.3f, \"TBps\": {\"staged_nt\": %.2f, \"pair\": %.2f, \"pair_copy\": %.2f, "
                   "\"tstore\": %.2f, \"tstore_copy\": %.2f, \"pair_prefetch\": %.2f}}\n", per_cus[pc], same[0], same[1], same[2], u[0], u[1], u[2], u[3], u[4], u[5],
                   gb / u[0], gb / u[1], gb / u[2], gb / u[3], gb / u[4], gb / u[5]);
        }
    }
    // ---- what THIS box's memory system gives a plain copy of the same byte volume (r is 48 n bytes, l the next allocation:
    // the calibration kernels read r / l as one 96 n-byte array only if they are adjacent, so they get their own buffers)
    {
        double *cin, *cout;
        (void)hipMalloc(&cin, n * 96); (void)hipMalloc(&cout, n * 76);
        (void)hipMemset(cin, 0, n * 96);
        const int per_cus[4] = {4, 8, 16, 64};
        for (int pc = 0; pc < 4; ++pc) {
            float c[4] = {1e9f, 1e9f, 1e9f, 1e9f};
            for (int round = 0; round < 5; ++round) {
                c[0] = fminf(c[0], run_cal<13>(cin, cout, n, 10, per_cus[pc])); c[1] = fminf(c[1], run_cal<14>(cin, cout, n, 10, per_cus[pc]));
                c[2] = fminf(c[2], run_cal<15>(cin, cout, n, 10, per_cus[pc])); c[3] = fminf(c[3], run_cal<16>(cin, cout, n, 10, per_cus[pc]));
            }
            printf("{\"calibration_blocks_per_cu\": %d, \"copy16_1to1_ms\": %.3f, \"planes_96to56_ms\": %.3f, \"copy16_1to1_nt_ms\": %.3f, \"planes_96to56_nt_ms\": %.3f, "
                   "\"TBps\": {\"copy16_1to1\": %.2f, \"planes_96to56\": %.2f, \"copy16_1to1_nt\": %.2f, \"planes_96to56_nt\": %.2f}}\n",
                   per_cus[pc], c[0], c[1], c[2], c[3], gb / c[0], gb / c[1], gb / c[2], gb / c[3]);
        }
        (void)hipFree(cin); (void)hipFree(cout);
    }
    return 0;
}
