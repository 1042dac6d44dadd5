// seqik_head.hip -- head / antenna angle kernel and its C ABI entry points (include/seqik.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "seqik_head.hpp"
#include "seqik_device_scope.hpp"
#include "seqik_hostctx.hpp"
#include "../../include/seqik.h"

extern "C" void seqik_set_error(int code, const char *msg);

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

// orders a wavefront's LDS writes before its reads of what OTHER lanes wrote (the hardware completes a wave's LDS
// operations in order; this is for the compiler)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One frame per lane, grid-stride.  Per frame 96 B in (two AoS records of 48 B) and 56 B out (seven SoA rows): the
// kernel is meant to run at the rate of a copy.  What that took (scripts/microbench/head_split.hip):
//   * every load of a frame is issued before its first store (the seven angles go to registers first): input and
//     output may alias as far as the compiler knows, so a store in the middle held the later loads back -- two memory
//     round trips per frame, 2.2 -> 1.75 ms for 64 M frames;
//   * STAGED: a wavefront's 64 records are one contiguous 3 KiB block per array; it comes in as three fully coalesced
//     16-byte-per-lane loads (1 KiB each, non-temporal: read once) into LDS and the lanes pick their records from
//     there; the outputs leave as non-temporal stores.  1.75 -> 1.67 ms; a copy with the same traffic takes 1.58 ms.
//     (Non-temporal loads WITHOUT the staging are a loss, 2.1 ms: the three strided loads of a lane then miss each
//     other's lines.)  Needs 16-byte aligned inputs and whole wavefronts; everything else takes the per-lane loads.
// GIVEN_ROLL: the derotation uses the caller's head roll (a.roll_in) -- its own instantiation, so that the sin / cos it
// needs stay out of the usual kernel's registers.
template <bool STAGED, bool GIVEN_ROLL = false>
__global__ void __launch_bounds__(256) seqik_head_kernel(seqik::HeadArgs a)
{
    __shared__ d2 s_stage[STAGED ? 4 * 384 : 1];  // per wavefront: 2 arrays x 3072 B = 384 x 16 B
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, n = a.n_frames;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool ant = a.compute_ant != 0;
    const int n_out = ant ? 7 : 3;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < n; t0 += stride) {
        const int64_t t = t0 + threadIdx.x;
        const int64_t w0 = t0 + wave * 64;  // first frame of this wavefront
        double out[7];
        if (STAGED && w0 + 64 <= n) {
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
            seqik::head_angles_compute(reinterpret_cast<const double *>(st) + lane * 6,
                                       reinterpret_cast<const double *>(st + 192) + lane * 6, a.neck + t * a.neck_stride,
                                       a.rest_head_pitch, a.rest_antenna_pitch, ant, out, GIVEN_ROLL ? a.roll_in + t : nullptr);
            wave_lds_fence();  // the next iteration's LDS writes stay behind these reads
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < n_out) __builtin_nontemporal_store(out[j], a.angles + j * n + t);
        } else if (t < n) {
            seqik::head_angles_compute(a.r_head + t * a.rec, a.l_head + t * a.rec, a.neck + t * a.neck_stride,
                                       a.rest_head_pitch, a.rest_antenna_pitch, ant, out, GIVEN_ROLL ? a.roll_in + t : nullptr);
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < n_out) a.angles[j * n + t] = out[j];
        }
    }
}

// angle_between_segments for general vectors: one pair per lane; a stride of 0 broadcasts one vector to every row
__global__ void __launch_bounds__(256) seqik_signed_angle_kernel(const double *v1, int64_t s1, const double *v2, int64_t s2,
                                                                 double ax, double ay, double az, int64_t n, double *out)
{
    const double axis[3] = {ax, ay, az};
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = seqik::signed_angle3(v1 + t * s1, v2 + t * s2, axis);
}

int hip_fail(hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    seqik_set_error(SEQIK_ERR_HIP, buf);
    return SEQIK_ERR_HIP;
}

#define HTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

}  // namespace

extern "C" {

int seqik_head_angles_device(const double *d_r_head, const double *d_l_head, int64_t n_frames, const double *d_neck,
                             int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch,
                             int32_t compute_ant, double *d_angles, void *hip_stream)
{
    return seqik_head_angles_ex_device(d_r_head, d_l_head, n_frames, 2, d_neck, neck_stride, rest_head_pitch,
                                       rest_antenna_pitch, compute_ant, nullptr, d_angles, hip_stream);
}

int seqik_head_angles_ex_device(const double *d_r_head, const double *d_l_head, int64_t n_frames, int32_t n_points,
                                const double *d_neck, int64_t neck_stride, double rest_head_pitch,
                                double rest_antenna_pitch, int32_t compute_ant, const double *d_head_roll,
                                double *d_angles, void *hip_stream)
{
    if (!d_r_head || !d_l_head || !d_neck || !d_angles || n_frames < 0 || (neck_stride != 0 && neck_stride != 3) ||
        n_points < 1) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_head_angles: bad argument");
        return SEQIK_ERR_BAD_ARG;
    }
    if (compute_ant && n_points < 2) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_head_angles: the antenna angles need two key points per side "
                                           "(antenna base and tip); pass compute_ant = 0 for single-point records");
        return SEQIK_ERR_BAD_ARG;
    }
    if (n_frames == 0) return SEQIK_OK;
    seqik::HeadArgs a;
    a.r_head = d_r_head; a.l_head = d_l_head; a.neck = d_neck; a.neck_stride = neck_stride;
    a.rec = 3 * (int64_t)n_points; a.roll_in = compute_ant ? d_head_roll : nullptr;
    a.rest_head_pitch = rest_head_pitch; a.rest_antenna_pitch = rest_antenna_pitch;
    a.angles = d_angles; a.n_frames = n_frames; a.compute_ant = compute_ant;
    int64_t blocks = (n_frames + 255) / 256;
    // (round 5: 64 workgroups per CU in the grid instead of 8 -- at most six are resident (LDS), the rest queue up and each
    // makes fewer grid-stride rounds: 16 M frames 0.483 -> 0.473 ms, 64 M 1.955 -> 1.902 ms; profiles/r05_head_blocks_per_cu.txt)
    static const int per_cu = getenv("SEQIK_HEAD_BLOCKS_PER_CU") ? atoi(getenv("SEQIK_HEAD_BLOCKS_PER_CU")) : 64;
    if (blocks > 256 * (int64_t)per_cu) blocks = 256 * (int64_t)per_cu;  // grid-stride beyond per_cu blocks per CU
    // staged loads need 16-byte aligned records and read the antenna tips too (only worth it when they are used)
    const bool staged = compute_ant && n_points == 2 && ((reinterpret_cast<uintptr_t>(d_r_head) | reinterpret_cast<uintptr_t>(d_l_head)) & 15) == 0;
    if (a.roll_in) hipLaunchKernelGGL((seqik_head_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    else if (staged) hipLaunchKernelGGL(seqik_head_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    else hipLaunchKernelGGL(seqik_head_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    HTRY(hipGetLastError());
    return SEQIK_OK;
}

int seqik_head_angles(const double *r_head, const double *l_head, int64_t n_frames, const double *neck,
                      int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch, int32_t compute_ant,
                      double *angles, const SeqikOptions *opt)
{
    return seqik_head_angles_ex(r_head, l_head, n_frames, 2, neck, neck_stride, rest_head_pitch, rest_antenna_pitch,
                                compute_ant, nullptr, angles, opt);
}

int seqik_head_angles_ex(const double *r_head, const double *l_head, int64_t n_frames, int32_t n_points, const double *neck,
                         int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch, int32_t compute_ant,
                         const double *head_roll, double *angles, const SeqikOptions *opt)
{
    if (!r_head || !l_head || !neck || !angles || n_frames < 0 || (neck_stride != 0 && neck_stride != 3) || n_points < 1) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_head_angles: bad argument");
        return SEQIK_ERR_BAD_ARG;
    }
    if (compute_ant && n_points < 2) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_head_angles: the antenna angles need two key points per side "
                                           "(antenna base and tip); pass compute_ant = 0 for single-point records");
        return SEQIK_ERR_BAD_ARG;
    }
    if (n_frames == 0) return SEQIK_OK;
    seqik::DeviceScope scope;
    HTRY(scope.enter(opt ? opt->device : -1));
    const int n_out = compute_ant ? 7 : 3;
    const size_t in_bytes = sizeof(double) * 3 * n_points * n_frames;
    const size_t roll_bytes = (head_roll && compute_ant) ? sizeof(double) * n_frames : 0;
    const size_t neck_bytes = sizeof(double) * (neck_stride ? 3 * n_frames : 3);
    const size_t out_bytes = sizeof(double) * 7 * n_frames;
    // a pooled context (stream + device arena) as the other host-buffer entry points: no hipMalloc / hipFree per call
    seqik::HostLeaseGuard g;
    int rc = seqik::host_lease_acquire(&g.lease);
    if (rc != SEQIK_OK) return rc;
    rc = seqik::host_lease_reserve(&g.lease, 2 * seqik::arena_padded(in_bytes) + seqik::arena_padded(neck_bytes) +
                                                 seqik::arena_padded(out_bytes) + seqik::arena_padded(roll_bytes));
    if (rc != SEQIK_OK) return rc;
    hipStream_t stream = g.lease.stream;
    char *p = g.lease.arena;
    double *d_r = reinterpret_cast<double *>(p); p += seqik::arena_padded(in_bytes);
    double *d_l = reinterpret_cast<double *>(p); p += seqik::arena_padded(in_bytes);
    double *d_n = reinterpret_cast<double *>(p); p += seqik::arena_padded(neck_bytes);
    double *d_a = reinterpret_cast<double *>(p); p += seqik::arena_padded(out_bytes);
    double *d_roll = roll_bytes ? reinterpret_cast<double *>(p) : nullptr;
    if (d_roll) HTRY(hipMemcpyAsync(d_roll, head_roll, roll_bytes, hipMemcpyHostToDevice, stream));
    HTRY(hipMemcpyAsync(d_r, r_head, in_bytes, hipMemcpyHostToDevice, stream));
    HTRY(hipMemcpyAsync(d_l, l_head, in_bytes, hipMemcpyHostToDevice, stream));
    HTRY(hipMemcpyAsync(d_n, neck, neck_bytes, hipMemcpyHostToDevice, stream));
    rc = seqik_head_angles_ex_device(d_r, d_l, n_frames, n_points, d_n, neck_stride, rest_head_pitch, rest_antenna_pitch,
                                     compute_ant, d_roll, d_a, stream);
    if (rc != SEQIK_OK) { (void)hipStreamSynchronize(stream); return rc; }
    HTRY(hipMemcpyAsync(angles, d_a, sizeof(double) * n_out * n_frames, hipMemcpyDeviceToHost, stream));
    HTRY(hipStreamSynchronize(stream));
    return SEQIK_OK;
}

int seqik_signed_angles(const double *v1, int64_t v1_stride, const double *v2, int64_t v2_stride, const double *axis,
                        int64_t n, double *out, const SeqikOptions *opt)
{
    if (!v1 || !v2 || !axis || !out || n < 0 || (v1_stride != 0 && v1_stride != 3) || (v2_stride != 0 && v2_stride != 3)) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_signed_angles: bad argument");
        return SEQIK_ERR_BAD_ARG;
    }
    if (n == 0) return SEQIK_OK;
    seqik::DeviceScope scope;
    HTRY(scope.enter(opt ? opt->device : -1));
    const size_t b1 = sizeof(double) * (v1_stride ? 3 * n : 3), b2 = sizeof(double) * (v2_stride ? 3 * n : 3);
    const size_t bo = sizeof(double) * n;
    seqik::HostLeaseGuard g;
    int rc = seqik::host_lease_acquire(&g.lease);
    if (rc != SEQIK_OK) return rc;
    rc = seqik::host_lease_reserve(&g.lease, seqik::arena_padded(b1) + seqik::arena_padded(b2) + seqik::arena_padded(bo));
    if (rc != SEQIK_OK) return rc;
    hipStream_t stream = g.lease.stream;
    char *p = g.lease.arena;
    double *d1 = reinterpret_cast<double *>(p); p += seqik::arena_padded(b1);
    double *d2v = reinterpret_cast<double *>(p); p += seqik::arena_padded(b2);
    double *d_o = reinterpret_cast<double *>(p);
    HTRY(hipMemcpyAsync(d1, v1, b1, hipMemcpyHostToDevice, stream));
    HTRY(hipMemcpyAsync(d2v, v2, b2, hipMemcpyHostToDevice, stream));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(seqik_signed_angle_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d1, v1_stride, d2v, v2_stride,
                       axis[0], axis[1], axis[2], n, d_o);
    HTRY(hipGetLastError());
    HTRY(hipMemcpyAsync(out, d_o, bo, hipMemcpyDeviceToHost, stream));
    HTRY(hipStreamSynchronize(stream));
    return SEQIK_OK;
}

}  // extern "C"
