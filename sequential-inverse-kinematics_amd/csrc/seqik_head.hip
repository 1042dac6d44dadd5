// seqik_head.hip -- head / antenna angle kernel and its C ABI entry points (include/seqik.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "seqik_head.hpp"
#include "seqik_device_scope.hpp"
#include "../../include/seqik.h"

extern "C" void seqik_set_error(int code, const char *msg);

namespace {

__global__ void __launch_bounds__(256) seqik_head_kernel(seqik::HeadArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.n_frames; t += stride)
        seqik::head_angles_frame(a, t);
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
    if (!d_r_head || !d_l_head || !d_neck || !d_angles || n_frames < 0 || (neck_stride != 0 && neck_stride != 3)) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_head_angles: bad argument");
        return SEQIK_ERR_BAD_ARG;
    }
    if (n_frames == 0) return SEQIK_OK;
    seqik::HeadArgs a;
    a.r_head = d_r_head; a.l_head = d_l_head; a.neck = d_neck; a.neck_stride = neck_stride;
    a.rest_head_pitch = rest_head_pitch; a.rest_antenna_pitch = rest_antenna_pitch;
    a.angles = d_angles; a.n_frames = n_frames; a.compute_ant = compute_ant;
    int64_t blocks = (n_frames + 255) / 256;
    static const int per_cu = getenv("SEQIK_HEAD_BLOCKS_PER_CU") ? atoi(getenv("SEQIK_HEAD_BLOCKS_PER_CU")) : 8;
    if (blocks > 256 * (int64_t)per_cu) blocks = 256 * (int64_t)per_cu;  // grid-stride beyond per_cu blocks per CU
    hipLaunchKernelGGL(seqik_head_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    HTRY(hipGetLastError());
    return SEQIK_OK;
}

int seqik_head_angles(const double *r_head, const double *l_head, int64_t n_frames, const double *neck,
                      int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch, int32_t compute_ant,
                      double *angles, const SeqikOptions *opt)
{
    if (!r_head || !l_head || !neck || !angles || n_frames < 0 || (neck_stride != 0 && neck_stride != 3)) {
        seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_head_angles: bad argument");
        return SEQIK_ERR_BAD_ARG;
    }
    if (n_frames == 0) return SEQIK_OK;
    seqik::DeviceScope scope;
    HTRY(scope.enter(opt ? opt->device : -1));
    const int n_out = compute_ant ? 7 : 3;
    const size_t in_bytes = sizeof(double) * 6 * n_frames;
    const size_t neck_bytes = sizeof(double) * (neck_stride ? 3 * n_frames : 3);
    double *d_r = nullptr, *d_l = nullptr, *d_n = nullptr, *d_a = nullptr;
    int rc = SEQIK_OK;
    do {
#define HB(expr) { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = hip_fail(e_, #expr); break; } }
        HB(hipMalloc(reinterpret_cast<void **>(&d_r), in_bytes));
        HB(hipMalloc(reinterpret_cast<void **>(&d_l), in_bytes));
        HB(hipMalloc(reinterpret_cast<void **>(&d_n), neck_bytes));
        HB(hipMalloc(reinterpret_cast<void **>(&d_a), sizeof(double) * 7 * n_frames));
        HB(hipMemcpy(d_r, r_head, in_bytes, hipMemcpyHostToDevice));
        HB(hipMemcpy(d_l, l_head, in_bytes, hipMemcpyHostToDevice));
        HB(hipMemcpy(d_n, neck, neck_bytes, hipMemcpyHostToDevice));
        rc = seqik_head_angles_device(d_r, d_l, n_frames, d_n, neck_stride, rest_head_pitch, rest_antenna_pitch,
                                      compute_ant, d_a, nullptr);
        if (rc != SEQIK_OK) break;
        HB(hipDeviceSynchronize());
        HB(hipMemcpy(angles, d_a, sizeof(double) * n_out * n_frames, hipMemcpyDeviceToHost));
#undef HB
    } while (0);
    (void)hipFree(d_r); (void)hipFree(d_l); (void)hipFree(d_n); (void)hipFree(d_a);
    return rc;
}

}  // extern "C"
