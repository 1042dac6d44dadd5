// Stand-in for the receiving side of an RCCL point-to-point transfer: a FEW long-lived workgroups that move a large
// block inside HBM (RCCL's receive kernel copies from its FIFO into the user buffer with one workgroup of 256 threads
// per channel).  Used by scripts/contention/gather_contention.py to see, on ONE GPU, how such a kernel fares next
// to solver launches that keep every SIMD slot occupied -- and whether reserving compute units for it helps.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) fake_recv_kernel(double2 *dst, const double2 *src, int64_t n16)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

extern "C" {

int fake_recv(void *dst, const void *src, int64_t bytes, int32_t workgroups, void *stream)
{
    hipLaunchKernelGGL(fake_recv_kernel, dim3(workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<double2 *>(dst), static_cast<const double2 *>(src), bytes / 16);
    return (int)hipGetLastError();
}

// A stream whose kernels may only run on the compute units set in `mask` (n_words 32-bit words).
int masked_stream_create(void **out, const uint32_t *mask, int32_t n_words, int32_t high_priority)
{
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, n_words, mask);
    (void)high_priority;
    *out = s;
    return (int)e;
}

int masked_stream_destroy(void *s) { return (int)hipStreamDestroy(static_cast<hipStream_t>(s)); }

}
