// seqik_align.hip -- the reductions of AlignPose on the GPU (include/seqik.h, "Alignment statistics").
//
// AlignPose.align_leg (seqikpy/alignment.py:436-487) is an affine map whose constants come from whole-recording
// statistics (seqikpy/alignment.py:83-87, 392-434):
//   fixed_coxa[a]   = mean of the 0.45 and 0.55 quantiles of the coxa coordinate a          (get_fixed_pos)
//   mean length[i]  = mean of the 0.45 and 0.55 quantiles of |key point i+1 - key point i|   (get_mean_length)
// i.e. per leg SEVEN per-frame series, of each of which np.quantile needs the order statistics around two ranks.
// On the host this costs ~5 s per million frames and six legs -- more than ten times the solve -- so for long
// recordings it is the bottleneck of "alignment fused into the kernel prologue" (BASELINE config 5).
//
// Here: a kernel extracts the seven series per leg from the RAW key points (with numpy's rounding sequence:
// sqrt((dx*dx + dy*dy) + dz*dz)), hipCUB radix-sorts each series (plain library sort; order statistics are
// exact whatever the algorithm), and the requested ranks are returned.  The caller applies numpy's own
// interpolation / mean / scale formulas to them (seqikpy_amd/alignment.py), so the affine constants are
// bit-identical to the reference's.  Slabs can be added one at a time (streaming): the order of the frames does
// not matter.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <new>
#include <vector>

#include "../../include/seqik.h"
#include "seqik_device_scope.hpp"

extern "C" void seqik_set_error(int code, const char *msg);

namespace {

int a_fail(int code, const char *what, const char *detail = "")
{
    char buf[384];
    snprintf(buf, sizeof(buf), "%s%s%s", what, detail[0] ? ": " : "", detail);
    seqik_set_error(code, buf);
    return code;
}

#define ATRY(expr)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return a_fail(SEQIK_ERR_HIP, #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kSeries = 7;  // coxa x, y, z; coxa, femur, tibia, tarsus length

// series[(leg * 7 + j) * capacity + offset + seq * n_frames + t]
__global__ void __launch_bounds__(256) seqik_align_extract_kernel(
    const double *pose, int64_t n_seq, int32_t n_legs, int64_t n_frames, int64_t pose_chain, int64_t pose_row,
    int64_t pose_frame, double *series, int64_t capacity, int64_t offset)
{
    const int64_t total = n_seq * n_legs * n_frames;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = i % n_frames;
        const int64_t c = i / n_frames;  // chain = seq * n_legs + leg
        const int leg = (int)(c % n_legs);
        const int64_t seq = c / n_legs;
        const double *p = pose + c * pose_chain + t * pose_frame;
        double kp[5][3];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int a = 0; a < 3; ++a) kp[r][a] = p[r * pose_row + a];
        double *out = series + (int64_t)leg * kSeries * capacity + offset + seq * n_frames + t;
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a * capacity] = kp[0][a];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double dx = kp[r + 1][0] - kp[r][0], dy = kp[r + 1][1] - kp[r][1], dz = kp[r + 1][2] - kp[r][2];
            // np.linalg.norm(np.diff(...), axis=2): sqrt(add.reduce(x * x)) = sqrt((dx*dx + dy*dy) + dz*dz)
            out[(3 + r) * capacity] = sqrt((dx * dx + dy * dy) + dz * dz);
        }
    }
}

__global__ void seqik_align_pick_kernel(const double *sorted, int64_t capacity, int64_t n, const int64_t *ranks,
                                        int32_t n_ranks, int32_t n_series, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_series * n_ranks) return;
    const int s = i / n_ranks, r = i % n_ranks;
    int64_t k = ranks[r];
    if (k < 0) k = 0;
    if (k > n - 1) k = n - 1;
    out[i] = sorted[(int64_t)s * capacity + k];
}

}  // namespace

struct SeqikAlignStats {
    int device = 0;
    int32_t n_legs = 0;
    int64_t capacity = 0, count = 0;  // frames per leg: room / filled
    double *d_series = nullptr;       // [n_legs][7][capacity]
    double *d_stage = nullptr;        // staging for host slabs
    size_t stage_bytes = 0;
};

extern "C" {

int seqik_align_stats_open(SeqikAlignStats **out, int32_t n_legs, int64_t capacity_frames, const SeqikOptions *opt)
{
    if (!out) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_open: null handle pointer");
    *out = nullptr;
    if (n_legs <= 0 || n_legs > 8 || capacity_frames <= 0)
        return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_open: bad sizes (n_legs 1..8, capacity_frames > 0)");
    SeqikAlignStats *s = new (std::nothrow) SeqikAlignStats;
    if (!s) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_open: out of host memory");
    s->n_legs = n_legs;
    s->capacity = capacity_frames;
    seqik::DeviceScope scope;
    hipError_t e = seqik::resolve_device(opt ? opt->device : -1, &s->device);
    if (e == hipSuccess) e = scope.enter(s->device);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void **>(&s->d_series), sizeof(double) * kSeries * n_legs * capacity_frames);
    if (e != hipSuccess) {
        delete s;
        return a_fail(SEQIK_ERR_HIP, "seqik_align_stats_open", hipGetErrorString(e));
    }
    *out = s;
    return SEQIK_OK;
}

int seqik_align_stats_add(SeqikAlignStats *s, const double *pose, int32_t pose_on_device, int64_t n_seq,
                          int64_t n_frames, const SeqikLayout *layout, void *hip_stream)
{
    if (!s || !pose) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_add: null pointer");
    if (n_seq < 0 || n_frames < 0) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_add: negative size");
    const int64_t add = n_seq * n_frames;
    if (add == 0) return SEQIK_OK;
    if (s->count + add > s->capacity) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_add: more frames than the capacity");
    seqik::DeviceScope scope;
    ATRY(scope.enter(s->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int64_t pc = n_frames * 15, pr = 3, pf = 15;
    if (layout) {
        if (layout->pose_chain < 0 || layout->pose_row <= 0 || layout->pose_frame <= 0)
            return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_add: layout strides must be positive");
        pc = layout->pose_chain; pr = layout->pose_row; pf = layout->pose_frame;
    }
    const double *d_pose = pose;
    if (!pose_on_device) {
        const size_t bytes = sizeof(double) * (size_t)pc * n_seq * s->n_legs;
        if (bytes > s->stage_bytes) {
            if (s->d_stage) { ATRY(hipStreamSynchronize(stream)); (void)hipFree(s->d_stage); s->d_stage = nullptr; s->stage_bytes = 0; }
            ATRY(hipMalloc(reinterpret_cast<void **>(&s->d_stage), bytes));
            s->stage_bytes = bytes;
        }
        ATRY(hipMemcpyAsync(s->d_stage, pose, bytes, hipMemcpyHostToDevice, stream));
        d_pose = s->d_stage;
    }
    const int64_t total = add * s->n_legs;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(seqik_align_extract_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_pose, n_seq, s->n_legs,
                       n_frames, pc, pr, pf, s->d_series, s->capacity, s->count);
    ATRY(hipGetLastError());
    if (!pose_on_device) ATRY(hipStreamSynchronize(stream));  // the staging buffer / the caller's slab may be reused
    s->count += add;
    return SEQIK_OK;
}

int seqik_align_stats_finish(SeqikAlignStats *s, const int64_t *ranks, int32_t n_ranks, double *out, void *hip_stream)
{
    if (!s || !ranks || !out) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_finish: null pointer");
    if (n_ranks <= 0 || n_ranks > 16) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_finish: n_ranks must be 1..16");
    if (s->count == 0) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_finish: no frames were added");
    if (s->count > 0x7fffffffLL) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_finish: more than 2^31 frames per leg");
    seqik::DeviceScope scope;
    ATRY(scope.enter(s->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int n_series = kSeries * s->n_legs;
    double *d_sorted = nullptr, *d_out = nullptr;
    int64_t *d_ranks = nullptr;
    void *d_tmp = nullptr;
    int rc = SEQIK_OK;
    do {
#define AB(expr) { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = a_fail(SEQIK_ERR_HIP, #expr, hipGetErrorString(e_)); break; } }
        AB(hipMalloc(reinterpret_cast<void **>(&d_sorted), sizeof(double) * (size_t)n_series * s->capacity));
        AB(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(double) * n_series * n_ranks));
        AB(hipMalloc(reinterpret_cast<void **>(&d_ranks), sizeof(int64_t) * n_ranks));
        AB(hipMemcpyAsync(d_ranks, ranks, sizeof(int64_t) * n_ranks, hipMemcpyHostToDevice, stream));
        size_t tmp_bytes = 0;
        AB(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, s->d_series, d_sorted, (int)s->count, 0, 64, stream));
        AB(hipMalloc(&d_tmp, tmp_bytes));
        bool ok = true;
        for (int i = 0; i < n_series && ok; ++i) {  // one full-width sort per series (a segmented sort would
            hipError_t e_ = hipcub::DeviceRadixSort::SortKeys(       // put one block on each of these huge segments)
                d_tmp, tmp_bytes, s->d_series + (int64_t)i * s->capacity, d_sorted + (int64_t)i * s->capacity, (int)s->count,
                0, 64, stream);
            if (e_ != hipSuccess) { rc = a_fail(SEQIK_ERR_HIP, "hipcub::DeviceRadixSort::SortKeys", hipGetErrorString(e_)); ok = false; }
        }
        if (!ok) break;
        hipLaunchKernelGGL(seqik_align_pick_kernel, dim3((n_series * n_ranks + 63) / 64), dim3(64), 0, stream, d_sorted,
                           s->capacity, s->count, d_ranks, n_ranks, n_series, d_out);
        AB(hipGetLastError());
        AB(hipMemcpyAsync(out, d_out, sizeof(double) * n_series * n_ranks, hipMemcpyDeviceToHost, stream));
        AB(hipStreamSynchronize(stream));
#undef AB
    } while (0);
    (void)hipFree(d_sorted); (void)hipFree(d_out); (void)hipFree(d_ranks); (void)hipFree(d_tmp);
    return rc;
}

int seqik_align_stats_reset(SeqikAlignStats *s)
{
    if (!s) return a_fail(SEQIK_ERR_BAD_ARG, "seqik_align_stats_reset: null handle");
    s->count = 0;
    return SEQIK_OK;
}

int seqik_align_stats_close(SeqikAlignStats *s)
{
    if (!s) return SEQIK_OK;
    seqik::DeviceScope scope;
    (void)scope.enter(s->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(s->d_series);
    (void)hipFree(s->d_stage);
    delete s;
    return SEQIK_OK;
}

}  // extern "C"
