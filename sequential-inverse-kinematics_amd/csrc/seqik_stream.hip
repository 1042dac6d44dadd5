// seqik_stream.hip -- slab-streaming pipeline over host buffers (include/seqik.h, "Streaming").
//
// BASELINE config 5: a recording set that does not have to fit (or live) in HBM is pushed through
// the stage kernels in slabs.  A slab is n_seq sequences x n_legs x n_frames; the reference's
// counterpart is one run_ik_and_fk call per piece of a recording (seqikpy/leg_inverse_kinematics.py:324)
// with AlignPose.align_leg applied beforehand (seqikpy/alignment.py:436-487; here fused into the
// kernel prologue through SeqikAffine).
//
// Pipeline: HIP streams for H2D copy, compute (one per slot, up to three, so that the tail of one slab's
// kernels overlaps the head of the next -- one when slabs are carried) and D2H copy, and n_slots
// device slots, chained by events only -- the host thread blocks solely when it wants to reuse a slot whose results
// have not reached the host yet:
//
//   submit(k):  wait done[slot]  ->  h2d: copy pose -> record up[slot]
//               compute: wait up[slot] -> 4 stage kernels (+ carry kernels) -> record solved[slot]
//               d2h: wait solved[slot] -> copy angles (+ FK) -> record done[slot]
//
// PCIe is full duplex, so slab k + 1 goes up while slab k is solved and slab k - 1 comes down.
// Host buffers should be pinned (seqik_host_alloc / seqik_host_register): hipMemcpyAsync from
// pageable memory is staged by the runtime and serialises the pipeline (it still gives the
// right answer).
//
// carry != 0: the slabs are consecutive pieces IN TIME of the same n_seq recordings; frame 0 of
// slab k + 1 is warm-started from the last frame of slab k (kept on the device, no host round
// trip), exactly as the reference's frame loop would have continued (:272).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/seqik.h"
#include "seqik_device_scope.hpp"

extern "C" void seqik_set_error(int code, const char *msg);

namespace {

int s_fail(int code, const char *what, const char *detail = "")
{
    char buf[384];
    snprintf(buf, sizeof(buf), "%s%s%s", what, detail[0] ? ": " : "", detail);
    seqik_set_error(code, buf);
    return code;
}

#define STRY(expr)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return s_fail(SEQIK_ERR_HIP, #expr, hipGetErrorString(e_)); \
    } while (0)

// last-frame joint angles of every chain -> init block [chain][7] of the next slab
__global__ void __launch_bounds__(256) seqik_carry_kernel(const double *angles, double *init, int64_t n_chains,
                                                          int64_t n_frames, int64_t ang_chain, int64_t ang_dof,
                                                          int64_t ang_frame)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chains * 7) return;
    const int64_t c = i / 7;
    const int dof = (int)(i - c * 7);
    init[i] = angles[c * ang_chain + dof * ang_dof + (n_frames - 1) * ang_frame];
}

struct Slot {
    double *d_pose = nullptr, *d_angles = nullptr, *d_fk = nullptr;
    hipEvent_t up = nullptr, solved = nullptr, done = nullptr;
    bool in_flight = false;
};

}  // namespace

struct SeqikStream {
    int device = 0;
    int32_t n_legs = 0;
    int64_t slab_seq = 0, n_frames = 0;
    bool want_fk = false, carry = false, generic = false;
    bool have_layout = false;
    SeqikLayout layout{};
    std::vector<SeqikLegParams> legs;
    std::vector<SeqikAffine> affine;
    SeqikOptions opt{};
    hipStream_t h2d = nullptr, compute[3] = {nullptr, nullptr, nullptr}, d2h = nullptr;
    int n_compute = 1;
    std::vector<Slot> slots;
    double *d_init = nullptr;   // carry: [slab_seq * n_legs][7]
    bool have_init = false;     // a previous slab has left its last frame in d_init
    int64_t carry_seq = -1;     // n_seq of the slabs of a carried run (must stay the same)
    int64_t submitted = 0;
};

namespace {

void destroy(SeqikStream *s)
{
    if (!s) return;
    seqik::DeviceScope scope;
    (void)scope.enter(s->device);
    for (hipStream_t c : s->compute) if (c) (void)hipStreamSynchronize(c);
    if (s->h2d) (void)hipStreamSynchronize(s->h2d);
    if (s->d2h) (void)hipStreamSynchronize(s->d2h);
    for (Slot &q : s->slots) {
        (void)hipFree(q.d_pose); (void)hipFree(q.d_angles); (void)hipFree(q.d_fk);
        if (q.up) (void)hipEventDestroy(q.up);
        if (q.solved) (void)hipEventDestroy(q.solved);
        if (q.done) (void)hipEventDestroy(q.done);
    }
    (void)hipFree(s->d_init);
    if (s->h2d) (void)hipStreamDestroy(s->h2d);
    for (hipStream_t c : s->compute) if (c) (void)hipStreamDestroy(c);
    if (s->d2h) (void)hipStreamDestroy(s->d2h);
    delete s;
}

int open_impl(SeqikStream *s)
{
    seqik::DeviceScope scope;
    STRY(seqik::resolve_device(s->device, &s->device));
    STRY(scope.enter(s->device));
    STRY(hipStreamCreateWithFlags(&s->h2d, hipStreamNonBlocking));
    // carried slabs depend on each other: one in-order stream.  Otherwise one compute stream per slot, up to three
    // (10 M frames x 6 legs with FK: 1 stream 0.47 s, 2 0.40 s, 3 0.38 s; without FK 2 0.26 s, 3 0.24 s).  That is
    // five streams with the copies: they only overlap on separate hardware queues, i.e. with GPU_MAX_HW_QUEUES >= 8
    // (HIP's default of 4 made the third compute stream a loss: 1.48e8 -> 1.19e8 leg-frames/s).
    s->n_compute = s->carry ? 1 : (int)(s->slots.size() < 3 ? s->slots.size() : 3);
    for (int i = 0; i < s->n_compute; ++i) STRY(hipStreamCreateWithFlags(&s->compute[i], hipStreamNonBlocking));
    STRY(hipStreamCreateWithFlags(&s->d2h, hipStreamNonBlocking));
    const size_t lf = (size_t)s->slab_seq * s->n_legs * s->n_frames;
    // a custom layout may pad a chain's block (chain stride > dense size): the slots hold whole chain blocks
    const size_t n_ch = (size_t)s->slab_seq * s->n_legs;
    const size_t pose_elems = s->have_layout ? (size_t)s->layout.pose_chain * n_ch : 15 * lf;
    const size_t ang_elems = s->have_layout ? (size_t)s->layout.ang_chain * n_ch : 7 * lf;
    for (Slot &q : s->slots) {
        STRY(hipMalloc(reinterpret_cast<void **>(&q.d_pose), sizeof(double) * pose_elems));
        STRY(hipMalloc(reinterpret_cast<void **>(&q.d_angles), sizeof(double) * ang_elems));
        if (s->want_fk) STRY(hipMalloc(reinterpret_cast<void **>(&q.d_fk), sizeof(double) * 27 * lf));
        STRY(hipEventCreateWithFlags(&q.up, hipEventDisableTiming));
        STRY(hipEventCreateWithFlags(&q.solved, hipEventDisableTiming));
        STRY(hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
    }
    if (s->carry) STRY(hipMalloc(reinterpret_cast<void **>(&s->d_init), sizeof(double) * 7 * s->slab_seq * s->n_legs));
    return SEQIK_OK;
}

}  // namespace

extern "C" {

void *seqik_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        s_fail(SEQIK_ERR_HIP, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void seqik_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int seqik_host_register(void *p, size_t bytes)
{
    if (!p) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_host_register: null pointer");
    STRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return SEQIK_OK;
}

int seqik_host_unregister(void *p)
{
    if (!p) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_host_unregister: null pointer");
    STRY(hipHostUnregister(p));
    return SEQIK_OK;
}

int seqik_stream_open(SeqikStream **out, int32_t n_legs, const SeqikLegParams *legs, const SeqikAffine *affine,
                      int64_t slab_seq, int64_t n_frames, const SeqikLayout *layout, int32_t want_fk,
                      int32_t n_slots, int32_t carry, int32_t generic, const SeqikOptions *opt)
{
    if (!out) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_open: null handle pointer");
    *out = nullptr;
    if (!legs || n_legs <= 0 || n_legs > 8 || slab_seq <= 0 || n_frames <= 0 || n_slots < 1 || n_slots > 16)
        return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_open: bad sizes (n_legs 1..8, slab_seq > 0, n_frames > 0, n_slots 1..16)");
    int rc = generic ? seqik_validate_legs_generic(legs, n_legs) : seqik_validate_legs(legs, n_legs, 1, 4);
    if (rc != SEQIK_OK) return rc;
    SeqikStream *s = new (std::nothrow) SeqikStream;
    if (!s) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_open: out of host memory");
    s->device = opt ? opt->device : -1;  // resolved to the current device in open_impl
    s->n_legs = n_legs; s->slab_seq = slab_seq; s->n_frames = n_frames;
    s->want_fk = want_fk != 0; s->carry = carry != 0; s->generic = generic != 0;
    s->legs.assign(legs, legs + n_legs);
    if (affine) s->affine.assign(affine, affine + n_legs);
    if (layout) {
        // every element a kernel touches must lie inside its chain's block, and the blocks of a slab are copied
        // as one contiguous piece of n_seq * n_legs chain strides
        const int64_t T = n_frames;
        const bool ok = layout->pose_row > 0 && layout->pose_frame > 0 && layout->ang_dof > 0 && layout->ang_frame > 0 &&
                        4 * layout->pose_row + (T - 1) * layout->pose_frame + 3 <= layout->pose_chain &&
                        6 * layout->ang_dof + (T - 1) * layout->ang_frame + 1 <= layout->ang_chain;
        if (!ok) {
            delete s;
            return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_open: layout strides must be positive and every key point / "
                                              "angle of a chain must lie inside its chain stride");
        }
        s->layout = *layout;
        s->have_layout = true;
    }
    if (opt) {
        s->opt = *opt;
        s->opt.stage_events = nullptr; s->opt.chunk_stats = nullptr; s->opt.chunk_flags = nullptr; s->opt.chunk_states = nullptr;
        s->opt.chunk_resume = 0; s->opt.frame_lead = 0;
    }
    s->slots.resize(n_slots);
    rc = open_impl(s);
    if (rc != SEQIK_OK) { destroy(s); return rc; }
    *out = s;
    return SEQIK_OK;
}

int seqik_stream_submit(SeqikStream *s, const double *pose, int64_t n_seq, double *angles, double *fk)
{
    if (!s || !pose || !angles) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_submit: null pointer");
    if (n_seq < 0 || n_seq > s->slab_seq) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_submit: n_seq exceeds the slab size");
    if (s->want_fk && !fk) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_submit: the stream was opened with want_fk but fk is null");
    if (n_seq == 0) return SEQIK_OK;
    if (s->carry) {
        if (s->carry_seq >= 0 && s->carry_seq != n_seq)
            return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_submit: a carried run needs the same n_seq in every slab");
        s->carry_seq = n_seq;
    }
    seqik::DeviceScope scope;
    STRY(scope.enter(s->device));
    Slot &q = s->slots[s->submitted % (int64_t)s->slots.size()];
    if (q.in_flight) { STRY(hipEventSynchronize(q.done)); q.in_flight = false; }
    const size_t lf = (size_t)n_seq * s->n_legs * s->n_frames;
    // a custom layout may leave gaps only inside a chain block (chain stride >= dense size), so the
    // slab is copied as the contiguous block of n_seq * n_legs chains in either case
    const size_t pose_elems = s->have_layout ? (size_t)s->layout.pose_chain * n_seq * s->n_legs : 15 * lf;
    const size_t ang_elems = s->have_layout ? (size_t)s->layout.ang_chain * n_seq * s->n_legs : 7 * lf;
    STRY(hipMemcpyAsync(q.d_pose, pose, sizeof(double) * pose_elems, hipMemcpyHostToDevice, s->h2d));
    STRY(hipEventRecord(q.up, s->h2d));
    hipStream_t compute = s->compute[s->submitted % s->n_compute];
    STRY(hipStreamWaitEvent(compute, q.up, 0));
    const SeqikLayout *lay = s->have_layout ? &s->layout : nullptr;
    const SeqikAffine *aff = s->affine.empty() ? nullptr : s->affine.data();
    const double *d_init = (s->carry && s->have_init) ? s->d_init : nullptr;
    int rc;
    if (s->generic)
        rc = seqik_solve_generic_device(q.d_pose, n_seq, s->n_legs, s->n_frames, s->legs.data(), q.d_angles, q.d_fk,
                                        nullptr, nullptr, d_init, lay, aff, &s->opt, compute);
    else
        rc = seqik_solve_seq_device(q.d_pose, n_seq, s->n_legs, s->n_frames, s->legs.data(), 1, 4, q.d_angles, q.d_fk,
                                    nullptr, nullptr, d_init, lay, aff, &s->opt, compute);
    if (rc != SEQIK_OK) return rc;
    if (s->carry) {
        const int64_t n_chains = n_seq * s->n_legs;
        const int64_t ac = lay ? lay->ang_chain : s->n_frames * 7, ad = lay ? lay->ang_dof : 1, af = lay ? lay->ang_frame : 7;
        const unsigned blocks = (unsigned)((n_chains * 7 + 255) / 256);
        hipLaunchKernelGGL(seqik_carry_kernel, dim3(blocks), dim3(256), 0, compute, q.d_angles, s->d_init, n_chains,
                           s->n_frames, ac, ad, af);
        STRY(hipGetLastError());
        s->have_init = true;
    }
    STRY(hipEventRecord(q.solved, compute));
    STRY(hipStreamWaitEvent(s->d2h, q.solved, 0));
    STRY(hipMemcpyAsync(angles, q.d_angles, sizeof(double) * ang_elems, hipMemcpyDeviceToHost, s->d2h));
    if (s->want_fk) STRY(hipMemcpyAsync(fk, q.d_fk, sizeof(double) * 27 * lf, hipMemcpyDeviceToHost, s->d2h));
    STRY(hipEventRecord(q.done, s->d2h));
    q.in_flight = true;
    s->submitted += 1;
    return SEQIK_OK;
}

int seqik_stream_wait(SeqikStream *s)
{
    if (!s) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_wait: null handle");
    seqik::DeviceScope scope;
    STRY(scope.enter(s->device));
    for (Slot &q : s->slots)
        if (q.in_flight) { STRY(hipEventSynchronize(q.done)); q.in_flight = false; }
    // every slab is back: a watchdog fault in one of them must not pass silently (this pipeline's own streams only)
    int rc = SEQIK_OK;
    for (int i = 0; i < s->n_compute; ++i) {
        const int r = seqik_check_faults_stream(s->compute[i]);
        if (r != SEQIK_OK) rc = r;
    }
    return rc;
}

int seqik_stream_reset_carry(SeqikStream *s)
{
    if (!s) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_reset_carry: null handle");
    s->have_init = false;
    s->carry_seq = -1;
    return SEQIK_OK;
}

int seqik_stream_set_carry(SeqikStream *s, const double *init, int64_t n_seq, int32_t on_device)
{
    if (!s || !init) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_set_carry: null pointer");
    if (!s->carry) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_set_carry: the stream was not opened with carry");
    if (n_seq <= 0 || n_seq > s->slab_seq) return s_fail(SEQIK_ERR_BAD_ARG, "seqik_stream_set_carry: n_seq exceeds the slab size");
    seqik::DeviceScope scope;
    STRY(scope.enter(s->device));
    // ordered on the compute stream: behind the slab before (whose carry kernel writes the same buffer), in front of the next
    STRY(hipMemcpyAsync(s->d_init, init, sizeof(double) * 7 * n_seq * s->n_legs,
                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->compute[0]));
    if (!on_device) STRY(hipStreamSynchronize(s->compute[0]));  // the caller's host buffer may go away
    s->have_init = true;
    s->carry_seq = n_seq;
    return SEQIK_OK;
}

int seqik_stream_close(SeqikStream *s)
{
    destroy(s);
    return SEQIK_OK;
}

}  // extern "C"
