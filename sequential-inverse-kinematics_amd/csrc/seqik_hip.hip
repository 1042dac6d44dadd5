// seqik_hip.hip -- HIP kernel launchers and the C ABI of libseqik_hip.so (include/seqik.h).
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "seqik_core.hpp"
#include "seqik_consts.hpp"

namespace {

constexpr int kMaxLegs = 8;   // LegConst table staged in LDS (a fly has 6 legs)
constexpr int kMaxBlock = 256;

// Register budget: waves per SIMD the stage kernels are compiled for (512 / N registers per lane).  With the
// closed-form trust-region step the single-launch kernel needs 180 VGPRs; capping it at 168 (three waves per
// SIMD) costs 12 spilled registers and still wins: 3.10e8 -> 3.47e8 solves/s (four waves = 128 registers spill
// 120 and lose: 3.0-3.2e8).  The third wave fills the issue slots the two others leave while they wait on
// dependent f64 chains and quarter-rate reciprocal / square-root seeds.
#ifndef SEQIK_WAVES_PER_EU
#define SEQIK_WAVES_PER_EU 3
#endif

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail(SEQIK_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

struct LegOrder {  // dispatch order of the legs, see chain_of_lane()
    uint8_t leg[8];
};

struct KernelArgs {
    const double *pose;
    double *angles;
    double *fk;
    int32_t *status;
    int32_t *nfev;
    const seqik::LegConst *legs;  // device, [n_legs]
    const double *init;           // nullable [n_chains][7]
    double *frames;               // workspace [n_chains][n_frames][12] (stage hand-off), may be null
    int64_t n_chains;             // n_seq * n_legs
    int64_t n_seq;
    int64_t n_frames;
    int32_t n_legs;
    int32_t lanes_per_wave;       // W: chains a wavefront carries (1..64), see chain_of_lane()
    LegOrder leg_order;           // dispatch order of the legs
    // element strides (SeqikLayout): pose (chain, key-point row, frame), angles (chain, dof, frame)
    int64_t pose_chain, pose_row, pose_frame;
    int64_t ang_chain, ang_dof, ang_frame;
};

// Lane -> chain mapping.  A wavefront carries W <= 64 chains OF THE SAME LEG (W consecutive sequences;
// chain = seq * n_legs + leg), and the waves of one leg are adjacent in the grid, legs in the order of
// KernelArgs::leg_order:
//   * same leg = same joint limits, seeds and segment lengths in all lanes, so the lanes of a wave take the
//     bound-reflection branches together and their costs are alike (legs differ systematically: on the
//     benchmark data a front leg needs ~20 trust-region passes per stage-1 solve, a hind leg ~16).  Measured
//     with the single-launch kernel and three batches in flight: 2.52e8 -> 2.64e8 solves/s.  (With one launch
//     per stage and two batches in flight it had been a loss, 2.26e8 -> 2.09e8: leg-pure waves finish at
//     different times and every stage kernel then drains unevenly; SeqikOptions.reserved[2] = 1 selects the
//     leg-interleaved mapping, W consecutive chains per wave.)
//   * leg order = longest first: launch() sorts the legs by the total width of their joint limits (wider
//     limits -> targets further from the warm start -> more passes), so the expensive waves are dispatched
//     first and the cheap ones fill the end of the launch.  One launch alone: 39.9 -> 34.3 ms; three batches
//     in flight: 2.64e8 -> 2.66e8.  It is a scheduling heuristic only.
//   * W < 64 only when there are few chains (pick_lanes_per_wave below): a pass of a wave costs the UNION of the
//     code paths its lanes take, so a handful of chains run best as one-lane waves; at a thousand chains and
//     more W = 64.
// W < 0 encodes the leg-interleaved mapping with |W| lanes.  Returns false for lanes that carry no chain.
__device__ __forceinline__ bool chain_of_lane(int64_t n_seq, int32_t n_legs, int32_t W, const LegOrder &order,
                                              int64_t &c, int &leg)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(g & 63);
    const int64_t wave = g >> 6;
    if (W < 0) {
        c = wave * (-W) + lane;
        leg = (int)(c % n_legs);
        return lane < -W && c < n_seq * n_legs;
    }
    const int64_t n_grp = (n_seq + W - 1) / W;  // waves per leg
    const int64_t slot = wave / n_grp;          // which leg, in dispatch order
    if (slot >= n_legs) return false;
    leg = order.leg[slot];
    const int64_t seq = (wave - slot * n_grp) * W + lane;
    c = seq * n_legs + leg;
    return lane < W && seq < n_seq;
}

// Dispatch order of the legs: descending total width of the joint limits (stable).
LegOrder make_leg_order(const SeqikLegParams *legs, int32_t n_legs)
{
    LegOrder o;
    double key[8];
    for (int l = 0; l < 8; ++l) { o.leg[l] = (uint8_t)l; key[l] = 0.0; }
    for (int l = 0; l < n_legs; ++l)
        for (int d = 0; d < 7; ++d) key[l] += legs[l].bounds[d][1] - legs[l].bounds[d][0];
    for (int i = 1; i < n_legs; ++i) {  // insertion sort, descending, stable
        const uint8_t v = o.leg[i];
        int j = i - 1;
        while (j >= 0 && key[o.leg[j]] < key[v]) { o.leg[j + 1] = o.leg[j]; --j; }
        o.leg[j + 1] = v;
    }
    return o;
}

// How many chains a wavefront should carry.  Measured (smooth synthetic data, 64 frames, single launch):
//     6 chains (one recording)      W = 1: 117 ms   W = 64 (one wave of 6 lanes): 137 ms
//     1 200 chains                  W = 1: 21.9   W = 3: 14.0   W = 64: 14.8 ms
//     6 000 chains                  W = 1: 30.4   W = 6: 26.9   W = 12: 19.4   W = 64: 15.9 ms
//     24 000 chains                 W = 3: 48.7   W = 24: 17.2  W = 64: 18.9 ms
// A pass of a wave costs the union of the code paths its lanes take, so a handful of chains run best one per
// wave; but every wave -- however thin -- occupies its SIMD's issue slots for the whole pass, and a thousand thin
// waves on all CUs run slower than the same chains in a hundred full waves on a few CUs (consistent with a
// power-limited clock), so from about a thousand chains on full waves win.
int pick_lanes_per_wave(int64_t n_chains, const SeqikOptions *opt)
{
    if (opt && opt->reserved[0] >= 1 && opt->reserved[0] <= 64) return opt->reserved[0];
    if (n_chains >= 1024) return 64;
    return (int)((n_chains + 255) / 256 < 1 ? 1 : (n_chains + 255) / 256);  // <= 256 thin waves of 1-4 lanes
}

// One lane per chain, one launch per stage (the reference's own loop order,
// leg_inverse_kinematics.py:373-385).  A workgroup is one or more independent wavefronts;
// the only shared data is the read-only per-leg constant table, staged once into LDS.
// Which chain a lane gets: chain_of_lane().
template <int STAGE, bool WANT_FK, bool WANT_DIAG, bool FROM_ANGLES, bool HANDOFF>
__global__ void __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(SEQIK_WAVES_PER_EU, SEQIK_WAVES_PER_EU)))
seqik_stage_kernel(KernelArgs a)
{
    __shared__ seqik::LegConst s_legs[kMaxLegs];
    {
        const int words = a.n_legs * (int)(sizeof(seqik::LegConst) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    int64_t c;
    int leg;
    if (!chain_of_lane(a.n_seq, a.n_legs, a.lanes_per_wave, a.leg_order, c, leg)) return;

    seqik::ChainIO io;
    io.pose = a.pose + c * a.pose_chain;
    io.pose_row = a.pose_row;
    io.pose_frame = a.pose_frame;
    io.angles = a.angles + c * a.ang_chain;
    io.ang_dof = a.ang_dof;
    io.ang_frame = a.ang_frame;
    io.fk = a.fk ? a.fk + c * a.n_frames * 27 : nullptr;
    io.status = a.status ? a.status + c * a.n_frames * 4 : nullptr;
    io.nfev = a.nfev ? a.nfev + c * a.n_frames * 4 : nullptr;
    io.init = a.init ? a.init + c * 7 : nullptr;
    io.frames = a.frames ? a.frames + c * a.n_frames * 12 : nullptr;
    io.n_frames = a.n_frames;
    seqik::run_stage<STAGE, WANT_FK, WANT_DIAG, FROM_ANGLES, HANDOFF>(s_legs[leg], io);
}

// All four stages in ONE launch: a wave takes its chains through stage 1, then 2, 3, 4 (each stage over all
// frames, exactly the per-stage kernels' code, so the register budget is the largest stage's, not the sum).
// The hand-off frames a lane writes are read back by the same lane.  Saves three kernel drains per call:
// a stage kernel ends with a tail in which ever fewer waves are resident, and the next stage cannot start
// before the last wave is gone; here every wave simply carries on.
template <bool WANT_FK>
__global__ void __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(SEQIK_WAVES_PER_EU, SEQIK_WAVES_PER_EU)))
seqik_fused_kernel(KernelArgs a)
{
    __shared__ seqik::LegConst s_legs[kMaxLegs];
    {
        const int words = a.n_legs * (int)(sizeof(seqik::LegConst) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    int64_t c;
    int leg;
    if (!chain_of_lane(a.n_seq, a.n_legs, a.lanes_per_wave, a.leg_order, c, leg)) return;
    seqik::ChainIO io;
    io.pose = a.pose + c * a.pose_chain;
    io.pose_row = a.pose_row;
    io.pose_frame = a.pose_frame;
    io.angles = a.angles + c * a.ang_chain;
    io.ang_dof = a.ang_dof;
    io.ang_frame = a.ang_frame;
    io.fk = a.fk ? a.fk + c * a.n_frames * 27 : nullptr;
    io.status = nullptr;
    io.nfev = nullptr;
    io.init = a.init ? a.init + c * 7 : nullptr;
    io.frames = a.frames + c * a.n_frames * 12;
    io.n_frames = a.n_frames;
    const seqik::LegConst &lc = s_legs[leg];
    seqik::run_stage<1, false, false, false, true>(lc, io);
    seqik::run_stage<2, WANT_FK, false, false, true>(lc, io);
    seqik::run_stage<3, WANT_FK, false, false, true>(lc, io);
    seqik::run_stage<4, WANT_FK, false, false, false>(lc, io);
}

// from_angles: first stage of a run that starts after stage 1; handoff: a later stage follows
template <int STAGE, bool FROM_ANGLES, bool HANDOFF>
void launch_stage2(const KernelArgs &a, bool fk, bool diag, dim3 grid, dim3 block, hipStream_t stream)
{
    if constexpr (STAGE >= 2) {
        if (fk && diag) { hipLaunchKernelGGL((seqik_stage_kernel<STAGE, true, true, FROM_ANGLES, HANDOFF>), grid, block, 0, stream, a); return; }
        if (fk) { hipLaunchKernelGGL((seqik_stage_kernel<STAGE, true, false, FROM_ANGLES, HANDOFF>), grid, block, 0, stream, a); return; }
    }
    if (diag) hipLaunchKernelGGL((seqik_stage_kernel<STAGE, false, true, FROM_ANGLES, HANDOFF>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((seqik_stage_kernel<STAGE, false, false, FROM_ANGLES, HANDOFF>), grid, block, 0, stream, a);
}

template <int STAGE>
void launch_stage(const KernelArgs &a, bool fk, bool diag, bool from_angles, bool handoff, dim3 grid, dim3 block,
                  hipStream_t stream)
{
    if constexpr (STAGE == 1) {
        if (handoff) launch_stage2<1, false, true>(a, fk, diag, grid, block, stream);
        else launch_stage2<1, false, false>(a, fk, diag, grid, block, stream);
    } else if constexpr (STAGE == 4) {
        if (from_angles) launch_stage2<4, true, false>(a, fk, diag, grid, block, stream);
        else launch_stage2<4, false, false>(a, fk, diag, grid, block, stream);
    } else {
        if (from_angles && handoff) launch_stage2<STAGE, true, true>(a, fk, diag, grid, block, stream);
        else if (from_angles) launch_stage2<STAGE, true, false>(a, fk, diag, grid, block, stream);
        else if (handoff) launch_stage2<STAGE, false, true>(a, fk, diag, grid, block, stream);
        else launch_stage2<STAGE, false, false>(a, fk, diag, grid, block, stream);
    }
}

struct GenericLegTable {
    seqik::GenericConst gc;
    seqik::LegAffine aff;
};

struct GenericKernelArgs {
    const double *pose;
    double *angles;
    double *fk;
    int32_t *status;
    int32_t *nfev;
    const GenericLegTable *legs;
    const double *init;
    int64_t n_chains, n_seq, n_frames;
    int32_t n_legs, lanes_per_wave;
    LegOrder leg_order;
    int64_t pose_chain, pose_row, pose_frame;
    int64_t ang_chain, ang_dof, ang_frame;
};

// Generic (single 9-link chain, 7 unknowns) IK: one lane per chain, one launch.
#ifndef SEQIK_GENERIC_WAVES_PER_EU
#define SEQIK_GENERIC_WAVES_PER_EU 1
#endif
template <bool WANT_DIAG>
__global__ void __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(SEQIK_GENERIC_WAVES_PER_EU, SEQIK_GENERIC_WAVES_PER_EU)))
seqik_generic_kernel(GenericKernelArgs a)
{
    __shared__ GenericLegTable s_legs[kMaxLegs];
    {
        const int words = a.n_legs * (int)(sizeof(GenericLegTable) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    int64_t c;
    int leg;
    if (!chain_of_lane(a.n_seq, a.n_legs, a.lanes_per_wave, a.leg_order, c, leg)) return;
    seqik::GenericIO io;
    io.pose = a.pose + c * a.pose_chain; io.pose_row = a.pose_row; io.pose_frame = a.pose_frame;
    io.angles = a.angles + c * a.ang_chain; io.ang_dof = a.ang_dof; io.ang_frame = a.ang_frame;
    io.fk = a.fk ? a.fk + c * a.n_frames * 27 : nullptr;
    io.status = a.status ? a.status + c * a.n_frames : nullptr;
    io.nfev = a.nfev ? a.nfev + c * a.n_frames : nullptr;
    io.init = a.init ? a.init + c * 7 : nullptr;
    io.n_frames = a.n_frames;
    seqik::run_generic<WANT_DIAG>(s_legs[leg].gc, s_legs[leg].aff, io);
}

// Device copy of the per-leg constant table.  Callers almost always pass the same legs on
// every call, so the table is cached per host thread and device: a repeat call is a pure
// kernel launch (no allocation, no copy).  When the contents change, the device is drained
// first because kernels still in flight read the old table in their prologue.
struct LegTableCache {
    seqik::LegConst *d = nullptr;
    int device = -1;
    std::vector<seqik::LegConst> h;
};
thread_local LegTableCache g_cache;

int device_leg_table(const SeqikLegParams *legs, const SeqikAffine *affine, int32_t n_legs, const seqik::LegConst **out)
{
    std::vector<seqik::LegConst> h(n_legs);
    memset(h.data(), 0, sizeof(seqik::LegConst) * n_legs);
    for (int l = 0; l < n_legs; ++l) seqik::make_leg_consts(legs[l], affine ? affine + l : nullptr, h[l]);
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    LegTableCache &c = g_cache;
    const bool same = c.d && c.device == dev && c.h.size() == h.size() &&
                      memcmp(c.h.data(), h.data(), sizeof(seqik::LegConst) * n_legs) == 0;
    if (!same) {
        HIP_TRY(hipDeviceSynchronize());
        if (c.d && c.device != dev) c.d = nullptr;  // belongs to another device: leave it (tiny)
        if (!c.d) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d), sizeof(seqik::LegConst) * kMaxLegs));
        HIP_TRY(hipMemcpy(c.d, h.data(), sizeof(seqik::LegConst) * n_legs, hipMemcpyHostToDevice));
        c.device = dev;
        c.h = h;
    }
    *out = c.d;
    return SEQIK_OK;
}

struct GenericTableCache {
    GenericLegTable *d = nullptr;
    int device = -1;
    std::vector<GenericLegTable> h;
};
thread_local GenericTableCache g_gen_cache;

int device_generic_table(const SeqikLegParams *legs, const SeqikAffine *affine, int32_t n_legs, const GenericLegTable **out)
{
    std::vector<GenericLegTable> h(n_legs);
    memset(h.data(), 0, sizeof(GenericLegTable) * n_legs);
    for (int l = 0; l < n_legs; ++l) {
        seqik::make_generic_consts(legs[l], h[l].gc);
        seqik::LegConst tmp;
        memset(&tmp, 0, sizeof(tmp));
        seqik::make_leg_consts(legs[l], affine ? affine + l : nullptr, tmp);
        h[l].aff = tmp.aff;
    }
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    GenericTableCache &c = g_gen_cache;
    const bool same = c.d && c.device == dev && c.h.size() == h.size() &&
                      memcmp(c.h.data(), h.data(), sizeof(GenericLegTable) * n_legs) == 0;
    if (!same) {
        HIP_TRY(hipDeviceSynchronize());
        if (c.d && c.device != dev) c.d = nullptr;
        if (!c.d) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d), sizeof(GenericLegTable) * kMaxLegs));
        HIP_TRY(hipMemcpy(c.d, h.data(), sizeof(GenericLegTable) * n_legs, hipMemcpyHostToDevice));
        c.device = dev;
        c.h = h;
    }
    *out = c.d;
    return SEQIK_OK;
}

// Stage hand-off workspace (12 doubles per leg-frame), one buffer per (device, stream): launches on one stream are
// ordered, so they can share a buffer; launches on different streams never do.  (Stream-ordered hipMallocAsync /
// hipFreeAsync per call was the first implementation: with several streams in flight the pool handed a block that a
// still-running kernel of another stream was using to the next launch.)  Grown on demand, kept until
// seqik_release_workspaces() or process exit; at most kMaxWorkspaces streams are remembered (least recently used
// evicted after draining its stream).
struct Workspace {
    int device = -1;
    hipStream_t stream = nullptr;
    double *d = nullptr;
    size_t bytes = 0;
    uint64_t last_use = 0;
};
constexpr int kMaxWorkspaces = 16;
std::mutex g_ws_mutex;
std::vector<Workspace> g_ws;
uint64_t g_ws_clock = 0;

int workspace_for(hipStream_t stream, size_t bytes, double **out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace *w = nullptr;
    for (auto &e : g_ws)
        if (e.device == dev && e.stream == stream) w = &e;
    if (!w) {
        if ((int)g_ws.size() >= kMaxWorkspaces) {
            size_t lru = 0;
            for (size_t i = 1; i < g_ws.size(); ++i)
                if (g_ws[i].last_use < g_ws[lru].last_use) lru = i;
            if (g_ws[lru].d) {
                HIP_TRY(hipSetDevice(g_ws[lru].device));
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipFree(g_ws[lru].d));
                HIP_TRY(hipSetDevice(dev));
            }
            g_ws.erase(g_ws.begin() + lru);
        }
        g_ws.push_back(Workspace());
        w = &g_ws.back();
        w->device = dev;
        w->stream = stream;
    }
    if (w->bytes < bytes) {
        if (w->d) {
            HIP_TRY(hipStreamSynchronize(stream));  // an earlier launch on this stream may still use the old buffer
            HIP_TRY(hipFree(w->d));
            w->d = nullptr;
            w->bytes = 0;
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w->d), bytes));
        w->bytes = bytes;
    }
    w->last_use = ++g_ws_clock;
    *out = w->d;
    return SEQIK_OK;
}

int check_args(int64_t n_seq, int32_t n_legs, int64_t n_frames, const SeqikLegParams *legs,
               int32_t first_stage, int32_t last_stage, const void *pose, const void *angles)
{
    if (!legs || !pose || !angles) return fail(SEQIK_ERR_BAD_ARG, "null pointer argument%s");
    if (n_seq < 0 || n_frames < 0 || n_legs <= 0 || n_legs > kMaxLegs)
        return fail(SEQIK_ERR_BAD_ARG, "bad sizes (n_legs must be 1..8)%s");
    if (first_stage < 1 || last_stage > 4 || first_stage > last_stage)
        return fail(SEQIK_ERR_BAD_STAGE, "Maximum stage number is 4 and the list should be strictly incremental.%s");
    return seqik_validate_legs(legs, n_legs, first_stage, last_stage);
}

int launch(const double *d_pose, int64_t n_seq, int32_t n_legs, int64_t n_frames, const SeqikLegParams *legs,
           const seqik::LegConst *d_legs, int32_t first_stage, int32_t last_stage, double *d_angles,
           double *d_fk, int32_t *d_status, int32_t *d_nfev, const double *d_init, const SeqikLayout *layout,
           const SeqikOptions *opt, hipStream_t stream)
{
    KernelArgs a;
    a.init = d_init;
    if (layout) {
        if (layout->pose_chain < 0 || layout->pose_row <= 0 || layout->pose_frame <= 0 || layout->ang_chain < 0 ||
            layout->ang_dof <= 0 || layout->ang_frame <= 0)
            return fail(SEQIK_ERR_BAD_ARG, "layout strides must be positive%s");
        a.pose_chain = layout->pose_chain; a.pose_row = layout->pose_row; a.pose_frame = layout->pose_frame;
        a.ang_chain = layout->ang_chain; a.ang_dof = layout->ang_dof; a.ang_frame = layout->ang_frame;
    } else {
        a.pose_chain = n_frames * 15; a.pose_row = 3; a.pose_frame = 15;
        a.ang_chain = n_frames * 7; a.ang_dof = 1; a.ang_frame = 7;
    }
    a.pose = d_pose; a.angles = d_angles; a.fk = d_fk; a.status = d_status; a.nfev = d_nfev;
    a.legs = d_legs;
    a.n_chains = n_seq * (int64_t)n_legs;
    a.n_frames = n_frames;
    a.n_legs = n_legs;
    if (a.n_chains == 0 || n_frames == 0) return SEQIK_OK;
    int block = (opt && opt->block_size > 0) ? opt->block_size : 64;
    if (block % 64 != 0 || block > kMaxBlock) return fail(SEQIK_ERR_BAD_ARG, "block_size must be a multiple of 64, <= 256%s");
    a.n_seq = n_seq;
    a.leg_order = make_leg_order(legs, n_legs);
    a.lanes_per_wave = pick_lanes_per_wave(a.n_chains, opt);
    int64_t n_waves = ((n_seq + a.lanes_per_wave - 1) / a.lanes_per_wave) * n_legs;  // leg-pure waves
    if (opt && opt->reserved[2] == 1) {  // leg-interleaved: |W| consecutive chains per wave
        n_waves = (a.n_chains + a.lanes_per_wave - 1) / a.lanes_per_wave;
        a.lanes_per_wave = -a.lanes_per_wave;
    }
    int64_t grid64 = (n_waves * 64 + block - 1) / block;
    if (grid64 > 0x7fffffffLL) return fail(SEQIK_ERR_BAD_ARG, "too many chains for one launch%s");
    const dim3 grid((unsigned)grid64), blk(block);
    const bool diag = d_status || d_nfev;
    const bool fk = d_fk && last_stage == 4;  // FK is the stage-4 chain's (leg_inverse_kinematics.py:279-282)
    if (!fk) a.fk = nullptr;
    // stage hand-off workspace: the frame after the active links of stage k is the prefix of stage k + 1
    a.frames = nullptr;
    static const bool pool_workspace = getenv("SEQIK_WORKSPACE_POOL") != nullptr;  // diagnosis only (see Workspace)
    if (last_stage > first_stage) {
        const size_t ws_bytes = sizeof(double) * 12 * a.n_chains * n_frames;
        if (pool_workspace) HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&a.frames), ws_bytes, stream));
        else if (int rc = workspace_for(stream, ws_bytes, &a.frames)) return rc;
    }
    const bool fused = !(opt && opt->reserved[1] == 1) && first_stage == 1 && last_stage == 4 && !diag;
    if (fused) {
        if (opt && opt->stage_events) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[0]), stream));
        if (fk) hipLaunchKernelGGL((seqik_fused_kernel<true>), grid, blk, 0, stream, a);
        else hipLaunchKernelGGL((seqik_fused_kernel<false>), grid, blk, 0, stream, a);
        HIP_TRY(hipGetLastError());
        if (opt && opt->stage_events)  // one kernel: [0] in front of it, [1..4] behind it
            for (int k = 1; k <= 4; ++k) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[k]), stream));
    }
    for (int stage = first_stage; stage <= last_stage && !fused; ++stage) {
        const bool from_angles = (stage == first_stage) && stage > 1;
        const bool handoff = stage < last_stage;
        if (opt && opt->stage_events) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[stage - 1]), stream));
        switch (stage) {
        case 1: launch_stage<1>(a, false, diag, from_angles, handoff, grid, blk, stream); break;
        case 2: launch_stage<2>(a, fk, diag, from_angles, handoff, grid, blk, stream); break;
        case 3: launch_stage<3>(a, fk, diag, from_angles, handoff, grid, blk, stream); break;
        default: launch_stage<4>(a, fk, diag, from_angles, handoff, grid, blk, stream); break;
        }
        HIP_TRY(hipGetLastError());
    }
    if (opt && opt->stage_events && !fused) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[4]), stream));
    if (a.frames && pool_workspace) HIP_TRY(hipFreeAsync(a.frames, stream));
    return SEQIK_OK;
}

}  // namespace

extern "C" {

int seqik_abi_version(void) { return SEQIK_ABI_VERSION; }

int seqik_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *seqik_last_error(void) { return g_err; }

int seqik_release_workspaces(void)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    for (auto &w : g_ws) {
        if (!w.d) continue;
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(w.d));
        w.d = nullptr;
    }
    g_ws.clear();
    HIP_TRY(hipSetDevice(prev));
    return SEQIK_OK;
}

int seqik_device_attributes(int32_t device, int32_t *compute_units, int32_t *clock_khz, int64_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SEQIK_OK;
}

// used by the other translation units of the library (seqik_head.hip)
void seqik_set_error(int code, const char *msg) { (void)fail(code, "%s", msg); }

int seqik_validate_legs(const SeqikLegParams *legs, int32_t n_legs, int32_t first_stage, int32_t last_stage)
{
    if (!legs || n_legs <= 0) return fail(SEQIK_ERR_BAD_ARG, "null legs%s");
    if (first_stage < 1 || last_stage > 4 || first_stage > last_stage)
        return fail(SEQIK_ERR_BAD_STAGE, "Maximum stage number is 4 and the list should be strictly incremental.%s");
    for (int l = 0; l < n_legs; ++l) {
        int rc = seqik::validate_leg(legs[l], first_stage, last_stage);
        if (rc == SEQIK_ERR_BAD_BOUNDS)
            return fail(rc, "Each lower bound must be strictly less than each upper bound.%s");
        if (rc == SEQIK_ERR_X0_OUT_OF_BOUNDS)
            return fail(rc, "Initial guess is outside of provided bounds%s");
    }
    return SEQIK_OK;
}

int seqik_solve_seq_device(const double *d_pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                           const SeqikLegParams *legs, int32_t first_stage, int32_t last_stage,
                           double *d_angles, double *d_fk, int32_t *d_status, int32_t *d_nfev,
                           const double *d_init_angles, const SeqikLayout *layout, const SeqikAffine *affine,
                           const SeqikOptions *opt, void *hip_stream)
{
    int rc = check_args(n_seq, n_legs, n_frames, legs, first_stage, last_stage, d_pose, d_angles);
    if (rc != SEQIK_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const seqik::LegConst *d_legs = nullptr;
    rc = device_leg_table(legs, affine, n_legs, &d_legs);
    if (rc != SEQIK_OK) return rc;
    return launch(d_pose, n_seq, n_legs, n_frames, legs, d_legs, first_stage, last_stage, d_angles, d_fk,
                  d_status, d_nfev, d_init_angles, layout, opt, stream);
}

int seqik_validate_legs_generic(const SeqikLegParams *legs, int32_t n_legs)
{
    if (!legs || n_legs <= 0) return fail(SEQIK_ERR_BAD_ARG, "null legs%s");
    for (int l = 0; l < n_legs; ++l) {
        int rc = seqik::validate_leg_generic(legs[l]);
        if (rc == SEQIK_ERR_BAD_BOUNDS)
            return fail(rc, "Each lower bound must be strictly less than each upper bound.%s");
        if (rc == SEQIK_ERR_X0_OUT_OF_BOUNDS)
            return fail(rc, "Initial guess is outside of provided bounds%s");
    }
    return SEQIK_OK;
}

int seqik_solve_generic_device(const double *d_pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                               const SeqikLegParams *legs, double *d_angles, double *d_fk, int32_t *d_status,
                               int32_t *d_nfev, const double *d_init_angles, const SeqikLayout *layout,
                               const SeqikAffine *affine, const SeqikOptions *opt, void *hip_stream)
{
    if (!legs || !d_pose || !d_angles) return fail(SEQIK_ERR_BAD_ARG, "null pointer argument%s");
    if (n_seq < 0 || n_frames < 0 || n_legs <= 0 || n_legs > kMaxLegs)
        return fail(SEQIK_ERR_BAD_ARG, "bad sizes (n_legs must be 1..8)%s");
    int rc = seqik_validate_legs_generic(legs, n_legs);
    if (rc != SEQIK_OK) return rc;
    GenericKernelArgs a;
    a.pose = d_pose; a.angles = d_angles; a.fk = d_fk; a.status = d_status; a.nfev = d_nfev; a.init = d_init_angles;
    a.n_chains = n_seq * (int64_t)n_legs; a.n_frames = n_frames; a.n_legs = n_legs;
    if (layout) {
        a.pose_chain = layout->pose_chain; a.pose_row = layout->pose_row; a.pose_frame = layout->pose_frame;
        a.ang_chain = layout->ang_chain; a.ang_dof = layout->ang_dof; a.ang_frame = layout->ang_frame;
    } else {
        a.pose_chain = n_frames * 15; a.pose_row = 3; a.pose_frame = 15;
        a.ang_chain = n_frames * 7; a.ang_dof = 1; a.ang_frame = 7;
    }
    if (a.n_chains == 0 || n_frames == 0) return SEQIK_OK;
    rc = device_generic_table(legs, affine, n_legs, &a.legs);
    if (rc != SEQIK_OK) return rc;
    int block = (opt && opt->block_size > 0) ? opt->block_size : 64;
    if (block % 64 != 0 || block > kMaxBlock) return fail(SEQIK_ERR_BAD_ARG, "block_size must be a multiple of 64, <= 256%s");
    a.n_seq = n_seq;
    a.leg_order = make_leg_order(legs, n_legs);
    a.lanes_per_wave = pick_lanes_per_wave(a.n_chains, opt);
    int64_t n_waves = ((n_seq + a.lanes_per_wave - 1) / a.lanes_per_wave) * n_legs;
    if (opt && opt->reserved[2] == 1) {
        n_waves = (a.n_chains + a.lanes_per_wave - 1) / a.lanes_per_wave;
        a.lanes_per_wave = -a.lanes_per_wave;
    }
    const dim3 grid((unsigned)((n_waves * 64 + block - 1) / block)), blk(block);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (d_status || d_nfev) hipLaunchKernelGGL(seqik_generic_kernel<true>, grid, blk, 0, stream, a);
    else hipLaunchKernelGGL(seqik_generic_kernel<false>, grid, blk, 0, stream, a);
    HIP_TRY(hipGetLastError());
    return SEQIK_OK;
}

int seqik_solve_generic(const double *pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                        const SeqikLegParams *legs, double *angles, double *fk, int32_t *status, int32_t *nfev,
                        const double *init_angles, const SeqikAffine *affine, const SeqikOptions *opt)
{
    if (!legs || !pose || !angles) return fail(SEQIK_ERR_BAD_ARG, "null pointer argument%s");
    if (n_seq < 0 || n_frames < 0 || n_legs <= 0 || n_legs > kMaxLegs)
        return fail(SEQIK_ERR_BAD_ARG, "bad sizes (n_legs must be 1..8)%s");
    int rc = seqik_validate_legs_generic(legs, n_legs);
    if (rc != SEQIK_OK) return rc;
    const int64_t n_lf = n_seq * (int64_t)n_legs * n_frames;
    if (n_lf == 0) return SEQIK_OK;
    if (opt) HIP_TRY(hipSetDevice(opt->device));
    double *d_pose = nullptr, *d_angles = nullptr, *d_fk = nullptr, *d_init = nullptr;
    int32_t *d_status = nullptr, *d_nfev = nullptr;
    int out = SEQIK_OK;
    do {
#define TB(expr) { hipError_t e_ = (expr); if (e_ != hipSuccess) { out = fail(SEQIK_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); break; } }
        TB(hipMalloc(reinterpret_cast<void **>(&d_pose), sizeof(double) * 15 * n_lf));
        TB(hipMalloc(reinterpret_cast<void **>(&d_angles), sizeof(double) * 7 * n_lf));
        if (fk) TB(hipMalloc(reinterpret_cast<void **>(&d_fk), sizeof(double) * 27 * n_lf));
        if (status) TB(hipMalloc(reinterpret_cast<void **>(&d_status), sizeof(int32_t) * n_lf));
        if (nfev) TB(hipMalloc(reinterpret_cast<void **>(&d_nfev), sizeof(int32_t) * n_lf));
        if (init_angles) {
            TB(hipMalloc(reinterpret_cast<void **>(&d_init), sizeof(double) * 7 * n_seq * n_legs));
            TB(hipMemcpy(d_init, init_angles, sizeof(double) * 7 * n_seq * n_legs, hipMemcpyHostToDevice));
        }
        TB(hipMemcpy(d_pose, pose, sizeof(double) * 15 * n_lf, hipMemcpyHostToDevice));
        out = seqik_solve_generic_device(d_pose, n_seq, n_legs, n_frames, legs, d_angles, d_fk, d_status, d_nfev, d_init,
                                         nullptr, affine, opt, nullptr);
        if (out != SEQIK_OK) break;
        TB(hipDeviceSynchronize());
        TB(hipMemcpy(angles, d_angles, sizeof(double) * 7 * n_lf, hipMemcpyDeviceToHost));
        if (fk) TB(hipMemcpy(fk, d_fk, sizeof(double) * 27 * n_lf, hipMemcpyDeviceToHost));
        if (status) TB(hipMemcpy(status, d_status, sizeof(int32_t) * n_lf, hipMemcpyDeviceToHost));
        if (nfev) TB(hipMemcpy(nfev, d_nfev, sizeof(int32_t) * n_lf, hipMemcpyDeviceToHost));
#undef TB
    } while (0);
    (void)hipFree(d_pose); (void)hipFree(d_angles); (void)hipFree(d_fk); (void)hipFree(d_status); (void)hipFree(d_nfev);
    (void)hipFree(d_init);
    return out;
}

int seqik_solve_seq(const double *pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                    const SeqikLegParams *legs, int32_t first_stage, int32_t last_stage,
                    double *angles, double *fk, int32_t *status, int32_t *nfev, const double *init_angles,
                    const SeqikAffine *affine, const SeqikOptions *opt)
{
    int rc = check_args(n_seq, n_legs, n_frames, legs, first_stage, last_stage, pose, angles);
    if (rc != SEQIK_OK) return rc;
    const int64_t n_lf = n_seq * (int64_t)n_legs * n_frames;  // leg-frames
    if (n_lf == 0) return SEQIK_OK;
    if (opt) HIP_TRY(hipSetDevice(opt->device));
    hipStream_t stream;
    HIP_TRY(hipStreamCreate(&stream));
    double *d_pose = nullptr, *d_angles = nullptr, *d_fk = nullptr, *d_init = nullptr;
    int32_t *d_status = nullptr, *d_nfev = nullptr;
    const bool want_fk = fk && last_stage == 4;
    int out = SEQIK_OK;
    do {
#define TRY_BREAK(expr)                                                                          \
    {                                                                                            \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { out = fail(SEQIK_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); break; } \
    }
        TRY_BREAK(hipMalloc(reinterpret_cast<void **>(&d_pose), sizeof(double) * 15 * n_lf));
        TRY_BREAK(hipMalloc(reinterpret_cast<void **>(&d_angles), sizeof(double) * 7 * n_lf));
        if (want_fk) TRY_BREAK(hipMalloc(reinterpret_cast<void **>(&d_fk), sizeof(double) * 27 * n_lf));
        if (status) TRY_BREAK(hipMalloc(reinterpret_cast<void **>(&d_status), sizeof(int32_t) * 4 * n_lf));
        if (nfev) TRY_BREAK(hipMalloc(reinterpret_cast<void **>(&d_nfev), sizeof(int32_t) * 4 * n_lf));
        if (init_angles) {
            const size_t ib = sizeof(double) * 7 * n_seq * n_legs;
            TRY_BREAK(hipMalloc(reinterpret_cast<void **>(&d_init), ib));
            TRY_BREAK(hipMemcpyAsync(d_init, init_angles, ib, hipMemcpyHostToDevice, stream));
        }
        TRY_BREAK(hipMemcpyAsync(d_pose, pose, sizeof(double) * 15 * n_lf, hipMemcpyHostToDevice, stream));
        // angles is in/out: earlier-stage columns are inputs when first_stage > 1
        TRY_BREAK(hipMemcpyAsync(d_angles, angles, sizeof(double) * 7 * n_lf, hipMemcpyHostToDevice, stream));
        if (d_status) TRY_BREAK(hipMemsetAsync(d_status, 0xff, sizeof(int32_t) * 4 * n_lf, stream));
        if (d_nfev) TRY_BREAK(hipMemsetAsync(d_nfev, 0, sizeof(int32_t) * 4 * n_lf, stream));
        out = seqik_solve_seq_device(d_pose, n_seq, n_legs, n_frames, legs, first_stage, last_stage, d_angles,
                                     d_fk, d_status, d_nfev, d_init, nullptr, affine, opt, stream);
        if (out != SEQIK_OK) break;
        TRY_BREAK(hipMemcpyAsync(angles, d_angles, sizeof(double) * 7 * n_lf, hipMemcpyDeviceToHost, stream));
        if (want_fk) TRY_BREAK(hipMemcpyAsync(fk, d_fk, sizeof(double) * 27 * n_lf, hipMemcpyDeviceToHost, stream));
        if (status) TRY_BREAK(hipMemcpyAsync(status, d_status, sizeof(int32_t) * 4 * n_lf, hipMemcpyDeviceToHost, stream));
        if (nfev) TRY_BREAK(hipMemcpyAsync(nfev, d_nfev, sizeof(int32_t) * 4 * n_lf, hipMemcpyDeviceToHost, stream));
        TRY_BREAK(hipStreamSynchronize(stream));
#undef TRY_BREAK
    } while (0);
    (void)hipFree(d_pose); (void)hipFree(d_angles); (void)hipFree(d_fk); (void)hipFree(d_status); (void)hipFree(d_nfev); (void)hipFree(d_init);
    (void)hipStreamDestroy(stream);
    return out;
}

}  // extern "C"
