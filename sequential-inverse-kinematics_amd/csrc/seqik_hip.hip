// seqik_hip.hip -- HIP kernel launchers and the C ABI of libseqik_hip.so (include/seqik.h).
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "seqik_core.hpp"
#include "seqik_consts.hpp"
#include "seqik_device_scope.hpp"
#include "seqik_hostctx.hpp"

#if SEQIK_BLOCK_CYCLES
namespace seqik { __device__ unsigned long long seqik_block_cycles[4][BLK_COUNT + 1]; }
// diagnostic builds only (scripts/block_cycles.py): copies the counters out (and zeroes them)
extern "C" int seqik_debug_block_cycles(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(seqik::seqik_block_cycles), sizeof(unsigned long long) * 4 * (seqik::BLK_COUNT + 1)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zero[4 * (seqik::BLK_COUNT + 1)] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(seqik::seqik_block_cycles), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
namespace seqik { __device__ unsigned long long seqik_block_entries[4][2 * CNT_COUNT]; }
// [stage][c] = how often a wavefront went through conditional part c of a pass, [stage][CNT_COUNT + c] = active lanes summed
extern "C" int seqik_debug_block_entries(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(seqik::seqik_block_entries), sizeof(unsigned long long) * 4 * 2 * seqik::CNT_COUNT) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zero[4 * 2 * seqik::CNT_COUNT] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(seqik::seqik_block_entries), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace {

constexpr int kMaxLegs = 8;   // LegConst table staged in LDS (a fly has 6 legs)
// Automatic choice of the stage pipeline (measured: profiles/r02_launch_shape.jsonl, r02_latency_configs.jsonl): with
// thin waves replicated (chain_of_wave_lane) it beats the lane-per-chain kernels for every call that does not fill the
// GPU -- serial walks 1.9-2.0x (config 4: 295 -> 149 ms), frame chunks 1.3-1.5x (config 4: 3.35 -> 2.54 ms), 8 196
// chains x 64 frames 8.2 -> 4.5 ms, 32 784 chains 8.9 -> 4.9 ms -- and loses once all SIMD slots are taken anyway
// (93 750 chains: 24.8 vs 32.5 ms: a workgroup of four stage waves holds its slots for the time of its slowest stage).
constexpr int64_t kPipeMaxChains = 40000;
constexpr int64_t kPipeMaxChunks = 40000;
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

// ---------------------------------------------------------------------------------------------------------------
// Device faults.  The reference reports every failure as a Python exception and never returns silent garbage
// (seqikpy/leg_inverse_kinematics.py:232-236: IKPy raises when scipy's status is -1).  The one condition a kernel of
// this library can detect by itself is the stage pipeline's watchdog (run_stage, PIPED: a lane that waited 2^24 passes
// for its neighbour wave -- impossible by construction, so if it happens something is broken).  The lane then fills the
// rest of its chain with NaN, frees its neighbours AND writes the stage number into a 32-bit word in pinned,
// device-mapped HOST memory.  No launch pays for it: the store sits in the branch that never runs.
// ABI 6: ONE WORD PER (device, stream) -- kFaultSlots of them in one mapped block; a launch reports to the word of the
// stream it was made on, so a host with one thread per GPU or per stream never consumes another thread's fault (with the
// single process-wide word of ABI 4-5 thread B entering an asynchronous call would read and clear thread A's fault, refuse
// its own valid launch, and A's blocking call would return SEQIK_OK with NaN data).  Every host entry point that
// synchronises reads and clears the word OF ITS STREAM afterwards and returns SEQIK_ERR_HIP with a message; the
// asynchronous device entry points report a fault EARLIER launches on the same stream left behind when they are entered;
// seqik_check_faults_stream(stream) serves callers that synchronise a stream themselves, seqik_check_faults() reads and
// clears EVERY word (single-threaded callers, end-of-job checks).  More than kFaultSlots - 1 distinct (device, stream)
// pairs in one process share the last word: the behaviour of ABI 5, conservative (a fault is never lost).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFaultSlots = 64;
std::once_flag g_fault_once;
std::atomic<int32_t *> g_fault_host{nullptr};  // host address of the block (read by threads that never launched)
int32_t *g_fault_device = nullptr;             // the same block as the GPUs address it (written once, under g_fault_once)
std::mutex g_fault_mu;                         // guards the (device, stream) -> slot table below
int g_fault_used = 0;
int g_fault_dev[kFaultSlots];
hipStream_t g_fault_stream[kFaultSlots];

void fault_block()
{
    std::call_once(g_fault_once, [] {
        void *h = nullptr;
        const size_t bytes = sizeof(int32_t) * kFaultSlots;
        if (hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return; }
        memset(h, 0, bytes);
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_fault_device = static_cast<int32_t *>(d);
        g_fault_host.store(static_cast<int32_t *>(h), std::memory_order_release);
    });
}

// slot of (current device, stream).  `assign` = false (the CHECK paths: seqik_check_faults_stream, the entry check of the device
// entry points): lookup only, -1 for a stream nothing was ever launched on -- a caller that polls or enters on transient
// streams (torch's pools hold 32 streams per device and priority) must not use the table up.  `assign` = true (a LAUNCH):
// found or assigned; the table never shrinks (streams are pooled by every caller in this tree); when it is full the last slot
// is shared.
int fault_slot(hipStream_t stream, bool assign)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> lock(g_fault_mu);
    for (int i = 0; i < g_fault_used; ++i)
        if (g_fault_dev[i] == dev && g_fault_stream[i] == stream) return i;
    if (!assign) return g_fault_used >= kFaultSlots - 1 ? kFaultSlots - 1 : -1;  // (table full: launches share the last word)
    if (g_fault_used < kFaultSlots - 1) {
        g_fault_dev[g_fault_used] = dev;
        g_fault_stream[g_fault_used] = stream;
        return g_fault_used++;
    }
    return kFaultSlots - 1;
}

int32_t *fault_word(hipStream_t stream)
{
    fault_block();
    if (!g_fault_device) return nullptr;  // no mapped host memory on this system -- the watchdog then only leaves its NaN
    return g_fault_device + fault_slot(stream, true);
}

constexpr int kNoSlot = -2;   // check_faults: a stream nothing was launched on has no fault to report

// reads and clears the fault word of `slot` (-1: every word; kNoSlot: none); SEQIK_OK or SEQIK_ERR_HIP with the message set
int check_faults(const char *where, int slot)
{
    int32_t *words = g_fault_host.load(std::memory_order_acquire);
    if (!words || slot == kNoSlot) return SEQIK_OK;   // nothing has been launched yet (at all / on that stream)
    int32_t v = 0;
    const int lo = slot < 0 ? 0 : slot, hi = slot < 0 ? kFaultSlots : slot + 1;
    for (int i = lo; i < hi; ++i) {
        const int32_t w = __atomic_exchange_n(words + i, 0, __ATOMIC_ACQ_REL);
        if (w != 0 && v == 0) v = w;
    }
    if (v == 0) return SEQIK_OK;
    snprintf(g_err, sizeof(g_err),
             "%s: stage pipeline watchdog: a lane of stage %d waited more than %d passes for its neighbour wave; the "
             "remaining frames of its chain hold NaN -- the results of the calls since the last check are invalid",
             where, (int)v, (int)seqik::PIPE_SPIN_LIMIT);
    return SEQIK_ERR_HIP;
}

int known_slot(hipStream_t stream)
{
    const int slot = fault_slot(stream, false);
    return slot < 0 ? kNoSlot : slot;
}

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
    int32_t lane_pairs;           // stage pipeline: thin waves split a pass over lane pairs (0 = off, for measurements)
    int32_t pool;                 // chain queue (seqik_fused_queue_kernel): sequences of one leg a wavefront owns (> 64), else 0
    int32_t *fault;               // host-visible fault word ("Device faults" below); the stage pipeline's watchdog writes it
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
//   * REPLICATION.  A wavefront that keeps fewer than 16 lanes active for more than a few tens of microseconds drops
//     into a mode in which its vector instructions issue ~4.6x slower once it shares its CU with other waves (measured:
//     scripts/microbench/exec_density.hip, exec_mode.hip, profiles/r02_sparse_exec_microbench.jsonl; the threshold is
//     the number of ACTIVE lanes, wherever they sit, exited or masked alike).  So a wave that carries W < 16 chains
//     runs every chain on R ~ 64 / W lanes: the replicas load the same operands, take the same branches and store the
//     same values to the same addresses -- no additional instruction is issued, and the wave stays in the fast mode.
//     R is even, and at least 2 up to W = 32: two adjacent replicas form a lane PAIR that shares the work of a pass on the
//     stage pipeline (seqik_core.hpp "Lane pairs"; lane_pairs() below).
//     Up to W = 8 it is a multiple of 8: the generic-chain kernel splits a pass over groups of 8 adjacent replicas
//     (seqik_generic.hpp "Lane groups"; lane_groups() below).
__device__ __forceinline__ int lane_replication(int W) { return W <= 8 ? ((64 / W) & ~7) : (W <= 32 ? ((64 / W) & ~1) : 1); }
__device__ __forceinline__ bool lane_pairs(int W) { return W <= 32; }
__device__ __forceinline__ bool lane_groups(int W) { return W <= 8; }

__device__ __forceinline__ bool chain_of_wave_lane(int64_t wave, int lane, int64_t n_seq, int32_t n_legs, int32_t W,
                                                   const LegOrder &order, int64_t &c, int &leg)
{
    if (W < 0) {
        lane /= lane_replication(-W);
        c = wave * (-W) + lane;
        leg = (int)(c % n_legs);
        return lane < -W && c < n_seq * n_legs;
    }
    lane /= lane_replication(W);
    const int64_t n_grp = (n_seq + W - 1) / W;  // waves per leg
    const int64_t slot = wave / n_grp;          // which leg, in dispatch order
    if (slot >= n_legs) return false;
    leg = order.leg[slot];
    const int64_t seq = (wave - slot * n_grp) * W + lane;
    c = seq * n_legs + leg;
    return lane < W && seq < n_seq;
}

__device__ __forceinline__ bool use_pairs(const KernelArgs &a)
{
    return a.lane_pairs != 0 && lane_pairs(a.lanes_per_wave < 0 ? -a.lanes_per_wave : a.lanes_per_wave);
}

__device__ __forceinline__ bool chain_of_lane(int64_t n_seq, int32_t n_legs, int32_t W, const LegOrder &order,
                                              int64_t &c, int &leg)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    return chain_of_wave_lane(g >> 6, (int)(g & 63), n_seq, n_legs, W, order, c, leg);
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
// Frame chunks (chunked = true) are short (12-72 frames), so a call is over before the power / issue argument for full
// waves applies: measured on the shipped recordings (scripts/latency_lanes.py, profiles/r02_latency_lanes.jsonl) the
// wall-clock is flat within ~10 % around  chunks / 256  lanes per wave up to several thousand chunks
// (1 500 chunks: W = 1 4.4, 3 3.8, 8 3.9, 64 5.0 ms; 3 000 chunks: W = 1 4.0, 6-16 3.6, 64 4.0 ms), so the thin-wave rule
// simply continues (1-16 chunks per wave) up to 4 096 chunks; from there on full waves (stage pipeline, 8 196 chains:
// 16 per group 5.0 ms, 64 per group 4.5 ms; 11 718 chains, three launches in flight: 46 per group 6.7, 64 per group 5.4).
// `chunked` also stands for "on the stage pipeline".
int pick_lanes_per_wave(int64_t n_chains, const SeqikOptions *opt, bool chunked = false)
{
    if (opt && opt->reserved[0] >= 1 && opt->reserved[0] <= 64) return opt->reserved[0];
    if (chunked && n_chains < 4096) return (int)((n_chains + 255) / 256 < 1 ? 1 : (n_chains + 255) / 256);
    if (n_chains >= 1024) return 64;
    return (int)((n_chains + 255) / 256 < 1 ? 1 : (n_chains + 255) / 256);  // <= 256 thin waves of 1-4 lanes
}

// Frame chunks of a call (SeqikOptions.frame_chunk / frame_halo / frame_lead): false = serial walk.
// Automatic choice (frame_chunk = -1): a function of the recording's LENGTH ALONE, so that a recording gets the same
// chunks -- and therefore the same result, bit for bit -- whether it is solved alone, beside other recordings in one
// call, or on another day (round-2 review: it used to depend on n_chains x n_frames).  Recordings shorter than 48 frames
// stay serial; otherwise the chunk length is the one that would cut a SIX-legged recording of that length into ~196 608
// pieces (three full waves on each of the 1024 SIMDs), rounded up to a multiple of 8 and kept within 8..64 frames: the
// run-in (8 frames by default) is extra work, so chunks are not made shorter than it, and a chunk longer than 64 frames
// gains nothing (BENCH sequence-length sweep).  A recording so short that chunks of 8 would leave most of the GPU
// idle (six legs x N / 8 <= 1024, i.e. up to 1365 frames) is cut finer still -- 4 frames after a run-in of 4: the
// additional run-in work lands on idle SIMDs and the longest serial piece halves (shipped recordings: RF x 100 frames
// 0.88 -> 0.53 ms, six legs x 1000 1.06 -> 0.89).  The price of determinism: a call with MANY short recordings pays that
// doubled run-in work although it would fill the GPU anyway; such callers pass frame_chunk = 0 or explicit values.
// For the BASELINE configs this reproduces the round-2 choices (config 1 / 2: 4 + 4, config 4: 8 + 8, 1M frames: 32 + 8).
bool pick_frame_chunks(const SeqikOptions *opt, int64_t n_frames, int32_t &chunk, int32_t &halo, int32_t &lead, int64_t &n_chunks)
{
    if (!opt || opt->frame_chunk == 0) return false;
    lead = opt->frame_lead > 0 ? opt->frame_lead : 0;
    if (lead >= n_frames) return false;
    const int64_t n = n_frames - lead;  // frames that are stored
    halo = opt->frame_halo > 0 ? opt->frame_halo : 8;
    int64_t c = opt->frame_chunk;
    if (c < 0) {
        if (n < 48) return false;
        c = ((6 * n / 196608 + 7) / 8) * 8;
        c = c < 8 ? 8 : (c > 64 ? 64 : c);
        if (c == 8 && 6 * ((n + 7) / 8) <= 1024) {
            c = 4;
            if (opt->frame_halo <= 0) halo = 4;
        }
    }
    if (c >= n && lead == 0 && !opt->chunk_resume) return false;
    if (c > (1 << 20)) c = 1 << 20;
    if (halo > (1 << 20)) halo = 1 << 20;
    chunk = (int32_t)c;
    n_chunks = (n + c - 1) / c;
    return true;
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
// (Round 5: a SPLIT instantiation -- 32 chains per wavefront on lane pairs, run_stage SPLIT -- was built and measured for
// ONE job of 93 750 chains: 26.2 ms against 19.5 ms with full wavefronts; pairs save 6 % of a thin wavefront's
// instructions, halving the chains per wavefront costs 40 %.  Not kept: EXPERIMENTS.md 5.3.)
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

// CHAIN QUEUE of the fused kernel (round 6; SeqikOptions.reserved[0] = 128, 192, ... 4096: chains per wavefront).
// A wavefront of seqik_fused_kernel lives as long as the slowest of its 64 chains and spends the tail of every stage with
// ever fewer active lanes -- in the body of a pass 52 of 64 lanes are active on the benchmark's iid poses, 31-47 on the smooth ones
// (profiles/r03_block_entries_*.json).  Here a wavefront owns a POOL of `pool` consecutive sequences of one leg; its 64 lanes
// start on the first 64 and a lane that has finished its chain takes the next one of the pool (run_stage<..., QUEUE>: ballot
// arithmetic, no atomics), stage by stage: the wave finishes stage s for the whole pool, then walks the pool again for stage
// s + 1.  A stage-(s + 1) chain may be walked by another lane than its stage s was, so the hand-off frames cross lanes OF
// THE SAME WAVEFRONT through the workspace in HBM: a device-scope fence between the stages makes the stores of stage s
// visible to the loads of stage s + 1 (the vector L1 is not coherent across the lanes' earlier reads of the same lines).
// The launch has 64 / pool as many wavefronts: a caller that wants the GPU full keeps pool / 64 as many calls in flight
// (bench.py calibrates (steps in flight, pool) together).  Bound of the gain, in issue cycles, from the oracle's pass counts
// (profiles/r06_queue_bound.json): pool 128 / 256 / 512 -> 6 / 10 / 13 % (iid), 10 / 17 / 22 % (smooth).  Same bits as every
// other launch path (tests/test_gpu_parity.py::test_chain_queue_of_the_fused_kernel_bit_for_bit, the soak).
template <bool WANT_FK>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SEQIK_WAVES_PER_EU, SEQIK_WAVES_PER_EU)))
seqik_fused_queue_kernel(KernelArgs a)
{
    __shared__ seqik::LegConst s_legs[kMaxLegs];
    {
        const int words = a.n_legs * (int)(sizeof(seqik::LegConst) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    // one wavefront per workgroup: wave = blockIdx.x; the waves of a leg are adjacent, legs in dispatch order (chain_of_lane)
    const int64_t n_grp = (a.n_seq + a.pool - 1) / a.pool;   // pools (= wavefronts) per leg
    const int64_t slot = (int64_t)blockIdx.x / n_grp;
    if (slot >= a.n_legs) return;
    const int leg = a.leg_order.leg[slot];
    const int64_t pool_first = ((int64_t)blockIdx.x - slot * n_grp) * a.pool;
    const int64_t pool_end = pool_first + a.pool < a.n_seq ? pool_first + a.pool : a.n_seq;
    if (pool_first + (int64_t)threadIdx.x >= pool_end) return;   // (only in the last pool of a leg: a lane with no chain to start on)
    const int64_t c0 = pool_first * a.n_legs + leg;   // first chain of the pool: wave-uniform, so are the pointers below
    seqik::ChainIO io;
    io.pose = a.pose + c0 * a.pose_chain;
    io.pose_row = a.pose_row;
    io.pose_frame = a.pose_frame;
    io.angles = a.angles + c0 * a.ang_chain;
    io.ang_dof = a.ang_dof;
    io.ang_frame = a.ang_frame;
    io.fk = a.fk ? a.fk + c0 * a.n_frames * 27 : nullptr;
    io.status = nullptr;
    io.nfev = nullptr;
    io.init = a.init ? a.init + c0 * 7 : nullptr;
    io.frames = a.frames + c0 * a.n_frames * 12;
    io.n_frames = a.n_frames;
    io.q_seq = (int32_t)threadIdx.x;
    io.q_next = 64;
    io.q_end = (int32_t)(pool_end - pool_first);
    io.q_pose = (uint32_t)(a.n_legs * a.pose_chain);   // (launch() has checked that these fit)
    io.q_ang = (uint32_t)(a.n_legs * a.ang_chain);
    io.q_fk = (uint32_t)((int64_t)a.n_legs * a.n_frames * 27);
    io.q_init = (uint32_t)(a.n_legs * 7);
    io.q_frames = (uint32_t)((int64_t)a.n_legs * a.n_frames * 12);
    const seqik::LegConst &lc = s_legs[leg];
    // every stage starts on the lane's FIRST chain again (run_stage works on a copy of io)
    seqik::run_stage<1, false, false, false, true, false, false, false, false, true>(lc, io);
    __threadfence();
    seqik::run_stage<2, WANT_FK, false, false, true, false, false, false, false, true>(lc, io);
    __threadfence();
    seqik::run_stage<3, WANT_FK, false, false, true, false, false, false, false, true>(lc, io);
    __threadfence();
    seqik::run_stage<4, WANT_FK, false, false, false, false, false, false, false, true>(lc, io);
}

// ---------------------------------------------------------------------------------------------------------------
// Stage pipeline: one WORKGROUP of four wavefronts per group of chains, wavefront k = stage k + 1.
//
// The lane-per-chain kernels above walk a chain stage by stage (all frames of stage 1, then all of stage 2, ...): fine
// when there are thousands of chains, but a handful of recordings (the reference's own use: one recording, 1-6 legs)
// then runs at the speed of ONE lane doing 4 x N solves one after the other.  The dependencies allow more: stage s of
// frame t needs stage s - 1 of frame t and stage s of frame t - 1, so the four stages can work on four consecutive
// frames at once.  Here wave k of a workgroup runs stage k + 1 for the workgroup's chains (lane = chain, as before) and
// receives the prefix frame of every time step from wave k - 1 through a two-slot ring in LDS (PipeLane in
// seqik_core.hpp): no HBM workspace, and the serial path per frame shrinks from the sum of the four stage solves to the
// longest of them (stage 1: ~40 % of the sum on the recordings) -- what BASELINE.json's north star sketches as "one
// wavefront per chain with link transforms staged in LDS", per stage.  Same run_stage bodies, same bits.
// ---------------------------------------------------------------------------------------------------------------
struct PipeShared {
    double ring[3][seqik::PIPE_DEPTH][12][64];  // [boundary][slot][element][lane]: lanes interleaved, conflict free
    int produced[3][64];
    int consumed[3][64];
};

// pairs (wave-uniform): the wave carries every chain on an even number of adjacent lanes, lanes 2k / 2k + 1 split the
// passes of the stages with two active joints between them
// LAT: run_stage's option of the same name (the latency build; the 256-register instantiations set it)
template <bool WANT_FK, bool CHUNK_SPEC_MODE, bool LAT>
__device__ __forceinline__ void pipe_run(const seqik::LegConst &lc, seqik::ChainIO &io, PipeShared &sh, int stage_wave, int lane,
                                         bool pairs, int32_t *fault, int base = 0)
{
    seqik::PipeLane &pl = io.pipe;
    pl.lane_stride = 64;
    pl.base = base;
    pl.fault = fault;
    pl.ring_in = stage_wave > 0 ? &sh.ring[stage_wave - 1][0][0][lane] : nullptr;
    pl.produced_in = stage_wave > 0 ? &sh.produced[stage_wave - 1][lane] : nullptr;
    pl.consumed_in = stage_wave > 0 ? &sh.consumed[stage_wave - 1][lane] : nullptr;
    pl.ring_out = stage_wave < 3 ? &sh.ring[stage_wave][0][0][lane] : nullptr;
    pl.produced_out = stage_wave < 3 ? &sh.produced[stage_wave][lane] : nullptr;
    pl.consumed_out = stage_wave < 3 ? &sh.consumed[stage_wave][lane] : nullptr;
    switch (stage_wave + (pairs && stage_wave < 3 ? 4 : 0)) {  // wave-uniform
    case 0: seqik::run_stage<1, false, false, false, true, CHUNK_SPEC_MODE, true, false, LAT>(lc, io); break;
    case 1: seqik::run_stage<2, WANT_FK, false, false, true, CHUNK_SPEC_MODE, true, false, LAT>(lc, io); break;
    case 2: seqik::run_stage<3, WANT_FK, false, false, true, CHUNK_SPEC_MODE, true, false, LAT>(lc, io); break;
    case 4: seqik::run_stage<1, false, false, false, true, CHUNK_SPEC_MODE, true, true, LAT>(lc, io); break;
    case 5: seqik::run_stage<2, WANT_FK, false, false, true, CHUNK_SPEC_MODE, true, true, LAT>(lc, io); break;
    case 6: seqik::run_stage<3, WANT_FK, false, false, true, CHUNK_SPEC_MODE, true, true, LAT>(lc, io); break;
    default: seqik::run_stage<4, WANT_FK, false, false, false, CHUNK_SPEC_MODE, true, false, LAT>(lc, io); break;  // one joint
    }
}

// Register budget: as the lane-per-chain kernels, 168 registers = three waves per SIMD = three workgroups per CU (the
// kernel would take 184): 46 872 chains 18.8 -> 16.0 ms, 23 436 chains with three launches in flight 9.0 -> 7.9 ms.
// WPE: waves per SIMD the kernel is compiled for.  3 (168 registers) for grids that put three workgroups on a CU; 2
// (256 registers, nothing spilled) for the small grids of the latency regime, where a SIMD never holds more than two
// of these waves anyway: serial walk of the shipped 6000-frame recording 144 -> 139 ms.
template <bool WANT_FK, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
seqik_pipe_kernel(KernelArgs a)
{
    __shared__ seqik::LegConst s_legs[kMaxLegs];
    __shared__ PipeShared sh;
    {
        const int words = a.n_legs * (int)(sizeof(seqik::LegConst) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < 3 * 64; i += blockDim.x) { (&sh.produced[0][0])[i] = 0; (&sh.consumed[0][0])[i] = 0; }
    }
    __syncthreads();  // the only barrier: from here on the four waves are coupled by the ring counters alone
    const int stage_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t c;
    int leg;
    if (!chain_of_wave_lane(blockIdx.x, lane, a.n_seq, a.n_legs, a.lanes_per_wave, a.leg_order, c, leg)) return;
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
    io.frames = nullptr;
    io.n_frames = a.n_frames;
    pipe_run<WANT_FK, false, WPE == 2>(s_legs[leg], io, sh, stage_wave, lane, use_pairs(a), a.fault);
}

// ---------------------------------------------------------------------------------------------------------------
// Frame chunks (SeqikOptions.frame_chunk, include/seqik.h): one long recording on the whole GPU.
//
// A chain of N frames is cut into K = ceil(N / C) chunks; "virtual chain" vc = (seq * K + k) * n_legs + leg is chunk k
// of real chain c = seq * n_legs + leg.  Launch sequence on the caller's stream, no host round trip:
//     chunk kernel  (SPEC)     every virtual chain: chunk 0 from the seeds / init at frame 0, chunk k >= 1 from the
//                              seeds at frame k C - h; the h run-in frames leave only the chunk's start_state (7 angles)
//     verify + decide kernels  first verification, one thread per chunk: per-chunk report, failures counted per chain;
//                              then one thread per chain: statistics, and in automatic mode the per-chain decision
//                              "speculation failed here: walk this chain serially"
//     R x { scan kernel        which chunks are inconsistent (|start_state - true last frame of chunk k-1| > tol in
//                              some joint)?  those whose predecessor is consistent go on the work list
//           chunk kernel (REPAIR)  re-solves the listed chunks from the true state (init = the stored angles of frame
//                              k C - 1; bit-identical to the serial continuation), which becomes its start_state }
//     scan kernel + chunk kernel (SWEEP)   one wave per real chain walks its chunks left to right and re-solves what is
//                              still inconsistent: terminates after at most K steps with every chunk consistent
//     pipe kernel (SERIAL)     automatic mode: the chains the verify kernel gave up on, frame by frame from their seeds
// Every kernel after a scan that found nothing returns at once (ChunkCtrl), so on well-posed data the tail costs a
// few empty launches.  Within a round no two adjacent chunks are rewritten, and a chunk is only ever read (its last
// frame, as warm start) while nobody writes it.
// ---------------------------------------------------------------------------------------------------------------
struct ChunkCtrl {
    int32_t count;    // chunks on the work list of this round
    int32_t pending;  // chunks this round's scan found inconsistent (listed or not)
};
constexpr int kMaxChunkRounds = 8;
constexpr int kCtrlSerial = kMaxChunkRounds + 1;  // ctrl entry whose `count` is the length of the serial list

// per-chunk report bits (SeqikOptions.chunk_flags)
enum : uint8_t { CHUNK_FLAG_FAILED_FIRST = 1, CHUNK_FLAG_REPAIRED = 2, CHUNK_FLAG_SWEPT = 4, CHUNK_FLAG_SERIAL = 8,
                 // INPUT of a lockstep round (chunk_resume = 4), set by the caller on chunk 0 of a chain: the last chunk of the slab
                 // to the LEFT (another GPU's) is inconsistent in this round, so chunk 0 -- if it is inconsistent itself -- must not be
                 // repaired yet (its predecessor is about to change), exactly as inside one call; cleared by the scan that reads it
                 CHUNK_FLAG_LEFT_BLOCKED = 0x80 };

struct ChunkArgs {
    int64_t n_chunks;      // K, chunks per chain
    int64_t n_vseq;        // n_seq * K
    int32_t chunk, halo;   // C, h
    int32_t lead;          // run-in frames in front of chunk 0 (SeqikOptions.frame_lead); chunk k stores [lead + k C, ...)
    int32_t k_first;       // first chunk that is verified: 1 normally; 0 when chunk 0 is speculative too (lead > 0 /
                           // resume) AND the true state in front of it is known (KernelArgs::init)
    double tol;
    double *start_state;   // [n_chains][K][7]: the warm start the stored frames of a chunk were computed from
    int32_t *worklist;     // [n_vchains]
    ChunkCtrl *ctrl;       // [kMaxChunkRounds + 2]
    int32_t *stats;        // nullable device int32[16] (SeqikOptions.chunk_stats)
    uint8_t *flags;        // nullable device [n_chains][K] (SeqikOptions.chunk_flags)
    int32_t *fail_count;   // [n_chains]: chunks that failed the first verification
    int32_t *chain_serial; // [n_chains]: 1 = the automatic mode's guard hands this chain to the serial walk
    int32_t *serial_list;  // [n_chains]
    int32_t guard;         // automatic mode: chains with more than one chunk in eight inconsistent are walked serially
    int32_t resume;        // chunk_resume call: the per-chunk report of the call it continues is kept (bits are added)
    int32_t round;         // entry of ctrl this launch writes (scan) / reads (repair, sweep)
    int32_t n_rounds;      // R
    int32_t left_blocked;  // lockstep round: honour CHUNK_FLAG_LEFT_BLOCKED on chunk 0 (needs ca.flags)
};

enum : int { CHUNK_SPEC = 0, CHUNK_REPAIR = 1, CHUNK_SWEEP = 2, CHUNK_SERIAL = 3 };

// does the warm start chunk k of real chain c was computed from differ from the stored last frame of chunk k - 1
// (chunk 0: from the caller's true state in front of the call, KernelArgs::init)?
__device__ __forceinline__ bool chunk_inconsistent(const KernelArgs &a, const ChunkArgs &ca, int64_t c, int64_t k)
{
    const double *ss = ca.start_state + (c * ca.n_chunks + k) * 7;
    bool bad = false;
    if (k == 0) {
        // chunk_resume = 2: the first chunk is an EXACT continuation (as a carried slab of a stream): anything but the
        // caller's state bit for bit counts as inconsistent, so it is re-solved from it once
        const double *init = a.init + c * 7;
        const double tol0 = (ca.resume == 2) ? 0.0 : ca.tol;
#pragma unroll
        for (int d = 0; d < 7; ++d) bad |= !(fabs(ss[d] - init[d]) <= tol0);
    } else {
        const double *ang = a.angles + c * a.ang_chain + (ca.lead + k * ca.chunk - 1) * a.ang_frame;
#pragma unroll
        for (int d = 0; d < 7; ++d) bad |= !(fabs(ss[d] - ang[d * a.ang_dof]) <= ca.tol);  // NaN counts as a mismatch
    }
    return bad;
}

// First verification, one thread per chunk (chunk-major, so the lanes of a wavefront sit on one chain and one atomic per
// wavefront counts its failures): which chunks start from a state that is not the true one?  Fills the per-chunk report and
// counts per chain.  seqik_chunk_decide_kernel (one thread per chain) then decides -- automatic mode -- PER CHAIN whether
// speculation is worth keeping: a chain with more than one chunk in eight inconsistent (random poses with several
// equivalent leg configurations do that: a run-in then lands in another configuration than the serial walk about half of
// the time) is put on the serial list; the scan / repair / sweep kernels leave it alone and
// seqik_chunk_pipe_kernel<CHUNK_SERIAL> walks it frame by frame from its seeds, which is the reference's result bit for bit.
// Per chain, so the decision for a recording does not depend on what else is in the call.
__global__ void __launch_bounds__(256) seqik_chunk_verify_kernel(KernelArgs a, ChunkArgs ca)
{
    const int64_t K = ca.n_chunks;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < a.n_chains * K;
    const int64_t c = live ? t / K : 0, k = live ? t - c * K : 0;
    const bool inc = live && k >= ca.k_first && chunk_inconsistent(a, ca, c, k);
    if (live && ca.flags) ca.flags[t] = (ca.resume ? ca.flags[t] : 0) | (inc ? CHUNK_FLAG_FAILED_FIRST : 0);
    // one atomic per (wavefront, chain): the first failing lane's chain is counted by ballot, a lane of another chain (a
    // wavefront that straddles a chain boundary) adds its own
    const unsigned long long m = __ballot(inc);
    if (m) {
        const int first = __ffsll((long long)m) - 1;
        const int64_t c0 = __shfl(c, first);
        const unsigned long long same = __ballot(inc && c == c0);
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&ca.fail_count[c0], (int)__popcll(same));
        else if (inc && c != c0) atomicAdd(&ca.fail_count[c], 1);
    }
}

__global__ void __launch_bounds__(256) seqik_chunk_decide_kernel(KernelArgs a, ChunkArgs ca)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n_chains) return;
    const int fails = ca.fail_count[c];
    if (ca.stats && fails) atomicAdd(&ca.stats[7], fails);
    if (ca.guard && (int64_t)fails * 8 > ca.n_chunks) {
        ca.chain_serial[c] = 1;
        ca.serial_list[atomicAdd(&ca.ctrl[kCtrlSerial].count, 1)] = (int32_t)c;
        if (ca.stats) { atomicAdd(&ca.stats[8], 1); atomicAdd(&ca.stats[9], (int32_t)ca.n_chunks); }
    }
}

__global__ void __launch_bounds__(256) seqik_chunk_scan_kernel(KernelArgs a, ChunkArgs ca)
{
    __shared__ int s_count, s_pending, s_base;
    if (ca.round > 0 && ca.ctrl[ca.round - 1].pending == 0) return;  // the previous scan found every chunk consistent
    if (threadIdx.x == 0) { s_count = 0; s_pending = 0; s_base = 0; }
    __syncthreads();
    const int64_t vc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool inc = false, ready = false;
    if (vc < ca.n_vseq * a.n_legs) {
        const int64_t vseq = vc / a.n_legs;
        const int leg = (int)(vc - vseq * a.n_legs);
        const int64_t seq = vseq / ca.n_chunks, k = vseq - seq * ca.n_chunks;
        const int64_t c = seq * a.n_legs + leg;
        if (k >= ca.k_first && !ca.chain_serial[c]) {
            inc = chunk_inconsistent(a, ca, c, k);
            // repaired now only if the chunk in front of it is not about to change
            if (inc) ready = !(k > ca.k_first && chunk_inconsistent(a, ca, c, k - 1));
            if (ca.left_blocked && k == 0 && ca.k_first == 0) {   // the chunk in front of chunk 0 lives on another GPU
                uint8_t *fl = ca.flags + c * ca.n_chunks;
                if (*fl & CHUNK_FLAG_LEFT_BLOCKED) { ready = false; *fl &= (uint8_t)~CHUNK_FLAG_LEFT_BLOCKED; }
            }
        }
    }
    int mine = -1;
    if (ready) mine = atomicAdd(&s_count, 1);
    if (inc) atomicAdd(&s_pending, 1);
    __syncthreads();
    if (threadIdx.x == 0) {  // one device atomic per workgroup
        if (s_count) s_base = atomicAdd(&ca.ctrl[ca.round].count, s_count);
        if (s_pending) atomicAdd(&ca.ctrl[ca.round].pending, s_pending);
        if (ca.stats && ca.round < ca.n_rounds && s_count) atomicAdd(&ca.stats[3 + (ca.round < 2 ? ca.round : 2)], s_count);
    }
    __syncthreads();
    if (ready) ca.worklist[s_base + mine] = (int32_t)vc;
}

// ChainIO of virtual chain vc (chunk k of real chain c): speculative (run-in from the seeds / the caller's init) or
// repair (from the stored last frame of chunk k - 1 -- chunk 0: from the caller's true state --, which also becomes the
// chunk's recorded start state)
__device__ __forceinline__ void chunk_io(const KernelArgs &a, const ChunkArgs &ca, int64_t vc, int leg, bool spec,
                                         seqik::ChainIO &io, uint8_t flag = 0)
{
    const int64_t K = ca.n_chunks, C = ca.chunk, N = a.n_frames, lead = ca.lead;
    const int64_t vseq = vc / a.n_legs;
    const int64_t seq = vseq / K, k = vseq - seq * K;
    const int64_t c = seq * a.n_legs + leg;
    io.pose = a.pose + c * a.pose_chain;
    io.pose_row = a.pose_row;
    io.pose_frame = a.pose_frame;
    io.angles = a.angles + c * a.ang_chain;
    io.ang_dof = a.ang_dof;
    io.ang_frame = a.ang_frame;
    io.fk = a.fk ? a.fk + c * N * 27 : nullptr;
    io.status = nullptr;
    io.nfev = nullptr;
    const int64_t ws_frames = C + (ca.halo > lead ? ca.halo : lead);
    io.frames = a.frames + vc * ws_frames * 12;
    io.t_store = lead + k * C;
    io.n_frames = lead + (k + 1) * C < N ? lead + (k + 1) * C : N;
    double *ss = ca.start_state + (c * K + k) * 7;
    if (spec) {
        const bool run_in = k > 0 || lead > 0;  // chunk 0 of a call with a lead is speculative like the others
        io.t_begin = (k > 0 && io.t_store > ca.halo) ? io.t_store - ca.halo : 0;
        io.init = (!run_in && a.init) ? a.init + c * 7 : nullptr;
        io.init_stride = 1;
        io.start_state = run_in ? ss : nullptr;
    } else {
        io.t_begin = io.t_store;
        if (k == 0) { io.init = a.init + c * 7; io.init_stride = 1; }
        else { io.init = io.angles + (io.t_store - 1) * a.ang_frame; io.init_stride = a.ang_dof; }
        io.start_state = nullptr;
#pragma unroll
        for (int d = 0; d < 7; ++d) ss[d] = io.init[d * io.init_stride];
        if (ca.flags && flag) ca.flags[c * K + k] |= flag;  // (replicas write the same byte)
    }
}

// ChainIO of real chain c walked serially from frame 0 (the guard's fallback): the whole call, everything stored
__device__ __forceinline__ void serial_io(const KernelArgs &a, int64_t c, seqik::ChainIO &io)
{
    io.pose = a.pose + c * a.pose_chain;
    io.pose_row = a.pose_row;
    io.pose_frame = a.pose_frame;
    io.angles = a.angles + c * a.ang_chain;
    io.ang_dof = a.ang_dof;
    io.ang_frame = a.ang_frame;
    io.fk = a.fk ? a.fk + c * a.n_frames * 27 : nullptr;
    io.status = nullptr;
    io.nfev = nullptr;
    io.frames = nullptr;
    io.t_begin = 0;
    io.t_store = 0;
    io.n_frames = a.n_frames;
    io.init = a.init ? a.init + c * 7 : nullptr;
    io.init_stride = 1;
    io.start_state = nullptr;
}

// Solves chunks: the four stage bodies back to back, as seqik_fused_kernel, over the frames of one chunk per lane.
//   CHUNK_SPEC    lane -> virtual chain by chain_of_lane() (leg-pure waves), run-in from the seeds
//   CHUNK_REPAIR  lanes take the entries of the work list of round ca.round (grid-stride), start from the true state
//   CHUNK_SWEEP   wave w = real chain w: verify 64 chunks at a time, re-solve the first inconsistent one (on all 64
//                 lanes, as replicas), continue behind it (the verification then sees the new last frame)
template <bool WANT_FK, int mode>
__global__ void __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(SEQIK_WAVES_PER_EU, SEQIK_WAVES_PER_EU)))
seqik_chunk_kernel(KernelArgs a, ChunkArgs ca)
{
    __shared__ seqik::LegConst s_legs[kMaxLegs];
    {
        const int words = a.n_legs * (int)(sizeof(seqik::LegConst) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(g & 63);
    const int64_t wave = g >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t K = ca.n_chunks, C = ca.chunk, N = a.n_frames;
    int64_t W = a.lanes_per_wave < 0 ? -a.lanes_per_wave : a.lanes_per_wave;

    int64_t cursor = 0, n_items = 0;
    bool spec_done = false;
    if (mode == CHUNK_REPAIR) {
        n_items = ca.ctrl[ca.round].count;
        // the listed chunks are spread over the waves of the grid as thinly as possible (a handful of repairs run one
        // per wave: a pass of a wave costs the union of the code paths its lanes take)
        W = (n_items + n_waves - 1) / n_waves;
        W = W < 1 ? 1 : (W > 64 ? 64 : W);
        const int cl = lane / lane_replication((int)W);  // thin waves: every item on 64 / W lanes (see chain_of_wave_lane)
        cursor = (cl < W) ? wave * W + cl : n_items;
    } else if (mode == CHUNK_SWEEP) {
        if (ca.ctrl[ca.round].pending == 0 || wave >= a.n_chains || ca.chain_serial[wave]) return;
        cursor = ca.k_first;  // wave-uniform: next chunk of real chain `wave` to verify
    }
    for (;;) {
        int64_t vc = -1;
        int leg = 0;
        if (mode == CHUNK_SPEC) {
            if (spec_done) break;
            spec_done = true;
            int64_t c_v;
            if (chain_of_lane(ca.n_vseq, a.n_legs, a.lanes_per_wave, a.leg_order, c_v, leg)) vc = c_v;
        } else if (mode == CHUNK_REPAIR) {
            if (cursor >= n_items) break;
            vc = ca.worklist[cursor];
            leg = (int)(vc % a.n_legs);
            cursor += n_waves * W;
        } else {
            const int64_t seq = wave / a.n_legs;
            leg = (int)(wave - seq * a.n_legs);
            bool found = false;
            while (cursor < K) {
                const int64_t kk = cursor + lane;
                const bool inc = kk < K && chunk_inconsistent(a, ca, wave, kk);
                const unsigned long long m = __ballot(inc);
                if (m) {
                    // every lane re-solves the first inconsistent chunk (replicas: same loads, same stores) -- a
                    // wavefront with one active lane would run in the slow sparse-EXEC mode (chain_of_wave_lane)
                    const int first = __ffsll((long long)m) - 1;
                    vc = (seq * K + cursor + first) * a.n_legs + leg;
                    cursor += first + 1;
                    found = true;
                    break;
                }
                cursor += 64;
            }
            if (!found) break;  // wave-uniform
        }
        if (vc >= 0) {
            seqik::ChainIO io;
            chunk_io(a, ca, vc, leg, mode == CHUNK_SPEC, io, mode == CHUNK_SWEEP ? CHUNK_FLAG_SWEPT : CHUNK_FLAG_REPAIRED);
            if (mode == CHUNK_SWEEP && ca.stats && lane == 0) atomicAdd(&ca.stats[6], 1);
            const seqik::LegConst &lc = s_legs[leg];
            seqik::run_stage<1, false, false, false, true, true>(lc, io);
            seqik::run_stage<2, WANT_FK, false, false, true, true>(lc, io);
            seqik::run_stage<3, WANT_FK, false, false, true, true>(lc, io);
            seqik::run_stage<4, WANT_FK, false, false, false, true>(lc, io);
        }
        if (mode == CHUNK_SWEEP) __threadfence();  // the next verification reads the frames just stored
    }
}

// The speculative pass of a chunked call on the stage pipeline (seqik_pipe_kernel): a workgroup of four waves per group
// of chunks.  In a workgroup the stage-1 wave of a chunk runs at most PIPE_DEPTH frames ahead of its stage-2 wave; all
// four store into the chunk's rows / start_state exactly what the lane-per-chunk kernel stores.
template <bool WANT_FK, int mode, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
seqik_chunk_pipe_kernel(KernelArgs a, ChunkArgs ca)
{
    static_assert(mode == CHUNK_SPEC || mode == CHUNK_REPAIR || mode == CHUNK_SERIAL, "the sweep stays on the lane-per-chunk kernel");
    __shared__ seqik::LegConst s_legs[kMaxLegs];
    __shared__ PipeShared sh;
    {
        const int words = a.n_legs * (int)(sizeof(seqik::LegConst) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < 3 * 64; i += blockDim.x) { (&sh.produced[0][0])[i] = 0; (&sh.consumed[0][0])[i] = 0; }
    }
    __syncthreads();
    const int stage_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (mode == CHUNK_SPEC) {
        int64_t vc;
        int leg;
        if (!chain_of_wave_lane(blockIdx.x, lane, ca.n_vseq, a.n_legs, a.lanes_per_wave, a.leg_order, vc, leg)) return;
        seqik::ChainIO io;
        chunk_io(a, ca, vc, leg, true, io);
        // the four waves of a chunk each record their own joints of the run-in's last frame (disjoint entries)
        pipe_run<WANT_FK, true, WPE == 2>(s_legs[leg], io, sh, stage_wave, lane, use_pairs(a), a.fault);
    } else {
        // the work list of round ca.round (SERIAL: the serial list), spread over the workgroups as thinly as possible; the
        // four stage waves of a workgroup walk the same entries in the same order (the ring counters of a lane keep
        // counting across entries)
        const int64_t n_items = (mode == CHUNK_SERIAL) ? ca.ctrl[kCtrlSerial].count : ca.ctrl[ca.round].count, n_groups = gridDim.x;
        int64_t W = (n_items + n_groups - 1) / n_groups;
        W = W < 1 ? 1 : (W > 64 ? 64 : W);
        const int cl = lane / lane_replication((int)W);
        int base = 0;
        for (int64_t cursor = (cl < W) ? (int64_t)blockIdx.x * W + cl : n_items; cursor < n_items; cursor += n_groups * W) {
            seqik::ChainIO io;
            int leg;
            if (mode == CHUNK_SERIAL) {
                const int64_t c = ca.serial_list[cursor];
                leg = (int)(c % a.n_legs);
                serial_io(a, c, io);
                if (ca.flags && stage_wave == 0)   // (replicas write the same bytes)
                    for (int64_t kk = lane; kk < ca.n_chunks; kk += 64) ca.flags[c * ca.n_chunks + kk] |= CHUNK_FLAG_SERIAL;
            } else {
                const int64_t vc = ca.worklist[cursor];
                leg = (int)(vc % a.n_legs);
                chunk_io(a, ca, vc, leg, false, io, CHUNK_FLAG_REPAIRED);
            }
            pipe_run<WANT_FK, true, WPE == 2>(s_legs[leg], io, sh, stage_wave, lane, a.lane_pairs != 0 && lane_pairs((int)W), a.fault, base);
            base += (int)(io.n_frames - io.t_begin);
        }
    }
}

// zeroes the control block / statistics of a chunked call (first thing on the stream)
__global__ void seqik_chunk_reset_kernel(ChunkArgs ca, int32_t n_chunks_total)
{
    const int i = threadIdx.x;
    if (i <= kCtrlSerial) { ca.ctrl[i].count = 0; ca.ctrl[i].pending = 0; }
    if (ca.stats && i < 16) ca.stats[i] = (i == 0) ? n_chunks_total : (i == 1) ? ca.chunk : (i == 2) ? ca.halo : 0;
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

using GenericLegTable = seqik::GenericLeg;   // { GenericConst gc; LegAffine aff; } (seqik_generic.hpp)

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
    int32_t lane_groups;  // thin waves split a pass over groups of 8 lanes (0 = off, for measurements)
    int64_t pose_chain, pose_row, pose_frame;
    int64_t ang_chain, ang_dof, ang_frame;
};

// Generic (single 9-link chain, 7 unknowns) IK: one lane per chain, one launch.
#ifndef SEQIK_GENERIC_WAVES_PER_EU
#define SEQIK_GENERIC_WAVES_PER_EU 1
#endif
// GROUPED: the thin-wave instantiation (every chain on a group of 8 lanes, seqik_generic.hpp "Lane groups") is a kernel of
// its own, so that its registers are allocated for it alone: compiled together with the one-lane code it inherited that
// code's demand (256 VGPRs + 92 AGPRs) and spent 7 % of a pass on v_accvgpr copies.
template <bool WANT_DIAG, bool GROUPED>
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
    seqik::run_generic<WANT_DIAG, GROUPED>(s_legs[leg].gc, s_legs[leg].aff, io);
}

// BATCHES of generic chains with more chains than the GPU has lanes: persistent wavefronts, one lane per chain, a lane
// that has finished its chain takes the next sequence of its leg from a per-leg counter (seqik_generic.hpp GenericQueue).
// The lanes of wavefront w start on leg  order[w mod n_legs]; a LANE that finds its leg's counter exhausted moves on to the
// next leg by itself (its constants come from LDS by a per-lane address), so no lane waits for its wavefront at a leg
// boundary and every wavefront helps to finish every leg.  The grid is at most one wavefront per SIMD (the kernel's
// register budget); a wavefront ends when all counters are exhausted.
template <bool WANT_DIAG>
__global__ void __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(SEQIK_GENERIC_WAVES_PER_EU, SEQIK_GENERIC_WAVES_PER_EU)))
seqik_generic_queue_kernel(GenericKernelArgs a, int32_t *counters)
{
    __shared__ GenericLegTable s_legs[kMaxLegs];
    {
        const int words = a.n_legs * (int)(sizeof(GenericLegTable) / sizeof(uint32_t));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.legs);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_legs);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    seqik::GenericIO io;
    io.pose = nullptr; io.pose_row = a.pose_row; io.pose_frame = a.pose_frame;
    io.angles = nullptr; io.ang_dof = a.ang_dof; io.ang_frame = a.ang_frame;
    io.fk = nullptr; io.status = nullptr; io.nfev = nullptr; io.init = nullptr;
    io.n_frames = a.n_frames;
    seqik::GenericQueue q;
    q.counters = counters; q.n_seq = a.n_seq; q.n_legs = a.n_legs; q.first = (int32_t)(wave % a.n_legs);
    for (int l = 0; l < 8; ++l) q.order[l] = a.leg_order.leg[l];
    q.table = s_legs;
    q.pose = a.pose; q.pose_chain = a.pose_chain; q.angles = a.angles; q.ang_chain = a.ang_chain;
    q.fk = a.fk; q.status = a.status; q.nfev = a.nfev; q.init = a.init;
    seqik::run_generic<WANT_DIAG, false, true>(s_legs[0].gc, s_legs[0].aff, io, &q);
}

// Device copies of the per-leg constant tables.  Callers almost always pass the same legs on every call, so the
// tables are cached: content-addressed and immutable (a table is never overwritten while kernels that read it may be
// in flight, so a change of legs needs no device drain), per device, shared by all host threads.  A repeat call is
// a pure kernel launch (no allocation, no copy).  seqik_release_workspaces() frees them.
template <typename T>
struct TableCache {
    struct Entry {
        int device;
        T *d;
        std::vector<T> h;
        uint64_t last_use;
    };
    static constexpr size_t kMaxEntries = 64;
    std::mutex mutex;
    std::vector<Entry> entries;
    uint64_t clock = 0;

    int get(const std::vector<T> &h, const T **out)
    {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mutex);
        for (Entry &e : entries)
            if (e.device == dev && e.h.size() == h.size() && memcmp(e.h.data(), h.data(), sizeof(T) * h.size()) == 0) {
                e.last_use = ++clock;
                *out = e.d;
                return SEQIK_OK;
            }
        if (entries.size() >= kMaxEntries) {  // evict the least recently used one (kernels in flight may read it)
            size_t lru = 0;
            for (size_t i = 1; i < entries.size(); ++i)
                if (entries[i].last_use < entries[lru].last_use) lru = i;
            {
                seqik::DeviceScope evict;  // restores the caller's device on every path out of this block
                HIP_TRY(evict.enter(entries[lru].device));
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipFree(entries[lru].d));
            }
            entries.erase(entries.begin() + lru);
        }
        Entry e;
        e.device = dev;
        e.d = nullptr;
        e.h = h;
        e.last_use = ++clock;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e.d), sizeof(T) * kMaxLegs));
        hipError_t err = hipMemcpy(e.d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            (void)hipFree(e.d);
            return fail(SEQIK_ERR_HIP, "hipMemcpy(leg table): %s", hipGetErrorString(err));
        }
        entries.push_back(e);
        *out = e.d;
        return SEQIK_OK;
    }

    int release()
    {
        std::lock_guard<std::mutex> lock(mutex);
        int prev = -1;
        HIP_TRY(hipGetDevice(&prev));
        for (Entry &e : entries) {
            HIP_TRY(hipSetDevice(e.device));
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(e.d));
        }
        entries.clear();
        HIP_TRY(hipSetDevice(prev));
        return SEQIK_OK;
    }
};
TableCache<seqik::LegConst> g_leg_tables;
TableCache<GenericLegTable> g_generic_tables;

int device_leg_table(const SeqikLegParams *legs, const SeqikAffine *affine, int32_t n_legs, const seqik::LegConst **out)
{
    std::vector<seqik::LegConst> h(n_legs);
    memset(h.data(), 0, sizeof(seqik::LegConst) * n_legs);
    for (int l = 0; l < n_legs; ++l) seqik::make_leg_consts(legs[l], affine ? affine + l : nullptr, h[l]);
    return g_leg_tables.get(h, out);
}

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
    return g_generic_tables.get(h, out);
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
// (Round 5: 16 -> 64.  A caller that keeps more streams in flight than workspaces are remembered pays an eviction -- a
// device-wide synchronisation, a free and an allocation -- on EVERY launch: 17 streams of 11 718-chain calls ran at 14.4 ms
// per call, the time of one call alone, against 2.0 ms on 16 streams; profiles/r05_stream_cliff.jsonl.)
constexpr int kMaxWorkspaces = 64;
std::mutex g_ws_mutex;
std::mutex g_queue_enqueue_mutex;   // chain queue: {zero the counters, launch} is one unit per stream (see the launch)
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

// Context of a host-buffer call (seqik_solve_seq, seqik_solve_generic): a stream and one grow-only device arena
// that the call's device buffers are carved from.  Contexts are pooled per device: a call takes a free one (or makes
// one) and gives it back, so repeated calls neither create streams nor call hipMalloc / hipFree (which drains the
// device), and -- because the stream lives on -- the hand-off workspace keyed by it (workspace_for) is reused instead
// of stranded.  Concurrent host threads get distinct contexts; seqik_release_workspaces() frees the idle ones.
struct HostCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    char *arena = nullptr;
    size_t arena_bytes = 0;
    bool busy = false;
};
std::mutex g_ctx_mutex;
std::vector<HostCtx *> g_ctx;

int acquire_ctx(HostCtx **out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    for (HostCtx *c : g_ctx)
        if (c->device == dev && !c->busy) { c->busy = true; *out = c; return SEQIK_OK; }
    HostCtx *c = new HostCtx;
    c->device = dev;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(SEQIK_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->busy = true;
    g_ctx.push_back(c);
    *out = c;
    return SEQIK_OK;
}

void release_ctx(HostCtx *c)
{
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    c->busy = false;
}

int ctx_reserve(HostCtx *c, size_t bytes)
{
    if (c->arena_bytes >= bytes) return SEQIK_OK;
    if (c->arena) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->arena));
        c->arena = nullptr;
        c->arena_bytes = 0;
    }
    const size_t want = bytes + bytes / 8;  // a little head room: slightly longer recordings do not reallocate
    if (hipMalloc(reinterpret_cast<void **>(&c->arena), want) == hipSuccess) { c->arena_bytes = want; return SEQIK_OK; }
    (void)hipGetLastError();
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->arena), bytes));
    c->arena_bytes = bytes;
    return SEQIK_OK;
}

// bump allocator over the arena (256-byte aligned pieces)
struct ArenaCursor {
    char *base;
    size_t off = 0;
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    template <typename T> T *take(size_t count) { T *p = reinterpret_cast<T *>(base + off); off += padded(sizeof(T) * count); return p; }
};

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
    a.fault = fault_word(stream);
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
    const bool diag = d_status || d_nfev;
    const bool fk = d_fk && last_stage == 4;  // FK is the stage-4 chain's (leg_inverse_kinematics.py:279-282)
    if (!fk) a.fk = nullptr;
    // frame chunks (SeqikOptions.frame_chunk): runs of all four stages without diagnostics only
    int32_t chunk = 0, halo = 0, lead = 0;
    int64_t n_chunks = 1;
    const bool chunked = first_stage == 1 && last_stage == 4 && !diag &&
                         pick_frame_chunks(opt, n_frames, chunk, halo, lead, n_chunks);
    if (!chunked && opt && (opt->frame_lead > 0 || opt->chunk_resume))
        return fail(SEQIK_ERR_BAD_ARG, "frame_lead / chunk_resume need frame chunks (frame_chunk != 0, all four stages, no diagnostics)%s");
    const int64_t n_vchains = a.n_chains * n_chunks;  // virtual chains = chunks (= chains when not chunked)
    if (n_vchains > 0x7fffffffLL) return fail(SEQIK_ERR_BAD_ARG, "too many frame chunks for one launch%s");
    // stage pipeline (four waves per group of chains, seqik_pipe_kernel): SeqikOptions.reserved[3] = 0 automatic (calls
    // with at most kPipeMaxChains chains / chunks: too few for the lane-per-chain kernels to fill the GPU, so the
    // serial path per frame is what counts), 1 = never, 2 = whenever applicable, 3 = as 2 without lane pairs, 4 / 5 = as
    // 2 / 3 but never the 256-register LATENCY build (run_stage<..., LAT>): the A/B switch of the tests that pin LAT's
    // register-carried Jacobian / gradient / scaling reuse against the plain instantiation, bit for bit
    const int pipe_opt = opt ? opt->reserved[3] : 0;
    const bool staged = opt && opt->reserved[1] == 1;  // "always one launch per stage" rules out the AUTOMATIC pipeline
    const bool piped = first_stage == 1 && last_stage == 4 && !diag &&
                       (pipe_opt >= 2 || (pipe_opt == 0 && !(staged && !chunked) &&
                                          n_vchains <= (chunked ? kPipeMaxChunks : kPipeMaxChains)));
    a.lane_pairs = (pipe_opt == 3 || pipe_opt == 5) ? 0 : 1;  // 3 / 5 = thin waves without lane pairs (measurements)
    a.lanes_per_wave = pick_lanes_per_wave(n_vchains, opt, chunked || piped);
    int64_t n_waves = ((n_seq * n_chunks + a.lanes_per_wave - 1) / a.lanes_per_wave) * n_legs;  // leg-pure waves
    if (opt && opt->reserved[2] == 1) {  // leg-interleaved: |W| consecutive chains per wave
        n_waves = (n_vchains + a.lanes_per_wave - 1) / a.lanes_per_wave;
        a.lanes_per_wave = -a.lanes_per_wave;
    }
    // chain queue of the fused kernel: SeqikOptions.reserved[0] in 65..4096 = chains a wavefront OWNS (a multiple of 64; its 64
    // lanes take them one after the other).  Only where seqik_fused_kernel would run: all four stages, no diagnostics, not
    // chunked, not piped, leg-pure wavefronts, at least 1024 chains; anywhere else the option is refused rather than ignored.
    a.pool = 0;
    if (opt && opt->reserved[0] > 64) {
        const bool plain_fused = first_stage == 1 && last_stage == 4 && !diag && !chunked && !piped && !staged && opt->reserved[2] != 1;
        const int64_t widest = n_legs * (a.pose_chain > a.ang_chain ? a.pose_chain : a.ang_chain) > (int64_t)n_legs * n_frames * 27
                                   ? n_legs * (a.pose_chain > a.ang_chain ? a.pose_chain : a.ang_chain) : (int64_t)n_legs * n_frames * 27;
        if (opt->reserved[0] > 4096 || opt->reserved[0] % 64 != 0 || !plain_fused || n_seq > 0x7fffffffLL || widest > 0xffffffffLL)
            return fail(SEQIK_ERR_BAD_ARG, "reserved[0] > 64 (chain queue: chains per wavefront) must be a multiple of 64 up to 4096 and "
                                           "needs the single-launch lane-per-chain path (all four stages, no diagnostics, no frame "
                                           "chunks, reserved[1] != 1, reserved[2] != 1, reserved[3] = 1 or more than 40 000 chains)%s");
        a.pool = opt->reserved[0];
        n_waves = ((n_seq + a.pool - 1) / a.pool) * n_legs;
        block = 64;
    }
    static const bool queue64 = getenv("SEQIK_QUEUE64") != nullptr;   // measurement: the queue kernel with pools of 64 (no pulls)
    if (queue64 && a.pool == 0 && a.lanes_per_wave == 64 && first_stage == 1 && last_stage == 4 && !diag && !chunked && !piped && !staged &&
        !(opt && opt->reserved[2] == 1)) {
        a.pool = 64;
        block = 64;
    }
    int64_t grid64 = (n_waves * 64 + block - 1) / block;
    if (grid64 > 0x7fffffffLL || n_waves > 0x7fffffffLL) return fail(SEQIK_ERR_BAD_ARG, "too many chains for one launch%s");
    const dim3 grid((unsigned)grid64), blk(block);
    const dim3 pipe_grid((unsigned)n_waves), pipe_blk(256);  // one workgroup (4 stage waves) per group of W chains
    // at most two workgroups per CU: the 256-register (latency) build of the pipeline kernels, unless switched off (4 / 5)
    const bool roomy = n_waves <= 2 * 256 && pipe_opt != 4 && pipe_opt != 5;
    // stage hand-off workspace: the frame after the active links of stage k is the prefix of stage k + 1
    a.frames = nullptr;
    static const bool pool_workspace = getenv("SEQIK_WORKSPACE_POOL") != nullptr;  // diagnosis only (see Workspace)
    const size_t ws_frames = chunked ? (size_t)chunk + (size_t)(halo > lead ? halo : lead) : 0;  // hand-off frames per chunk
    if (chunked) {
        const size_t ws_bytes = sizeof(double) * (12 * ws_frames + 7) * n_vchains + 128 +
                                sizeof(int32_t) * ((size_t)n_vchains + 3 * (size_t)a.n_chains);
        if (int rc = workspace_for(stream, ws_bytes, &a.frames)) return rc;
    } else if (last_stage > first_stage && !piped) {
        const size_t ws_bytes = sizeof(double) * 12 * a.n_chains * n_frames;
        if (pool_workspace) HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&a.frames), ws_bytes, stream));
        else if (int rc = workspace_for(stream, ws_bytes, &a.frames)) return rc;
    }
    const bool fused = (piped || !staged) && first_stage == 1 && last_stage == 4 && !diag;
    if (chunked) {
        ChunkArgs ca;
        ca.n_chunks = n_chunks; ca.n_vseq = n_seq * n_chunks; ca.chunk = chunk; ca.halo = halo; ca.lead = lead;
        ca.tol = (opt->chunk_tol > 0) ? opt->chunk_tol : (opt->chunk_tol < 0 ? 0.0 : 1e-6);
        ca.n_rounds = (opt->chunk_rounds > 0) ? (opt->chunk_rounds < kMaxChunkRounds ? opt->chunk_rounds : kMaxChunkRounds) : 3;
        ca.stats = opt->chunk_stats;
        ca.flags = opt->chunk_flags;
        ca.round = 0;
        // chunk_resume: 0 = a whole call; 1 / 2 = resume (2: chunk 0 must continue d_init exactly); LOCKSTEP pieces of ONE call spread
        // over the GPUs of a frame-sharded recording, so that the ranks together run exactly the rounds one GPU would run:
        // 3 = the speculative pass and the first verification only; 4 = ONE {scan, repair} round (chunk 0 against d_init, held back
        // where CHUNK_FLAG_LEFT_BLOCKED says the chunk in front of it is about to change); 5 = the final scan + serial sweep only
        const int mode = opt->chunk_resume;
        if (mode < 0 || mode > 5) return fail(SEQIK_ERR_BAD_ARG, "chunk_resume must be 0 .. 5%s");
        const bool resume = mode == 1 || mode == 2 || mode == 4 || mode == 5;
        ca.resume = resume ? (mode == 2 ? 2 : 1) : 0;
        if ((mode >= 3) && opt->frame_chunk <= 0)
            return fail(SEQIK_ERR_BAD_ARG, "chunk_resume 3 / 4 / 5 (lockstep pieces) need an explicit frame_chunk > 0%s");
        if (mode == 3 && !opt->chunk_states)
            return fail(SEQIK_ERR_BAD_ARG, "chunk_resume = 3 needs chunk_states (the rounds that follow are other calls)%s");
        if (mode == 4) ca.n_rounds = 1;          // one round, no sweep (the loop below stops in front of it)
        if (mode == 5) ca.n_rounds = 0;          // straight to the final scan + sweep
        ca.left_blocked = (mode == 4 && opt->chunk_flags) ? 1 : 0;
        if (resume && !opt->chunk_states)
            return fail(SEQIK_ERR_BAD_ARG, "chunk_resume needs the chunk_states of the call it resumes%s");
        // chunk 0 is verified (and repaired) like the others when it started from a run-in and the caller says what the
        // true state in front of it is
        ca.k_first = ((lead > 0 || resume) && d_init) ? 0 : 1;
        // the guard against failed speculation belongs to the automatic mode of a whole recording
        ca.guard = (opt->frame_chunk == -1 && lead == 0 && !resume) ? 1 : 0;
        // carve the workspace: [hand-off frames] | start states (unless the caller keeps them) | control block |
        // work list | serial flags | serial list.  (The stage pipeline hands frames over through LDS, but the sweep at the
        // end of a piped call runs on the lane-per-chunk kernel and uses the hand-off frames.)
        char *base = reinterpret_cast<char *>(a.frames);
        size_t off = sizeof(double) * 12 * (size_t)n_vchains * ws_frames;
        ca.start_state = reinterpret_cast<double *>(base + off); off += sizeof(double) * 7 * (size_t)n_vchains;
        if (opt->chunk_states) ca.start_state = opt->chunk_states;
        ca.ctrl = reinterpret_cast<ChunkCtrl *>(base + off); off += 128;
        ca.worklist = reinterpret_cast<int32_t *>(base + off); off += sizeof(int32_t) * (size_t)n_vchains;
        ca.chain_serial = reinterpret_cast<int32_t *>(base + off); off += sizeof(int32_t) * (size_t)a.n_chains;
        ca.fail_count = reinterpret_cast<int32_t *>(base + off); off += sizeof(int32_t) * (size_t)a.n_chains;  // (zeroed with chain_serial)
        ca.serial_list = reinterpret_cast<int32_t *>(base + off);
        if (opt->stage_events) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[0]), stream));
        hipLaunchKernelGGL(seqik_chunk_reset_kernel, dim3(1), dim3(64), 0, stream, ca, (int32_t)(n_chunks * a.n_chains));
        HIP_TRY(hipMemsetAsync(ca.chain_serial, 0, sizeof(int32_t) * 2 * (size_t)a.n_chains, stream));
        if (resume) {
            // nothing is solved speculatively: angles / chunk_states are a previous call's, d_init the true state
        } else if (piped) {
            if (roomy) {
                if (fk) hipLaunchKernelGGL((seqik_chunk_pipe_kernel<true, CHUNK_SPEC, 2>), pipe_grid, pipe_blk, 0, stream, a, ca);
                else hipLaunchKernelGGL((seqik_chunk_pipe_kernel<false, CHUNK_SPEC, 2>), pipe_grid, pipe_blk, 0, stream, a, ca);
            } else if (fk) hipLaunchKernelGGL((seqik_chunk_pipe_kernel<true, CHUNK_SPEC, SEQIK_WAVES_PER_EU>), pipe_grid, pipe_blk, 0, stream, a, ca);
            else hipLaunchKernelGGL((seqik_chunk_pipe_kernel<false, CHUNK_SPEC, SEQIK_WAVES_PER_EU>), pipe_grid, pipe_blk, 0, stream, a, ca);
        } else if (fk) hipLaunchKernelGGL((seqik_chunk_kernel<true, CHUNK_SPEC>), grid, blk, 0, stream, a, ca);
        else hipLaunchKernelGGL((seqik_chunk_kernel<false, CHUNK_SPEC>), grid, blk, 0, stream, a, ca);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(seqik_chunk_verify_kernel, dim3((unsigned)((n_vchains + 255) / 256)), dim3(256), 0, stream, a, ca);
        hipLaunchKernelGGL(seqik_chunk_decide_kernel, dim3((unsigned)((a.n_chains + 255) / 256)), dim3(256), 0, stream, a, ca);
        const dim3 scan_grid((unsigned)((n_vchains + 255) / 256)), scan_blk(256);
        const int64_t rep_waves = n_waves < 4096 ? n_waves : 4096;  // the work list is walked grid-stride
        const dim3 rep_grid((unsigned)((rep_waves * 64 + block - 1) / block));
        for (int r = 0; r <= ca.n_rounds && mode != 3; ++r) {
            if (mode == 4 && r == ca.n_rounds) break;   // lockstep round: the sweep is another call (chunk_resume = 5)
            ca.round = r;
            hipLaunchKernelGGL(seqik_chunk_scan_kernel, scan_grid, scan_blk, 0, stream, a, ca);
            if (r < ca.n_rounds && piped) {
                const dim3 rep_pipe_grid((unsigned)(n_waves < 1024 ? n_waves : 1024));
                if (roomy) {
                    if (fk) hipLaunchKernelGGL((seqik_chunk_pipe_kernel<true, CHUNK_REPAIR, 2>), rep_pipe_grid, pipe_blk, 0, stream, a, ca);
                    else hipLaunchKernelGGL((seqik_chunk_pipe_kernel<false, CHUNK_REPAIR, 2>), rep_pipe_grid, pipe_blk, 0, stream, a, ca);
                } else if (fk) hipLaunchKernelGGL((seqik_chunk_pipe_kernel<true, CHUNK_REPAIR, SEQIK_WAVES_PER_EU>), rep_pipe_grid, pipe_blk, 0, stream, a, ca);
                else hipLaunchKernelGGL((seqik_chunk_pipe_kernel<false, CHUNK_REPAIR, SEQIK_WAVES_PER_EU>), rep_pipe_grid, pipe_blk, 0, stream, a, ca);
            } else if (r < ca.n_rounds) {
                if (fk) hipLaunchKernelGGL((seqik_chunk_kernel<true, CHUNK_REPAIR>), rep_grid, blk, 0, stream, a, ca);
                else hipLaunchKernelGGL((seqik_chunk_kernel<false, CHUNK_REPAIR>), rep_grid, blk, 0, stream, a, ca);
            } else {  // serial sweep: one wave per real chain
                const dim3 sweep_grid((unsigned)a.n_chains), sweep_blk(64);
                if (fk) hipLaunchKernelGGL((seqik_chunk_kernel<true, CHUNK_SWEEP>), sweep_grid, sweep_blk, 0, stream, a, ca);
                else hipLaunchKernelGGL((seqik_chunk_kernel<false, CHUNK_SWEEP>), sweep_grid, sweep_blk, 0, stream, a, ca);
            }
            HIP_TRY(hipGetLastError());
        }
        if (ca.guard) {  // the chains the first verification gave up on: the serial walk, on the stage pipeline
            const dim3 ser_grid((unsigned)(a.n_chains < 1024 ? a.n_chains : 1024));
            if (a.n_chains <= 2 * 256) {
                if (fk) hipLaunchKernelGGL((seqik_chunk_pipe_kernel<true, CHUNK_SERIAL, 2>), ser_grid, pipe_blk, 0, stream, a, ca);
                else hipLaunchKernelGGL((seqik_chunk_pipe_kernel<false, CHUNK_SERIAL, 2>), ser_grid, pipe_blk, 0, stream, a, ca);
            } else if (fk) hipLaunchKernelGGL((seqik_chunk_pipe_kernel<true, CHUNK_SERIAL, SEQIK_WAVES_PER_EU>), ser_grid, pipe_blk, 0, stream, a, ca);
            else hipLaunchKernelGGL((seqik_chunk_pipe_kernel<false, CHUNK_SERIAL, SEQIK_WAVES_PER_EU>), ser_grid, pipe_blk, 0, stream, a, ca);
            HIP_TRY(hipGetLastError());
        }
        if (opt->stage_events)
            for (int k = 1; k <= 4; ++k) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[k]), stream));
        return SEQIK_OK;
    }
    if (fused) {
        if (opt && opt->stage_events) HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opt->stage_events[0]), stream));
        if (piped) {
            if (roomy) {
                if (fk) hipLaunchKernelGGL((seqik_pipe_kernel<true, 2>), pipe_grid, pipe_blk, 0, stream, a);
                else hipLaunchKernelGGL((seqik_pipe_kernel<false, 2>), pipe_grid, pipe_blk, 0, stream, a);
            } else if (fk) hipLaunchKernelGGL((seqik_pipe_kernel<true, SEQIK_WAVES_PER_EU>), pipe_grid, pipe_blk, 0, stream, a);
            else hipLaunchKernelGGL((seqik_pipe_kernel<false, SEQIK_WAVES_PER_EU>), pipe_grid, pipe_blk, 0, stream, a);
        } else if (a.pool > 0) {
            if (fk) hipLaunchKernelGGL((seqik_fused_queue_kernel<true>), grid, blk, 0, stream, a);
            else hipLaunchKernelGGL((seqik_fused_queue_kernel<false>), grid, blk, 0, stream, a);
        } else if (fk) hipLaunchKernelGGL((seqik_fused_kernel<true>), grid, blk, 0, stream, a);
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

// the pooled context for the other translation units (seqik_hostctx.hpp)
namespace seqik {

int host_lease_acquire(HostLease *lease)
{
    HostCtx *c = nullptr;
    const int rc = acquire_ctx(&c);
    if (rc != SEQIK_OK) return rc;
    lease->ctx = c;
    lease->stream = c->stream;
    lease->arena = c->arena;
    return SEQIK_OK;
}

int host_lease_reserve(HostLease *lease, size_t bytes)
{
    HostCtx *c = static_cast<HostCtx *>(lease->ctx);
    const int rc = ctx_reserve(c, bytes);
    lease->arena = c->arena;
    return rc;
}

void host_lease_release(HostLease *lease)
{
    if (lease->ctx) {
        // also on a failing path: work queued so far may still use the arena / the caller's buffers
        (void)hipStreamSynchronize(static_cast<HostCtx *>(lease->ctx)->stream);
        release_ctx(static_cast<HostCtx *>(lease->ctx));
    }
    lease->ctx = nullptr;
}

}  // namespace seqik

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
    bool any_busy = false;
    {
        std::lock_guard<std::mutex> ctx_lock(g_ctx_mutex);
        for (size_t i = 0; i < g_ctx.size();) {
            HostCtx *c = g_ctx[i];
            if (c->busy) { any_busy = true; ++i; continue; }  // a call is running on another thread
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->arena) HIP_TRY(hipFree(c->arena));
            HIP_TRY(hipStreamDestroy(c->stream));
            delete c;
            g_ctx.erase(g_ctx.begin() + i);
        }
    }
    HIP_TRY(hipSetDevice(prev));
    // the cached leg tables stay while a host-buffer call is running on another thread: it may have fetched its table
    // pointer already and not launched yet (they are freed by the next release that finds every context idle)
    if (any_busy) return SEQIK_OK;
    if (int rc = g_leg_tables.release()) return rc;
    if (int rc = g_generic_tables.release()) return rc;
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

int seqik_check_faults(void) { return check_faults("seqik_check_faults", -1); }

int seqik_check_faults_stream(void *hip_stream)
{
    if (!g_fault_host.load(std::memory_order_acquire)) return SEQIK_OK;  // nothing has been launched yet
    return check_faults("seqik_check_faults_stream", known_slot(static_cast<hipStream_t>(hip_stream)));
}

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
        if (rc == SEQIK_ERR_BAD_ARG)
            return fail(rc, "a joint limit is non-zero but smaller than 2^-600 in magnitude: not supported (DESIGN.md, floating-point contract)%s");
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
    // asynchronous: a fault of THIS launch cannot be known yet; one an earlier launch left behind is reported now
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (g_fault_host.load(std::memory_order_acquire) &&
        (rc = check_faults("seqik_solve_seq_device (fault of an earlier launch on this stream)", known_slot(stream))) != SEQIK_OK)
        return rc;
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
        if (rc == SEQIK_ERR_BAD_ARG)
            return fail(rc, "a joint limit is non-zero but smaller than 2^-600 in magnitude: not supported (DESIGN.md, floating-point contract)%s");
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
    a.lane_groups = (opt && opt->reserved[3] == 3) ? 0 : 1;  // 3: thin waves without lane groups (measurements)
    int64_t n_waves = ((n_seq + a.lanes_per_wave - 1) / a.lanes_per_wave) * n_legs;
    if (opt && opt->reserved[2] == 1) {
        n_waves = (a.n_chains + a.lanes_per_wave - 1) / a.lanes_per_wave;
        a.lanes_per_wave = -a.lanes_per_wave;
    }
    const dim3 grid((unsigned)((n_waves * 64 + block - 1) / block)), blk(block);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int W_abs = a.lanes_per_wave < 0 ? -a.lanes_per_wave : a.lanes_per_wave;
    const bool grouped = a.lane_groups != 0 && W_abs <= 8;  // lane_groups(W): the replication is then a multiple of 8
    // Chain queue (SeqikOptions.reserved[1]: 0 = automatic, 1 = never, 2 = whenever full wavefronts are used): pays when the
    // batch has several times more chains than the GPU has lanes for this kernel (one wavefront per SIMD) -- with fewer
    // there is nothing to pull and the launch lasts as long as its slowest chain either way.  Measured on windows of the
    // shipped recording (bench.py generic_batches, static -> queue): 1 chain per lane 120.2 -> 120.8 ms, 2: 142.8 -> 156.8,
    // 4: 220.8 -> 191.4 (1.15 x), 8: 358.3 -> 261.1 (1.37 x); the bound from the oracle's pass counts at a constant pass time
    // says 1.00 / 1.21 / 1.38 / 1.57 (profiles/r05_generic_queue_bound.json) -- the static launch does better than that
    // model at 2 because its passes get faster as the GPU drains (7.5 us with 128 wavefronts resident, 10.5 with 1 024),
    // while the queue keeps every wavefront alive to the end.  Automatic: from 4 chains per lane on.
    const int queue_opt = opt ? opt->reserved[1] : 0;
    if (W_abs == 64 && a.lanes_per_wave > 0 && queue_opt != 1) {
        int dev = 0, cus = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int64_t slots = (int64_t)cus * 4 * SEQIK_GENERIC_WAVES_PER_EU;   // resident wavefronts of this kernel
        if (queue_opt == 2 || a.n_chains >= 4 * 64 * slots) {
            double *ws = nullptr;
            if (int rc2 = workspace_for(stream, 256, &ws)) return rc2;
            int32_t *counters = reinterpret_cast<int32_t *>(ws);
            // The per-leg counters live in the stream's workspace: zeroing them and launching the kernel that consumes them must
            // reach the stream as ONE unit.  Two host threads launching on the same stream (both passing stream 0, say) could
            // otherwise enqueue memset A, memset B, kernel A, kernel B -- kernel B would find every counter exhausted and retire
            // without writing a result, and the call would still return SEQIK_OK (round-5 advice).  Both are enqueues (~10 us).
            std::lock_guard<std::mutex> enqueue_lock(g_queue_enqueue_mutex);
            HIP_TRY(hipMemsetAsync(counters, 0, sizeof(int32_t) * kMaxLegs, stream));
            const int64_t q_waves = n_waves < slots ? n_waves : slots;
            const dim3 q_grid((unsigned)q_waves), q_blk(64);
            if (d_status || d_nfev) hipLaunchKernelGGL((seqik_generic_queue_kernel<true>), q_grid, q_blk, 0, stream, a, counters);
            else hipLaunchKernelGGL((seqik_generic_queue_kernel<false>), q_grid, q_blk, 0, stream, a, counters);
            HIP_TRY(hipGetLastError());
            return SEQIK_OK;
        }
    }
    if (d_status || d_nfev) {
        if (grouped) hipLaunchKernelGGL((seqik_generic_kernel<true, true>), grid, blk, 0, stream, a);
        else hipLaunchKernelGGL((seqik_generic_kernel<true, false>), grid, blk, 0, stream, a);
    } else if (grouped) hipLaunchKernelGGL((seqik_generic_kernel<false, true>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((seqik_generic_kernel<false, false>), grid, blk, 0, stream, a);
    HIP_TRY(hipGetLastError());
    return SEQIK_OK;
}

// device buffers of one host-buffer call, carved from a pooled context's arena
struct HostCall {
    HostCtx *ctx = nullptr;
    hipStream_t stream = nullptr;
    // (on every exit path, also the failing ones: copies and kernels queued so far may still be writing into the arena
    // or the caller's buffers; the context goes back to the pool only once its stream is idle)
    ~HostCall()
    {
        if (!ctx) return;
        if (stream) (void)hipStreamSynchronize(stream);
        release_ctx(ctx);
    }
};

#define TRY_OUT(expr)                                                                            \
    {                                                                                            \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(SEQIK_ERR_HIP, #expr ": %s", hipGetErrorString(e_));  \
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
    const size_t n_lf = (size_t)n_seq * n_legs * n_frames;
    if (n_lf == 0) return SEQIK_OK;
    seqik::DeviceScope scope;
    HIP_TRY(scope.enter(opt ? opt->device : -1));
    HostCall call;
    if ((rc = acquire_ctx(&call.ctx)) != SEQIK_OK) return rc;
    const size_t n_ch = (size_t)n_seq * n_legs;
    const size_t need = ArenaCursor::padded(sizeof(double) * 15 * n_lf) + ArenaCursor::padded(sizeof(double) * 7 * n_lf) +
                        (fk ? ArenaCursor::padded(sizeof(double) * 27 * n_lf) : 0) +
                        (status ? ArenaCursor::padded(sizeof(int32_t) * n_lf) : 0) +
                        (nfev ? ArenaCursor::padded(sizeof(int32_t) * n_lf) : 0) +
                        (init_angles ? ArenaCursor::padded(sizeof(double) * 7 * n_ch) : 0);
    if ((rc = ctx_reserve(call.ctx, need)) != SEQIK_OK) return rc;
    hipStream_t stream = call.ctx->stream;
    call.stream = stream;
    ArenaCursor cur{call.ctx->arena};
    double *d_pose = cur.take<double>(15 * n_lf), *d_angles = cur.take<double>(7 * n_lf);
    double *d_fk = fk ? cur.take<double>(27 * n_lf) : nullptr;
    int32_t *d_status = status ? cur.take<int32_t>(n_lf) : nullptr, *d_nfev = nfev ? cur.take<int32_t>(n_lf) : nullptr;
    double *d_init = init_angles ? cur.take<double>(7 * n_ch) : nullptr;
    if (d_init) TRY_OUT(hipMemcpyAsync(d_init, init_angles, sizeof(double) * 7 * n_ch, hipMemcpyHostToDevice, stream));
    TRY_OUT(hipMemcpyAsync(d_pose, pose, sizeof(double) * 15 * n_lf, hipMemcpyHostToDevice, stream));
    rc = seqik_solve_generic_device(d_pose, n_seq, n_legs, n_frames, legs, d_angles, d_fk, d_status, d_nfev, d_init,
                                    nullptr, affine, opt, stream);
    if (rc != SEQIK_OK) { (void)hipStreamSynchronize(stream); return rc; }
    TRY_OUT(hipMemcpyAsync(angles, d_angles, sizeof(double) * 7 * n_lf, hipMemcpyDeviceToHost, stream));
    if (fk) TRY_OUT(hipMemcpyAsync(fk, d_fk, sizeof(double) * 27 * n_lf, hipMemcpyDeviceToHost, stream));
    if (status) TRY_OUT(hipMemcpyAsync(status, d_status, sizeof(int32_t) * n_lf, hipMemcpyDeviceToHost, stream));
    if (nfev) TRY_OUT(hipMemcpyAsync(nfev, d_nfev, sizeof(int32_t) * n_lf, hipMemcpyDeviceToHost, stream));
    TRY_OUT(hipStreamSynchronize(stream));
    return SEQIK_OK;
}

int seqik_solve_seq(const double *pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                    const SeqikLegParams *legs, int32_t first_stage, int32_t last_stage,
                    double *angles, double *fk, int32_t *status, int32_t *nfev, const double *init_angles,
                    const SeqikAffine *affine, const SeqikOptions *opt)
{
    int rc = check_args(n_seq, n_legs, n_frames, legs, first_stage, last_stage, pose, angles);
    if (rc != SEQIK_OK) return rc;
    const size_t n_lf = (size_t)n_seq * n_legs * n_frames;  // leg-frames
    if (n_lf == 0) return SEQIK_OK;
    seqik::DeviceScope scope;
    HIP_TRY(scope.enter(opt ? opt->device : -1));
    HostCall call;
    if ((rc = acquire_ctx(&call.ctx)) != SEQIK_OK) return rc;
    const bool want_fk = fk && last_stage == 4;
    // frame chunks: statistics and the per-chunk report are produced on the device and copied back
    int32_t pl_chunk = 0, pl_halo = 0, pl_lead = 0;
    int64_t pl_k = 0;
    const bool diag = status || nfev;
    const bool chunked = first_stage == 1 && last_stage == 4 && !diag && pick_frame_chunks(opt, n_frames, pl_chunk, pl_halo, pl_lead, pl_k);
    if (opt && (opt->chunk_states || opt->chunk_resume))
        return fail(SEQIK_ERR_BAD_ARG, "chunk_states / chunk_resume: device entry point only (seqik_solve_seq_device)%s");
    const bool want_stats = opt && opt->chunk_stats;
    const bool want_flags = opt && opt->chunk_flags && chunked;
    const size_t n_ch = (size_t)n_seq * n_legs;
    const size_t n_flags = want_flags ? n_ch * (size_t)pl_k : 0;
    const size_t need = ArenaCursor::padded(sizeof(double) * 15 * n_lf) + ArenaCursor::padded(sizeof(double) * 7 * n_lf) +
                        (want_fk ? ArenaCursor::padded(sizeof(double) * 27 * n_lf) : 0) +
                        (status ? ArenaCursor::padded(sizeof(int32_t) * 4 * n_lf) : 0) +
                        (nfev ? ArenaCursor::padded(sizeof(int32_t) * 4 * n_lf) : 0) +
                        (init_angles ? ArenaCursor::padded(sizeof(double) * 7 * n_ch) : 0) + 256 + ArenaCursor::padded(n_flags);
    if ((rc = ctx_reserve(call.ctx, need)) != SEQIK_OK) return rc;
    hipStream_t stream = call.ctx->stream;
    call.stream = stream;
    ArenaCursor cur{call.ctx->arena};
    double *d_pose = cur.take<double>(15 * n_lf), *d_angles = cur.take<double>(7 * n_lf);
    double *d_fk = want_fk ? cur.take<double>(27 * n_lf) : nullptr;
    int32_t *d_status = status ? cur.take<int32_t>(4 * n_lf) : nullptr, *d_nfev = nfev ? cur.take<int32_t>(4 * n_lf) : nullptr;
    double *d_init = init_angles ? cur.take<double>(7 * n_ch) : nullptr;
    int32_t *d_stats = cur.take<int32_t>(16);
    uint8_t *d_flags = want_flags ? cur.take<uint8_t>(n_flags) : nullptr;
    if (d_init) TRY_OUT(hipMemcpyAsync(d_init, init_angles, sizeof(double) * 7 * n_ch, hipMemcpyHostToDevice, stream));
    TRY_OUT(hipMemcpyAsync(d_pose, pose, sizeof(double) * 15 * n_lf, hipMemcpyHostToDevice, stream));
    // angles is in/out: columns of stages that do not run are inputs (earlier stages) or stay as they are
    if (first_stage > 1 || last_stage < 4)
        TRY_OUT(hipMemcpyAsync(d_angles, angles, sizeof(double) * 7 * n_lf, hipMemcpyHostToDevice, stream));
    if (d_status) TRY_OUT(hipMemsetAsync(d_status, 0xff, sizeof(int32_t) * 4 * n_lf, stream));
    if (d_nfev) TRY_OUT(hipMemsetAsync(d_nfev, 0, sizeof(int32_t) * 4 * n_lf, stream));
    SeqikOptions dev_opt;
    if (opt) dev_opt = *opt; else memset(&dev_opt, 0, sizeof(dev_opt));
    dev_opt.chunk_flags = d_flags;
    if (want_stats) {
        TRY_OUT(hipMemsetAsync(d_stats, 0, sizeof(int32_t) * 16, stream));  // stays zero when the call is not chunked
        dev_opt.chunk_stats = d_stats;
    }
    rc = seqik_solve_seq_device(d_pose, n_seq, n_legs, n_frames, legs, first_stage, last_stage, d_angles,
                                d_fk, d_status, d_nfev, d_init, nullptr, affine, opt ? &dev_opt : nullptr, stream);
    if (rc != SEQIK_OK) { (void)hipStreamSynchronize(stream); return rc; }
    if (want_stats) TRY_OUT(hipMemcpyAsync(opt->chunk_stats, d_stats, sizeof(int32_t) * 16, hipMemcpyDeviceToHost, stream));
    if (want_flags) TRY_OUT(hipMemcpyAsync(opt->chunk_flags, d_flags, n_flags, hipMemcpyDeviceToHost, stream));
    TRY_OUT(hipMemcpyAsync(angles, d_angles, sizeof(double) * 7 * n_lf, hipMemcpyDeviceToHost, stream));
    if (want_fk) TRY_OUT(hipMemcpyAsync(fk, d_fk, sizeof(double) * 27 * n_lf, hipMemcpyDeviceToHost, stream));
    if (status) TRY_OUT(hipMemcpyAsync(status, d_status, sizeof(int32_t) * 4 * n_lf, hipMemcpyDeviceToHost, stream));
    if (nfev) TRY_OUT(hipMemcpyAsync(nfev, d_nfev, sizeof(int32_t) * 4 * n_lf, hipMemcpyDeviceToHost, stream));
    TRY_OUT(hipStreamSynchronize(stream));
    return check_faults("seqik_solve_seq", known_slot(stream));
}

// Self-test hook of the floating-point contract: q[i] = div_(a[i], b[i]), r[i] = sqrt_(a[i]) on the device.
__global__ void seqik_selftest_div_sqrt_kernel(const double *a, const double *b, double *q, double *r, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { q[i] = seqik::div_(a[i], b[i]); r[i] = seqik::sqrt_(a[i]); }
}

int seqik_selftest_div_sqrt(const double *a, const double *b, double *q, double *r, int64_t n)
{
    if (!a || !b || !q || !r || n < 0) return fail(SEQIK_ERR_BAD_ARG, "null pointer argument%s");
    if (n == 0) return SEQIK_OK;
    double *d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * 4 * n));
    hipError_t e = hipMemcpy(d, a, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(seqik_selftest_div_sqrt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d, d + n, d + 2 * n, d + 3 * n, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(q, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(r, d + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SEQIK_ERR_HIP, "seqik_selftest_div_sqrt: %s", hipGetErrorString(e));
    return SEQIK_OK;
}

// r[i] = sqrt_pos_(a[i]): the select-free square root the Coleman-Li distances go through (run_stage, run_generic)
__global__ void seqik_selftest_sqrt_pos_kernel(const double *a, double *r, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = seqik::sqrt_pos_(a[i]);
}

int seqik_selftest_sqrt_pos(const double *a, double *r, int64_t n)
{
    if (!a || !r || n < 0) return fail(SEQIK_ERR_BAD_ARG, "null pointer argument%s");
    if (n == 0) return SEQIK_OK;
    double *d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * 2 * n));
    hipError_t e = hipMemcpy(d, a, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(seqik_selftest_sqrt_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d, d + n, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(r, d + n, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SEQIK_ERR_HIP, "seqik_selftest_sqrt_pos: %s", hipGetErrorString(e));
    return SEQIK_OK;
}

int seqik_frame_chunk_plan(int64_t n_frames, const SeqikOptions *opt, int32_t *chunk, int32_t *halo, int64_t *n_chunks)
{
    int32_t c = 0, h = 0, lead = 0;
    int64_t k = 0;
    const bool chunked = n_frames > 0 && pick_frame_chunks(opt, n_frames, c, h, lead, k);
    if (chunk) *chunk = chunked ? c : 0;
    if (halo) *halo = chunked ? h : 0;
    if (n_chunks) *n_chunks = chunked ? k : 0;
    return SEQIK_OK;
}
#undef TRY_OUT

}  // extern "C"
