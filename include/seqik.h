/*
 * seqik.h -- C ABI of libseqik_hip.so: batched sequential leg inverse kinematics on
 * AMD Instinct MI355X (gfx950).
 *
 * The reference (NeLy-EPFL/sequential-inverse-kinematics, `seqikpy` 1.0.2) is pure
 * Python and has no FFI; its narrowest seam for this path is the per-frame pair
 *     LegInvKinBase.calculate_ik(chain, target_pos, initial_angles)
 *     LegInvKinBase.calculate_fk(chain, joint_angles)
 *         seqikpy/leg_inverse_kinematics.py:62-77
 * called from the serial frame loop of
 *     LegInvKinSeq.calculate_ik_stage         seqikpy/leg_inverse_kinematics.py:200-322
 * under
 *     LegInvKinSeq.run_ik_and_fk              seqikpy/leg_inverse_kinematics.py:324-403
 * with chains built by
 *     KinematicChainSeq.create_leg_chain_stage_1..4   seqikpy/kinematic_chain.py:152-421.
 * A batched replacement has to sit at the run_ik_and_fk level: the entry points
 * below take whole recordings (sequences x legs x frames) and return what
 * run_ik_and_fk returns.  INTEGRATION.md shows the ctypes binding a seqikpy
 * maintainer would add.
 *
 * Conventions: plain pointers and sizes, all arrays dense C order, float64 unless
 * noted; the caller owns every buffer; nothing is retained after return.  Functions
 * return 0 on success or a negative SEQIK_ERR_* code; seqik_last_error() gives a
 * thread-local message.  The host wrapper maps SEQIK_ERR_X0_OUT_OF_BOUNDS /
 * SEQIK_ERR_BAD_BOUNDS / SEQIK_ERR_BAD_STAGE to ValueError, the exception type the
 * reference raises in those cases (scipy least_squares: "Initial guess is outside of
 * provided bounds"; seqikpy/leg_inverse_kinematics.py:232-236, 350-353).
 */
#ifndef SEQIK_H
#define SEQIK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEQIK_ABI_VERSION 7

#define SEQIK_OK 0
#define SEQIK_ERR_HIP (-1)               /* HIP runtime error (no device, launch failure, ...) */
#define SEQIK_ERR_X0_OUT_OF_BOUNDS (-2)  /* a stage seed lies outside its [lb, ub] */
#define SEQIK_ERR_BAD_BOUNDS (-3)        /* lb >= ub for some joint */
#define SEQIK_ERR_BAD_ARG (-4)           /* null pointer, negative size, ... */
#define SEQIK_ERR_BAD_STAGE (-5)         /* stages not within 1..4 / not consecutive */

/* Joint (DOF) order used by every angle array of this ABI.  Axis naming follows the
 * reference: "yaw" = rotation about X, "pitch" about Y, "roll" about Z
 * (seqikpy/kinematic_chain.py:176,184,237). */
enum {
    SEQIK_DOF_THC_YAW = 0,
    SEQIK_DOF_THC_PITCH = 1,
    SEQIK_DOF_THC_ROLL = 2,
    SEQIK_DOF_CTR_PITCH = 3,
    SEQIK_DOF_CTR_ROLL = 4,
    SEQIK_DOF_FTI_PITCH = 5,
    SEQIK_DOF_TITA_PITCH = 6,
    SEQIK_NDOF = 7
};

/* Everything KinematicChainSeq + INITIAL_ANGLES hold for one leg:
 *   seg     body_size["<leg>_Coxa|Femur|Tibia|Tarsus"]   (seqikpy/utils.py:89-123)
 *   bounds  bounds_dof["<leg>_<dof>"] = (lb, ub), DOF order above (seqikpy/data.py:24-41)
 *   seeds   initial_angles[leg]["stage_1".."stage_4"] concatenated: 4 + 6 + 8 + 9 values,
 *           one per link of the stage chain (seqikpy/data.py:4-22)                        */
typedef struct SeqikLegParams {
    double seg[4];
    double bounds[7][2];
    double seeds[27];
} SeqikLegParams;

typedef struct SeqikOptions {
    int32_t device;       /* HIP device ordinal (host-buffer entry points only); -1 = the calling thread's
                             current device */
    int32_t block_size;   /* threads per workgroup, multiple of 64; 0 = default (64) */
    void **stage_events;  /* nullable: 5 hipEvent_t handles, recorded on the launch stream in front of the
                             stage-1..4 kernels ([0]..[3], only for stages that run) and behind the last one
                             ([4]) -- lets a caller time the individual kernels of one call (see reserved[1]) */
    int32_t reserved[4];  /* [0]: chains per wavefront, 1..64 (0 = automatic: one to four while there are fewer than
                             1024 chains, 64 from there on); 128, 192, ... 4096 (multiples of 64; since round 6) = CHAIN
                             QUEUE of the single-launch lane-per-chain kernel: a wavefront OWNS that many chains of one
                             leg, its 64 lanes start on the first 64 and a lane that has finished its chain takes the
                             next one instead of idling until the wavefront's slowest chain is done (stage by stage).
                             The launch then has 64 / [0] as many wavefronts: a caller that wants the GPU full keeps
                             [0] / 64 as many calls in flight.  Only for runs of all four stages without diagnostics,
                             frame chunks, [1] = 1 or [2] = 1, on the lane-per-chain path ([3] = 1, or more than
                             40 000 chains); refused with SEQIK_ERR_BAD_ARG elsewhere;
                             [1]: 0 = a run of all four stages without diagnostics is ONE launch (every wave takes its
                             chains through stages 1, 2, 3, 4 in turn; stage_events[0] is then recorded in front of that
                             kernel and [1]..[4] behind it), 1 = always one launch per stage (this also switches the
                             AUTOMATIC stage pipeline of [3] off; an explicit [3] >= 2 and frame chunks still use it);
                             seqik_solve_generic*: chain queue of a batch of full wavefronts (persistent wavefronts whose
                             lanes take the next chain of their leg from a counter when they have finished one): 0 =
                             automatic (batches with at least four chains per lane of the GPU), 1 = never, 2 = always;
                             [2]: 0 = all lanes of a wavefront carry the same leg, 1 = consecutive chains (legs
                             interleaved);
                             [3]: stage pipeline -- a workgroup of four wavefronts per group of chains, wavefront k
                             running stage k of frame t while wavefront k-1 is already at frame t+1, the prefix frames
                             handed over through LDS: 0 = automatic (runs of all four stages without diagnostics over
                             at most 40 000 chains / frame chunks, where the serial path per frame is what counts),
                             1 = never, 2 = whenever applicable, 3 = as 2 but without lane pairs (a wavefront that
                             carries at most 32 chains runs each on two or more lanes, and neighbouring lanes split
                             the finite-difference columns and the trial point's sin / cos of a pass between them;
                             3 is for measurements), 4 / 5 = as 2 / 3 but never the 256-register latency build of
                             the pipeline kernels (tests: that build against the plain one).  None of these changes a
                             result bit. */
    /* ---- frame chunks (ABI 2): ONE long recording on the whole GPU -------------------------------------------
     * The reference walks a recording serially because frame t is warm-started from frame t-1
     * (seqikpy/leg_inverse_kinematics.py:259-282).  With frame_chunk != 0 a run of all four stages (no status /
     * nfev requested) cuts every chain into chunks of frame_chunk frames that are solved concurrently, entirely
     * on the device:
     *   1. speculative launch: chunk 0 starts from the seeds (or init_angles) as the reference does, chunk k >= 1
     *      from the seeds frame_halo frames early (a run-in whose results are dropped);
     *   2. verification: the state the run-in reached (7 joint angles) is compared with the true last frame of
     *      chunk k-1; a chunk is consistent when they agree to chunk_tol rad in every joint;
     *   3. repair: inconsistent chunks whose predecessor is consistent are re-solved from the true state (bit
     *      identical to what the serial walk does from there), the successors are verified again -- chunk_rounds
     *      such rounds in parallel, then one serial sweep per chain that leaves every chunk consistent.
     * Every stored frame was therefore computed by the reference's algorithm from a warm start within chunk_tol of
     * the serial trajectory's; the result equals the serial one to about chunk_tol where the solver is well-posed
     * (measured: <= 1.1e-5 rad on the shipped recordings with the defaults, the reference's own run-to-run noise
     * being ~5e-5); inside kinematic-singularity episodes, where the reference itself is chaotic, a
     * 1e-6 difference can pick the other branch.  frame_chunk = 0 keeps the serial walk (bit-exact). */
    int32_t frame_chunk;  /* 0 = serial (default: the reference's walk, bit-exact); > 0 = frames per chunk;
                             -1 = automatic: a function of n_frames ALONE (seqik_frame_chunk_plan), so a recording gets
                             the same chunks -- hence the same bits -- alone, beside other recordings in one call, or in a
                             longer batch: serial below 48 frames, 4 frames after a run-in of 4 up to 1365 frames, then
                             8..64 frames after a run-in of 8.  The automatic mode also guards itself PER CHAIN, on the
                             device: a chain of which more than one chunk in eight fails the first verification (data
                             with several equivalent leg configurations, where a run-in does not find the serial
                             trajectory) is walked serially instead -- the reference's result bit for bit for that
                             chain; chunk_stats[8] counts such chains, chunk_flags marks their chunks */
    int32_t frame_halo;   /* run-in frames of a speculative chunk; 0 = default (8) */
    double chunk_tol;     /* consistency tolerance in rad; 0 = default (1e-6); negative = 0 (a chunk is accepted
                             only if the run-in reproduced the true state bit for bit) */
    int32_t chunk_rounds; /* parallel repair rounds before the serial sweep; 0 = default (3) */
    int32_t frame_lead;   /* ABI 3.  > 0: the call is a SLAB of a longer recording (frame-sharding over GPUs): its first
                             frame_lead frames are the run-in of chunk 0 (solved from the seeds, not stored), chunk k
                             stores frames [frame_lead + k C, frame_lead + (k + 1) C).  Chunk 0 is then speculative like
                             the others; it is verified / repaired against d_init_angles (the true state of the frame in
                             front of frame_lead) when that is given -- typically in a second, chunk_resume call */
    int32_t *chunk_stats; /* nullable int32[16], HOST memory for the host-buffer entry points, DEVICE memory for the
                             _device ones: [0] chunks, [1] frames per chunk, [2] run-in frames, [3..5] chunks
                             re-solved in repair rounds 1..3 (later rounds are added to [5]), [6] chunks re-solved
                             by the serial sweep, [7] chunks found inconsistent by the first verification, [8] chains
                             the automatic mode walked serially instead, [9] their chunks, [10..15] zero */
    /* ---- ABI 3 ---------------------------------------------------------------------------------------------------- */
    uint8_t *chunk_flags; /* nullable [n_seq][n_legs][K] (K from seqik_frame_chunk_plan; host / device memory as
                             chunk_stats): per chunk, bit 0 = failed the first verification, bit 1 = re-solved in a
                             repair round, bit 2 = re-solved by the sweep, bit 3 = its chain was walked serially
                             (bit 7 of a chain's first entry: INPUT of chunk_resume = 4, see there).
                             Lets a caller see WHERE a recording is chaotic (kinematic-singularity episodes) */
    double *chunk_states; /* nullable [n_seq][n_legs][K][7], DEVICE entry point only, in/out: the warm start the stored
                             frames of every chunk were computed from.  Kept by the caller between a call and the
                             chunk_resume call that continues it */
    int32_t chunk_resume; /* 1: no speculative pass -- d_angles / d_fk / chunk_states hold the result of an earlier call
                             with the same geometry, d_init_angles the TRUE state in front of chunk 0; only
                             verification, repair rounds and sweep run (chunk 0 included).  Needs an explicit
                             frame_chunk > 0 and chunk_states.  2: as 1, and chunk 0 is accepted only if its run-in
                             reproduced d_init_angles bit for bit (otherwise re-solved from it): the slab then continues
                             the frames in front of it EXACTLY, like a carried slab of a stream.
                             ABI 7 -- LOCKSTEP pieces of one chunked call, for a recording whose slabs live on several GPUs
                             (seqikpy_amd/frame_sharding.py): the ranks run the speculative pass, every {scan, repair} round
                             and the sweep of ONE call together, exchanging the slabs' end states in between, so the result is
                             the one-GPU call's bit for bit whatever the number of ranks.  3: the speculative pass and the
                             first verification only (needs chunk_states).  4: ONE {scan, repair} round: d_init_angles =
                             the CURRENT last frame of the slab to the left (NULL on the first slab), and where the caller
                             has set bit 7 (0x80) of chunk_flags[chain][0] -- "the last chunk of the slab to the left is
                             itself inconsistent in this round" -- an inconsistent chunk 0 is held back, as a chunk whose
                             predecessor is about to change is inside one call (the bit is cleared).  5: the final scan and
                             the serial sweep only (d_init_angles = the FINAL last frame of the slab to the left) */
    int32_t pad2_;
} SeqikOptions;

/* Element (double) strides of the device buffers of seqik_solve_seq_device.  Chain c = seq * n_legs + leg.
 *   key point (c, row, t):  pose   + c * pose_chain + row * pose_row + t * pose_frame   (x, y, z contiguous)
 *   angle     (c, dof, t):  angles + c * ang_chain  + dof * ang_dof  + t * ang_frame
 * NULL layout = the dense arrays of seqik_solve_seq: pose [chain][frame][5][3], angles [chain][frame][7].
 * The HBM-friendly "planar" layout (what bench.py uses) stores every key-point row and
 * every joint as its own time series -- pose [chain][5][frame][3], angles [chain][7][frame] -- so each
 * stage kernel touches exactly the rows it needs; the per-joint series are also the reference's own
 * output format (joint_angles_dict["Angle_<leg>_<dof>"] is an (N,) array). */
typedef struct SeqikLayout {
    int64_t pose_chain, pose_row, pose_frame;
    int64_t ang_chain, ang_dof, ang_frame;
} SeqikLayout;

/* Optional fused alignment of one leg (AlignPose.align_leg, seqikpy/alignment.py:436-487).  When an
 * array of these is passed, `pose` holds RAW (un-aligned) key points and the kernels compute
 *     aligned[row] = (raw[row] - fixed_coxa) * scale + template_coxa   (rows 1..4),  aligned[0] = template_coxa
 * in their prologue, with the reference's rounding sequence, before forming the stage target.
 *   fixed_coxa     AlignPose.get_fixed_pos(raw[:, 0, :])   (quantile statistics, computed by the caller)
 *   scale          AlignPose.find_scale_leg(...)
 *   template_coxa  body_template["<leg>_Coxa"]                                                         */
typedef struct SeqikAffine {
    double fixed_coxa[3];
    double scale;
    double template_coxa[3];
} SeqikAffine;

int seqik_abi_version(void);
int seqik_device_count(void);
const char *seqik_last_error(void);
/* Compute units, peak shader clock (kHz) and HBM size of a device (any out pointer may be NULL). */
int seqik_device_attributes(int32_t device, int32_t *compute_units, int32_t *clock_khz, int64_t *hbm_bytes);
/* The library keeps one stage hand-off workspace (96 B per leg-frame of the largest call so far) per HIP stream it
 * has launched on.  This drains the devices and frees them all (they are re-created on demand).  No reference
 * counterpart: the reference allocates per frame in Python. */
int seqik_release_workspaces(void);

/* Validates `legs` exactly as the reference would fail at frame 0 (bounds order,
 * seeds inside bounds).  Returns SEQIK_OK or the error code; no GPU needed. */
int seqik_validate_legs(const SeqikLegParams *legs, int32_t n_legs, int32_t first_stage, int32_t last_stage);

/*
 * LegInvKinSeq.run_ik_and_fk for n_seq independent recordings ("sequences") of
 * n_frames frames and n_legs legs each; HOST buffers, blocking.
 *
 *   pose    [n_seq][n_legs][n_frames][5][3]  aligned key points; row 0 = Thorax-Coxa origin,
 *           row k = end effector of stage k  (aligned_pos["<leg>_leg"], leg_inverse_kinematics.py:371-375)
 *   legs    [n_legs]
 *   first_stage..last_stage  consecutive stages to run, 1 <= first <= last <= 4
 *   angles  [n_seq][n_legs][n_frames][7]  in/out: columns of stages < first_stage are READ
 *           (earlier results, as joint_angles_dict in the reference), columns of the stages run
 *           are written, others untouched
 *   fk      nullable [n_seq][n_legs][n_frames][9][3]; written only when last_stage == 4
 *           (stage-4 forward kinematics + origin, leg_inverse_kinematics.py:279-282)
 *   status, nfev  nullable int32 [n_seq][n_legs][n_frames][4]: scipy termination status /
 *           trial-evaluation count per (frame, stage); requesting either costs one extra
 *           Jacobian per solve
 *   init_angles  nullable [n_seq][n_legs][7]: joint angles of the frame that PRECEDES frame 0 of each
 *           chain; frame 0 is then warm-started from them instead of from the seeds (continuation of a
 *           recording that is processed in pieces; the inert seed entries still come from `legs`)
 *   affine  nullable [n_legs]: fuse AlignPose.align_leg into the kernels (pose is then RAW)
 * Frame t of a chain is warm-started from frame t-1 of the same chain; frame 0 from the seeds.
 */
int seqik_solve_seq(const double *pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                    const SeqikLegParams *legs, int32_t first_stage, int32_t last_stage,
                    double *angles, double *fk, int32_t *status, int32_t *nfev,
                    const double *init_angles /* nullable [n_seq][n_legs][7] */,
                    const SeqikAffine *affine /* nullable [n_legs] */, const SeqikOptions *opt);

/*
 * Same computation on DEVICE buffers of the current HIP device, enqueued on `hip_stream`
 * (a hipStream_t passed as void*, NULL = default stream) and NOT synchronised; `legs` is a
 * host pointer (copied before return); `layout` describes d_pose / d_angles (NULL = dense, as
 * above); d_fk is always [chain][frame][9][3], d_status / d_nfev [chain][frame][4].
 * This is the entry point the benchmark times.
 */
/* Self-test of the floating-point contract: q[i] = a[i] / b[i] and r[i] = sqrt(a[i]) computed on the device by
 * the kernels' own division / square root (the gfx950 expansions without the range-scaling steps).  Host buffers of n
 * doubles each.  The GPU tests compare the results with IEEE division / square root over the operand range the
 * contract states, and with the special values.  No reference counterpart. */
int seqik_selftest_div_sqrt(const double *a, const double *b, double *q, double *r, int64_t n);
/* ABI 4.  r[i] = the select-free square root the kernels apply to the Coleman-Li distances v = x - lb | ub - x | 1 of a
 * strictly feasible x (`sqrt_pos_`, csrc/seqik_core.hpp): positive finite operands only.  A joint limit of exactly 0 makes
 * v the smallest subnormal (the trial point scipy's make_strictly_feasible(rstep = 0) pins next to the limit); the GPU tests
 * check that value, the other subnormal powers of two and the per-leg values  lb_in - lb,  ub - ub_in  of the shipped
 * limits against IEEE sqrt.  No reference counterpart. */
int seqik_selftest_sqrt_pos(const double *a, double *r, int64_t n);

/* ABI 4 / 6.  Device faults.  The reference reports every failure as a Python exception (IKPy raises on scipy status -1,
 * seqikpy/leg_inverse_kinematics.py:62-69 -> ikpy); it never returns silent garbage.  The one failure a kernel of this
 * library can detect by itself -- the stage pipeline's watchdog: a lane that waited 2^24 passes for its neighbour wave,
 * impossible by construction -- fills the rest of that chain with NaN and sets a fault word in mapped host memory: since
 * ABI 6 ONE WORD PER (device, stream), so that host threads that drive different GPUs or streams never consume each
 * other's faults (more than 63 distinct (device, stream) pairs in a process share the last word).  The blocking entry
 * points (seqik_solve_seq, seqik_stream_wait) read and clear the words of THEIR streams after they have synchronised and
 * return SEQIK_ERR_HIP with a message; seqik_solve_seq_device, which does not synchronise, reports a fault left by EARLIER
 * launches ON THE SAME STREAM when it is entered.  Callers of the device entry point call, after synchronising,
 * seqik_check_faults_stream(stream) (that stream of the current device only; thread-safe next to other streams' users) or
 * seqik_check_faults() (EVERY word: single-threaded callers, end-of-job checks): SEQIK_OK, or SEQIK_ERR_HIP (message
 * in seqik_last_error(); the words read are cleared). */
int seqik_check_faults(void);
int seqik_check_faults_stream(void *hip_stream);

/* The frame chunks a call over recordings of n_frames frames would use with these options (frame_chunk / frame_halo /
 * frame_lead): frames per chunk, run-in frames, chunks per chain K -- all 0 when the call would be walked serially.
 * A function of n_frames and the options alone (not of the number of recordings or legs).  No GPU needed. */
int seqik_frame_chunk_plan(int64_t n_frames, const SeqikOptions *opt, int32_t *chunk, int32_t *halo, int64_t *n_chunks);

int seqik_solve_seq_device(const double *d_pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                           const SeqikLegParams *legs, int32_t first_stage, int32_t last_stage,
                           double *d_angles, double *d_fk, int32_t *d_status, int32_t *d_nfev,
                           const double *d_init_angles /* nullable [n_seq][n_legs][7] */,
                           const SeqikLayout *layout, const SeqikAffine *affine /* nullable [n_legs] */,
                           const SeqikOptions *opt, void *hip_stream);

/*
 * LegInvKinGeneric.run_ik_and_fk (seqikpy/leg_inverse_kinematics.py:545-613): one 9-link chain per leg
 * (KinematicChainGeneric, seqikpy/kinematic_chain.py:464-530), 7 unknowns, target = the claw (pose row 4),
 * frame t warm-started from frame t-1, frame 0 from legs[l].seeds[18..26] = initial_angles["stage_4"]
 * applied positionally to the links Base, ThC_roll, ThC_yaw, ThC_pitch, CTr_pitch, CTr_roll, FTi, TiTa, Claw.
 * Arrays and options as for seqik_solve_seq; angles come back in this ABI's DOF order; status / nfev are
 * [n_seq][n_legs][n_frames] (one solve per frame).  The problem is rank-deficient (3 equations): the
 * reference's angles depend on LAPACK round-off, only the claw position is comparable.
 */
int seqik_validate_legs_generic(const SeqikLegParams *legs, int32_t n_legs);
int seqik_solve_generic(const double *pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                        const SeqikLegParams *legs, double *angles, double *fk, int32_t *status, int32_t *nfev,
                        const double *init_angles, const SeqikAffine *affine, const SeqikOptions *opt);
int seqik_solve_generic_device(const double *d_pose, int64_t n_seq, int32_t n_legs, int64_t n_frames,
                               const SeqikLegParams *legs, double *d_angles, double *d_fk, int32_t *d_status,
                               int32_t *d_nfev, const double *d_init_angles, const SeqikLayout *layout,
                               const SeqikAffine *affine, const SeqikOptions *opt, void *hip_stream);

/*
 * HeadInverseKinematics.compute_head_angles (seqikpy/head_inverse_kinematics.py:103-140) for n_frames
 * frames: closed-form head roll / pitch / yaw and, per side, antenna yaw / pitch.
 *   r_head, l_head  [n_frames][2][3]  aligned antenna base and tip (aligned_pos["R_head"], ["L_head"])
 *   neck            [3] (neck_stride = 0: the template's fixed neck, the usual case) or
 *                   [n_frames][3] (neck_stride = 3)
 *   rest_head_pitch, rest_antenna_pitch   zero-pose angles of the body template
 *                   (get_rest_head_pitch / get_rest_antenna_pitch, :304-328; computed by the caller)
 *   compute_ant     0: only the three head angles
 *   angles          [7][n_frames] (or [3][n_frames]): rows in the reference's dict order -- head roll,
 *                   head pitch, head yaw, antenna yaw L, antenna pitch L, antenna yaw R, antenna pitch R
 * The _device variant takes device pointers, enqueues on `hip_stream` and does not synchronise.
 */
int seqik_head_angles(const double *r_head, const double *l_head, int64_t n_frames, const double *neck,
                      int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch, int32_t compute_ant,
                      double *angles, const SeqikOptions *opt);
int seqik_head_angles_device(const double *d_r_head, const double *d_l_head, int64_t n_frames, const double *d_neck,
                             int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch,
                             int32_t compute_ant, double *d_angles, void *hip_stream);

/* ABI 5.  The same with the two things the reference's per-quantity methods allow beyond compute_head_angles:
 *   n_points   key points per head record: r_head / l_head are [n_frames][n_points][3] and only point 0 (and, for the
 *              antennae, point 1) is read.  1 = one head key point per side, e.g. a bristle, with compute_ant = 0
 *              (seqikpy/head_inverse_kinematics.py:26); compute_ant != 0 with n_points < 2 is SEQIK_ERR_BAD_ARG.
 *   head_roll  nullable [n_frames]: the antenna vectors are derotated by THIS head roll instead of the frame's own --
 *              compute_antenna_pitch(side, head_roll) / compute_antenna_yaw(side, head_roll), :242-307, take it as an
 *              argument.  Rows 0-2 are the frame's own head angles either way.
 * seqik_head_angles(...) == seqik_head_angles_ex(..., n_points = 2, head_roll = NULL). */
int seqik_head_angles_ex(const double *r_head, const double *l_head, int64_t n_frames, int32_t n_points, const double *neck,
                         int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch, int32_t compute_ant,
                         const double *head_roll, double *angles, const SeqikOptions *opt);
int seqik_head_angles_ex_device(const double *d_r_head, const double *d_l_head, int64_t n_frames, int32_t n_points,
                                const double *d_neck, int64_t neck_stride, double rest_head_pitch,
                                double rest_antenna_pitch, int32_t compute_ant, const double *d_head_roll,
                                double *d_angles, void *hip_stream);

/* ABI 5.  HeadInverseKinematics.angle_between_segments (seqikpy/head_inverse_kinematics.py:163-182) for general vectors:
 * out[i] = acos(v1_i . v2_i / (|v1_i| |v2_i|)), negated unless det([axis, v1_i, v2_i]) > 0.  v1 / v2 are [n][3] (stride 3)
 * or one vector [3] used for every row (stride 0); axis [3]; host buffers. */
int seqik_signed_angles(const double *v1, int64_t v1_stride, const double *v2, int64_t v2_stride, const double *axis,
                        int64_t n, double *out, const SeqikOptions *opt);

/*
 * Peer gather ("RCCL over xGMI only for the final joint-angle gather").  One process per GPU; the rank
 * that collects the joint angles allocates its receive buffers with seqik_peer_alloc and publishes their handles
 * (SEQIK_PEER_HANDLE_BYTES opaque bytes, sent through any channel, e.g. torch.distributed); every other rank
 * maps its slot with seqik_peer_open and pushes its angle block into it with seqik_peer_copy after each solve:
 * a copy-engine transfer over the direct xGMI link, ordered on the given HIP stream, no compute unit busy on
 * either GPU.  Completion / reuse of a slot is signalled by the caller (seqikpy_amd/peer_gather.py: an 8-byte
 * all-reduce enqueued behind the copy).  The reference has no counterpart (its parallel example returns pickled
 * dictionaries from a multiprocessing.Pool, examples/example_leg_inv_kinematics_parallel.py:186-187).
 *
 *   seqik_peer_alloc / _free     device memory that can be exported (one allocation per call)
 *   seqik_peer_export            handle of such an allocation
 *   seqik_peer_open / _close     map / unmap another process's allocation on the current device
 *   seqik_peer_copy              asynchronous device-to-device copy (either side may be a mapped pointer)
 */
#define SEQIK_PEER_HANDLE_BYTES 64
int seqik_peer_alloc(void **d_ptr, size_t bytes);
int seqik_peer_free(void *d_ptr);
int seqik_peer_export(const void *d_ptr, unsigned char *handle);
int seqik_peer_open(const unsigned char *handle, void **d_ptr);
int seqik_peer_close(void *d_ptr);
int seqik_peer_copy(void *d_dst, const void *d_src, size_t bytes, void *hip_stream);

/*
 * Streaming (BASELINE config 5): recordings that do not have to fit or live in HBM are pushed through
 * the kernels in SLABS of n_seq sequences x n_legs x n_frames from host buffers.  The reference's
 * counterpart is one AlignPose.align_pose + LegInvKinSeq.run_ik_and_fk call per piece of a recording
 * (seqikpy/alignment.py:345, seqikpy/leg_inverse_kinematics.py:324).  Upload, the four stage kernels
 * and download of consecutive slabs overlap on three HIP streams over n_slots device slots.
 *
 *   seqik_host_alloc / _free      pinned host memory (hipHostMalloc); slabs in pinned memory are
 *   seqik_host_register / _unregister   copied asynchronously, pageable ones still work but serialise
 *   seqik_stream_open    legs / affine / layout as in seqik_solve_seq_device (copied); slab_seq = the largest
 *                        n_seq a submit may carry; want_fk: also return the stage-4 FK; n_slots: slabs in
 *                        flight (2-3 hide the copies); carry != 0: consecutive slabs are consecutive pieces
 *                        IN TIME of the same n_seq recordings -- frame 0 of a slab is warm-started from the
 *                        last frame of the slab before it (device-resident hand-over), so the result equals
 *                        one call over the concatenated recording; generic != 0: LegInvKinGeneric chains;
 *                        opt->device selects the GPU
 *   seqik_stream_submit  enqueues one slab: pose [n_seq][n_legs][n_frames][5][3] (or `layout`) in, angles
 *                        [n_seq][n_legs][n_frames][7] (or `layout`) and fk [n_seq][n_legs][n_frames][9][3]
 *                        out.  Returns as soon as the work is queued; blocks only while all slots are busy.
 *                        The host buffers must stay valid and untouched until seqik_stream_wait returns
 *                        (or until n_slots further slabs have been submitted).
 *   seqik_stream_wait    blocks until every submitted slab's results are in its host buffers
 *   seqik_stream_reset_carry   the next slab starts new recordings (frame 0 from the seeds again)
 *   seqik_stream_set_carry     the next slab continues from a given state (a recording whose earlier frames were
 *                        solved elsewhere: another GPU, another call)
 *   seqik_stream_close   drains and frees everything
 */
typedef struct SeqikStream SeqikStream;
void *seqik_host_alloc(size_t bytes);
void seqik_host_free(void *p);
int seqik_host_register(void *p, size_t bytes);
int seqik_host_unregister(void *p);
int seqik_stream_open(SeqikStream **out, int32_t n_legs, const SeqikLegParams *legs, const SeqikAffine *affine,
                      int64_t slab_seq, int64_t n_frames, const SeqikLayout *layout, int32_t want_fk,
                      int32_t n_slots, int32_t carry, int32_t generic, const SeqikOptions *opt);
int seqik_stream_submit(SeqikStream *s, const double *pose, int64_t n_seq, double *angles, double *fk);
int seqik_stream_wait(SeqikStream *s);
int seqik_stream_reset_carry(SeqikStream *s);
/* carried streams: the next slab is warm-started from `init` [n_seq][n_legs][7] (the joint angles of the frame in front of
 * it; host memory, or device memory of the stream's GPU when on_device != 0) instead of the seeds / the slab before */
int seqik_stream_set_carry(SeqikStream *s, const double *init, int64_t n_seq, int32_t on_device);
int seqik_stream_close(SeqikStream *s);

/*
 * Alignment statistics: the whole-recording reductions behind AlignPose.align_leg (seqikpy/alignment.py:83-87,
 * 392-434), i.e. the constants of SeqikAffine.  Per leg seven per-frame series -- coxa x, y, z (get_fixed_pos) and
 * the lengths of coxa, femur, tibia, tarsus (get_mean_length) -- are extracted from the RAW key points on the GPU
 * and sorted; _finish returns the requested order statistics, to which the caller applies numpy's own quantile
 * interpolation / mean / scale formulas (seqikpy_amd/alignment.py does), so the constants are bit-identical to the
 * reference's.  Host cost avoided: ~5 s per million frames x 6 legs.
 *   _open     room for capacity_frames frames per leg on device opt->device
 *   _add      appends n_seq x n_frames frames of every leg: pose [n_seq][n_legs][n_frames][5][3] or `layout`
 *             (pose strides only), host memory (pose_on_device = 0, blocking) or device memory (enqueued on
 *             hip_stream); slabs may come in any order
 *   _finish   sorts; out [n_legs][7][n_ranks] = value at the 0-based ranks `ranks[i]` of each ascending series
 *             (series order: coxa x, y, z, then the four segment lengths); blocking
 *   _reset    forget the frames added so far;  _close  free everything
 */
typedef struct SeqikAlignStats SeqikAlignStats;
int seqik_align_stats_open(SeqikAlignStats **out, int32_t n_legs, int64_t capacity_frames, const SeqikOptions *opt);
int seqik_align_stats_add(SeqikAlignStats *s, const double *pose, int32_t pose_on_device, int64_t n_seq,
                          int64_t n_frames, const SeqikLayout *layout, void *hip_stream);
int seqik_align_stats_finish(SeqikAlignStats *s, const int64_t *ranks, int32_t n_ranks, double *out, void *hip_stream);
int seqik_align_stats_reset(SeqikAlignStats *s);
int seqik_align_stats_close(SeqikAlignStats *s);

#ifdef __cplusplus
}
#endif
#endif /* SEQIK_H */
