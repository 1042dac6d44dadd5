// seqik_core.hpp -- per-chain sequential leg-IK solver, CDNA4 (gfx950) device code.
//
// One *chain* = one (sequence, leg).  The path replaced is
//   LegInvKinSeq.calculate_ik_stage         seqikpy/leg_inverse_kinematics.py:200-322
//   KinematicChainSeq.create_leg_chain_*    seqikpy/kinematic_chain.py:152-421
//   LegInvKinBase.calculate_ik/calculate_fk seqikpy/leg_inverse_kinematics.py:62-77
//   ikpy Chain.inverse_kinematics -> scipy.optimize.least_squares(method="trf")
// i.e. for every frame four bounded trust-region-reflective least-squares problems with
// 2 (stages 1-3) or 1 (stage 4) effective unknowns, frame t warm-started from frame t-1.
//
// Decomposition (same loop order as the reference, :373-385): STAGE BY STAGE, and inside a
// stage ONE LANE PER CHAIN walking its frames serially (run_stage<STAGE>).  A lane runs a flat
// state machine whose step ("pass") is one outer TRF iteration, so the 64 lanes of a
// wavefront never wait for each other at frame boundaries -- only the total work per chain
// has to balance -- and because all lanes of a wave are in the same stage, the stage
// properties (number of unknowns, rotation axes, rank handling) are compile-time constants.
// seqik_hip.hip calls run_stage<1..4> back to back from one kernel (default) or from one
// kernel per stage (diagnostics, stage subsets).  No MFMA: the sub-problems are 5x2 / 4x1.
//
// Structure that is exploited (none of it changes the arithmetic, see below):
//   * links that cannot move the end effector (base, "fixed" links, last link) have
//     exactly-zero Jacobian columns: they are never evaluated;
//   * the product of the fixed links in front of the active ones is a per-frame constant,
//     rebuilt from the stored angles of the earlier stages (2-6 sin/cos pairs) -- what the
//     reference does by re-building the whole ikpy chain per frame, kinematic_chain.py:200+;
//   * every link matrix is [axis rotation | (0,0,-length)]: products are written out per axis.
//
// Floating-point contract: every operation below is one IEEE binary64 op or an explicit fused
// multiply-add (fma_ = v_fma_f64); sums run as acc = fma(a_i, b_i, acc) in the same index order
// as the generic restatement in oracle/seqik_oracle.c (terms that are exactly 0 and factors
// that are exactly 1 dropped), quotients by a common denominator use one shared reciprocal,
// sin/cos is the same Cody-Waite + fdlibm-polynomial routine.  Built with -ffp-contract=off
// (nothing else gets fused) the kernels therefore reproduce the oracle bit for bit; tests/
// check exactly that.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define SEQIK_HD __host__ __device__ __forceinline__

namespace seqik {

enum : int { AXIS_X = 0, AXIS_Y = 1, AXIS_Z = 2 };

SEQIK_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---------------------------------------------------------------------------
// IEEE division and square root, as the compiler expands them for gfx950 minus the range scaling.
//
// hipcc expands a binary64 `a / b` into 11 instructions -- v_div_scale x 2, v_rcp_f64, four Newton multiply-adds, a
// multiply, a residual multiply-add, v_div_fmas, v_div_fixup -- and `sqrt(x)` into 18 (scale test + v_ldexp, v_rsq_f64,
// two multiplies, seven multiply-adds, v_ldexp back, a zero / infinity test with two selects).  The v_div_scale /
// v_div_fmas / v_ldexp steps only move operands whose exponents are within ~2^250 of the limits of the format into a
// range where the Newton iteration cannot over- or underflow: for every other operand they are the identity, and the
// remaining sequence -- the same instructions on the same values -- gives the same, correctly rounded result.  div_() and
// sqrt_() are those sequences without the scaling (9 and 13 instructions: v_div_fixup, which supplies the IEEE results for
// zero / infinite / NaN operands, and the zero / infinity selects of the square root stay).  A pass of the solver holds
// ~25 divisions and ~14 square roots: 7 % fewer vector instructions per benchmark step.
//   Valid while |a|, |b|, |a / b| lie in [2^-767, 2^767] or are 0 / inf / NaN (division) and x >= 2^-767 or x == 0
// (square root).  Angles, segment lengths, residuals, Jacobians, trust-region radii and multipliers of this solver live
// between ~1e-100 and ~1e+20 for any key points a camera can produce (the smallest quantity, the Levenberg-Marquardt
// multiplier after scipy's ten thousandfold reductions, is ~1e-50 and enters as its square).  The oracle divides and takes
// roots with the host's IEEE operations; every parity test and the soak therefore also checks this equivalence.
//   ONE quantity leaves that range: the Coleman-Li distance v = |x - bound| next to a joint limit of exactly 0 (the
// shipped tables have three: CTr_pitch ub, FTi_pitch lb, TiTa_pitch ub).  A trial point that lands on such a limit is moved
// to next_toward(0, .) = +-2^-1074 (strictly_feasible0), so the next pass takes sqrt_pos_(2^-1074).  For that operand -- and
// for every even power of two down there -- the sequence is still exact: v_rsq_f64 returns the power of two 2^537, g = x y
// = 2^-537 is normal and exact, h g = 1/2 exactly so the first correction is zero, and g g - x = 0 exactly so the others
// are too.  The GPU tier checks this on the device (seqik_selftest_sqrt_pos: 2^-1074, the even powers of two up to
// 2^-700, and lb_in - lb / ub - ub_in of every shipped limit).  A general subnormal operand would NOT be rounded correctly
// (fma(-g, g, x) underflows); the solver cannot produce one from limits that are 0 or of ordinary size -- from x =
// 2^-1074 the next iterate is x + d p_h with d = sqrt(v) = 2^-537 -- and seqik_validate_legs refuses non-zero limits below
// 2^-600 in magnitude, the only other way to get there (limit_too_small, seqik_consts.hpp).
// SEQIK_IEEE_DIV_SQRT=1 builds the kernels with the compiler's full expansions (A/B check, tests/tools/soak_parity.py).
// ---------------------------------------------------------------------------
#ifndef SEQIK_IEEE_DIV_SQRT
#define SEQIK_IEEE_DIV_SQRT 0
#endif

#ifndef SEQIK_FAST_PATHS
#define SEQIK_FAST_PATHS 1   // wave-uniform fast paths, see "Wave-uniform fast paths" below; 0 = plain code (A/B)
#endif
// "does any active lane of this wavefront ...?" (one ballot; on the host: this lane)
SEQIK_HD bool wave_any(bool c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(c) != 0ull;
#else
    return c;
#endif
}

// The chain queue of run_stage<..., QUEUE>: which of the active lanes of this wavefront satisfy c (a bit mask; on the host: this
// lane is lane 0), and how many set bits of a mask lie below this lane.
SEQIK_HD unsigned long long wave_ballot(bool c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(c);
#else
    return c ? 1ull : 0ull;
#endif
}

SEQIK_HD int lane_rank_in(unsigned long long mask)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
#else
    (void)mask;
    return 0;
#endif
}

SEQIK_HD int mask_count(unsigned long long mask)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(mask);
#else
    return (int)__builtin_popcountll(mask);
#endif
}

SEQIK_HD double div_(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !SEQIK_IEEE_DIV_SQRT
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    return __builtin_amdgcn_div_fixup(q, b, a);
#else
    return a / b;
#endif
}

SEQIK_HD double sqrt_(double x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !SEQIK_IEEE_DIV_SQRT
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    // +-0 and +inf are their own square roots (the iteration makes NaN of them).  (A wave-uniform branch around the two
    // selects was measured: 67 more branches split the blocks the scheduler works on, 12.15 -> 12.25 ms per step.)
    return __builtin_amdgcn_class(x, 0x260) ? x : g;
#else
    return sqrt(x);
#endif
}
// sqrt_ for arguments known to be positive and finite (the Coleman-Li distances v = ub - x, x - lb or 1 of a strictly
// feasible x; normal, or 2^-1074 next to a limit of exactly 0 -- see above): the iteration alone, without the two selects
// for +-0 / +inf.  Same bits as sqrt_ there (NaN stays NaN).
#ifndef SEQIK_SQRT_POS
#define SEQIK_SQRT_POS 1
#endif
SEQIK_HD double sqrt_pos_(double x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !SEQIK_IEEE_DIV_SQRT && SEQIK_SQRT_POS
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
#else
    return sqrt_(x);
#endif
}
enum : int { STATUS_NONE = -99 };

// ---------------------------------------------------------------------------
// DIAGNOSTIC BUILD ONLY (-DSEQIK_BLOCK_CYCLES=1, scripts/block_cycles.py): where a wavefront's time goes inside run_stage.
// A stamp (s_memtime, shader clock) at every block boundary; the cycles since the previous stamp are charged to the
// block that just ended -- WAVE time, whatever the number of active lanes, which is what a pass costs.  Per-wave sums
// are added to seqik_block_cycles[stage - 1][block] when the stage ends.  The product build has none of this.
// ---------------------------------------------------------------------------
#ifndef SEQIK_BLOCK_CYCLES
#define SEQIK_BLOCK_CYCLES 0
#endif
enum : int { BLK_LOOP = 0, BLK_NEW_SOLVE, BLK_FD_JACOBIAN, BLK_SCALING, BLK_TR_STEP, BLK_IN_BOUNDS, BLK_REFLECTIVE,
             BLK_TRIAL_EVAL, BLK_POST_TRIAL, BLK_FINISHED, BLK_PIPE_WAIT, BLK_COUNT };
// conditional parts of a pass whose executions are counted (wave level: how often a wavefront went through them; lane
// level: how many lanes were active when it did)
enum : int { CNT_NEW_SOLVE = 0, CNT_FEASIBLE_SLOW, CNT_START_EVAL, CNT_BODY, CNT_FIRST_PASS, CNT_TR, CNT_REFLECTIVE, CNT_ACCEPT,
             CNT_FINISHED, CNT_FD_SLOW, CNT_ROOT_ITER, CNT_ROOT_EVAL, CNT_GN_STEP, CNT_BODY_LT16, CNT_BODY_LT32, CNT_COUNT };
#if SEQIK_BLOCK_CYCLES && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long seqik_block_cycles[4][BLK_COUNT + 1];  // [..][BLK_COUNT] = passes (wave level)
extern __device__ unsigned long long seqik_block_entries[4][2 * CNT_COUNT]; // [..][c] wave entries, [..][CNT_COUNT + c] lanes
// per-wavefront tallies in LDS (a counter incremented inside a divergent branch would otherwise become one per lane)
__device__ __forceinline__ unsigned long long *blk_tally()
{
    __shared__ unsigned long long tally[16][2 * CNT_COUNT];
    return tally[threadIdx.x >> 6];
}
template <int C>
__device__ __forceinline__ void blk_count()
{
    const unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == (int)__ffsll((long long)m) - 1) {
        unsigned long long *t = blk_tally();
        t[C] += 1;
        t[CNT_COUNT + C] += (unsigned long long)__popcll(m);
    }
}
#define SEQIK_BLK_COUNT(c) blk_count<c>()
// (every stamp names the block that ENDS there, so each accumulator is indexed by a constant and lives in scalar
// registers: an array indexed by a "current block" variable went to scratch memory, and the s_waitcnt vmcnt(0) of its
// read-modify-write then charged the drain of the stores just issued to whatever stamp came next)
struct BlockClock {
    unsigned long long acc[BLK_COUNT];
    unsigned long long last, passes;
    __device__ __forceinline__ void start()
    {
#pragma unroll
        for (int i = 0; i < BLK_COUNT; ++i) acc[i] = 0;
        passes = 0;
        if ((int)(threadIdx.x & 63) == (int)__ffsll((long long)__ballot(1)) - 1) {
            unsigned long long *t = blk_tally();
#pragma unroll
            for (int i = 0; i < 2 * CNT_COUNT; ++i) t[i] = 0;
        }
        last = __builtin_amdgcn_s_memtime();
    }
    template <int BLK>
    __device__ __forceinline__ void end_of()
    {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[BLK] += now - last;
        last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void finish(int stage)
    {
        end_of<BLK_LOOP>();
        if ((int)(threadIdx.x & 63) == (int)__ffsll((long long)__ballot(1)) - 1) {
#pragma unroll
            for (int i = 0; i < BLK_COUNT; ++i) atomicAdd(&seqik_block_cycles[stage - 1][i], acc[i]);
            atomicAdd(&seqik_block_cycles[stage - 1][BLK_COUNT], passes);
            const unsigned long long *t = blk_tally();
#pragma unroll
            for (int i = 0; i < 2 * CNT_COUNT; ++i) atomicAdd(&seqik_block_entries[stage - 1][i], t[i]);
        }
    }
};
#define SEQIK_BLK_DECL BlockClock blk_clock; blk_clock.start();
#define SEQIK_BLK_END_OF(b) blk_clock.end_of<b>()
#define SEQIK_BLK_PASS() (blk_clock.passes += 1)
#define SEQIK_BLK_END(stage) blk_clock.finish(stage)
#else
#define SEQIK_BLK_DECL
#define SEQIK_BLK_END_OF(b) ((void)0)
#define SEQIK_BLK_PASS() ((void)0)
#define SEQIK_BLK_END(stage) ((void)0)
#endif
#ifndef SEQIK_BLK_COUNT
#define SEQIK_BLK_COUNT(c) ((void)0)
#endif

// ---------------------------------------------------------------------------
// Per-(leg, stage) constants, built on the host by make_leg_consts().
// ---------------------------------------------------------------------------
struct StageConst {
    double lb[2], ub[2];   // bounds of the active link(s), link order
    double seed[2];        // initial_angles["stage_k"] entries of the active link(s)
    double tz_a, tz_b;     // origin_translation z of active link a / b (0 or -segment length)
    double tz_last;        // origin_translation z of the last (inert) link
    double x_pre_sq;       // sum of squares of the (strictly feasible) seed entries in FRONT of the
                           // active links, accumulated in link order as acc = fma(x, x, acc) from 0.0
    double x_suf;          // (strictly feasible) seed entry of the last link
    double thr_lb[2], thr_ub[2];  // 1e-10 * max(1, |lb|), 1e-10 * max(1, |ub|): make_strictly_feasible's thresholds at
                                  // a frame start (constants of the leg; computed on the host with the same operations)
    double lb_in[2], ub_in[2];    // next_toward(lb, ub), next_toward(ub, lb): where make_strictly_feasible(rstep = 0) puts
                                  // a trial point that landed on / beyond a bound
    double gate_lb[2], gate_ub[2];  // 0.0 where the bound is finite, NaN where it is not: `g > gate_lb` is
                                    // `g > 0 && isfinite(lb)` of CL_scaling_vector in one compare (cl_scaling_gated)
    // A warm start that lies within thr of a bound -- a joint the previous frame left pinned on its limit -- is moved by
    // make_strictly_feasible(rstep = 1e-10) to a constant of the leg: lb + thr_lb or ub - thr_ub (the midpoint of tight
    // limits).  That constant and its sin / cos (computed on the host with this file's sincos_cw) are kept here, so such
    // a frame start costs selects instead of two sin / cos evaluations (strictly_feasible_pinned, run_stage).
    double lb_out[2], ub_out[2];
    double thr_lb_g[2], thr_ub_g[2];  // thr_lb / thr_ub, NaN where the bound is infinite (the comparison is then false)
    double sc_lb[2][2], sc_ub[2][2];  // [joint][sin, cos] of lb_out / ub_out
    int32_t max_nfev;      // 100 * number of links of the stage chain (4, 6, 8, 9)
    int32_t pad_;
};

// Optional fused alignment (AlignPose.align_leg, seqikpy/alignment.py:436-487): the key points
// handed to the kernel are RAW and  aligned = (raw - fixed_coxa) * scale + template_coxa  for rows
// 1..4, aligned row 0 = template_coxa.
struct LegAffine {
    double fixed_coxa[3];
    double scale;
    double template_coxa[3];
    int32_t enabled;
    int32_t pad_;
};

struct LegConst {
    StageConst st[4];
    LegAffine aff;
};

// Compile-time description of a stage chain (kinematic_chain.py:152-421):
//   stage 1: Base | yaw(X) pitch(Y)            | CTr_pitch(-coxa)
//   stage 2: Base yaw pitch | roll(Z) CTr_pitch(Y,-coxa) | FTi(-femur)
//   stage 3: Base .. CTr_pitch | CTr_roll(Z) FTi(Y,-femur) | TiTa(-tibia)
//   stage 4: Base .. FTi | TiTa(Y,-tibia)      | Claw(-tarsus)
template <int STAGE>
struct StageTraits {
    static constexpr int NA = (STAGE == 4) ? 1 : 2;
    static constexpr int AXIS_A = (STAGE == 1) ? AXIS_X : (STAGE == 4 ? AXIS_Y : AXIS_Z);
    // Stage 1 has no inert link between the base and the active links; LAPACK then returns
    // exactly-zero singular values for the zero columns and scipy's m < n logic ("never full
    // rank") applies verbatim.  See the ACTIVE SET note in oracle/seqik_oracle.c.
    static constexpr bool DEFICIENT = (STAGE == 1);
};

// cumulative frame: rotation (row major) + translation
struct Frame {
    double r[9];
    double t[3];
};

// ---------------------------------------------------------------------------
// sin / cos -- same algorithm and constants as oracle_sincos()
// ---------------------------------------------------------------------------
SEQIK_HD void sincos_cw(double x, double &sn, double &cs)
{
    const double INVPIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_2 = 6.07710050630396597660e-11;
    const double PIO2_2T = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = rint(x * INVPIO2);
    double t = fma_(-fn, PIO2_1, x);
    double w = fn * PIO2_2;
    double r = t - w;
    w = fma_(fn, PIO2_2T, -((t - r) - w));
    double y0 = r - w;
    double y1 = (r - y0) - w;

    double z = y0 * y0;
    double v = z * y0;
    double rs = fma_(z, fma_(z, fma_(z, fma_(z, S6, S5), S4), S3), S2);
    double sa = fma_(-v, rs, 0.5 * y1);
    double sb = fma_(z, sa, -y1);
    double ks = y0 - fma_(-v, S1, sb);
    double zz = z * z;
    double p1 = fma_(z, fma_(z, C3, C2), C1);
    double p2 = fma_(z, fma_(z, C6, C5), C4);
    double rc = fma_(zz * zz, p2, z * p1);
    double hz = 0.5 * z;
    double wc = 1.0 - hz;
    double kc = wc + (((1.0 - wc) - hz) + fma_(z, rc, -(y0 * y1)));

    int q = ((int)fn) & 3;
    double s_sel = (q & 1) ? kc : ks;
    double c_sel = (q & 1) ? ks : kc;
    sn = (q & 2) ? -s_sel : s_sel;
    cs = ((q == 1) || (q == 2)) ? -c_sel : c_sel;
}

// next representable double after b in the direction of `toward` (b != toward)
SEQIK_HD double next_toward(double b, double toward)
{
    union { double d; uint64_t u; } v;
    v.d = b;
    if (b == 0.0) {
        v.u = 1ull;  // smallest subnormal
        return (toward > 0.0) ? v.d : -v.d;
    }
    bool up = toward > b;
    bool positive = b > 0.0;
    if (up == positive) v.u += 1; else v.u -= 1;
    return v.d;
}

SEQIK_HD bool is_finite(double x) { return (x - x) == 0.0; }

// ---------------------------------------------------------------------------
// Wave-uniform fast paths.  make_strictly_feasible and the finite-difference step of _numdiff spend most of their
// instructions on what happens AT or BEYOND a bound (14 compares and 10 selects per strictly_feasible call); a point that
// lies strictly inside its bounds comes back unchanged.  Since a wavefront issues an instruction whenever ANY of its lanes
// needs it, the cheap test "is some lane of this wavefront not strictly inside?" (wave_any: one ballot) decides for the
// whole wavefront whether the full logic runs; when it does, every lane runs it and gets the value it would have got
// anyway.  Same values in both cases, so nothing changes in the bits.  SEQIK_FAST_PATHS=0 compiles the plain calls (A/B).
// ---------------------------------------------------------------------------

// out = in @ [R_axis(s, c) | (0, 0, tz)]   (one ikpy link frame appended on the right)
template <int AXIS>
SEQIK_HD void frame_mul_link(Frame &out, const Frame &in, double s, double c, double tz)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) out.t[i] = in.r[3 * i + 2] * tz + in.t[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a0 = in.r[3 * i], a1 = in.r[3 * i + 1], a2 = in.r[3 * i + 2];
        if constexpr (AXIS == AXIS_X) {
            out.r[3 * i] = a0;
            out.r[3 * i + 1] = fma_(a2, s, a1 * c);
            out.r[3 * i + 2] = fma_(a2, c, a1 * (-s));
        } else if constexpr (AXIS == AXIS_Y) {
            out.r[3 * i] = fma_(a2, -s, a0 * c);
            out.r[3 * i + 1] = a1;
            out.r[3 * i + 2] = fma_(a2, c, a0 * s);
        } else {
            out.r[3 * i] = fma_(a1, s, a0 * c);
            out.r[3 * i + 1] = fma_(a1, c, a0 * (-s));
            out.r[3 * i + 2] = a2;
        }
    }
}

SEQIK_HD void frame_identity(Frame &f)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) f.r[i] = 0.0;
    f.r[0] = f.r[4] = f.r[8] = 1.0;
    f.t[0] = f.t[1] = f.t[2] = 0.0;
}

// _lsq/common.py:make_strictly_feasible on one entry
SEQIK_HD double strictly_feasible(double x, double lb, double ub, double rstep)
{
    int active = 0;
    if (rstep == 0.0) {
        if (x <= lb) active = -1;
        if (x >= ub) active = 1;
    } else {
        double lower_dist = x - lb;
        double upper_dist = ub - x;
        double lower_threshold = rstep * fmax(1.0, fabs(lb));
        double upper_threshold = rstep * fmax(1.0, fabs(ub));
        if (is_finite(lb) && lower_dist <= fmin(upper_dist, lower_threshold)) active = -1;
        if (is_finite(ub) && upper_dist <= fmin(lower_dist, upper_threshold)) active = 1;
    }
    double xn = x;
    if (active == -1)
        xn = (rstep == 0.0) ? next_toward(lb, ub) : lb + rstep * fmax(1.0, fabs(lb));
    else if (active == 1)
        xn = (rstep == 0.0) ? next_toward(ub, lb) : ub - rstep * fmax(1.0, fabs(ub));
    if (xn < lb || xn > ub) xn = 0.5 * (lb + ub);
    return xn;
}

// make_strictly_feasible(rstep = 0) with the two replacement values precomputed (StageConst::lb_in / ub_in): 2 compares and
// 4 selects instead of strictly_feasible()'s 28 instructions.  Same result: for lb < ub (validated before any launch) the
// replacement values lie inside [lb, ub], so the final "outside the bounds -> midpoint" test of the general routine never
// fires, and a NaN stays a NaN in both.
SEQIK_HD double strictly_feasible0(double x, double lb, double ub, double lb_in, double ub_in)
{
    double xn = x;
    if (x <= lb) xn = lb_in;
    if (x >= ub) xn = ub_in;
    return xn;
}

// make_strictly_feasible(rstep = 1e-10) with its thresholds precomputed (StageConst::thr_lb / thr_ub): the same operations
// as strictly_feasible(x, lb, ub, 1e-10) minus the two max / multiply pairs
SEQIK_HD double strictly_feasible_thr(double x, double lb, double ub, double thr_lb, double thr_ub)
{
    const double lower_dist = x - lb;
    const double upper_dist = ub - x;
    int active = 0;
    if (is_finite(lb) && lower_dist <= fmin(upper_dist, thr_lb)) active = -1;
    if (is_finite(ub) && upper_dist <= fmin(lower_dist, thr_ub)) active = 1;
    double xn = x;
    if (active == -1) xn = lb + thr_lb;
    else if (active == 1) xn = ub - thr_ub;
    if (xn < lb || xn > ub) xn = 0.5 * (lb + ub);
    return xn;
}

// The same result from the per-leg constants of StageConst (lb_out / ub_out hold the replacement values with the
// midpoint rule applied; `a <= fmin(b, c)` is `a <= b && a <= c`; an infinite bound has a NaN threshold, which makes its
// comparison false as `isfinite(bound) &&` does).  side: -1 / +1 = moved to the lower / upper replacement value, 0 = x
// is returned as it is.  Only for x inside [lb, ub] (the general routine sends an x outside to the midpoint).
SEQIK_HD double strictly_feasible_pinned(double x, double lb, double ub, double thr_lb_g, double thr_ub_g, double lb_out,
                                         double ub_out, int &side)
{
    const double lower_dist = x - lb;
    const double upper_dist = ub - x;
    double xn = x;
    side = 0;
    if (lower_dist <= upper_dist && lower_dist <= thr_lb_g) { xn = lb_out; side = -1; }
    if (upper_dist <= lower_dist && upper_dist <= thr_ub_g) { xn = ub_out; side = 1; }
    return xn;
}

// ---------------------------------------------------------------------------
// Small fixed-size linear algebra.  Vectors have 2 slots; slot 1 is unused when NA == 1.
// ---------------------------------------------------------------------------
template <int NA>
SEQIK_HD double norm2v(const double *a)
{
    double acc = a[0] * a[0];
    if constexpr (NA == 2) acc = fma_(a[1], a[1], acc);
    return sqrt_(acc);
}

template <int NA>
SEQIK_HD double dot2v(const double *a, const double *b)
{
    double acc = a[0] * b[0];
    if constexpr (NA == 2) acc = fma_(a[1], b[1], acc);
    return acc;
}

SEQIK_HD double dot3(const double *a, const double *b)
{
    return fma_(a[2], b[2], fma_(a[1], b[1], a[0] * b[0]));
}

template <int NA>
SEQIK_HD void matvec32(const double Jh[3][2], const double *s, double *out)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double acc = Jh[k][0] * s[0];
        if constexpr (NA == 2) acc = fma_(Jh[k][1], s[1], acc);
        out[k] = acc;
    }
}

template <int NA>
SEQIK_HD double diag_form(const double *a, const double *diag, const double *b)
{
    double acc = (a[0] * diag[0]) * b[0];
    if constexpr (NA == 2) acc = fma_(a[1] * diag[1], b[1], acc);
    return acc;
}

// _lsq/common.py:CL_scaling_vector on one entry
SEQIK_HD void cl_scaling(double x, double g, double lb, double ub, double &v, double &dv)
{
    v = 1.0; dv = 0.0;
    if (g < 0 && is_finite(ub)) { v = ub - x; dv = -1.0; }
    if (g > 0 && is_finite(lb)) { v = x - lb; dv = 1.0; }
}

// the same with the finiteness of the bounds folded into the comparands (StageConst::gate_lb / gate_ub: a comparison with
// NaN is false, as `isfinite(bound)` would make the conjunction): two compares instead of two compares, two subtractions
// and two more compares
SEQIK_HD void cl_scaling_gated(double x, double g, double lb, double ub, double gate_lb, double gate_ub, double &v, double &dv)
{
    v = 1.0; dv = 0.0;
    if (g < gate_ub) { v = ub - x; dv = -1.0; }
    if (g > gate_lb) { v = x - lb; dv = 1.0; }
}

// _numdiff.py: the nominal 2-point step, before _adjust_scheme_to_bounds
SEQIK_HD double fd_step_nominal(double x)
{
    const double RSTEP = 1.4901161193847656e-08;
    double sign = (x >= 0.0) ? 1.0 : -1.0;
    return RSTEP * sign * fmax(1.0, fabs(x));
}

// does x + h leave [lb, ub]?  If not, _adjust_scheme_to_bounds('1-sided') leaves h alone: |h| <= the distance to the
// bound it points at, hence `fitting` holds and `violated` does not (fd_step below returns the nominal step).
SEQIK_HD bool fd_step_violates(double x, double h, double lb, double ub)
{
    const double xh = x + h;
    return (xh < lb) || (xh > ub);
}

// _numdiff.py: 2-point step with _adjust_scheme_to_bounds('1-sided')
SEQIK_HD double fd_step(double x, double lb, double ub)
{
    const double RSTEP = 1.4901161193847656e-08;
    double sign = (x >= 0.0) ? 1.0 : -1.0;
    double h = RSTEP * sign * fmax(1.0, fabs(x));
    double lower_dist = x - lb;
    double upper_dist = ub - x;
    double xh = x + h;
    bool violated = (xh < lb) || (xh > ub);
    bool fitting = fabs(h) <= fmax(lower_dist, upper_dist);
    if (violated && fitting) h = -h;
    else if (!fitting) h = (upper_dist >= lower_dist) ? upper_dist : -lower_dist;
    return h;
}

// SVD of the augmented matrix [[J_h], [diag(q)]] restricted to the active columns, by
// one-sided Jacobi (same sweep / threshold / ordering rules as oracle jacobi_svd).
// Out: s (descending), V (2x2), uf = U^T f.
template <int NA>
SEQIK_HD void svd_active(const double Jh[3][2], const double *q, const double *f, double *s, double V[2][2],
                         double *uf)
{
    if constexpr (NA == 1) {
        // 4 x 1: singular value = column norm, V = [1]
        double acc = Jh[0][0] * Jh[0][0];
        acc = fma_(Jh[1][0], Jh[1][0], acc);
        acc = fma_(Jh[2][0], Jh[2][0], acc);
        acc = fma_(q[0], q[0], acc);
        double sv0 = sqrt_(acc);
        double inv0 = (sv0 > 0.0) ? div_(1.0, sv0) : 0.0;
        double u0 = (Jh[0][0] * inv0) * f[0];
        u0 = fma_(Jh[1][0] * inv0, f[1], u0);
        u0 = fma_(Jh[2][0] * inv0, f[2], u0);
        s[0] = sv0; s[1] = 0.0; uf[0] = u0; uf[1] = 0.0;
        V[0][0] = 1.0; V[0][1] = 0.0; V[1][0] = 0.0; V[1][1] = 1.0;
    } else {
        const double TOL = 8.881784197001252e-16;
        double A[5][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) { A[k][0] = Jh[k][0]; A[k][1] = Jh[k][1]; }
        A[3][0] = q[0]; A[3][1] = 0.0;
        A[4][0] = 0.0;  A[4][1] = q[1];
        V[0][0] = 1.0; V[0][1] = 0.0; V[1][0] = 0.0; V[1][1] = 1.0;
        // squared column norms of the final A: the sums of the last (non-rotating) sweep are the
        // same sums the oracle recomputes for the singular values
        double alpha = 0.0, beta = 0.0;
        bool have_norms = false;
        for (int sweep = 0; sweep < 30; ++sweep) {
            double gamma = 0.0;
            alpha = 0.0; beta = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                alpha = fma_(A[i][0], A[i][0], alpha);
                beta = fma_(A[i][1], A[i][1], beta);
                gamma = fma_(A[i][0], A[i][1], gamma);
            }
            have_norms = true;
            if (gamma == 0.0) break;
            if (fabs(gamma) <= TOL * sqrt_(alpha * beta)) break;
            have_norms = false;
            double zeta = div_(beta - alpha, 2.0 * gamma);
            double t = div_(1.0, fabs(zeta) + sqrt_(fma_(zeta, zeta, 1.0)));
            if (zeta < 0.0) t = -t;
            double c = div_(1.0, sqrt_(fma_(t, t, 1.0)));
            double sn = c * t;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                double ap = A[i][0], aq = A[i][1];
                A[i][0] = fma_(c, ap, -(sn * aq));
                A[i][1] = fma_(sn, ap, c * aq);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                double vp = V[i][0], vq = V[i][1];
                V[i][0] = fma_(c, vp, -(sn * vq));
                V[i][1] = fma_(sn, vp, c * vq);
            }
        }
        if (!have_norms) {
            alpha = 0.0; beta = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) { alpha = fma_(A[i][0], A[i][0], alpha); beta = fma_(A[i][1], A[i][1], beta); }
        }
        double sv0 = sqrt_(alpha);
        double sv1 = sqrt_(beta);
        double inv0 = (sv0 > 0.0) ? div_(1.0, sv0) : 0.0;
        double inv1 = (sv1 > 0.0) ? div_(1.0, sv1) : 0.0;
        double u0 = (A[0][0] * inv0) * f[0], u1 = (A[0][1] * inv1) * f[0];
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            u0 = fma_(A[k][0] * inv0, f[k], u0);
            u1 = fma_(A[k][1] * inv1, f[k], u1);
        }
        // descending order (stable): swap only if the second is strictly larger
        if (sv0 < sv1) {
            s[0] = sv1; s[1] = sv0; uf[0] = u1; uf[1] = u0;
            double t0 = V[0][0], t1 = V[1][0];
            V[0][0] = V[0][1]; V[1][0] = V[1][1];
            V[0][1] = t0; V[1][1] = t1;
        } else {
            s[0] = sv0; s[1] = sv1; uf[0] = u0; uf[1] = u1;
        }
    }
}

// phi = ||suf / (s^2 + alpha)|| - Delta and the Newton ratio phi / phi' (oracle phi_and_ratio)
template <int NA>
SEQIK_HD void phi_and_ratio(double alpha, const double *suf, const double *s, double Delta, double &phi, double &ratio)
{
    double r0 = div_(1.0, fma_(s[0], s[0], alpha));
    double t0 = suf[0] * r0;
    double acc = (t0 * t0) * r0;
    double nn = t0 * t0;
    if constexpr (NA == 2) {
        double r1 = div_(1.0, fma_(s[1], s[1], alpha));
        double t1 = suf[1] * r1;
        acc = fma_(t1 * t1, r1, acc);
        nn = fma_(t1, t1, nn);
    }
    double p_norm = sqrt_(nn);
    phi = p_norm - Delta;
    ratio = div_(-(phi * p_norm), acc);
}

template <int NA>
SEQIK_HD void apply_V_neg(const double V[2][2], const double *tmp, double *p)
{
    {
        double acc = V[0][0] * tmp[0];
        if constexpr (NA == 2) acc = fma_(V[0][1], tmp[1], acc);
        p[0] = -acc;
    }
    if constexpr (NA == 2) {
        double acc = fma_(V[1][1], tmp[1], V[1][0] * tmp[0]);
        p[1] = -acc;
    } else {
        p[1] = 0.0;
    }
}

// _lsq/common.py:solve_lsq_trust_region on the active set (m = 3 residuals >= NA)
template <int NA, bool DEFICIENT>
SEQIK_HD void solve_lsq_trust_region(const double *uf, const double *s, const double V[2][2], double Delta,
                                     double &alpha_io, double *p)
{
    const double EPS = 2.220446049250313e-16;
    double suf[2] = {s[0] * uf[0], 0.0};
    if constexpr (NA == 2) suf[1] = s[1] * uf[1];
    double tmp[2] = {0.0, 0.0};
    bool full_rank = false;
    if constexpr (!DEFICIENT) {
        double threshold = EPS * 3 * s[0];
        full_rank = s[NA - 1] > threshold;
        if (full_rank) {
            tmp[0] = div_(uf[0], s[0]);
            if constexpr (NA == 2) tmp[1] = div_(uf[1], s[1]);
            apply_V_neg<NA>(V, tmp, p);
            if (norm2v<NA>(p) <= Delta) { alpha_io = 0.0; return; }
        }
    }
    const double inv_Delta = div_(1.0, Delta);
    double alpha_upper = norm2v<NA>(suf) * inv_Delta;
    double alpha_lower = 0.0;
    if (full_rank) {
        double phi, ratio;
        phi_and_ratio<NA>(0.0, suf, s, Delta, phi, ratio);
        alpha_lower = -ratio;
    }
    double alpha = alpha_io;
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt_(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt_(alpha_lower * alpha_upper));
        double phi, ratio;
        phi_and_ratio<NA>(alpha, suf, s, Delta, phi, ratio);
        if (phi < 0) alpha_upper = alpha;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio * inv_Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    tmp[0] = div_(suf[0], fma_(s[0], s[0], alpha));
    if constexpr (NA == 2) tmp[1] = div_(suf[1], fma_(s[1], s[1], alpha));
    apply_V_neg<NA>(V, tmp, p);
    double scale = div_(Delta, norm2v<NA>(p));
    p[0] = p[0] * scale;
    if constexpr (NA == 2) p[1] = p[1] * scale;
    alpha_io = alpha;
}

// ---------------------------------------------------------------------------
// Trust-region step of the 2-unknown stages (1-3) in closed form (mirrors oracle solve_tr_2x2 operation for
// operation).  scipy's solve_lsq_trust_region works on the SVD of A = [[J_h], [diag(sqrt_(diag_h))]] (5 x 2); all it
// needs is p(alpha) = -(A^T A + alpha I)^-1 J_h^T f, ||p||, phi'(alpha) = -p^T (A^T A + alpha I)^-1 p / ||p|| and, for
// its rank test, the extreme singular values.  A^T A = J_h^T J_h + diag(diag_h) is 2 x 2: the inverse by cofactors,
// the singular values from its eigenvalues (lambda_max = tr/2 + sqrt_(((a-c)/2)^2 + b^2), lambda_min = det /
// lambda_max).  Root search, bracket updates and the final rescaling are scipy's.  On the shipped recordings this
// follows the SVD-based variant to 6e-6 rad with the same evaluation counts on 99.8 % of the solves and the same
// distance to the reference; it removes the one-sided Jacobi sweeps (3 divisions + 3 square roots
// per sweep) and one of the two reciprocals of every root-search iteration: 22 % fewer instructions per step.
// ---------------------------------------------------------------------------
// q = (A^T A + alpha I)^-1 r with aa = a + alpha, cc = c + alpha already formed
SEQIK_HD void tr2_apply(double aa, double b, double cc, const double *r, double *q)
{
    double det = fma_(aa, cc, -(b * b));
    double inv = div_(1.0, det);
    q[0] = fma_(cc, r[0], -(b * r[1])) * inv;
    q[1] = fma_(aa, r[1], -(b * r[0])) * inv;
}

SEQIK_HD void tr2_phi(double aa, double b, double cc, const double *r, double Delta, double *pp, double &phi,
                      double &ratio)
{
    double q[2];
    tr2_apply(aa, b, cc, r, pp);
    double p_norm = sqrt_(fma_(pp[1], pp[1], pp[0] * pp[0]));
    tr2_apply(aa, b, cc, pp, q);
    double acc = fma_(pp[1], q[1], pp[0] * q[0]);
    phi = p_norm - Delta;
    ratio = div_(-(phi * p_norm), acc);
}

template <bool DEFICIENT>
SEQIK_HD void solve_tr_2x2(const double Jh[3][2], const double *diag_h, const double *f, double Delta,
                           double &alpha_io, double *p)
{
    double a = diag_h[0], b = 0.0, c = diag_h[1], r[2], pp[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a = fma_(Jh[k][0], Jh[k][0], a);
        b = fma_(Jh[k][0], Jh[k][1], b);
        c = fma_(Jh[k][1], Jh[k][1], c);
    }
    r[0] = fma_(Jh[2][0], f[2], fma_(Jh[1][0], f[1], Jh[0][0] * f[0]));
    r[1] = fma_(Jh[2][1], f[2], fma_(Jh[1][1], f[1], Jh[0][1] * f[0]));
    bool full_rank = false;
    if constexpr (!DEFICIENT) {
        double h = 0.5 * (a - c);
        double lmax = fma_(0.5, a + c, sqrt_(fma_(h, h, b * b)));
        double lmin = div_(fma_(a, c, -(b * b)), lmax);
        full_rank = lmin > 4.437342591868191e-31 * lmax;  // (3 eps)^2: s_min > eps * m * s_max
        if (full_rank) {
            tr2_apply(a + 0.0, b, c + 0.0, r, pp);
            if (sqrt_(fma_(pp[1], pp[1], pp[0] * pp[0])) <= Delta) { SEQIK_BLK_COUNT(CNT_GN_STEP); p[0] = -pp[0]; p[1] = -pp[1]; alpha_io = 0.0; return; }
        }
    }
    const double inv_Delta = div_(1.0, Delta);
    double alpha_upper = sqrt_(fma_(r[1], r[1], r[0] * r[0])) * inv_Delta;
    double alpha_lower = 0.0;
    double phi = 0.0, ratio = 0.0;
    if (full_rank) {
        tr2_phi(a + 0.0, b, c + 0.0, r, Delta, pp, phi, ratio);
        alpha_lower = -ratio;
    }
    double alpha = alpha_io;
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt_(alpha_lower * alpha_upper));
    // SHORTCUT of the m < n root search (mirrors oracle solve_tr_2x2, where it is derived): when the Gauss-Newton
    // step lies well inside the trust region scipy's ten iterations are ten resets alpha <- 0.001 alpha_upper plus
    // one Newton step from the last alpha, so only that alpha is evaluated; otherwise the verbatim loop runs.
    bool shortcut = false;
    if (!full_rank) {
        double au = alpha_upper, a_k = alpha;
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            if (a_k < 0.0 || a_k > au) a_k = fmax(0.001 * au, 0.0);
            au = a_k;
            if (it < 9) a_k = -1.0;
        }
        tr2_phi(a + a_k, b, c + a_k, r, Delta, pp, phi, ratio);
        if (phi < 0 && !(fabs(phi) < 0.01 * Delta)) {
            alpha = a_k - (phi + Delta) * ratio * inv_Delta;
            shortcut = true;
        }
    }
    // phi / ratio depend on alpha only through a + alpha and c + alpha.  In the rank-deficient stage 1 scipy's
    // search shrinks alpha a thousandfold per iteration; once it is below half an ulp of a and c the sums stop
    // changing and the evaluation (two divisions, a square root) would repeat itself bit for bit -- measured on the
    // benchmark data: in the last three of the ten iterations for 99.9 % of the solves -- so it is skipped.
    double aa_prev = __builtin_nan(""), cc_prev = __builtin_nan("");
    for (int it = 0; it < 10 && !shortcut; ++it) {
        SEQIK_BLK_COUNT(CNT_ROOT_ITER);
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt_(alpha_lower * alpha_upper));
        const double aa = a + alpha, cc = c + alpha;
        if (!(aa == aa_prev && cc == cc_prev)) {
            SEQIK_BLK_COUNT(CNT_ROOT_EVAL);
            tr2_phi(aa, b, cc, r, Delta, pp, phi, ratio);
            aa_prev = aa;
            cc_prev = cc;
        }
        if (phi < 0) alpha_upper = alpha;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio * inv_Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    tr2_apply(a + alpha, b, c + alpha, r, pp);
    double scale = div_(Delta, sqrt_(fma_(pp[1], pp[1], pp[0] * pp[0])));
    p[0] = -(pp[0] * scale);
    p[1] = -(pp[1] * scale);
    alpha_io = alpha;
}

template <int NA>
SEQIK_HD bool in_bounds2(const double *x, const double *lb, const double *ub)
{
    bool ok = (x[0] >= lb[0]) && (x[0] <= ub[0]);
    if constexpr (NA == 2) ok = ok && (x[1] >= lb[1]) && (x[1] <= ub[1]);
    return ok;
}

// _lsq/common.py:step_size_to_bound
template <int NA>
SEQIK_HD double step_size_to_bound(const double *x, const double *s, const double *lb, const double *ub, int *hits)
{
    const double INF = __builtin_huge_val();
    double steps[2] = {INF, INF};
    if (s[0] != 0.0) {
        double inv_s = div_(1.0, s[0]);
        steps[0] = fmax((lb[0] - x[0]) * inv_s, (ub[0] - x[0]) * inv_s);
    }
    if constexpr (NA == 2)
        if (s[1] != 0.0) {
            double inv_s = div_(1.0, s[1]);
            steps[1] = fmax((lb[1] - x[1]) * inv_s, (ub[1] - x[1]) * inv_s);
        }
    double min_step = fmin(steps[0], steps[1]);
    if (hits) {
        int sg0 = (s[0] > 0) - (s[0] < 0);
        int sg1 = (s[1] > 0) - (s[1] < 0);
        hits[0] = (steps[0] == min_step) ? sg0 : 0;
        hits[1] = (NA == 2 && steps[1] == min_step) ? sg1 : 0;
    }
    return min_step;
}

// _lsq/common.py:evaluate_quadratic
template <int NA>
SEQIK_HD double evaluate_quadratic(const double Jh[3][2], const double *g, const double *s, const double *diag)
{
    double Js[3];
    matvec32<NA>(Jh, s, Js);
    double q = dot3(Js, Js);
    q = q + diag_form<NA>(s, diag, s);
    double l = dot2v<NA>(s, g);
    return fma_(0.5, q, l);
}

// _lsq/common.py:minimize_quadratic_1d
SEQIK_HD double minimize_quadratic_1d(double a, double b, double lb, double ub, double c, double &y_out)
{
    double tbest = lb;
    double ybest = fma_(lb, fma_(a, lb, b), c);
    {
        double y = fma_(ub, fma_(a, ub, b), c);
        if (y < ybest) { ybest = y; tbest = ub; }
    }
    if (a != 0) {
        double extremum = div_(-0.5 * b, a);
        if (lb < extremum && extremum < ub) {
            double y = fma_(extremum, fma_(a, extremum, b), c);
            if (y < ybest) { ybest = y; tbest = extremum; }
        }
    }
    y_out = ybest;
    return tbest;
}

// _lsq/trf.py:select_step when x + p leaves the bounds (the in-bounds case is handled by
// the caller).  p, p_h are clobbered.  Not inlined: it is the cold path of stages 2-4.
template <int NA>
SEQIK_HD double select_step_reflective(
    const double *x, const double Jh[3][2], const double *diag_h, const double *g_h, double *p, double *p_h,
    const double *d, double Delta, const double *lb, const double *ub, double theta, double *step, double *step_h)
{
    const double INF = __builtin_huge_val();
    int hits[2];
    double p_stride = step_size_to_bound<NA>(x, p, lb, ub, hits);
    double r_h[2], r[2], x_on_bound[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        r_h[i] = p_h[i];
        if (hits[i] != 0) r_h[i] = r_h[i] * -1.0;
        r[i] = d[i] * r_h[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        p[i] = p[i] * p_stride;
        p_h[i] = p_h[i] * p_stride;
        x_on_bound[i] = x[i] + p[i];
    }
    // intersect_trust_region(p_h, r_h, Delta) -> positive root
    double to_tr;
    {
        double a = dot2v<NA>(r_h, r_h);
        double b = dot2v<NA>(p_h, r_h);
        double c = fma_(-Delta, Delta, dot2v<NA>(p_h, p_h));
        double dd = sqrt_(fma_(b, b, -(a * c)));
        double q = -(b + copysign(dd, b));
        double t1 = div_(q, a);
        double t2 = div_(c, q);
        to_tr = (t1 < t2) ? t2 : t1;
    }
    double to_bound = step_size_to_bound<NA>(x_on_bound, r, lb, ub, nullptr);
    double r_stride = fmin(to_bound, to_tr);
    double r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = div_((1 - theta) * p_stride, r_stride);
        r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        // build_quadratic_1d(J_h, g_h, r_h, s0=p_h, diag=diag_h)
        double v[3], u[3];
        matvec32<NA>(Jh, r_h, v);
        double a = dot3(v, v);
        a = a + diag_form<NA>(r_h, diag_h, r_h);
        a = a * 0.5;
        double b = dot2v<NA>(g_h, r_h);
        matvec32<NA>(Jh, p_h, u);
        b = b + dot3(u, v);
        double c = fma_(0.5, dot3(u, u), dot2v<NA>(g_h, p_h));
        b = b + diag_form<NA>(p_h, diag_h, r_h);
        c = fma_(0.5, diag_form<NA>(p_h, diag_h, p_h), c);
        r_stride = minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c, r_value);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            r_h[i] = r_h[i] * r_stride;
            r_h[i] = r_h[i] + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = INF;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { p[i] = p[i] * theta; p_h[i] = p_h[i] * theta; }
    double p_value = evaluate_quadratic<NA>(Jh, g_h, p_h, diag_h);

    double ag_h[2], ag[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    to_tr = div_(Delta, norm2v<NA>(ag_h));
    to_bound = step_size_to_bound<NA>(x, ag, lb, ub, nullptr);
    double ag_stride = (to_bound < to_tr) ? theta * to_bound : to_tr;
    double ag_value;
    {
        double v[3];
        matvec32<NA>(Jh, ag_h, v);
        double a = dot3(v, v);
        a = a + diag_form<NA>(ag_h, diag_h, ag_h);
        a = a * 0.5;
        double b = dot2v<NA>(g_h, ag_h);
        ag_stride = minimize_quadratic_1d(a, b, 0.0, ag_stride, 0.0, ag_value);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { ag_h[i] = ag_h[i] * ag_stride; ag[i] = ag[i] * ag_stride; }

    if (p_value < r_value && p_value < ag_value) {
        step[0] = p[0]; step[1] = p[1]; step_h[0] = p_h[0]; step_h[1] = p_h[1];
        return -p_value;
    } else if (r_value < p_value && r_value < ag_value) {
        step[0] = r[0]; step[1] = r[1]; step_h[0] = r_h[0]; step_h[1] = r_h[1];
        return -r_value;
    }
    step[0] = ag[0]; step[1] = ag[1]; step_h[0] = ag_h[0]; step_h[1] = ag_h[1];
    return -ag_value;
}

// ---------------------------------------------------------------------------
// The same three functions written for LATENCY instead of instruction count (run_stage<..., LAT>: the 256-register build
// of the stage pipeline, where a wavefront carries one or a few chains and its time is the length of its dependent
// chains, not its instruction count).  A basic block is the scheduler's horizon: a division inside `if (s[i] != 0)` or
// `if (a != 0)` cannot overlap with the next one.  Here every quotient is formed unconditionally and selected, and
// everything that does not depend on the strides comes first, in one block.  Same operations on the same operands wherever
// the original uses a value; the discarded quotients (zero direction, a == 0) are +-inf / NaN and never selected.
// Measured with the s_memtime stamps on config 4's serial walk: the reflective branch was 2 864 cycles per entry, 25 % of the
// critical stage-1 wavefront (scripts/block_cycles.py --recording --pipeline 2; EXPERIMENTS.md 4.11).  The lane-per-chain
// kernels keep the compact forms above: they are bound by issue, and these issue more.
// ---------------------------------------------------------------------------
template <int NA>
SEQIK_HD double step_size_to_bound_ilp(const double *x, const double *s, const double *lb, const double *ub, bool *hit)
{
    const double INF = __builtin_huge_val();
    const double inv0 = div_(1.0, s[0]);
    const double st0 = fmax((lb[0] - x[0]) * inv0, (ub[0] - x[0]) * inv0);
    double steps[2] = {(s[0] != 0.0) ? st0 : INF, INF};
    if constexpr (NA == 2) {
        const double inv1 = div_(1.0, s[1]);
        const double st1 = fmax((lb[1] - x[1]) * inv1, (ub[1] - x[1]) * inv1);
        steps[1] = (s[1] != 0.0) ? st1 : INF;
    }
    const double min_step = fmin(steps[0], steps[1]);
    if (hit) {
        hit[0] = (steps[0] == min_step) && (s[0] != 0.0);
        hit[1] = (NA == 2) && (steps[1] == min_step) && (s[1] != 0.0);
    }
    return min_step;
}

// minimize_quadratic_1d with the extremum -b / (2a) formed by the caller (unconditionally, early)
SEQIK_HD double minimize_quadratic_1d_ext(double a, double b, double lb, double ub, double c, double extremum, double &y_out)
{
    double tbest = lb;
    double ybest = fma_(lb, fma_(a, lb, b), c);
    {
        double y = fma_(ub, fma_(a, ub, b), c);
        if (y < ybest) { ybest = y; tbest = ub; }
    }
    if (a != 0 && lb < extremum && extremum < ub) {
        double y = fma_(extremum, fma_(a, extremum, b), c);
        if (y < ybest) { ybest = y; tbest = extremum; }
    }
    y_out = ybest;
    return tbest;
}

template <int NA>
SEQIK_HD double select_step_reflective_ilp(
    const double *x, const double Jh[3][2], const double *diag_h, const double *g_h, double *p, double *p_h,
    const double *d, double Delta, const double *lb, const double *ub, double theta, double *step, double *step_h)
{
    const double INF = __builtin_huge_val();
    // ---- stride-independent part of the three candidates ----------------------------------------------------------------
    bool hit[2];
    const double p_stride = step_size_to_bound_ilp<NA>(x, p, lb, ub, hit);
    double r_h[2], r[2], x_on_bound[2], pt_h[2], ag_h[2], ag[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        r_h[i] = hit[i] ? p_h[i] * -1.0 : p_h[i];
        r[i] = d[i] * r_h[i];
        p[i] = p[i] * p_stride;
        p_h[i] = p_h[i] * p_stride;
        x_on_bound[i] = x[i] + p[i];
        pt_h[i] = p_h[i] * theta;
        ag_h[i] = -g_h[i];
        ag[i] = d[i] * ag_h[i];
    }
    const double a_tr = dot2v<NA>(r_h, r_h);
    const double b_tr = dot2v<NA>(p_h, r_h);
    const double c_tr = fma_(-Delta, Delta, dot2v<NA>(p_h, p_h));
    const double to_bound_r = step_size_to_bound_ilp<NA>(x_on_bound, r, lb, ub, nullptr);
    double v[3], u[3], w[3];
    matvec32<NA>(Jh, r_h, v);
    matvec32<NA>(Jh, p_h, u);
    double a_q = dot3(v, v);
    a_q = a_q + diag_form<NA>(r_h, diag_h, r_h);
    a_q = a_q * 0.5;
    double b_q = dot2v<NA>(g_h, r_h);
    b_q = b_q + dot3(u, v);
    double c_q = fma_(0.5, dot3(u, u), dot2v<NA>(g_h, p_h));
    b_q = b_q + diag_form<NA>(p_h, diag_h, r_h);
    c_q = fma_(0.5, diag_form<NA>(p_h, diag_h, p_h), c_q);
    const double ext_q = div_(-0.5 * b_q, a_q);
    const double p_value = evaluate_quadratic<NA>(Jh, g_h, pt_h, diag_h);
    const double to_tr_ag = div_(Delta, norm2v<NA>(ag_h));
    const double to_bound_ag = step_size_to_bound_ilp<NA>(x, ag, lb, ub, nullptr);
    matvec32<NA>(Jh, ag_h, w);
    double a_ag = dot3(w, w);
    a_ag = a_ag + diag_form<NA>(ag_h, diag_h, ag_h);
    a_ag = a_ag * 0.5;
    const double b_ag = dot2v<NA>(g_h, ag_h);
    const double ext_ag = div_(-0.5 * b_ag, a_ag);
    // ---- strides ---------------------------------------------------------------------------------------------------------
    double to_tr;
    {
        double dd = sqrt_(fma_(b_tr, b_tr, -(a_tr * c_tr)));
        double q = -(b_tr + copysign(dd, b_tr));
        double t1 = div_(q, a_tr);
        double t2 = div_(c_tr, q);
        to_tr = (t1 < t2) ? t2 : t1;
    }
    double r_stride = fmin(to_bound_r, to_tr);
    const double stride_l = div_((1 - theta) * p_stride, r_stride);
    double r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = stride_l;
        r_stride_u = (r_stride == to_bound_r) ? theta * to_bound_r : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        r_stride = minimize_quadratic_1d_ext(a_q, b_q, r_stride_l, r_stride_u, c_q, ext_q, r_value);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            r_h[i] = r_h[i] * r_stride;
            r_h[i] = r_h[i] + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = INF;
    }
    double ag_stride = (to_bound_ag < to_tr_ag) ? theta * to_bound_ag : to_tr_ag;
    double ag_value;
    ag_stride = minimize_quadratic_1d_ext(a_ag, b_ag, 0.0, ag_stride, 0.0, ext_ag, ag_value);
#pragma unroll
    for (int i = 0; i < 2; ++i) { ag_h[i] = ag_h[i] * ag_stride; ag[i] = ag[i] * ag_stride; }
    const bool take_p = p_value < r_value && p_value < ag_value;
    const bool take_r = !take_p && r_value < p_value && r_value < ag_value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        step[i] = take_p ? p[i] * theta : (take_r ? r[i] : ag[i]);
        step_h[i] = take_p ? pt_h[i] : (take_r ? r_h[i] : ag_h[i]);
    }
    return -(take_p ? p_value : (take_r ? r_value : ag_value));
}

// ---------------------------------------------------------------------------
// Forward kinematics of the active part of a stage chain
// ---------------------------------------------------------------------------
template <int STAGE>
struct StageProblem {
    Frame pre;          // product of the links in front of the active ones
    double target[3];
    double tz_a, tz_b, tz_last;
};

// Frame after the active links for given sin/cos pairs.
template <int STAGE>
SEQIK_HD void frame_after_active(const StageProblem<STAGE> &P, double sa, double ca, double sb, double cb,
                                 Frame &after)
{
    using T = StageTraits<STAGE>;
    if constexpr (T::NA == 2) {
        Frame f1;
        frame_mul_link<T::AXIS_A>(f1, P.pre, sa, ca, P.tz_a);
        frame_mul_link<AXIS_Y>(after, f1, sb, cb, P.tz_b);
    } else {
        frame_mul_link<T::AXIS_A>(after, P.pre, sa, ca, P.tz_a);
    }
}

// End-effector residual for given sin/cos pairs of the active joints.
// pe (nullable): the end-effector position itself, f = pe - target
template <int STAGE>
SEQIK_HD void residual_sc(const StageProblem<STAGE> &P, double sa, double ca, double sb, double cb, double *f,
                          double *pe = nullptr)
{
    Frame after;
    frame_after_active<STAGE>(P, sa, ca, sb, cb, after);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double e = after.r[3 * i + 2] * P.tz_last + after.t[i];
        if (pe) pe[i] = e;
        f[i] = e - P.target[i];
    }
}

// End-effector residual at (xa, xb); also returns the sin/cos pairs.
template <int STAGE>
SEQIK_HD void eval_residual(const StageProblem<STAGE> &P, double xa, double xb, double *f,
                            double &sa, double &ca, double &sb, double &cb, double *pe = nullptr)
{
    sincos_cw(xa, sa, ca);
    if constexpr (StageTraits<STAGE>::NA == 2) sincos_cw(xb, sb, cb);
    else { sb = 0.0; cb = 1.0; }
    residual_sc<STAGE>(P, sa, ca, sb, cb, f, pe);
}

// 2-point finite-difference Jacobian of the active columns: J[k][j].  Column a perturbs
// link a and reuses link b's sin/cos, column b the other way round.
template <int STAGE>
SEQIK_HD void fd_jacobian(const StageProblem<STAGE> &P, const double *x, const double *f0, const double *lb,
                          const double *ub, double sa, double ca, double sb, double cb, double J[3][2])
{
    using T = StageTraits<STAGE>;
#if SEQIK_FAST_PATHS
    double h_a = fd_step_nominal(x[0]), h_b = (T::NA == 2) ? fd_step_nominal(x[1]) : 0.0;
    if (wave_any(fd_step_violates(x[0], h_a, lb[0], ub[0]) || (T::NA == 2 && fd_step_violates(x[1], h_b, lb[1], ub[1])))) {
        SEQIK_BLK_COUNT(CNT_FD_SLOW);
        h_a = fd_step(x[0], lb[0], ub[0]);
        if constexpr (T::NA == 2) h_b = fd_step(x[1], lb[1], ub[1]);
    }
#else
    double h_a = fd_step(x[0], lb[0], ub[0]), h_b = (T::NA == 2) ? fd_step(x[1], lb[1], ub[1]) : 0.0;
#endif
    {
        double h = h_a;
        double x1 = x[0] + h;
        double dx = x1 - x[0];
        double s1, c1, f1[3];
        sincos_cw(x1, s1, c1);
        residual_sc<STAGE>(P, s1, c1, sb, cb, f1);
        double inv_dx = div_(1.0, dx);
#pragma unroll
        for (int i = 0; i < 3; ++i) J[i][0] = (f1[i] - f0[i]) * inv_dx;
    }
    if constexpr (T::NA == 2) {
        double h = h_b;
        double x1 = x[1] + h;
        double dx = x1 - x[1];
        double s1, c1, f1[3];
        sincos_cw(x1, s1, c1);
        residual_sc<STAGE>(P, sa, ca, s1, c1, f1);
        double inv_dx = div_(1.0, dx);
#pragma unroll
        for (int i = 0; i < 3; ++i) J[i][1] = (f1[i] - f0[i]) * inv_dx;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) J[i][1] = 0.0;
    }
}

// ---------------------------------------------------------------------------
// Lane pairs (SPLIT instantiations of run_stage)
// ---------------------------------------------------------------------------
// A thin wave carries every chain on R >= 2 ADJACENT lanes (seqik_hip.hip, chain_of_wave_lane: replication): the
// replicas hold the same state and take the same branches.  A SPLIT stage uses lanes 2k and 2k + 1 of such a group as a
// pair: wherever a pass applies the SAME code to the two active joints one after the other -- the two finite-difference
// columns (perturbed sin / cos + chain product each) and the two sin / cos of the trial point -- the even lane takes
// joint a, the odd lane joint b, and they swap results with a DPP move (quad_perm [1, 0, 3, 2]: no LDS, no barrier).
// Each value is computed by the same operations on the same operands as in the unsplit code, so the bits are the same.
// Work that differs by candidate (select_step's three candidates, the root loop) cannot be split this way: lanes of
// one wave that run different code run it one after the other.
SEQIK_HD double pair_swap(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
#else
    return v;
#endif
}

SEQIK_HD bool pair_or(bool v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return v || (__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true) != 0);
#else
    return v;
#endif
}

SEQIK_HD bool pair_is_odd()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (threadIdx.x & 1) != 0;
#else
    return false;
#endif
}

// fd_jacobian with one column per lane of a pair (stages with two active joints)
template <int STAGE>
SEQIK_HD void fd_jacobian_pair(const StageProblem<STAGE> &P, const double *x, const double *f0, const double *lb,
                               const double *ub, double sa, double ca, double sb, double cb, bool odd, double J[3][2])
{
    static_assert(StageTraits<STAGE>::NA == 2, "one active joint: nothing to split");
    const double xj = odd ? x[1] : x[0];
#if SEQIK_FAST_PATHS
    double h = fd_step_nominal(xj);
    if (wave_any(fd_step_violates(xj, h, odd ? lb[1] : lb[0], odd ? ub[1] : ub[0]))) {
        SEQIK_BLK_COUNT(CNT_FD_SLOW);
        h = fd_step(xj, odd ? lb[1] : lb[0], odd ? ub[1] : ub[0]);
    }
#else
    double h = fd_step(xj, odd ? lb[1] : lb[0], odd ? ub[1] : ub[0]);
#endif
    double x1 = xj + h;
    double dx = x1 - xj;
    double s1, c1, f1[3];
    sincos_cw(x1, s1, c1);
    residual_sc<STAGE>(P, odd ? sa : s1, odd ? ca : c1, odd ? s1 : sb, odd ? c1 : cb, f1);
    double inv_dx = div_(1.0, dx);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double mine = (f1[i] - f0[i]) * inv_dx;
        const double other = pair_swap(mine);
        J[i][0] = odd ? other : mine;
        J[i][1] = odd ? mine : other;
    }
}

// eval_residual with one sin / cos per lane of a pair
template <int STAGE>
SEQIK_HD void eval_residual_pair(const StageProblem<STAGE> &P, double xa, double xb, bool odd, double *f,
                                 double &sa, double &ca, double &sb, double &cb, double *pe = nullptr)
{
    static_assert(StageTraits<STAGE>::NA == 2, "one active joint: nothing to split");
    double s1, c1;
    sincos_cw(odd ? xb : xa, s1, c1);
    const double s2 = pair_swap(s1), c2 = pair_swap(c1);
    sa = odd ? s2 : s1; ca = odd ? c2 : c1;
    sb = odd ? s1 : s2; cb = odd ? c1 : c2;
    residual_sc<STAGE>(P, sa, ca, sb, cb, f, pe);
}

// ---------------------------------------------------------------------------
// Stage driver
// ---------------------------------------------------------------------------
// Stage pipeline (PIPED instantiations, seqik_hip.hip "Stage pipeline"): the four stages of a chain run in four
// wavefronts of one workgroup, stage k + 1 following stage k one frame behind.  The hand-off frame (12 doubles) goes
// through a small ring in LDS instead of the HBM workspace; two counters per stage boundary and lane order the
// accesses:  produced = frames the upstream stage has published, consumed = frames the downstream stage has taken.
//   upstream, before starting frame i:   wait until consumed + PIPE_DEPTH > i   (slot i % PIPE_DEPTH is free again)
//   upstream, frame i solved:            write slot, then produced = i + 1 (release)
//   downstream, before starting frame i: wait until produced > i (acquire), read slot, then consumed = i + 1
// "wait" = the lane sits out the pass (flat state machine: the other lanes of the wave carry on); a wave whose lanes
// all wait sleeps a few cycles.  No cycle of waits exists: stage 1 only waits for stage 2 to free a slot, stage 4
// waits for nobody downstream, so the oldest unfinished frame can always advance; every wave leaves its loop when its
// lanes have done all frames.
constexpr int PIPE_DEPTH = 2;
#ifndef SEQIK_PIPE_SPIN_LIMIT
#define SEQIK_PIPE_SPIN_LIMIT (1 << 24)  // watchdog: ~1 s of waiting on one frame (a frame takes ~10-100 us)
#endif
constexpr int PIPE_SPIN_LIMIT = SEQIK_PIPE_SPIN_LIMIT;  // (-DSEQIK_PIPE_SPIN_LIMIT=1: the diagnostic build the watchdog test trips)
struct PipeLane {
    // element k of slot j of this lane at ring[(j * 12 + k) * lane_stride]
    double *ring_in, *ring_out;     // LDS; null for the first / last stage
    int *produced_in, *consumed_in; // counters of the boundary in front of this stage
    int *produced_out, *consumed_out;
    int lane_stride;                // lanes interleaved in LDS (bank-conflict free)
    int base;                       // frames this lane's four waves have put through the ring before this chain
                                    // (a lane that solves several chains in turn keeps counting)
    int32_t *fault;                 // nullable: host-visible word the watchdog reports to (seqik_hip.hip "Device faults")
};

SEQIK_HD int pipe_load(const int *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    return *p;
#endif
}

SEQIK_HD void pipe_store(int *p, int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p = v;
#endif
}

struct ChainIO {
    const double *pose;     // key point (row, t) of this chain at pose + row * pose_row + t * pose_frame (x, y, z)
    int64_t pose_row, pose_frame;
    double *angles;         // angle (dof, t) at angles + dof * ang_dof + t * ang_frame
    int64_t ang_dof, ang_frame;  // DOF order yaw, pitch, roll, CTr_pitch, CTr_roll, FTi, TiTa
    double *fk;             // nullable [n_frames][9][3]
    int32_t *status;        // nullable [n_frames][4]
    int32_t *nfev;          // nullable [n_frames][4]
    const double *init;     // nullable [7]: warm start of frame 0 (the joint angles of the frame that
                            // precedes this chain's first frame) instead of the stage seeds
    double *frames;         // workspace [n_frames][12]: the frame after the active links at the solution of
                            // stage k (rotation 9 + translation 3) = prefix frame of stage k + 1
    int64_t n_frames;       // frames [0, n_frames) are solved (CHUNKED: [t_begin, n_frames))
    // --- CHUNKED instantiations only (a piece of a recording, seqik_hip.hip "frame chunks") -----------------
    int64_t t_begin;        // first frame solved; the workspace is indexed by t - t_begin
    int64_t t_store;        // first frame whose angles / FK / diagnostics are stored: [t_begin, t_store) is a
                            // run-in ("halo") that only brings the warm start close to the serial trajectory
    int64_t init_stride;    // init is read as init[dof * init_stride] (lets it point into an angle array)
    double *start_state;    // nullable [7]: receives the angles of frame t_store - 1 (the state the run-in reached
                            // = the warm start the stored frames were computed from) when t_store > t_begin
    // --- PIPED instantiations only ----------------------------------------------------------------------------
    PipeLane pipe;
    // --- QUEUE instantiations only (chain queue of the fused kernel: seqik_fused_queue_kernel in seqik_hip.hip) ---------------
    // (pose / angles / fk / init / frames then point at the FIRST chain of the wavefront's pool: the same in every lane)
    int32_t q_seq;          // the chain of the pool this lane starts on (0 .. 63)
    int32_t q_next;         // next chain of the POOL that no lane has taken yet (the same value in every lane; 64 at the start)
    int32_t q_end;          // chains in the pool
    uint32_t q_pose, q_ang, q_fk, q_init, q_frames;  // element strides from one chain of the pool to the next (32 bits: one
                            // v_mad_u64_u32 per pointer; the launcher refuses the queue when a stride does not fit)
};

// Prefix frame of STAGE from the angles of the earlier stages: the "fixed" links of
// kinematic_chain.py:215-241, 276-316, 353-401, multiplied in link order.
template <int STAGE>
SEQIK_HD void build_prefix(Frame &pre, const LegConst &lc, const double *ang, int64_t ang_dof, double *coxa_end)
{
    frame_identity(pre);
    if constexpr (STAGE >= 2) {
        double sn, cs;
        Frame tmp;
        sincos_cw(ang[0], sn, cs);
        frame_mul_link<AXIS_X>(tmp, pre, sn, cs, lc.st[0].tz_a);
        sincos_cw(ang[ang_dof], sn, cs);
        frame_mul_link<AXIS_Y>(pre, tmp, sn, cs, lc.st[0].tz_b);
    }
    if constexpr (STAGE >= 3) {
        double sn, cs;
        Frame tmp;
        sincos_cw(ang[2 * ang_dof], sn, cs);
        frame_mul_link<AXIS_Z>(tmp, pre, sn, cs, lc.st[1].tz_a);
        sincos_cw(ang[3 * ang_dof], sn, cs);
        frame_mul_link<AXIS_Y>(pre, tmp, sn, cs, lc.st[1].tz_b);
        if (coxa_end) { coxa_end[0] = pre.t[0]; coxa_end[1] = pre.t[1]; coxa_end[2] = pre.t[2]; }
    }
    if constexpr (STAGE >= 4) {
        double sn, cs;
        Frame tmp;
        sincos_cw(ang[4 * ang_dof], sn, cs);
        frame_mul_link<AXIS_Z>(tmp, pre, sn, cs, lc.st[2].tz_a);
        sincos_cw(ang[5 * ang_dof], sn, cs);
        frame_mul_link<AXIS_Y>(pre, tmp, sn, cs, lc.st[2].tz_b);
    }
}

// Runs stage STAGE over all frames of one chain.
//   WANT_FK   : also write this stage's rows of the stage-4 forward kinematics + origin
//               (leg_inverse_kinematics.py:279-282): stage 2 -> rows 4, 5 (coxa end), stage 3 ->
//               row 6 (femur end), stage 4 -> rows 0-3 (origin), 7 (tibia end), 8 (claw).
//   FROM_ANGLES : the prefix frame is rebuilt from the stored angles of the earlier stages
//               (first stage of a run that starts after stage 1); otherwise it is read from the
//               workspace where the previous stage's kernel left it.  With WANT_FK such a
//               first stage also writes the FK rows of the stages that are not run.
//   HANDOFF   : leave the frame after the active links in the workspace for the next stage.
//   WANT_DIAG : also produce scipy's status / nfev (one extra Jacobian per solve: scipy
//               re-evaluates it after the last accepted step and may overwrite the status
//               with 1 = gtol).
//   CHUNKED   : the lane solves frames [io.t_begin, io.n_frames) of its chain and stores only those from
//               io.t_store on (ChainIO); false = the whole chain from frame 0, everything stored.
//   PIPED     : the hand-off goes through the LDS ring of io.pipe (stage pipeline, see PipeLane) instead of io.frames.
//   SPLIT     : the lane and its neighbour lane ^ 1 carry the same chain and share the work of a pass ("Lane pairs").
//   LAT       : the LATENCY build, set by the 256-register instantiations of the stage pipeline (small grids: serial walks,
//               short recordings, where a wavefront carries one or a few chains): (i) Jacobian, gradient and scaling are
//               kept across passes and not re-derived after a rejected trial when no lane of the wavefront has moved (see
//               jac_valid below; the 168-register build would spill the 14 doubles); (ii) the reflective select_step and
//               the post-trial block in their branch-free forms (select_step_reflective_ilp).  Same values.
//   QUEUE     : CHAIN QUEUE (round 6).  The wavefront owns a POOL of chains of one leg (io.q_next .. io.q_end, sequences; the
//               lanes start on the pool's first 64) and a lane that has finished the last frame of its chain takes the next
//               chain nobody has taken yet instead of idling until the slowest lane of the wavefront is done: the tail of
//               a wavefront -- ever fewer active lanes issuing whole passes -- is paid once per pool instead of once per 64
//               chains.  Which lane walks a chain does not enter its arithmetic: same bits.  The hand-out needs no atomic and no
//               LDS: the lanes that finish in the same pass number themselves by a ballot (lane_rank_in), every lane of the
//               wavefront advances its copy of q_next by the same count.  Costs one ballot and one scalar branch per pass.
template <int STAGE, bool WANT_FK, bool WANT_DIAG, bool FROM_ANGLES, bool HANDOFF, bool CHUNKED = false, bool PIPED = false,
          bool SPLIT = false, bool LAT = false, bool QUEUE = false>
SEQIK_HD void run_stage(const LegConst &lc, const ChainIO &io_arg)
{
    static_assert(!QUEUE || (!CHUNKED && !PIPED && !SPLIT && !LAT && !WANT_DIAG && !FROM_ANGLES),
                  "the chain queue belongs to the fused lane-per-chain kernel");
    // QUEUE: io_arg points at the first chain of the wavefront's pool (wave-uniform: scalar registers); the lane's current chain
    // is q_seq, and its pointers are formed from that where a frame starts and where it ends -- they are not kept in vector
    // registers across the body of a pass (the first version of the queue, which re-based per-lane pointers, spilled 11 more
    // registers than the plain kernel and gave back what the queue saved)
    ChainIO io_q;
    if constexpr (QUEUE) io_q = io_arg;
    const ChainIO &io = QUEUE ? io_q : io_arg;
    int32_t q_next = QUEUE ? io_arg.q_next : 0, q_seq = QUEUE ? io_arg.q_seq : 0;
    auto queue_point_at = [&](int32_t seq) {
        if constexpr (QUEUE) {
            const uint32_t u = (uint32_t)seq;
            io_q.pose = io_arg.pose + (uint64_t)u * io_arg.q_pose;
            io_q.angles = io_arg.angles + (uint64_t)u * io_arg.q_ang;
            if constexpr (WANT_FK) io_q.fk = io_arg.fk + (uint64_t)u * io_arg.q_fk;   // (WANT_FK: the launcher passes a buffer)
            io_q.frames = io_arg.frames + (uint64_t)u * io_arg.q_frames;
        }
    };
    queue_point_at(q_seq);
    static_assert(!LAT || (PIPED && !WANT_DIAG), "LAT is a stage-pipeline option");
    constexpr bool PAIRED = SPLIT && StageTraits<STAGE>::NA == 2;
    const bool odd = PAIRED && pair_is_odd();
    static_assert(!(PIPED && FROM_ANGLES), "the stage pipeline starts at stage 1");
    static_assert(STAGE > 1 || !FROM_ANGLES, "stage 1 has no prefix");
    static_assert(!(CHUNKED && FROM_ANGLES), "frame chunks start at stage 1 (the run-in has no stored angles)");
    static_assert(STAGE < 4 || !HANDOFF, "stage 4 is the last one");
    using T = StageTraits<STAGE>;
    constexpr int NA = T::NA;
    constexpr int DOF0 = 2 * (STAGE - 1);  // angle columns written: DOF0 (and DOF0 + 1)
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    const StageConst &sc = lc.st[STAGE - 1];
    const double *lb = sc.lb;
    const double *ub = sc.ub;
    const int max_nfev = sc.max_nfev;

    StageProblem<STAGE> P;
    P.tz_a = sc.tz_a; P.tz_b = sc.tz_b; P.tz_last = sc.tz_last;
    frame_identity(P.pre);
    P.target[0] = P.target[1] = P.target[2] = 0.0;

    // x carries the solution from frame to frame: it is the warm start of the next solve
    double x[2] = {sc.seed[0], (NA == 2) ? sc.seed[1] : 0.0}, f[3] = {0.0, 0.0, 0.0};
    if (io.init) {
        const int64_t is = CHUNKED ? io.init_stride : 1;
        const double *init_c = QUEUE ? io_arg.init + (uint64_t)(uint32_t)q_seq * io_arg.q_init : io.init;
        x[0] = init_c[DOF0 * is];
        if constexpr (NA == 2) x[1] = init_c[(DOF0 + 1) * is];
    }
    double cost = 0.0, Delta = 0.0, alpha = 0.0;
    double sa = 0.0, ca = 1.0, sb = 0.0, cb = 1.0;  // sin/cos of the active joints at x
    int nfev = 0, status = STATUS_NONE;
    bool first_pass = true, new_solve = true;
    double coxa_end[3] = {0.0, 0.0, 0.0};  // stage 4 + FK only
    const int64_t t_first = CHUNKED ? io.t_begin : 0;
    int64_t t = t_first;
    // (Tried and dropped: per-frame addresses as running pointers advanced with t instead of `base + t * stride` with
    // run-time 64-bit strides -- 16 fewer vector instructions per stage, all v_mul_lo_u32 / v_mad_u64_u32 of the frame-start
    // and frame-end blocks: 11.971 -> 11.967 ms per benchmark step.  Those blocks wait for memory; their arithmetic is free.)
    int pipe_spins = 0;  // PIPED: consecutive passes this lane sat out (watchdog only)
    double pe[3] = {0.0, 0.0, 0.0};  // stage 1: end-effector position at x (see the new-solve block)
    bool have_pe = false;
    // Jacobian, gradient, Coleman-Li scaling and its square root at the current x.  scipy leaves its inner loop
    // (`while actual_reduction <= 0`) only on an accepted step, so after a REJECTED trial these are what they were; the
    // pass loop re-derives them anyway (bit-identical values) because in a full wavefront some lane has always just
    // accepted.  On the stage pipeline a wavefront carries one or a few chains: when NO lane of it has moved since they
    // were formed (jac_valid on every lane) the finite differences, the gradient, the scaling and the two square roots are
    // skipped for the whole wavefront.  Stage 1 -- the critical stage of a serial walk -- rejects 32 % (RF) / 10 % (LF) of its
    // trials on the shipped recording (serial walk of config 4: 122.8 -> 118.2 ms).  LAT instantiations only; same values
    // either way.
    double J[3][2], g[2], v[2], dv[2], d[2];
    bool jac_valid = false;

    // (Tried and dropped: loading the NEXT frame's inputs (key point, origin, prefix frame: 18 doubles) into staging
    // registers at the top of every pass and moving them into place when a lane finishes its frame, so that no lane
    // waits for the memory round trip of a frame start -- every pass has a few lanes starting one.  The generated code
    // did what was intended (loads at the top, `s_waitcnt vmcnt(8)` behind the finished frame's stores), needs 36 more
    // registers, i.e. two waves per SIMD: 15.1 ms per benchmark step against 15.2 without it at two waves, 13.7 at the
    // usual three (20.8 with the staging registers spilled at three).  The frame-start loads are not what waves wait for.
    // The same for the key point and origin alone (six doubles) on the stage pipeline's thin wavefronts: serial walk of
    // the shipped 6000-frame recording 144 -> 174 ms -- vector memory returns in order, so the spill reloads of every
    // pass then wait for that pass's look-ahead loads instead of one round trip per frame; on the 256-register build, which
    // has no spills, still 140 -> 144 ms: six loads and their wait in every pass cost more than one round trip per frame.)
    // (Tried and dropped: letting finished lanes wait until 4 / 8 / 16 of them have gathered, so that the wavefront goes
    // through the end-of-frame and start-of-frame blocks -- ~150-250 instructions that it otherwise executes in almost
    // every pass for two or three lanes -- less often: 14.05 / 13.79 / 13.90 ms per benchmark step against 13.83; what
    // the blocks save is spent on the additional passes of the waiting lanes.)
    // (Tried and dropped: keeping finished lanes in the loop and letting all 64 lanes execute a burst of dummy
    // multiply-adds per pass while fewer than 16 lanes are still working, to keep the wave out of the slow sparse-EXEC
    // mode of scripts/microbench/exec_*.hip during the end-of-stage tail: 14.2 -> 15.0 ms per benchmark step.)
    SEQIK_BLK_DECL
    while (t < io.n_frames) {
        SEQIK_BLK_PASS();
        SEQIK_BLK_END_OF(BLK_LOOP);
        if constexpr (PIPED) {
            // may this lane start frame t?  (only asked at a frame boundary; a lane in the middle of a solve runs on)
            bool stall = false;
            if (new_solve) {
                const int i = io.pipe.base + (int)(t - t_first);
                if constexpr (STAGE > 1) stall = pipe_load(io.pipe.produced_in) <= i;
                if constexpr (HANDOFF) stall = stall || (pipe_load(io.pipe.consumed_out) + PIPE_DEPTH <= i);
            }
            // a pair starts a frame together (each lane has its own ring and counters, written by its own replica in
            // the neighbouring waves in the same instruction; this makes the pair independent of how LDS orders them)
            if constexpr (SPLIT) stall = pair_or(stall);
#if defined(__HIP_DEVICE_COMPILE__)
            if (__ballot(stall) == __ballot(1)) __builtin_amdgcn_s_sleep(2);  // every lane still in the loop waits
#endif
            if (stall) {
                if (++pipe_spins > PIPE_SPIN_LIMIT) {
                    // Cannot happen by construction (see PipeLane); should it ever, never hang the GPU: mark this
                    // lane's remaining results NaN, release the neighbours, leave -- and say so: the word behind
                    // io.pipe.fault lives in host memory, every host entry point turns it into SEQIK_ERR_HIP.
                    if (io.pipe.fault) *(volatile int32_t *)io.pipe.fault = STAGE;
                    const double nan = __builtin_nan("");
                    for (int64_t tt = t; tt < io.n_frames; ++tt) {
                        if (CHUNKED && tt < io.t_store) continue;
                        io.angles[tt * io.ang_frame + DOF0 * io.ang_dof] = nan;
                        if constexpr (NA == 2) io.angles[tt * io.ang_frame + (DOF0 + 1) * io.ang_dof] = nan;
                    }
                    if constexpr (HANDOFF) pipe_store(io.pipe.produced_out, 0x3fffffff);
                    if constexpr (STAGE > 1) pipe_store(io.pipe.consumed_in, 0x3fffffff);
                    break;
                }
                continue;
            }
            pipe_spins = 0;
        }
        if constexpr (PIPED) SEQIK_BLK_END_OF(BLK_PIPE_WAIT);
        if (new_solve) {
            SEQIK_BLK_COUNT(CNT_NEW_SOLVE);
            queue_point_at(q_seq);
            const double *org = io.pose + t * io.pose_frame;
            const double *kp = org + STAGE * io.pose_row;
            if constexpr (STAGE > 1) {
                if constexpr (FROM_ANGLES) {
                    build_prefix<STAGE>(P.pre, lc, io.angles + t * io.ang_frame, io.ang_dof, WANT_FK ? coxa_end : nullptr);
                } else if constexpr (PIPED) {
                    const int fi = io.pipe.base + (int)(t - t_first);
                    const double *w = io.pipe.ring_in + (fi % PIPE_DEPTH) * 12 * io.pipe.lane_stride;
#pragma unroll
                    for (int i = 0; i < 9; ++i) P.pre.r[i] = w[i * io.pipe.lane_stride];
#pragma unroll
                    for (int i = 0; i < 3; ++i) P.pre.t[i] = w[(9 + i) * io.pipe.lane_stride];
                    pipe_store(io.pipe.consumed_in, fi + 1);
                } else {
                    const double *w = io.frames + (t - t_first) * 12;
#pragma unroll
                    for (int i = 0; i < 9; ++i) P.pre.r[i] = w[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) P.pre.t[i] = w[9 + i];
                }
            }
            if (lc.aff.enabled) {
                // fused AlignPose.align_leg, then target = aligned key point - aligned origin
                // (three separately rounded operations, as numpy evaluates them)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    double al = (kp[a] - lc.aff.fixed_coxa[a]) * lc.aff.scale + lc.aff.template_coxa[a];
                    P.target[a] = al - lc.aff.template_coxa[a];
                }
            } else {
                P.target[0] = kp[0] - org[0];
                P.target[1] = kp[1] - org[1];
                P.target[2] = kp[2] - org[2];
            }
#if SEQIK_FAST_PATHS
            // scipy makes the start point strictly feasible (make_strictly_feasible, rstep = 1e-10) and evaluates the
            // residual there.  Three cases, same values in each:
            //  * first frame of the chain (have_pe false, every lane of the wave at once): the general routine;
            //  * the warm start lies further than the threshold from its limits: it is not moved, its sin / cos are the
            //    ones the last accepted trial (or start evaluation) of the previous frame left in sa .. cb, and in stage 1,
            //    which has no frame-dependent prefix, so is the chain position pe;
            //  * a joint the previous frame left pinned on a limit: it is moved to a constant of the leg whose sin / cos
            //    the host computed (StageConst::sc_lb / sc_ub) -- selects instead of sin / cos evaluations.  On the
            //    benchmark's iid poses some lane of a wavefront is in this case in 30-60 % of its passes
            //    (profiles/r03_block_entries_*.json).
            if (!have_pe) {
                SEQIK_BLK_COUNT(CNT_START_EVAL);
                x[0] = strictly_feasible_thr(x[0], lb[0], ub[0], sc.thr_lb[0], sc.thr_ub[0]);
                if constexpr (NA == 2) x[1] = strictly_feasible_thr(x[1], lb[1], ub[1], sc.thr_lb[1], sc.thr_ub[1]);
                eval_residual<STAGE>(P, x[0], x[1], f, sa, ca, sb, cb, (STAGE == 1) ? pe : nullptr);
                have_pe = true;  // (stages 2-4 use the flag for "sa .. cb belong to x")
            } else {
                // (x is a result of this solver here: inside its limits)
                double xs0 = x[0], xs1 = x[1];
                int side0 = 0, side1 = 0;
                if (wave_any(!((x[0] - lb[0] > sc.thr_lb[0]) && (ub[0] - x[0] > sc.thr_ub[0]) &&
                               (NA == 1 || ((x[1] - lb[1] > sc.thr_lb[1]) && (ub[1] - x[1] > sc.thr_ub[1])))))) {
                    SEQIK_BLK_COUNT(CNT_FEASIBLE_SLOW);
                    xs0 = strictly_feasible_pinned(x[0], lb[0], ub[0], sc.thr_lb_g[0], sc.thr_ub_g[0], sc.lb_out[0], sc.ub_out[0], side0);
                    if constexpr (NA == 2)
                        xs1 = strictly_feasible_pinned(x[1], lb[1], ub[1], sc.thr_lb_g[1], sc.thr_ub_g[1], sc.lb_out[1], sc.ub_out[1], side1);
                }
                const bool moved0 = xs0 != x[0], moved1 = (NA == 2) && xs1 != x[1];
                if (moved0) { sa = (side0 < 0) ? sc.sc_lb[0][0] : sc.sc_ub[0][0]; ca = (side0 < 0) ? sc.sc_lb[0][1] : sc.sc_ub[0][1]; x[0] = xs0; }
                if constexpr (NA == 2)
                    if (moved1) { sb = (side1 < 0) ? sc.sc_lb[1][0] : sc.sc_ub[1][0]; cb = (side1 < 0) ? sc.sc_lb[1][1] : sc.sc_ub[1][1]; x[1] = xs1; }
                if constexpr (STAGE == 1) {
                    if (moved0 || moved1) {
                        SEQIK_BLK_COUNT(CNT_START_EVAL);
                        residual_sc<STAGE>(P, sa, ca, sb, cb, f, pe);
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i) f[i] = pe[i] - P.target[i];
                    }
                } else {
                    residual_sc<STAGE>(P, sa, ca, sb, cb, f);  // the prefix frame is another one in every frame
                }
            }
#else
            {
                const double xs0 = strictly_feasible(x[0], lb[0], ub[0], 1e-10);
                const double xs1 = (NA == 2) ? strictly_feasible(x[1], lb[1], ub[1], 1e-10) : x[1];
                if (have_pe && xs0 == x[0] && xs1 == x[1]) {
                    if constexpr (STAGE == 1) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) f[i] = pe[i] - P.target[i];
                    } else {
                        residual_sc<STAGE>(P, sa, ca, sb, cb, f);
                    }
                } else {
                    x[0] = xs0;
                    if constexpr (NA == 2) x[1] = xs1;
                    eval_residual<STAGE>(P, x[0], x[1], f, sa, ca, sb, cb, (STAGE == 1) ? pe : nullptr);
                    have_pe = true;
                }
            }
#endif
            cost = 0.5 * dot3(f, f);
            nfev = 1;
            alpha = 0.0;
            status = STATUS_NONE;
            first_pass = true;
            new_solve = false;
            jac_valid = false;
        }

        bool finished = false;
        SEQIK_BLK_END_OF(BLK_NEW_SOLVE);
        if (WANT_DIAG || status == STATUS_NONE) {
            SEQIK_BLK_COUNT(CNT_BODY);
#if SEQIK_BLOCK_CYCLES && defined(__HIP_DEVICE_COMPILE__)
            if (__popcll(__ballot(1)) < 16) SEQIK_BLK_COUNT(CNT_BODY_LT16);   // passes in which a wave is "thin"
            if (__popcll(__ballot(1)) < 32) SEQIK_BLK_COUNT(CNT_BODY_LT32);
#endif
            // ---- top of scipy's outer loop: J, g, scaling, gtol test --------------------
            bool derive = true;
            if constexpr (LAT) derive = wave_any(!jac_valid);   // wave-uniform: nobody moved since the last pass
            if (derive) {
            if constexpr (PAIRED) fd_jacobian_pair<STAGE>(P, x, f, lb, ub, sa, ca, sb, cb, odd, J);
            else fd_jacobian<STAGE>(P, x, f, lb, ub, sa, ca, sb, cb, J);
            SEQIK_BLK_END_OF(BLK_FD_JACOBIAN);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                g[j] = fma_(J[2][j], f[2], fma_(J[1][j], f[1], J[0][j] * f[0]));
            }
#if SEQIK_FAST_PATHS
            cl_scaling_gated(x[0], g[0], lb[0], ub[0], sc.gate_lb[0], sc.gate_ub[0], v[0], dv[0]);
            if constexpr (NA == 2) cl_scaling_gated(x[1], g[1], lb[1], ub[1], sc.gate_lb[1], sc.gate_ub[1], v[1], dv[1]);
#else
            cl_scaling(x[0], g[0], lb[0], ub[0], v[0], dv[0]);
            if constexpr (NA == 2) cl_scaling(x[1], g[1], lb[1], ub[1], v[1], dv[1]);
#endif
            else { v[1] = 1.0; dv[1] = 0.0; }
            // d = sqrt_(v): needed by the trust-region scaling below and by Delta_0 (computed once, here)
            d[0] = sqrt_pos_(v[0]);
            d[1] = (NA == 2) ? sqrt_pos_(v[1]) : 1.0;
            jac_valid = true;
            }
            if (first_pass) {
                SEQIK_BLK_COUNT(CNT_FIRST_PASS);
                // Delta_0 = || x0 / sqrt_(v) || over ALL links (inert entries: v = 1)
                double acc = sc.x_pre_sq;
                // (x is still the start point x0 here: no step has been taken yet)
                double t0 = div_(x[0], d[0]);
                acc = fma_(t0, t0, acc);
                if constexpr (NA == 2) { double t1 = div_(x[1], d[1]); acc = fma_(t1, t1, acc); }
                acc = fma_(sc.x_suf, sc.x_suf, acc);
                Delta = sqrt_(acc);
                if (Delta == 0) Delta = 1.0;
                first_pass = false;
            }
            double g_norm = fabs(g[0] * v[0]);
            if constexpr (NA == 2) g_norm = fmax(g_norm, fabs(g[1] * v[1]));
            if (g_norm < gtol) status = 1;

            if (status != STATUS_NONE || nfev == max_nfev) {
                finished = true;
            } else {
                SEQIK_BLK_COUNT(CNT_TR);
                // ---- trust-region sub-problem -------------------------------------------
                double diag_h[2], g_h[2], Jh[3][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    diag_h[j] = g[j] * dv[j] * 1.0;
                    g_h[j] = d[j] * g[j];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) { Jh[k][0] = J[k][0] * d[0]; Jh[k][1] = J[k][1] * d[1]; }
                double theta = fmax(0.995, 1 - g_norm);

                // ---- ONE trial step per pass.  scipy's inner `while actual_reduction <= 0` loop is
                // unrolled over passes: after a rejected trial x, f and therefore J, g, the scaling
                // and the SVD are unchanged, so re-deriving them at the top of the next pass gives
                // bit-identical values, and no lane ever makes the other 63 wait in an inner loop.
                double p_h[2], p[2], step[2], step_h[2];
                SEQIK_BLK_END_OF(BLK_SCALING);
                if constexpr (NA == 2) {
                    solve_tr_2x2<T::DEFICIENT>(Jh, diag_h, f, Delta, alpha, p_h);
                } else {  // one unknown: the "SVD" is a column norm
                    double q[2] = {sqrt_(diag_h[0]), 0.0}, s[2], V[2][2], uf[2];
                    svd_active<1>(Jh, q, f, s, V, uf);
                    solve_lsq_trust_region<1, false>(uf, s, V, Delta, alpha, p_h);
                }
                SEQIK_BLK_END_OF(BLK_TR_STEP);
                p[0] = d[0] * p_h[0]; p[1] = d[1] * p_h[1];
                double predicted_reduction;
                double xp[2] = {x[0] + p[0], x[1] + p[1]};
                if (in_bounds2<NA>(xp, lb, ub)) {
                    predicted_reduction = -evaluate_quadratic<NA>(Jh, g_h, p_h, diag_h);
                    step[0] = p[0]; step[1] = p[1]; step_h[0] = p_h[0]; step_h[1] = p_h[1];
                } else {
                    SEQIK_BLK_END_OF(BLK_IN_BOUNDS);
                    SEQIK_BLK_COUNT(CNT_REFLECTIVE);
                    if constexpr (LAT)
                        predicted_reduction = select_step_reflective_ilp<NA>(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub,
                                                                             theta, step, step_h);
                    else
                        predicted_reduction = select_step_reflective<NA>(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub,
                                                                         theta, step, step_h);
                    SEQIK_BLK_END_OF(BLK_REFLECTIVE);
                }
                SEQIK_BLK_END_OF(BLK_IN_BOUNDS);
                double x_new[2] = {0.0, 0.0}, f_new[3], sa_n, ca_n, sb_n, cb_n, pe_n[3] = {0.0, 0.0, 0.0};
#if SEQIK_FAST_PATHS
                x_new[0] = strictly_feasible0(x[0] + step[0], lb[0], ub[0], sc.lb_in[0], sc.ub_in[0]);
                if constexpr (NA == 2) x_new[1] = strictly_feasible0(x[1] + step[1], lb[1], ub[1], sc.lb_in[1], sc.ub_in[1]);
#else
                x_new[0] = strictly_feasible(x[0] + step[0], lb[0], ub[0], 0.0);
                if constexpr (NA == 2) x_new[1] = strictly_feasible(x[1] + step[1], lb[1], ub[1], 0.0);
#endif
                if constexpr (PAIRED)
                    eval_residual_pair<STAGE>(P, x_new[0], x_new[1], odd, f_new, sa_n, ca_n, sb_n, cb_n, (STAGE == 1) ? pe_n : nullptr);
                else
                    eval_residual<STAGE>(P, x_new[0], x_new[1], f_new, sa_n, ca_n, sb_n, cb_n, (STAGE == 1) ? pe_n : nullptr);
                nfev += 1;
                SEQIK_BLK_END_OF(BLK_TRIAL_EVAL);
                double step_h_norm = norm2v<NA>(step_h);
                double cost_new = 0.5 * dot3(f_new, f_new);
                double actual_reduction = cost - cost_new;
                double ratio;
                if constexpr (LAT) {  // the quotient formed unconditionally (overlaps with the norms), then selected
                    const double q = div_(actual_reduction, predicted_reduction);
                    ratio = (predicted_reduction > 0) ? q : ((predicted_reduction == 0 && actual_reduction == 0) ? 1.0 : 0.0);
                } else {
                    if (predicted_reduction > 0) ratio = div_(actual_reduction, predicted_reduction);
                    else if (predicted_reduction == 0 && actual_reduction == 0) ratio = 1;
                    else ratio = 0;
                }
                double Delta_new = Delta;
                if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
                else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
                double step_norm = norm2v<NA>(step);
                // ||x|| over all links: inert prefix, active, inert last link
                double xn = sc.x_pre_sq;
                xn = fma_(x[0], x[0], xn);
                if constexpr (NA == 2) xn = fma_(x[1], x[1], xn);
                xn = fma_(sc.x_suf, sc.x_suf, xn);
                xn = sqrt_(xn);
                bool ftol_ok = (actual_reduction < ftol * cost) && (ratio > 0.25);
                bool xtol_ok = step_norm < xtol * (xtol + xn);
                if (ftol_ok && xtol_ok) status = 4;
                else if (ftol_ok) status = 2;
                else if (xtol_ok) status = 3;
                if constexpr (LAT) {
                    const double alpha_scaled = alpha * div_(Delta, Delta_new);
                    if (status == STATUS_NONE) { alpha = alpha_scaled; Delta = Delta_new; }
                } else if (status == STATUS_NONE) {
                    alpha = alpha * div_(Delta, Delta_new);
                    Delta = Delta_new;
                }
                if (actual_reduction > 0) {
                    SEQIK_BLK_COUNT(CNT_ACCEPT);
                    jac_valid = false;
                    x[0] = x_new[0]; x[1] = x_new[1];
                    f[0] = f_new[0]; f[1] = f_new[1]; f[2] = f_new[2];
                    cost = cost_new;
                    sa = sa_n; ca = ca_n; sb = sb_n; cb = cb_n;
                    if constexpr (STAGE == 1) { pe[0] = pe_n[0]; pe[1] = pe_n[1]; pe[2] = pe_n[2]; }
                }
                if (!WANT_DIAG && (status != STATUS_NONE || nfev == max_nfev)) finished = true;
            }
        } else {
            finished = true;
        }

        SEQIK_BLK_END_OF(BLK_POST_TRIAL);
        if (finished) {
            SEQIK_BLK_COUNT(CNT_FINISHED);
            queue_point_at(q_seq);
            // ---- solve done: store, advance to the next frame -------------------------------
            const bool stored = !CHUNKED || t >= io.t_store;  // run-in frames leave nothing but the hand-off
            if (stored) {
                double *ang = io.angles + t * io.ang_frame;
                ang[DOF0 * io.ang_dof] = x[0];
                if constexpr (NA == 2) ang[(DOF0 + 1) * io.ang_dof] = x[1];
                if constexpr (WANT_DIAG) {
                    if (io.status) io.status[t * 4 + STAGE - 1] = (status == STATUS_NONE) ? 0 : status;
                    if (io.nfev) io.nfev[t * 4 + STAGE - 1] = nfev;
                }
            } else if (CHUNKED && t == io.t_store - 1 && io.start_state) {
                io.start_state[DOF0] = x[0];
                if constexpr (NA == 2) io.start_state[DOF0 + 1] = x[1];
            }
            if constexpr ((WANT_FK && STAGE >= 2) || HANDOFF) {
                Frame after;  // frame after the active links at the solution
                frame_after_active<STAGE>(P, sa, ca, sb, cb, after);
                if constexpr (HANDOFF && PIPED) {
                    const int fi = io.pipe.base + (int)(t - t_first);
                    double *w = io.pipe.ring_out + (fi % PIPE_DEPTH) * 12 * io.pipe.lane_stride;
#pragma unroll
                    for (int i = 0; i < 9; ++i) w[i * io.pipe.lane_stride] = after.r[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) w[(9 + i) * io.pipe.lane_stride] = after.t[i];
                    pipe_store(io.pipe.produced_out, fi + 1);
                } else if constexpr (HANDOFF) {
                    double *w = io.frames + (t - t_first) * 12;
#pragma unroll
                    for (int i = 0; i < 9; ++i) w[i] = after.r[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) w[9 + i] = after.t[i];
                }
                if (WANT_FK && STAGE >= 2 && stored) {
                    // (the origin is read into registers BEFORE the first store: pose and fk may alias as far as the
                    // compiler knows, and re-reading origin[a] after every store made each of the up to 21 stores below
                    // wait for a memory round trip of its own: 14.1 -> 13.7 ms per benchmark step)
                    const double *origin_p = lc.aff.enabled ? lc.aff.template_coxa : io.pose + t * io.pose_frame;
                    const double origin[3] = {origin_p[0], origin_p[1], origin_p[2]};
                    double *fk = io.fk + t * 27;
                    if constexpr (STAGE == 2) {
                        for (int a = 0; a < 3; ++a) { fk[12 + a] = after.t[a] + origin[a]; fk[15 + a] = after.t[a] + origin[a]; }
                    } else if constexpr (STAGE == 3) {
                        if constexpr (FROM_ANGLES)  // stage 2 was not run: its rows (coxa end) come from the prefix
                            for (int a = 0; a < 3; ++a) { fk[12 + a] = P.pre.t[a] + origin[a]; fk[15 + a] = P.pre.t[a] + origin[a]; }
                        for (int a = 0; a < 3; ++a) fk[18 + a] = after.t[a] + origin[a];
                    } else {
                        if constexpr (FROM_ANGLES)
                            for (int a = 0; a < 3; ++a) {
                                fk[12 + a] = coxa_end[a] + origin[a];
                                fk[15 + a] = coxa_end[a] + origin[a];
                                fk[18 + a] = P.pre.t[a] + origin[a];
                            }
                        for (int i = 0; i < 4; ++i)
                            for (int a = 0; a < 3; ++a) fk[3 * i + a] = 0.0 + origin[a];
                        for (int a = 0; a < 3; ++a) {
                            fk[21 + a] = after.t[a] + origin[a];
                            fk[24 + a] = (after.r[3 * a + 2] * P.tz_last + after.t[a]) + origin[a];
                        }
                    }
                }
            }
            t += 1;
            new_solve = true;
        }
        SEQIK_BLK_END_OF(BLK_FINISHED);
        if constexpr (QUEUE) {
            // (all lanes that are still in the loop are converged here)
            const bool chain_done = t >= io.n_frames;
            const unsigned long long done_mask = wave_ballot(chain_done);
            if (done_mask != 0ull) {  // wave-uniform; once per chain and lane
                if (chain_done) {
                    const int32_t s_new = q_next + lane_rank_in(done_mask);
                    if (s_new < io_arg.q_end) {
                        q_seq = s_new;
                        // a new chain starts as every chain does (function entry): seeds / init, first frame, general start
                        x[0] = sc.seed[0];
                        x[1] = (NA == 2) ? sc.seed[1] : 0.0;
                        if (io_arg.init) {
                            const double *init_c = io_arg.init + (uint64_t)(uint32_t)s_new * io_arg.q_init;
                            x[0] = init_c[DOF0];
                            if constexpr (NA == 2) x[1] = init_c[DOF0 + 1];
                        }
                        have_pe = false;
                        t = 0;
                    }
                }
                q_next += mask_count(done_mask);
            }
        }
    }
    SEQIK_BLK_END(STAGE);
}

}  // namespace seqik
