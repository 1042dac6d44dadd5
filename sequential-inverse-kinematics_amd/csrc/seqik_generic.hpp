// seqik_generic.hpp -- "generic" (single-chain) leg inverse kinematics, one lane per chain.
//
// Replaces LegInvKinGeneric.calculate_ik_stage / run_ik_and_fk
// (seqikpy/leg_inverse_kinematics.py:406-613) over the 9-link chain of
// KinematicChainGeneric.create_leg_chain (seqikpy/kinematic_chain.py:464-530):
//   Base | ThC_roll(Z) ThC_yaw(X) ThC_pitch(Y) CTr_pitch(Y,-coxa) CTr_roll(Z) FTi(Y,-femur) TiTa(Y,-tibia) | Claw(-tarsus)
// One bounded trust-region least-squares solve per frame with 7 unknowns and 3 residuals (the claw
// position), warm-started from the previous frame.
//
// The problem is rank-deficient by construction (7 unknowns, 3 equations).  In the reference the
// step inside the null space is decided by LAPACK round-off, so only the claw position
// -- not the individual angles -- can be compared with the reference (the reference does not reproduce its
// own angles under a 1-ulp change of its input: profiles/r04_perturbation_generic.json).  What IS exact:
// this file mirrors the generic code path of oracle/seqik_oracle.c (oracle_generic_leg) operation for
// operation, so kernel == oracle bit for bit, as for the sequential stages.
//
// Since the angles are not pinned, the association order of the forward kinematics is the restatement's to choose
// (round-3 review): the claw position is evaluated RIGHT TO LEFT as a 3-vector pushed through the seven links
// (generic_claw: 30 instructions) instead of as the last column of the left-to-right product of 3 x 4 frames
// (generic_chain: 150, now only used once per frame for the stored joint positions), and the trust-region step uses
// the push-through identity twice (woodbury_phi).  A pass is ~30 % shorter; oracle_generic_leg makes the same choices.
#pragma once
#include "seqik_core.hpp"

namespace seqik {

constexpr int GN = 7;  // active links

struct GenericConst {
    double lb[GN], ub[GN];   // bounds in LINK order: roll, yaw, pitch, CTr_pitch, CTr_roll, FTi, TiTa
    double lb_in[GN], ub_in[GN];  // next_toward(lb, ub), next_toward(ub, lb): make_strictly_feasible(rstep=0) replacements
    double gate_lb[GN], gate_ub[GN];  // 0.0 / NaN: isfinite(bound) folded into the sign test of the gradient (cl_scaling_gated)
    double seed[GN];         // initial_angles["stage_4"][1..7], applied positionally to the links
    double tz[GN];           // origin_translation z per link: 0, 0, 0, -coxa, 0, -femur, -tibia
    double tz_claw;          // -tarsus
    double x_pre_sq;         // fma(x_base, x_base, 0)
    double x_suf;            // strictly feasible seed of the claw link
    int32_t max_nfev;        // 900
    int32_t pad_;
};

// cumulative chain for given sin/cos of the 7 joints; returns the frame after TiTa and, optionally,
// the translations after CTr_pitch (coxa end) and FTi (femur end) for the FK output
SEQIK_HD void generic_chain(const GenericConst &gc, const double *sn, const double *cs, Frame &out, double *coxa_end,
                            double *femur_end)
{
    Frame a, b;
    frame_identity(a);
    frame_mul_link<AXIS_Z>(b, a, sn[0], cs[0], gc.tz[0]);
    frame_mul_link<AXIS_X>(a, b, sn[1], cs[1], gc.tz[1]);
    frame_mul_link<AXIS_Y>(b, a, sn[2], cs[2], gc.tz[2]);
    frame_mul_link<AXIS_Y>(a, b, sn[3], cs[3], gc.tz[3]);
    if (coxa_end) { coxa_end[0] = a.t[0]; coxa_end[1] = a.t[1]; coxa_end[2] = a.t[2]; }
    frame_mul_link<AXIS_Z>(b, a, sn[4], cs[4], gc.tz[4]);
    frame_mul_link<AXIS_Y>(a, b, sn[5], cs[5], gc.tz[5]);
    if (femur_end) { femur_end[0] = a.t[0]; femur_end[1] = a.t[1]; femur_end[2] = a.t[2]; }
    frame_mul_link<AXIS_Y>(out, a, sn[6], cs[6], gc.tz[6]);
}

// v <- [R_axis(s, c) | (0, 0, tz)] v : one link applied to a position vector.  Row r of the oracle's general form
//     acc = M[r][3]; acc = fma(M[r][2], v[2], acc); acc = fma(M[r][1], v[1], acc); acc = fma(M[r][0], v[0], acc)
// (chain_end_effector_rtl) with the exact zeros and ones of the axis rotation dropped.
template <int AXIS>
SEQIK_HD void link_apply(double s, double c, double tz, double *v)
{
    const double x = v[0], y = v[1], z = v[2];
    if constexpr (AXIS == AXIS_X) {          // [1 0 0 | 0] [0 c -s | 0] [0 s c | tz]
        v[1] = fma_(c, y, (-s) * z);
        v[2] = fma_(s, y, fma_(c, z, tz));
    } else if constexpr (AXIS == AXIS_Y) {   // [c 0 s | 0] [0 1 0 | 0] [-s 0 c | tz]
        v[0] = fma_(c, x, s * z);
        v[2] = fma_(-s, x, fma_(c, z, tz));
    } else {                                 // [c -s 0 | 0] [s c 0 | 0] [0 0 1 | tz]
        v[0] = fma_(c, x, (-s) * y);
        v[1] = fma_(s, x, c * y);
        v[2] = z + tz;
    }
}

// link I of the generic chain: roll(Z) yaw(X) pitch(Y) CTr_pitch(Y) CTr_roll(Z) FTi(Y) TiTa(Y)
template <int I>
SEQIK_HD void generic_link_apply(const GenericConst &gc, double s, double c, double *v)
{
    constexpr int AXIS = (I == 0 || I == 4) ? AXIS_Z : (I == 1 ? AXIS_X : AXIS_Y);
    link_apply<AXIS>(s, c, gc.tz[I], v);
}

// links FROM - 1 .. 0 applied to v (the part of the chain in FRONT of link FROM)
template <int FROM>
SEQIK_HD void generic_links_below(const GenericConst &gc, const double *sn, const double *cs, double *v)
{
    if constexpr (FROM > 0) {
        generic_link_apply<FROM - 1>(gc, sn[FROM - 1], cs[FROM - 1], v);
        generic_links_below<FROM - 1>(gc, sn, cs, v);
    }
}

// Claw position relative to the origin: v = (0, 0, tz_claw) (the claw link), then links 6 .. 0.
// suf (nullable, [GN][3]): suf[k] = the vector after links 6 .. k have been applied, k = 1 .. 6, and suf[0]... is not
// kept (it is the result); used by the finite-difference columns, which share the part behind the perturbed link.
SEQIK_HD void generic_claw(const GenericConst &gc, const double *sn, const double *cs, double *v, double (*suf)[3] = nullptr)
{
    v[0] = 0.0; v[1] = 0.0; v[2] = gc.tz_claw;
#define SEQIK_GEN_STEP(I)                                                     \
    generic_link_apply<I>(gc, sn[I], cs[I], v);                               \
    if (suf) { suf[I][0] = v[0]; suf[I][1] = v[1]; suf[I][2] = v[2]; }
    SEQIK_GEN_STEP(6) SEQIK_GEN_STEP(5) SEQIK_GEN_STEP(4) SEQIK_GEN_STEP(3) SEQIK_GEN_STEP(2) SEQIK_GEN_STEP(1)
#undef SEQIK_GEN_STEP
    generic_link_apply<0>(gc, sn[0], cs[0], v);
}

// the same with joint jm's sin / cos replaced by (s1, c1): one finite-difference column per lane of a group
SEQIK_HD void generic_claw_perturbed(const GenericConst &gc, const double *sn, const double *cs, int jm, double s1, double c1, double *v)
{
    v[0] = 0.0; v[1] = 0.0; v[2] = gc.tz_claw;
#define SEQIK_GEN_STEP(I) generic_link_apply<I>(gc, (jm == I) ? s1 : sn[I], (jm == I) ? c1 : cs[I], v);
    SEQIK_GEN_STEP(6) SEQIK_GEN_STEP(5) SEQIK_GEN_STEP(4) SEQIK_GEN_STEP(3) SEQIK_GEN_STEP(2) SEQIK_GEN_STEP(1) SEQIK_GEN_STEP(0)
#undef SEQIK_GEN_STEP
}

SEQIK_HD void generic_residual(const GenericConst &gc, const double *sn, const double *cs, const double *target, double *f,
                               double (*suf)[3] = nullptr)
{
    double v[3];
    generic_claw(gc, sn, cs, v, suf);
#pragma unroll
    for (int i = 0; i < 3; ++i) f[i] = v[i] - target[i];
}

// column J of the finite-difference Jacobian's perturbed residual, from the shared suffix: links 6 .. J + 1 at the base
// angles (suf[J + 1], or the claw link alone for J = 6), link J at (s1, c1), links J - 1 .. 0 at the base angles -- the same
// operations on the same values as generic_claw_perturbed(jm = J)
template <int J>
SEQIK_HD void generic_residual_column(const GenericConst &gc, const double *sn, const double *cs, const double (*suf)[3],
                                      double s1, double c1, const double *target, double *f1)
{
    double v[3];
    if constexpr (J == GN - 1) { v[0] = 0.0; v[1] = 0.0; v[2] = gc.tz_claw; }
    else { v[0] = suf[J + 1][0]; v[1] = suf[J + 1][1]; v[2] = suf[J + 1][2]; }
    generic_link_apply<J>(gc, s1, c1, v);
    generic_links_below<J>(gc, sn, cs, v);
#pragma unroll
    for (int i = 0; i < 3; ++i) f1[i] = v[i] - target[i];
}

SEQIK_HD double vnorm7(const double *a)
{
    double acc = 0.0;
    for (int i = 0; i < GN; ++i) acc = fma_(a[i], a[i], acc);
    return sqrt_(acc);
}

SEQIK_HD double vdot7(const double *a, const double *b)
{
    double acc = 0.0;
    for (int i = 0; i < GN; ++i) acc = fma_(a[i], b[i], acc);
    return acc;
}

SEQIK_HD void matvec37(const double Jh[3][GN], const double *s, double *out)
{
    for (int k = 0; k < 3; ++k) {
        double acc = 0.0;
        for (int i = 0; i < GN; ++i) acc = fma_(Jh[k][i], s[i], acc);
        out[k] = acc;
    }
}

SEQIK_HD double diag_form7(const double *a, const double *diag, const double *b)
{
    double acc = 0.0;
    for (int i = 0; i < GN; ++i) acc = fma_(a[i] * diag[i], b[i], acc);
    return acc;
}

// ---------------------------------------------------------------------------
// Lane groups (GROUPED instantiation of run_generic): a thin wave carries every chain on a multiple of 8 adjacent
// lanes that hold the same state (seqik_hip.hip, lane_replication).  Lanes 0..6 of a group of 8 each take ONE of the
// seven joints wherever a pass runs the same code once per joint -- the seven finite-difference columns (a perturbed
// sin / cos and the whole 7-link chain product each: half of a pass) and the seven sin / cos of the trial point -- and
// the group exchanges the results with ds_swizzle broadcasts (the LDS crossbar, no memory, no barrier).  Every value
// is produced by the same operations on the same operands as in the one-lane code: same bits.  Lane 7 repeats joint 6.
// ---------------------------------------------------------------------------
template <int C>
SEQIK_HD double group8_bcast(double v)  // the value lane C of this lane's group of 8 holds
{
#if defined(__HIP_DEVICE_COMPILE__)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, (C << 5) | 0x18);  // bit mode: source lane = (lane & 0x18) | C
    hi = __builtin_amdgcn_ds_swizzle(hi, (C << 5) | 0x18);
    return __hiloint2double(hi, lo);
#else
    return v;
#endif
}

SEQIK_HD void group8_gather(double v, double *out /* [GN] */)
{
    out[0] = group8_bcast<0>(v); out[1] = group8_bcast<1>(v); out[2] = group8_bcast<2>(v); out[3] = group8_bcast<3>(v);
    out[4] = group8_bcast<4>(v); out[5] = group8_bcast<5>(v); out[6] = group8_bcast<6>(v);
}

SEQIK_HD int group8_joint()
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int l = threadIdx.x & 7;
    return l < GN ? l : GN - 1;
#else
    return 0;
#endif
}

SEQIK_HD double pick7(const double *a, int j)  // a[j] without indexing a register array by a lane-dependent value
{
    double r = a[0];
#pragma unroll
    for (int i = 1; i < GN; ++i) r = (j == i) ? a[i] : r;
    return r;
}

// ---------------------------------------------------------------------------
// Trust-region step without an SVD (mirrors oracle solve_tr_woodbury operation for operation).
//
// scipy's solve_lsq_trust_region works on the SVD of A = [[J_h], [diag(sqrt_(diag_h))]] (10 x 7).  All it needs from
// it is p(alpha) = -(A^T A + alpha I)^-1 J_h^T f, ||p|| and phi'(alpha) = -p^T (A^T A + alpha I)^-1 p / ||p||.  With
// B = diag(diag_h + alpha), W = B^-1 and A^T A = diag(diag_h) + J_h^T J_h the push-through identity gives
//     (B + J_h^T J_h)^-1 r = W r - W J_h^T (I_3 + J_h W J_h^T)^-1 J_h W r,
// one symmetric 3 x 3 inverse per alpha (by cofactors: no pivoting, fine for the indefinite matrices the last,
// possibly negative alpha of scipy's m < n root search can produce).  Root search, bracket updates and the final
// rescaling of p to the trust radius are scipy's (_lsq/common.py:solve_lsq_trust_region, m < n branch);
// alpha_upper = ||s * uf|| / Delta = ||J_h^T f|| / Delta.  The one-sided Jacobi SVD this replaces (21 rotations x
// ~8 sweeps per pass, 119 matrix entries live) was > 95 % of this kernel's instructions and the reason it needed
// 489 registers.
// ---------------------------------------------------------------------------
SEQIK_HD void sym3_inverse(const double *m /* 00 01 02 11 12 22 */, double *inv)
{
    double c00 = fma_(m[3], m[5], -(m[4] * m[4]));
    double c01 = fma_(m[2], m[4], -(m[1] * m[5]));
    double c02 = fma_(m[1], m[4], -(m[2] * m[3]));
    double c11 = fma_(m[0], m[5], -(m[2] * m[2]));
    double c12 = fma_(m[1], m[2], -(m[0] * m[4]));
    double c22 = fma_(m[0], m[3], -(m[1] * m[1]));
    double det = fma_(m[2], c02, fma_(m[1], c01, m[0] * c00));
    double r = div_(1.0, det);
    inv[0] = c00 * r; inv[1] = c01 * r; inv[2] = c02 * r; inv[3] = c11 * r; inv[4] = c12 * r; inv[5] = c22 * r;
}

// pp = (B + J_h^T J_h)^-1 J_h^T f (the un-negated step), and -- when WANT_PHI -- phi and the Newton ratio phi / phi'.
// Push-through twice (oracle woodbury_phi, form 1):  pp = W J_h^T y  with  y = (I + J_h W J_h^T)^-1 f,  and
//     pp^T (B + J_h^T J_h)^-1 pp = pp^T W pp - u^T (I + J_h W J_h^T)^-1 u,   u = J_h W pp.
// jm >= 0: lane group, this lane divides for joint jm only (the seven quotients are the same code on different data)
template <bool WANT_PHI>
SEQIK_HD void woodbury_phi(const double Jh[3][GN], const double *diag_h, const double *f, double alpha, double Delta,
                           double *pp, double &phi, double &ratio, int jm = -1)
{
    double W[GN], M[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 1.0}, Minv[6];
    if (jm >= 0) {
        group8_gather(div_(1.0, pick7(diag_h, jm) + alpha), W);
    } else {
#pragma unroll
        for (int c = 0; c < GN; ++c) W[c] = div_(1.0, diag_h[c] + alpha);
    }
#pragma unroll
    for (int c = 0; c < GN; ++c) {
        double w0 = W[c] * Jh[0][c], w1 = W[c] * Jh[1][c], w2 = W[c] * Jh[2][c];
        M[0] = fma_(w0, Jh[0][c], M[0]); M[1] = fma_(w0, Jh[1][c], M[1]); M[2] = fma_(w0, Jh[2][c], M[2]);
        M[3] = fma_(w1, Jh[1][c], M[3]); M[4] = fma_(w1, Jh[2][c], M[4]); M[5] = fma_(w2, Jh[2][c], M[5]);
    }
    sym3_inverse(M, Minv);
    double y[3];
    y[0] = fma_(Minv[2], f[2], fma_(Minv[1], f[1], Minv[0] * f[0]));
    y[1] = fma_(Minv[4], f[2], fma_(Minv[3], f[1], Minv[1] * f[0]));
    y[2] = fma_(Minv[5], f[2], fma_(Minv[4], f[1], Minv[2] * f[0]));
#pragma unroll
    for (int c = 0; c < GN; ++c) {
        double z = fma_(Jh[2][c], y[2], fma_(Jh[1][c], y[1], Jh[0][c] * y[0]));
        pp[c] = W[c] * z;
    }
    if constexpr (WANT_PHI) {
        double p_norm = vnorm7(pp);
        double wa[GN], u[3], t[3];
#pragma unroll
        for (int c = 0; c < GN; ++c) wa[c] = W[c] * pp[c];
        double s1 = vdot7(wa, pp);
        matvec37(Jh, wa, u);
        t[0] = fma_(Minv[2], u[2], fma_(Minv[1], u[1], Minv[0] * u[0]));
        t[1] = fma_(Minv[4], u[2], fma_(Minv[3], u[1], Minv[1] * u[0]));
        t[2] = fma_(Minv[5], u[2], fma_(Minv[4], u[1], Minv[2] * u[0]));
        double s2 = fma_(u[2], t[2], fma_(u[1], t[1], u[0] * t[0]));
        double acc = s1 - s2;
        phi = p_norm - Delta;
        ratio = div_(-(phi * p_norm), acc);
    }
}

// solve_lsq_trust_region with m = 3 < n = 7 (never full rank), SVD-free
SEQIK_HD void solve_tr_woodbury(const double Jh[3][GN], const double *diag_h, const double *f, double Delta,
                                double &alpha_io, double *p, int jm = -1)
{
    double rhs[GN], pp[GN];
#pragma unroll
    for (int c = 0; c < GN; ++c) rhs[c] = fma_(Jh[2][c], f[2], fma_(Jh[1][c], f[1], Jh[0][c] * f[0]));
    const double inv_Delta = div_(1.0, Delta);
    double alpha_upper = vnorm7(rhs) * inv_Delta;
    double alpha_lower = 0.0;
    double alpha = alpha_io;
    if (alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt_(alpha_lower * alpha_upper));
    // shortcut of the m < n root search (see seqik_core.hpp solve_tr_2x2 / oracle solve_tr_2x2): Gauss-Newton step
    // inside the trust region = ten resets alpha <- 0.001 alpha_upper + one Newton step from the last alpha
    bool shortcut = false;
    {
        double au = alpha_upper, a_k = alpha;
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            if (a_k < 0.0 || a_k > au) a_k = fmax(0.001 * au, 0.0);
            au = a_k;
            if (it < 9) a_k = -1.0;
        }
        double phi, ratio;
        woodbury_phi<true>(Jh, diag_h, f, a_k, Delta, pp, phi, ratio, jm);
        if (phi < 0 && !(fabs(phi) < 0.01 * Delta)) {
            alpha = a_k - (phi + Delta) * ratio * inv_Delta;
            shortcut = true;
        }
    }
    for (int it = 0; it < 10 && !shortcut; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt_(alpha_lower * alpha_upper));
        double phi, ratio;
        woodbury_phi<true>(Jh, diag_h, f, alpha, Delta, pp, phi, ratio, jm);
        if (phi < 0) alpha_upper = alpha;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio * inv_Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    double unused_phi, unused_ratio;
    woodbury_phi<false>(Jh, diag_h, f, alpha, Delta, pp, unused_phi, unused_ratio, jm);
    double scale = div_(Delta, vnorm7(pp));
#pragma unroll
    for (int c = 0; c < GN; ++c) p[c] = -(pp[c] * scale);
    alpha_io = alpha;
}

// _lsq/common.py:step_size_to_bound.  Written WITHOUT branches: the reference-shaped call is one wavefront walking one
// chain, its time is the latency of its dependent operations, and seven divisions inside seven `if (s[i] != 0)` blocks run one
// after the other (a basic block is the scheduler's horizon), while seven unconditional ones overlap.  Same values: for
// s[i] == 0 the quotient (+-inf or NaN) is discarded by the select, as the branch skipped it.  (Measured with the s_memtime
// stamps, scripts/generic_block_cycles.py: select_step7 was 44 % of a pass of the shipped recording before this.)
// hit (nullable): this joint's step equals the minimum and its direction is not 0 (scipy's `hits` as a flag).
// jm >= 0: lane group -- this lane forms the quotients of joint jm only and the group exchanges the seven step lengths.
SEQIK_HD double step_size_to_bound7(const double *x, const double *s, const double *lb, const double *ub, bool *hit, int jm = -1)
{
    const double INF = __builtin_huge_val();
    double steps[GN];
    if (jm >= 0) {
        const double sj = pick7(s, jm), xj = pick7(x, jm);
        const double inv_s = div_(1.0, sj);
        const double st = fmax((lb[jm] - xj) * inv_s, (ub[jm] - xj) * inv_s);
        group8_gather((sj != 0.0) ? st : INF, steps);
    } else {
#pragma unroll
        for (int i = 0; i < GN; ++i) {
            const double inv_s = div_(1.0, s[i]);
            const double st = fmax((lb[i] - x[i]) * inv_s, (ub[i] - x[i]) * inv_s);
            steps[i] = (s[i] != 0.0) ? st : INF;
        }
    }
    // the minimum is exact in any order: a tree instead of a chain
    const double m01 = fmin(steps[0], steps[1]), m23 = fmin(steps[2], steps[3]), m45 = fmin(steps[4], steps[5]);
    const double min_step = fmin(fmin(m01, m23), fmin(m45, steps[6]));
    if (hit) {
#pragma unroll
        for (int i = 0; i < GN; ++i) hit[i] = (steps[i] == min_step) && (s[i] != 0.0);
    }
    return min_step;
}

SEQIK_HD double evaluate_quadratic7(const double Jh[3][GN], const double *g, const double *s, const double *diag)
{
    double Js[3];
    matvec37(Jh, s, Js);
    double q = dot3(Js, Js);
    q = q + diag_form7(s, diag, s);
    return fma_(0.5, q, vdot7(s, g));
}

// _lsq/trf.py:select_step (in-bounds case included); p, p_h are clobbered.
// The three candidates -- the interior step scaled by theta, the step reflected at the first bound it hits, the
// constrained Cauchy step -- are scipy's, operation for operation (oracle select_step).  What is specific to this file is the
// ORDER the independent pieces are written in: everything that does not depend on the strides (the seven-term dot
// products, the three J_h products, the three bound distances with their 21 divisions) comes first, in one basic block,
// so that the scheduler can overlap the chains; the scalar stride logic follows.  No value changes.
SEQIK_HD double select_step7(const double *x, const double Jh[3][GN], const double *diag_h, const double *g_h, double *p,
                             double *p_h, const double *d, double Delta, const double *lb, const double *ub, double theta,
                             double *step, double *step_h, int jm = -1)
{
    const double INF = __builtin_huge_val();
    bool inb = true;
#pragma unroll
    for (int i = 0; i < GN; ++i) {
        double xp = x[i] + p[i];
        if (!(xp >= lb[i] && xp <= ub[i])) inb = false;
    }
    if (inb) {
        double p_value = evaluate_quadratic7(Jh, g_h, p_h, diag_h);
#pragma unroll
        for (int i = 0; i < GN; ++i) { step[i] = p[i]; step_h[i] = p_h[i]; }
        return -p_value;
    }
    // ---- stride-independent part of all three candidates ----------------------------------------------------------
    bool hit[GN];
    const double p_stride = step_size_to_bound7(x, p, lb, ub, hit, jm);
    double r_h[GN], r[GN], x_on_bound[GN], pt_h[GN], ag_h[GN], ag[GN];
#pragma unroll
    for (int i = 0; i < GN; ++i) {
        r_h[i] = hit[i] ? p_h[i] * -1.0 : p_h[i];
        r[i] = d[i] * r_h[i];
        p[i] = p[i] * p_stride;
        p_h[i] = p_h[i] * p_stride;
        x_on_bound[i] = x[i] + p[i];
        pt_h[i] = p_h[i] * theta;      // the interior step, kept strictly inside
        ag_h[i] = -g_h[i];
        ag[i] = d[i] * ag_h[i];
    }
    // reflected step: distance to the trust-region boundary (the quadratic's coefficients) and to the bounds
    const double a_tr = vdot7(r_h, r_h);
    const double b_tr = vdot7(p_h, r_h);
    const double c_tr = fma_(-Delta, Delta, vdot7(p_h, p_h));
    const double to_bound_r = step_size_to_bound7(x_on_bound, r, lb, ub, nullptr, jm);
    // ... and its 1-d model along r_h from p_h (build_quadratic_1d with s0 = p_h)
    double v[3], u[3], w[3];
    matvec37(Jh, r_h, v);
    matvec37(Jh, p_h, u);
    double a_q = dot3(v, v);
    a_q = a_q + diag_form7(r_h, diag_h, r_h);
    a_q = a_q * 0.5;
    double b_q = vdot7(g_h, r_h);
    b_q = b_q + dot3(u, v);
    double c_q = fma_(0.5, dot3(u, u), vdot7(g_h, p_h));
    b_q = b_q + diag_form7(p_h, diag_h, r_h);
    c_q = fma_(0.5, diag_form7(p_h, diag_h, p_h), c_q);
    // interior step scaled by theta
    const double p_value = evaluate_quadratic7(Jh, g_h, pt_h, diag_h);
    // anti-gradient (constrained Cauchy) step
    const double to_tr_ag = div_(Delta, vnorm7(ag_h));
    const double to_bound_ag = step_size_to_bound7(x, ag, lb, ub, nullptr, jm);
    matvec37(Jh, ag_h, w);
    double a_ag = dot3(w, w);
    a_ag = a_ag + diag_form7(ag_h, diag_h, ag_h);
    a_ag = a_ag * 0.5;
    const double b_ag = vdot7(g_h, ag_h);
    // ---- strides ---------------------------------------------------------------------------------------------------
    double to_tr;
    {
        double dd = sqrt_(fma_(b_tr, b_tr, -(a_tr * c_tr)));
        double q = -(b_tr + copysign(dd, b_tr));
        double t1 = div_(q, a_tr);
        double t2 = div_(c_tr, q);
        to_tr = (t1 < t2) ? t2 : t1;
    }
    double r_stride = fmin(to_bound_r, to_tr);
    double r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = div_((1 - theta) * p_stride, r_stride);
        r_stride_u = (r_stride == to_bound_r) ? theta * to_bound_r : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        r_stride = minimize_quadratic_1d(a_q, b_q, r_stride_l, r_stride_u, c_q, r_value);
#pragma unroll
        for (int i = 0; i < GN; ++i) {
            r_h[i] = r_h[i] * r_stride;
            r_h[i] = r_h[i] + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = INF;
    }
    double ag_stride = (to_bound_ag < to_tr_ag) ? theta * to_bound_ag : to_tr_ag;
    double ag_value;
    ag_stride = minimize_quadratic_1d(a_ag, b_ag, 0.0, ag_stride, 0.0, ag_value);
#pragma unroll
    for (int i = 0; i < GN; ++i) { ag_h[i] = ag_h[i] * ag_stride; ag[i] = ag[i] * ag_stride; }

    const bool take_p = p_value < r_value && p_value < ag_value;
    const bool take_r = !take_p && r_value < p_value && r_value < ag_value;
#pragma unroll
    for (int i = 0; i < GN; ++i) {  // value selects, no pointer select: the candidates stay in registers
        // (p * theta is formed again here instead of being kept since the top: 14 multiplications against 56 register copies)
        step[i] = take_p ? p[i] * theta : (take_r ? r[i] : ag[i]);
        step_h[i] = take_p ? p_h[i] * theta : (take_r ? r_h[i] : ag_h[i]);
    }
    return -(take_p ? p_value : (take_r ? r_value : ag_value));
}

// finite-difference column J of the one-lane code (_numdiff.py 2-point scheme, as fd_jacobian of the sequential stages)
template <int J>
SEQIK_HD void fd_column(const GenericConst &gc, const double *x, const double *sn, const double *cs, const double (*suf)[3],
                        const double *target, const double *f, double Jm[3][GN])
{
    double h = fd_step(x[J], gc.lb[J], gc.ub[J]);
    double x1 = x[J] + h;
    double dx = x1 - x[J];
    double s1, c1, f1[3];
    sincos_cw(x1, s1, c1);
    generic_residual_column<J>(gc, sn, cs, suf, s1, c1, target, f1);
    double inv_dx = div_(1.0, dx);
#pragma unroll
    for (int k = 0; k < 3; ++k) Jm[k][J] = (f1[k] - f[k]) * inv_dx;
}

struct GenericIO {
    const double *pose;     // key point (row, t) at pose + row * pose_row + t * pose_frame
    int64_t pose_row, pose_frame;
    double *angles;         // angle (dof, t) at angles + dof * ang_dof + t * ang_frame, DOF order of seqik.h
    int64_t ang_dof, ang_frame;
    double *fk;             // nullable [n_frames][9][3]
    int32_t *status;        // nullable [n_frames]
    int32_t *nfev;          // nullable [n_frames]
    const double *init;     // nullable [7] in DOF order: warm start of frame 0
    int64_t n_frames;
};

// Chain queue of ONE leg (batches of generic chains; seqik_hip.hip seqik_generic_queue_kernel).  A wavefront of the
// one-lane-per-chain instantiation lives as long as its slowest lane and the launch as long as its slowest wavefront; the
// iteration counts of the 7-unknown problem are heavy-tailed (mean lane 2 010 passes, mean wavefront 4 030, slowest chain
// 11 243 on windows of the shipped recording), so in a batch with several times more chains than the GPU has lanes half of
// the lane-passes are spent waiting.  With a queue a lane that has finished its chain takes the next sequence of its leg
// from an atomic counter (submission order); when its leg has none left it moves on to the next leg (a lane never returns
// to a leg it found exhausted) and exits when all are exhausted: the chains are independent, every chain is walked by the
// same code from the same start, so the results are the static launch's bit for bit.  Every lane reaches the exit (the
// counters only grow, a lane tries every leg at most once more), so the grid drains.  Reference: one run_ik_and_fk call per recording and leg
// (seqikpy/leg_inverse_kinematics.py:545-613); the queue is the batching of many such calls.
struct GenericLeg {          // one leg's constants as the kernels keep them in LDS
    GenericConst gc;
    LegAffine aff;
};

SEQIK_HD const GenericConst *leg_consts(const GenericLeg *table, int leg) { return &table[leg].gc; }
SEQIK_HD const LegAffine *leg_affine(const GenericLeg *table, int leg) { return &table[leg].aff; }

struct GenericQueue {
    int32_t *counters;       // [n_legs] next sequence of every leg (device memory, zeroed by the launcher)
    int64_t n_seq;           // sequences of the batch; chain of (sequence s, leg l) = s * n_legs + l
    int32_t n_legs, first;   // the lane starts on leg order[first] and moves on, leg by leg, when a leg's counter is exhausted
    uint8_t order[8];        // dispatch order of the legs
    const GenericLeg *table; // [n_legs] constants of every leg (LDS): the lane's leg changes while it runs
    const double *pose; int64_t pose_chain;   // bases and per-chain strides of the batch's buffers
    double *angles; int64_t ang_chain;
    double *fk;              // nullable, [chain][n_frames][9][3]
    int32_t *status, *nfev;  // nullable, [chain][n_frames]
    const double *init;      // nullable, [chain][7]
};

// link index -> DOF index of the ABI (yaw, pitch, roll, CTr_pitch, CTr_roll, FTi, TiTa)
SEQIK_HD int generic_link_dof(int link)
{
    return link == 0 ? 2 : (link == 1 ? 0 : (link == 2 ? 1 : link));
}

// gc must live in addressable memory (LDS): the GROUPED code reads gc.lb[j] / gc.ub[j] with a lane-dependent j
// QUEUED: `io_in` only carries n_frames and the strides; the lane takes its chains from `queue` (GenericQueue above) until
// the leg's counter is exhausted.  One lane per chain only (a lane group would have to pull as a group).
template <bool WANT_DIAG, bool GROUPED = false, bool QUEUED = false>
SEQIK_HD void run_generic(const GenericConst &gc0, const LegAffine &aff0, const GenericIO &io_in, const GenericQueue *queue = nullptr)
{
    static_assert(!(GROUPED && QUEUED), "the chain queue is for the one-lane-per-chain instantiation");
    // (QUEUED: the lane's leg -- and with it the constants -- changes while it runs; otherwise these never move)
    const GenericConst *gcp = &gc0;
    const LegAffine *affp = &aff0;
#define gc (*gcp)
#define aff (*affp)
    int legs_tried = 0;
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    const int jm = GROUPED ? group8_joint() : 0;  // the joint this lane takes in grouped sections
    GenericIO io = io_in;
    double x[GN], f[3] = {0.0, 0.0, 0.0}, sn[GN], cs[GN], target[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < GN; ++i) {
        x[i] = (!QUEUED && io.init) ? io.init[generic_link_dof(i)] : gc.seed[i];
        sn[i] = 0.0; cs[i] = 1.0;
    }
    double cost = 0.0, Delta = 0.0, alpha = 0.0;
    int nfev = 0, status = STATUS_NONE;
    bool first_pass = true, new_solve = true;
    int64_t t = QUEUED ? io.n_frames : 0;   // QUEUED: "my chain is finished" -- the first pass pulls the first chain

    // (diagnostic build only, -DSEQIK_BLOCK_CYCLES=1: the s_memtime stamps of run_stage, scripts/generic_block_cycles.py;
    // block names as used here: new_solve, fd_jacobian = step + sin / cos + chain + gather of the columns, scaling_gtol =
    // gradient, Coleman-Li scaling, radius, d, J_h, tr_step = solve_tr_woodbury, in_bounds = select_step7, reflective =
    // trial point + its sin / cos + gather, trial_eval = chain + cost, post_trial, finished)
    SEQIK_BLK_DECL
    while (QUEUED || t < io.n_frames) {
        if constexpr (QUEUED) {
            if (t >= io.n_frames) {     // this lane's chain is done (or it has none yet): take the next sequence of its leg,
                int64_t sq = -1;        // or of the next leg that still has one
                int leg = 0;
                while (legs_tried < queue->n_legs) {
                    leg = queue->order[(queue->first + legs_tried) % queue->n_legs];
#ifdef __HIP_DEVICE_COMPILE__
                    sq = (int64_t)atomicAdd(queue->counters + leg, 1);
#else
                    sq = (int64_t)(queue->counters[leg])++;
#endif
                    if (sq < queue->n_seq) break;
                    sq = -1;
                    legs_tried += 1;
                }
                if (sq < 0) break;      // every leg exhausted: the lane retires (reached by every lane: the counters only grow)
                gcp = leg_consts(queue->table, leg);
                affp = leg_affine(queue->table, leg);
                const int64_t c = sq * queue->n_legs + leg;
                io.pose = queue->pose + c * queue->pose_chain;
                io.angles = queue->angles + c * queue->ang_chain;
                io.fk = queue->fk ? queue->fk + c * io.n_frames * 27 : nullptr;
                io.status = queue->status ? queue->status + c * io.n_frames : nullptr;
                io.nfev = queue->nfev ? queue->nfev + c * io.n_frames : nullptr;
                io.init = queue->init ? queue->init + c * 7 : nullptr;
                for (int i = 0; i < GN; ++i) x[i] = io.init ? io.init[generic_link_dof(i)] : gc.seed[i];
                t = 0;
                new_solve = true;
            }
        }
        SEQIK_BLK_PASS();
        SEQIK_BLK_END_OF(BLK_LOOP);
        if (new_solve) {
            const double *org = io.pose + t * io.pose_frame;
            const double *kp = org + 4 * io.pose_row;  // the claw is the end effector (:587)
            if (aff.enabled) {
                for (int a = 0; a < 3; ++a) {
                    double al = (kp[a] - aff.fixed_coxa[a]) * aff.scale + aff.template_coxa[a];
                    target[a] = al - aff.template_coxa[a];
                }
            } else {
                for (int a = 0; a < 3; ++a) target[a] = kp[a] - org[a];
            }
            if constexpr (GROUPED) {
                const double xj = strictly_feasible(pick7(x, jm), gc.lb[jm], gc.ub[jm], 1e-10);
                double s1, c1;
                sincos_cw(xj, s1, c1);
                group8_gather(xj, x); group8_gather(s1, sn); group8_gather(c1, cs);
            } else {
                for (int i = 0; i < GN; ++i) {
                    x[i] = strictly_feasible(x[i], gc.lb[i], gc.ub[i], 1e-10);
                    sincos_cw(x[i], sn[i], cs[i]);
                }
            }
            generic_residual(gc, sn, cs, target, f);
            cost = 0.5 * dot3(f, f);
            nfev = 1;
            alpha = 0.0;
            status = STATUS_NONE;
            first_pass = true;
            new_solve = false;
        }

        bool finished = false;
        SEQIK_BLK_END_OF(BLK_NEW_SOLVE);
        if (WANT_DIAG || status == STATUS_NONE) {
            // ---- 2-point finite-difference Jacobian: column j perturbs joint j only ---------------
            double d[GN], diag_h[GN], g_h[GN], Jh[3][GN], ga[GN];
            if constexpr (GROUPED) {
                // One joint per lane of the group, and everything that is per-joint stays on that lane: the column of J, its
                // gradient entry, the Coleman-Li scaling, d, diag_h, g_h and the column of J_h.  Only what the 3 x 3 algebra and
                // the norms need from ALL joints is exchanged afterwards (J_h, d, diag_h, g_h, |g v|: 7 gathers instead of
                // forming 172 instructions' worth of per-joint values seven times over on every lane).  Same operations on
                // the same operands as the one-lane code below.
                const double xj = pick7(x, jm);
                double h = fd_step(xj, gc.lb[jm], gc.ub[jm]);
                double x1 = xj + h;
                double dx = x1 - xj;
                double s1, c1, v1[3], col[3];
                sincos_cw(x1, s1, c1);
                generic_claw_perturbed(gc, sn, cs, jm, s1, c1, v1);
                double inv_dx = div_(1.0, dx);
#pragma unroll
                for (int k = 0; k < 3; ++k) col[k] = ((v1[k] - target[k]) - f[k]) * inv_dx;
                SEQIK_BLK_END_OF(BLK_FD_JACOBIAN);
                const double gj = fma_(col[2], f[2], fma_(col[1], f[1], fma_(col[0], f[0], 0.0)));
                double vj, dvj;
                cl_scaling_gated(xj, gj, gc.lb[jm], gc.ub[jm], gc.gate_lb[jm], gc.gate_ub[jm], vj, dvj);
                const double rj = sqrt_pos_(vj);
                if (first_pass) {   // (wave-uniform: the group's lanes carry the same solve)
                    double tq[GN];
                    group8_gather(div_(xj, rj), tq);
                    double acc = gc.x_pre_sq;
                    for (int j = 0; j < GN; ++j) acc = fma_(tq[j], tq[j], acc);
                    acc = fma_(gc.x_suf, gc.x_suf, acc);
                    Delta = sqrt_(acc);
                    if (Delta == 0) Delta = 1.0;
                    first_pass = false;
                }
                const double dj = rj * 1.0;
                group8_gather(fabs(gj * vj), ga);
                group8_gather(dj, d);
                group8_gather(gj * dvj * 1.0, diag_h);
                group8_gather(dj * gj, g_h);
#pragma unroll
                for (int k = 0; k < 3; ++k) group8_gather(col[k] * dj, Jh[k]);
            } else {
                double J[3][GN], g[GN], v[GN], dv[GN];
                // the columns share the part of the chain behind the perturbed link (suf, at the base angles)
                double suf[GN][3], vb[3];
                generic_claw(gc, sn, cs, vb, suf);
                fd_column<0>(gc, x, sn, cs, suf, target, f, J); fd_column<1>(gc, x, sn, cs, suf, target, f, J);
                fd_column<2>(gc, x, sn, cs, suf, target, f, J); fd_column<3>(gc, x, sn, cs, suf, target, f, J);
                fd_column<4>(gc, x, sn, cs, suf, target, f, J); fd_column<5>(gc, x, sn, cs, suf, target, f, J);
                fd_column<6>(gc, x, sn, cs, suf, target, f, J);
                SEQIK_BLK_END_OF(BLK_FD_JACOBIAN);
                for (int j = 0; j < GN; ++j) {
                    g[j] = fma_(J[2][j], f[2], fma_(J[1][j], f[1], fma_(J[0][j], f[0], 0.0)));
                    cl_scaling_gated(x[j], g[j], gc.lb[j], gc.ub[j], gc.gate_lb[j], gc.gate_ub[j], v[j], dv[j]);
                }
                if (first_pass) {
                    double acc = gc.x_pre_sq;
                    for (int j = 0; j < GN; ++j) { double tj = div_(x[j], sqrt_pos_(v[j])); acc = fma_(tj, tj, acc); }
                    acc = fma_(gc.x_suf, gc.x_suf, acc);
                    Delta = sqrt_(acc);
                    if (Delta == 0) Delta = 1.0;
                    first_pass = false;
                }
                for (int j = 0; j < GN; ++j) {
                    ga[j] = fabs(g[j] * v[j]);
                    d[j] = sqrt_pos_(v[j]) * 1.0;
                    diag_h[j] = g[j] * dv[j] * 1.0;
                    g_h[j] = d[j] * g[j];
                }
                for (int k = 0; k < 3; ++k)
                    for (int j = 0; j < GN; ++j) Jh[k][j] = J[k][j] * d[j];
            }
            // ||g * v||_inf: the maximum is exact in any order (the products are >= +0 after fabs), so a tree of fmax gives
            // the oracle's running `if (a > g_norm) g_norm = a` value with a third of the dependent steps
            double g_norm = fmax(fmax(fmax(ga[0], ga[1]), fmax(ga[2], ga[3])), fmax(fmax(ga[4], ga[5]), fmax(ga[6], 0.0)));
            if (g_norm < gtol) status = 1;

            if (status != STATUS_NONE || nfev == gc.max_nfev) {
                finished = true;
            } else {
                double theta = fmax(0.995, 1 - g_norm);

                double p_h[GN], p[GN], step[GN], step_h[GN];
                SEQIK_BLK_END_OF(BLK_SCALING);
                solve_tr_woodbury(Jh, diag_h, f, Delta, alpha, p_h, GROUPED ? jm : -1);
                SEQIK_BLK_END_OF(BLK_TR_STEP);
                for (int j = 0; j < GN; ++j) p[j] = d[j] * p_h[j];
                double predicted_reduction = select_step7(x, Jh, diag_h, g_h, p, p_h, d, Delta, gc.lb, gc.ub, theta, step, step_h, GROUPED ? jm : -1);
                SEQIK_BLK_END_OF(BLK_IN_BOUNDS);
                double x_new[GN], sn_n[GN], cs_n[GN], f_new[3];
                if constexpr (GROUPED) {
                    const double xj = strictly_feasible0(pick7(x, jm) + pick7(step, jm), gc.lb[jm], gc.ub[jm], gc.lb_in[jm], gc.ub_in[jm]);
                    double s1, c1;
                    sincos_cw(xj, s1, c1);
                    group8_gather(xj, x_new); group8_gather(s1, sn_n); group8_gather(c1, cs_n);
                } else {
                    for (int j = 0; j < GN; ++j) {
                        x_new[j] = strictly_feasible0(x[j] + step[j], gc.lb[j], gc.ub[j], gc.lb_in[j], gc.ub_in[j]);
                        sincos_cw(x_new[j], sn_n[j], cs_n[j]);
                    }
                }
                SEQIK_BLK_END_OF(BLK_REFLECTIVE);
                generic_residual(gc, sn_n, cs_n, target, f_new);
                nfev += 1;
                double cost_new = 0.5 * dot3(f_new, f_new);
                SEQIK_BLK_END_OF(BLK_TRIAL_EVAL);
                double step_h_norm = vnorm7(step_h);
                double actual_reduction = cost - cost_new;
                // (the quotient is formed unconditionally and selected: a division inside a branch cannot overlap with the
                // norms next to it; same value wherever scipy uses it)
                const double ar_over_pr = div_(actual_reduction, predicted_reduction);
                double ratio;
                if (predicted_reduction > 0) ratio = ar_over_pr;
                else if (predicted_reduction == 0 && actual_reduction == 0) ratio = 1;
                else ratio = 0;
                double Delta_new = Delta;
                if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
                else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
                double step_norm = vnorm7(step);
                double xn = gc.x_pre_sq;
                for (int j = 0; j < GN; ++j) xn = fma_(x[j], x[j], xn);
                xn = fma_(gc.x_suf, gc.x_suf, xn);
                xn = sqrt_(xn);
                bool ftol_ok = (actual_reduction < ftol * cost) && (ratio > 0.25);
                bool xtol_ok = step_norm < xtol * (xtol + xn);
                if (ftol_ok && xtol_ok) status = 4;
                else if (ftol_ok) status = 2;
                else if (xtol_ok) status = 3;
                const double alpha_scaled = alpha * div_(Delta, Delta_new);
                if (status == STATUS_NONE) {
                    alpha = alpha_scaled;
                    Delta = Delta_new;
                }
                if (actual_reduction > 0) {
                    for (int j = 0; j < GN; ++j) { x[j] = x_new[j]; sn[j] = sn_n[j]; cs[j] = cs_n[j]; }
                    f[0] = f_new[0]; f[1] = f_new[1]; f[2] = f_new[2];
                    cost = cost_new;
                }
                if (!WANT_DIAG && (status != STATUS_NONE || nfev == gc.max_nfev)) finished = true;
            }
        } else {
            finished = true;
        }

        SEQIK_BLK_END_OF(BLK_POST_TRIAL);
        if (finished) {
            double *ang = io.angles + t * io.ang_frame;
            for (int j = 0; j < GN; ++j) ang[generic_link_dof(j) * io.ang_dof] = x[j];
            if constexpr (WANT_DIAG) {
                if (io.status) io.status[t] = (status == STATUS_NONE) ? 0 : status;
                if (io.nfev) io.nfev[t] = nfev;
            }
            if (io.fk) {
                const double *origin_p = aff.enabled ? aff.template_coxa : io.pose + t * io.pose_frame;
                const double origin[3] = {origin_p[0], origin_p[1], origin_p[2]};  // read before the first store (run_stage)
                double *fk = io.fk + t * 27;
                Frame e;
                double coxa_end[3], femur_end[3];
                generic_chain(gc, sn, cs, e, coxa_end, femur_end);
                for (int i = 0; i < 4; ++i)
                    for (int a = 0; a < 3; ++a) fk[3 * i + a] = 0.0 + origin[a];
                for (int a = 0; a < 3; ++a) {
                    fk[12 + a] = coxa_end[a] + origin[a];
                    fk[15 + a] = coxa_end[a] + origin[a];
                    fk[18 + a] = femur_end[a] + origin[a];
                    fk[21 + a] = e.t[a] + origin[a];
                    fk[24 + a] = (e.r[3 * a + 2] * gc.tz_claw + e.t[a]) + origin[a];
                }
            }
            t += 1;
            new_solve = true;
        }
        SEQIK_BLK_END_OF(BLK_FINISHED);
    }
    SEQIK_BLK_END(1);
#undef gc
#undef aff
}

}  // namespace seqik
