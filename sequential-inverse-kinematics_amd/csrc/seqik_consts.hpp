// seqik_consts.hpp -- host side: validation of SeqikLegParams and construction of the
// per-(leg, stage) device constants.  Mirrors what the reference derives per stage in
// KinematicChainSeq.create_leg_chain_stage_1..4 (seqikpy/kinematic_chain.py:152-421):
// which links are revolute, their axes, translations and bounds.
#pragma once
#include "seqik_core.hpp"
#include "seqik_generic.hpp"
#include "../../include/seqik.h"

namespace seqik {

// links per stage chain and, per stage, the link index of the first active link
static const int kStageLinks[4] = {4, 6, 8, 9};
static const int kFirstActive[4] = {1, 3, 5, 7};
static const int kNumActive[4] = {2, 2, 2, 1};
static const int kSeedOffset[4] = {0, 4, 10, 18};

// DOF whose bounds apply to link i of the stage-k chain (0 = base: unbounded; 8 = claw: +-pi)
// stage chains list the DOFs in order yaw, pitch, roll, CTr_pitch, CTr_roll, FTi, TiTa, except that
// stage 1 ends with CTr_pitch and stage 2 ends with FTi_pitch (kinematic_chain.py:188-196, 250-257).
inline int link_dof(int stage, int link)
{
    if (link == 0) return -1;
    if (stage == 1) { static const int m[4] = {-1, 0, 1, 3}; return m[link]; }
    if (stage == 2) { static const int m[6] = {-1, 0, 1, 2, 3, 5}; return m[link]; }
    if (link == 8) return -2;  // claw
    return link - 1;
}

inline void link_bounds(const SeqikLegParams &lp, int stage, int link, double &lb, double &ub)
{
    const double PI = 3.141592653589793;
    int dof = link_dof(stage, link);
    if (dof == -1) { lb = -__builtin_huge_val(); ub = __builtin_huge_val(); }
    else if (dof == -2) { lb = -PI; ub = PI; }
    else { lb = lp.bounds[dof][0]; ub = lp.bounds[dof][1]; }
}

// Limits the range-scaling-free square root of seqik_core.hpp cannot serve: a limit of exactly 0 is fine (the Coleman-Li
// distance next to it is 2^-1074, whose root is exact), ordinary magnitudes are fine (the distance is at least one ulp of
// the limit), a non-zero limit below 2^-600 in magnitude would put general subnormal distances under the root.  No joint
// has such a limit; refusing it keeps the floating-point contract closed.
inline bool limit_too_small(double b) { return b != 0.0 && fabs(b) < 0x1p-600; }

// scipy least_squares argument checks, in scipy's order, for the stages that will run.
inline int validate_leg(const SeqikLegParams &lp, int first_stage, int last_stage)
{
    for (int d = 0; d < 7; ++d)
        if (limit_too_small(lp.bounds[d][0]) || limit_too_small(lp.bounds[d][1])) return SEQIK_ERR_BAD_ARG;
    for (int stage = first_stage; stage <= last_stage; ++stage) {
        int n = kStageLinks[stage - 1];
        for (int i = 0; i < n; ++i) {
            double lb, ub;
            link_bounds(lp, stage, i, lb, ub);
            if (!(lb < ub)) return SEQIK_ERR_BAD_BOUNDS;
        }
        for (int i = 0; i < n; ++i) {
            double lb, ub;
            link_bounds(lp, stage, i, lb, ub);
            double x = lp.seeds[kSeedOffset[stage - 1] + i];
            if (!(x >= lb && x <= ub)) return SEQIK_ERR_X0_OUT_OF_BOUNDS;
        }
    }
    return SEQIK_OK;
}

inline void make_leg_consts(const SeqikLegParams &lp, const SeqikAffine *aff, LegConst &lc)
{
    lc.aff.enabled = aff ? 1 : 0;
    lc.aff.pad_ = 0;
    lc.aff.scale = aff ? aff->scale : 1.0;
    for (int a = 0; a < 3; ++a) {
        lc.aff.fixed_coxa[a] = aff ? aff->fixed_coxa[a] : 0.0;
        lc.aff.template_coxa[a] = aff ? aff->template_coxa[a] : 0.0;
    }
    for (int stage = 1; stage <= 4; ++stage) {
        StageConst &sc = lc.st[stage - 1];
        const int n = kStageLinks[stage - 1];
        const int a0 = kFirstActive[stage - 1];
        const int na = kNumActive[stage - 1];
        const double *seed = lp.seeds + kSeedOffset[stage - 1];
        sc.max_nfev = 100 * n;
        sc.pad_ = 0;
        // translations: CTr_pitch carries -coxa, FTi -femur, TiTa -tibia, claw -tarsus
        switch (stage) {
        case 1: sc.tz_a = 0.0; sc.tz_b = 0.0; sc.tz_last = -lp.seg[0]; break;
        case 2: sc.tz_a = 0.0; sc.tz_b = -lp.seg[0]; sc.tz_last = -lp.seg[1]; break;
        case 3: sc.tz_a = 0.0; sc.tz_b = -lp.seg[1]; sc.tz_last = -lp.seg[2]; break;
        default: sc.tz_a = -lp.seg[2]; sc.tz_b = 0.0; sc.tz_last = -lp.seg[3]; break;
        }
        for (int j = 0; j < 2; ++j) {
            if (j < na) {
                link_bounds(lp, stage, a0 + j, sc.lb[j], sc.ub[j]);
                sc.seed[j] = seed[a0 + j];
            } else {
                sc.lb[j] = -1.0; sc.ub[j] = 1.0; sc.seed[j] = 0.0;
            }
            // make_strictly_feasible's thresholds for rstep = 1e-10 (the same two operations the device function performs)
            sc.thr_lb[j] = 1e-10 * fmax(1.0, fabs(sc.lb[j]));
            sc.thr_ub[j] = 1e-10 * fmax(1.0, fabs(sc.ub[j]));
            // ... and its replacement values for rstep = 0 (the device function's own next_toward)
            sc.lb_in[j] = next_toward(sc.lb[j], sc.ub[j]);
            sc.ub_in[j] = next_toward(sc.ub[j], sc.lb[j]);
            // CL_scaling_vector's `isfinite(bound)` folded into the comparand of the gradient's sign test
            sc.gate_lb[j] = is_finite(sc.lb[j]) ? 0.0 : __builtin_nan("");
            sc.gate_ub[j] = is_finite(sc.ub[j]) ? 0.0 : __builtin_nan("");
            // where make_strictly_feasible(rstep = 1e-10) puts a warm start that lies within the threshold of a limit
            // (strictly_feasible_thr's operations, midpoint rule included), and the sin / cos there
            sc.thr_lb_g[j] = is_finite(sc.lb[j]) ? sc.thr_lb[j] : __builtin_nan("");
            sc.thr_ub_g[j] = is_finite(sc.ub[j]) ? sc.thr_ub[j] : __builtin_nan("");
            double lo = sc.lb[j] + sc.thr_lb[j], hi = sc.ub[j] - sc.thr_ub[j];
            if (lo < sc.lb[j] || lo > sc.ub[j]) lo = 0.5 * (sc.lb[j] + sc.ub[j]);
            if (hi < sc.lb[j] || hi > sc.ub[j]) hi = 0.5 * (sc.lb[j] + sc.ub[j]);
            sc.lb_out[j] = lo;
            sc.ub_out[j] = hi;
            sincos_cw(lo, sc.sc_lb[j][0], sc.sc_lb[j][1]);
            sincos_cw(hi, sc.sc_ub[j][0], sc.sc_ub[j][1]);
        }
        // inert entries of the start vector, made strictly feasible as scipy does, then the
        // partial sums of squares that ||x0 / sqrt(v)|| and ||x|| need (link order, from 0.0)
        double acc = 0.0;
        for (int i = 0; i < a0; ++i) {
            double lb, ub;
            link_bounds(lp, stage, i, lb, ub);
            double xi = strictly_feasible(seed[i], lb, ub, 1e-10);
            acc = fma_(xi, xi, acc);
        }
        sc.x_pre_sq = acc;
        {
            double lb, ub;
            link_bounds(lp, stage, n - 1, lb, ub);
            sc.x_suf = strictly_feasible(seed[n - 1], lb, ub, 1e-10);
        }
    }
}

// ---- generic chain (KinematicChainGeneric.create_leg_chain, kinematic_chain.py:464-530) -------------
// link i (1..7) of the generic chain carries DOF: roll, yaw, pitch, CTr_pitch, CTr_roll, FTi, TiTa
static const int kGenericLinkDof[GN] = {2, 0, 1, 3, 4, 5, 6};

inline void generic_link_bounds(const SeqikLegParams &lp, int link /*0..8*/, double &lb, double &ub)
{
    const double PI = 3.141592653589793;
    if (link == 0) { lb = -__builtin_huge_val(); ub = __builtin_huge_val(); }
    else if (link == 8) { lb = -PI; ub = PI; }
    else { lb = lp.bounds[kGenericLinkDof[link - 1]][0]; ub = lp.bounds[kGenericLinkDof[link - 1]][1]; }
}

// LegInvKinGeneric seeds the chain with initial_angles["stage_4"] (leg_inverse_kinematics.py:588),
// applied positionally: seeds[18 + i] is the start value of link i.
inline int validate_leg_generic(const SeqikLegParams &lp)
{
    for (int d = 0; d < 7; ++d)
        if (limit_too_small(lp.bounds[d][0]) || limit_too_small(lp.bounds[d][1])) return SEQIK_ERR_BAD_ARG;
    for (int i = 0; i < 9; ++i) {
        double lb, ub;
        generic_link_bounds(lp, i, lb, ub);
        if (!(lb < ub)) return SEQIK_ERR_BAD_BOUNDS;
    }
    for (int i = 0; i < 9; ++i) {
        double lb, ub;
        generic_link_bounds(lp, i, lb, ub);
        double x = lp.seeds[18 + i];
        if (!(x >= lb && x <= ub)) return SEQIK_ERR_X0_OUT_OF_BOUNDS;
    }
    return SEQIK_OK;
}

inline void make_generic_consts(const SeqikLegParams &lp, GenericConst &gc)
{
    const double tz[GN] = {0.0, 0.0, 0.0, -lp.seg[0], 0.0, -lp.seg[1], -lp.seg[2]};
    for (int i = 0; i < GN; ++i) {
        generic_link_bounds(lp, i + 1, gc.lb[i], gc.ub[i]);
        gc.lb_in[i] = next_toward(gc.lb[i], gc.ub[i]);
        gc.ub_in[i] = next_toward(gc.ub[i], gc.lb[i]);
        gc.gate_lb[i] = is_finite(gc.lb[i]) ? 0.0 : __builtin_nan("");
        gc.gate_ub[i] = is_finite(gc.ub[i]) ? 0.0 : __builtin_nan("");
        gc.seed[i] = lp.seeds[18 + 1 + i];
        gc.tz[i] = tz[i];
    }
    gc.tz_claw = -lp.seg[3];
    double lb, ub;
    generic_link_bounds(lp, 0, lb, ub);
    double xb = strictly_feasible(lp.seeds[18], lb, ub, 1e-10);
    gc.x_pre_sq = fma_(xb, xb, 0.0);
    generic_link_bounds(lp, 8, lb, ub);
    gc.x_suf = strictly_feasible(lp.seeds[26], lb, ub, 1e-10);
    gc.max_nfev = 900;
    gc.pad_ = 0;
}

}  // namespace seqik
