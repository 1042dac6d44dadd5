/*
 * ORACLE / TEST INFRASTRUCTURE ONLY -- this file is the parity checker, never
 * the product.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.
 *
 * Plain-C, scalar, double-precision restatement of the reference hot path
 *
 *   LegInvKinSeq.run_ik_and_fk / calculate_ik_stage
 *        (seqikpy/leg_inverse_kinematics.py:200-322, 324-403)
 *   KinematicChainSeq.create_leg_chain_stage_{1..4}, KinematicChainGeneric
 *        (seqikpy/kinematic_chain.py:152-421, 464-532)
 *   LegInvKinBase.calculate_ik / calculate_fk
 *        (seqikpy/leg_inverse_kinematics.py:62-77)
 *
 * plus the two third-party layers the reference delegates the arithmetic to,
 * neither of which lives under /root/reference:
 *
 *   ikpy==3.3.4 (pinned at setup.py:14): Chain.forward_kinematics /
 *        Chain.inverse_kinematics, URDFLink frame matrices
 *        (published semantics restated in SURVEY.md Appendix A);
 *   scipy.optimize.least_squares, method 'trf' with bounds, 2-point finite
 *        difference Jacobian, tr_solver 'exact', all-default tolerances
 *        (scipy 1.15.3: optimize/_lsq/least_squares.py, _lsq/trf.py
 *        ::trf_bounds/select_step, _lsq/common.py, _numdiff.py
 *        ::approx_derivative/_adjust_scheme_to_bounds/_dense_difference).
 *
 * It is deliberately GENERIC and un-optimised: chains are lists of n links
 * with full 4x4 homogeneous matrices multiplied left to right and the Jacobian has
 * all n columns.  The HIP kernel exploits the structure (2 / 2 / 2 / 1 effective
 * unknowns); this file does not.
 *
 * WHERE THE RESTATEMENT CHOOSES AN EQUIVALENT FORM (each kept switchable, see the
 * oracle_set_* hooks and tests/test_oracle_golden.py::test_algorithm_variants_kept_as_hooks):
 *   - inert (exactly-zero) Jacobian columns are removed from the trust-region
 *     sub-problem (ACTIVE SET note in oracle_least_squares);
 *   - the sub-problem itself -- scipy: SVD of the (3+n) x n augmented matrix --
 *     is solved in closed form for two unknowns (solve_tr_2x2) and through 3 x 3
 *     solves for the generic chain (solve_tr_woodbury); a one-sided Jacobi SVD
 *     (jacobi_svd) remains for one unknown and as the test-hook variant;
 *   - scipy's ten-iteration root search is short-cut when the Gauss-Newton step
 *     lies inside the trust region (bit-identical to the verbatim loop on every
 *     input tried, see solve_tr_2x2).
 * None of them moves the distances to the reference's shipped outputs in the third
 * significant digit.
 *
 * PARITY PINNING: tests/test_oracle_golden.py checks this file against
 *   (1) the shipped outputs of the reference pipeline
 *       data/anipose_220525_aJO_Fly001_001/pose-3d/{leg_joint_angles,
 *       forward_kinematics}.pkl (cut to tests/golden/ *.npz), and
 *   (2) outputs of the reference's own LegInvKinSeq/KinematicChainSeq source
 *       run in the build container over an ikpy shim + real scipy
 *       (oracle/gen_golden.py), including per-solve (status, nfev).
 *
 * Floating-point conventions (shared with the HIP kernel so that the two can
 * be compared BIT FOR BIT): every operation is a single IEEE-754 binary64
 * + - * / sqrt or an explicit fused multiply-add FMA(a, b, c) = round(a*b + c)
 * (compile with -ffp-contract=off so that nothing else is fused), sums run in
 * index order as acc = FMA(a_i, b_i, acc), several quotients by one denominator
 * are formed as products with its reciprocal, sin/cos come from oracle_sincos()
 * below (Cody-Waite reduction + the classic fdlibm kernel polynomials), never
 * from libm.  numpy / BLAS / LAPACK make the same kind of choices (FMA, SIMD
 * summation order) inside the reference, unspecified; the differences are at the
 * 1-ulp level.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define FMA(a, b, c) fma((a), (b), (c))

#define MAXN 9
#define NRES 3
#define MAXROWS (NRES + MAXN)

#define SEQIK_OK 0
#define SEQIK_ERR_X0_OUT_OF_BOUNDS (-2)  /* scipy: "Initial guess is outside of provided bounds" */
#define SEQIK_ERR_BAD_BOUNDS (-3)        /* scipy: "Each lower bound must be strictly less ..." */
#define SEQIK_ERR_BAD_ARG (-4)

/* How the exactly-zero Jacobian columns of the inert links enter the trust-region
 * sub-problem (see the ACTIVE SET note in oracle_least_squares). */
#define NULL_ACTIVE_ONLY 0 /* zero columns removed: scipy's own full-rank logic on the active set */
#define NULL_EXACT_ZERO 1  /* zero columns kept as exactly-zero singular values (m < n: never full rank) */
/* Branch statistics of the trust-region step (what share of the solves takes which path: used to reason about
 * divergence in the lane-per-chain kernels).  Compiled in only with -DORACLE_STATS (make stats ->
 * _build/libseqik_oracle_stats.so, used by tests/tools/branch_stats.py); not thread-safe.
 * [kind][0] calls, [1] Gauss-Newton step inside the region, [2] shortcut, [3] root loop entered, [4] sum of its
 * iterations, [5] select_step calls, [6] of which reflective, [8 + k] loops that ran k iterations;
 * kind 0 = stage 1 (rank deficient), 1 = stages 2-3. */
#ifdef ORACLE_STATS
static int g_stats_on = 0;
static long long g_stats[2][24];
void oracle_stats_reset(int on) { g_stats_on = on; memset(g_stats, 0, sizeof g_stats); }
void oracle_stats_get(long long *out) { memcpy(out, g_stats, sizeof g_stats); }
static int g_stats_kind = 0;
#define STAT(i) do { if (g_stats_on) g_stats[g_stats_kind][i] += 1; } while (0)
#define STAT_KIND(k) do { if (g_stats_on) g_stats_kind = (k); } while (0)
#define STAT_LOOP(n) do { if (g_stats_on) { g_stats[g_stats_kind][4] += (n); g_stats[g_stats_kind][8 + (n)] += 1; } } while (0)
#else
#define STAT(i) do { } while (0)
#define STAT_KIND(k) do { } while (0)
#define STAT_LOOP(n) do { (void)(n); } while (0)
#endif
static int g_tr2_shortcut = 1;        /* 0 = scipy's ten-iteration root search verbatim (test hook) */
static int g_closed_form_2x2 = 1;     /* 2 unknowns: closed-form trust-region step (solve_tr_2x2); 0 = one-sided Jacobi SVD (test hook) */
#define NULL_WOODBURY 2    /* active columns, m < n: the trust-region step from 3x3 solves instead of an SVD */

/* ------------------------------------------------------------------ */
/* sin / cos: fdlibm algorithm (Sun Microsystems, public algorithm):   */
/* k = rint(x * 2/pi); two-step Cody-Waite reduction to y0 + y1;        */
/* __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4].    */
/* Valid (< 1 ulp) for |x| < ~1e5, far beyond any joint bound (|x|<=pi) */
/* ------------------------------------------------------------------ */
static const double INVPIO2 = 6.36619772367581382433e-01;
static const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
static const double PIO2_2 = 6.07710050630396597660e-11;  /* second 33 bits */
static const double PIO2_2T = 2.02226624879595063154e-21; /* pi/2 - (PIO2_1 + PIO2_2) */

static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

void oracle_sincos(double x, double *sn, double *cs)
{
    double fn = rint(x * INVPIO2);
    /* two-step Cody-Waite reduction (always both steps: branch-free, exact when fn == 0) */
    double t = FMA(-fn, PIO2_1, x);
    double w = fn * PIO2_2;
    double r = t - w;
    w = FMA(fn, PIO2_2T, -((t - r) - w));
    double y0 = r - w;
    double y1 = (r - y0) - w;

    /* kernel sin(y0 + y1) */
    double z = y0 * y0;
    double v = z * y0;
    double rs = FMA(z, FMA(z, FMA(z, FMA(z, S6, S5), S4), S3), S2);
    double sa = FMA(-v, rs, 0.5 * y1);
    double sb = FMA(z, sa, -y1);
    double ks = y0 - FMA(-v, S1, sb);
    /* kernel cos(y0 + y1) */
    double zz = z * z;
    double p1 = FMA(z, FMA(z, C3, C2), C1);
    double p2 = FMA(z, FMA(z, C6, C5), C4);
    double rc = FMA(zz * zz, p2, z * p1);
    double hz = 0.5 * z;
    double wc = 1.0 - hz;
    double kc = wc + (((1.0 - wc) - hz) + FMA(z, rc, -(y0 * y1)));

    int q = ((int)fn) & 3;
    double s_out, c_out;
    if (q == 0) { s_out = ks; c_out = kc; }
    else if (q == 1) { s_out = kc; c_out = -ks; }
    else if (q == 2) { s_out = -ks; c_out = -kc; }
    else { s_out = -kc; c_out = ks; }
    *sn = s_out;
    *cs = c_out;
}

/* ------------------------------------------------------------------ */
/* IKPy layer: links, frame matrices, forward kinematics (Appendix A)  */
/* ------------------------------------------------------------------ */
typedef struct {
    int n;
    int is_origin[MAXN];
    int has_rot[MAXN];       /* joint_type == "revolute" and rotation is not None */
    double trans[MAXN][3];   /* origin_translation */
    double rpy[MAXN][3];     /* origin_orientation (roll, pitch, yaw) */
    double axis[MAXN][3];    /* rotation axis (un-normalised, may be 0) */
    double lb[MAXN], ub[MAXN];
    double base[MAXN][16];   /* T(trans) . H(RPY) -- filled by chain_prepare */
    int rtl;                 /* residual(): end-effector position by the right-to-left VECTOR recursion (generic chain) */
} oracle_chain;

static void mat4_identity(double *m)
{
    memset(m, 0, 16 * sizeof(double));
    m[0] = m[5] = m[10] = m[15] = 1.0;
}

/* c = a @ b, inner index ascending: fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 * b0))) */
static void mat4_mul(const double *a, const double *b, double *c)
{
    double out[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = a[4 * i + 0] * b[0 + j];
            acc = FMA(a[4 * i + 1], b[4 + j], acc);
            acc = FMA(a[4 * i + 2], b[8 + j], acc);
            acc = FMA(a[4 * i + 3], b[12 + j], acc);
            out[4 * i + j] = acc;
        }
    memcpy(c, out, sizeof(out));
}

static void mat3_mul(const double *a, const double *b, double *c)
{
    double out[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = a[3 * i + 0] * b[0 + j];
            acc = FMA(a[3 * i + 1], b[3 + j], acc);
            acc = FMA(a[3 * i + 2], b[6 + j], acc);
            out[3 * i + j] = acc;
        }
    memcpy(c, out, sizeof(out));
}

static void homogeneous_from_rot(const double *r, double *h)
{
    mat4_identity(h);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) h[4 * i + j] = r[3 * i + j];
}

/* IKPy geometry.rpy_matrix(roll, pitch, yaw) = Rz(yaw) . Ry(pitch) . Rx(roll) */
static void rpy_matrix(double roll, double pitch, double yaw, double *out)
{
    double s, c;
    oracle_sincos(roll, &s, &c);
    double rx[9] = {1, 0, 0, 0, c, -s, 0, s, c};
    oracle_sincos(pitch, &s, &c);
    double ry[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    oracle_sincos(yaw, &s, &c);
    double rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    double tmp[9];
    mat3_mul(ry, rx, tmp);
    mat3_mul(rz, tmp, out);
}

/* IKPy geometry.axis_rotation_matrix: un-normalised Rodrigues form */
static void axis_rotation_matrix(const double *ax, double theta, double *out)
{
    double s, c;
    oracle_sincos(theta, &s, &c);
    double x = ax[0], y = ax[1], z = ax[2];
    out[0] = x * x + (1 - x * x) * c;
    out[1] = x * y * (1 - c) - z * s;
    out[2] = x * z * (1 - c) + y * s;
    out[3] = x * y * (1 - c) + z * s;
    out[4] = y * y + (1 - y * y) * c;
    out[5] = y * z * (1 - c) - x * s;
    out[6] = x * z * (1 - c) - y * s;
    out[7] = y * z * (1 - c) + x * s;
    out[8] = z * z + (1 - z * z) * c;
}

static void chain_prepare(oracle_chain *ch)
{
    for (int i = 0; i < ch->n; ++i) {
        if (ch->is_origin[i]) { mat4_identity(ch->base[i]); continue; }
        double t[16], r[9], h[16];
        mat4_identity(t);
        t[3] = ch->trans[i][0]; t[7] = ch->trans[i][1]; t[11] = ch->trans[i][2];
        rpy_matrix(ch->rpy[i][0], ch->rpy[i][1], ch->rpy[i][2], r);
        homogeneous_from_rot(r, h);
        mat4_mul(t, h, ch->base[i]);
    }
}

static void link_frame_matrix(const oracle_chain *ch, int i, double theta, double *out)
{
    if (ch->is_origin[i] || !ch->has_rot[i]) { memcpy(out, ch->base[i], 16 * sizeof(double)); return; }
    double r[9], h[16];
    axis_rotation_matrix(ch->axis[i], theta, r);
    homogeneous_from_rot(r, h);
    mat4_mul(ch->base[i], h, out);
}

/* Chain.forward_kinematics(q, full_kinematics): frames[i] = cumulative product */
static void chain_fk(const oracle_chain *ch, const double *q, double *last, double *all /* nullable [n][16] */)
{
    double frame[16], link[16];
    mat4_identity(frame);
    for (int i = 0; i < ch->n; ++i) {
        link_frame_matrix(ch, i, q[i], link);
        mat4_mul(frame, link, frame);
        if (all) memcpy(all + 16 * i, frame, sizeof(frame));
    }
    if (last) memcpy(last, frame, sizeof(frame));
}

/* The end-effector position FK(x)[:3,3] = (M_0 M_1 ... M_{n-1}) e_4 evaluated RIGHT TO LEFT as a vector,
 *     v = e_4;  v <- M_i v  for i = n-1 .. 0,
 * instead of as the last column of the left-to-right matrix product of chain_fk: 12 multiply-adds per link instead of 64,
 * the same real number, another association order, so the last bits differ.  Used for the GENERIC chain only
 * (oracle_chain.rtl, set by build_generic_chain): its seven joint angles cannot be pinned against the reference anyway
 * -- the reference does not reproduce them itself (profiles/r04_perturbation_generic.json) -- while its claw is pinned to
 * 1e-6 (tests/test_generic.py).  The sequential stages keep chain_fk, which the shipped outputs pin.  Each row runs
 *     acc = M[r][3];  acc = FMA(M[r][2], v[2], acc);  acc = FMA(M[r][1], v[1], acc);  acc = FMA(M[r][0], v[0], acc)
 * (translation first, column index descending): with the exact zeros and ones of an axis rotation dropped this is what
 * csrc/seqik_generic.hpp writes out per axis (link_apply), two multiply-adds per changed coordinate.
 * oracle_set_generic_rtl(0) restores the matrix product (test hook, tests/test_oracle_golden.py). */
static int g_generic_rtl = 1;
void oracle_set_generic_rtl(int on) { g_generic_rtl = on; }
static void chain_end_effector_rtl(const oracle_chain *ch, const double *q, double *pos)
{
    double v[3] = {0.0, 0.0, 0.0}, m[16];
    for (int i = ch->n - 1; i >= 0; --i) {
        link_frame_matrix(ch, i, q[i], m);
        double w[3];
        for (int r = 0; r < 3; ++r) {
            double acc = m[4 * r + 3];
            acc = FMA(m[4 * r + 2], v[2], acc);
            acc = FMA(m[4 * r + 1], v[1], acc);
            acc = FMA(m[4 * r + 0], v[0], acc);
            w[r] = acc;
        }
        v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
    }
    pos[0] = v[0]; pos[1] = v[1]; pos[2] = v[2];
}

/* residual of Chain.inverse_kinematics: FK(x)[:3,3] - target */
static void residual(const oracle_chain *ch, const double *x, const double *target, double *f)
{
    if (ch->rtl) {
        double pos[3];
        chain_end_effector_rtl(ch, x, pos);
        f[0] = pos[0] - target[0];
        f[1] = pos[1] - target[1];
        f[2] = pos[2] - target[2];
        return;
    }
    double frame[16];
    chain_fk(ch, x, frame, NULL);
    f[0] = frame[3] - target[0];
    f[1] = frame[7] - target[1];
    f[2] = frame[11] - target[2];
}

/* ------------------------------------------------------------------ */
/* scipy layer                                                          */
/* ------------------------------------------------------------------ */
static double vnorm(const double *x, int n)
{
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc = FMA(x[i], x[i], acc);
    return sqrt(acc);
}

static double vdot(const double *a, const double *b, int n)
{
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc = FMA(a[i], b[i], acc);
    return acc;
}

static int in_bounds(const double *x, const double *lb, const double *ub, int n)
{
    for (int i = 0; i < n; ++i)
        if (!(x[i] >= lb[i] && x[i] <= ub[i])) return 0;
    return 1;
}

/* _lsq/common.py:make_strictly_feasible + find_active_constraints */
static void make_strictly_feasible(double *x, const double *lb, const double *ub, int n, double rstep)
{
    for (int i = 0; i < n; ++i) {
        int active = 0;
        if (rstep == 0.0) {
            if (x[i] <= lb[i]) active = -1;
            if (x[i] >= ub[i]) active = 1;
        } else {
            double lower_dist = x[i] - lb[i];
            double upper_dist = ub[i] - x[i];
            double lower_threshold = rstep * fmax(1.0, fabs(lb[i]));
            double upper_threshold = rstep * fmax(1.0, fabs(ub[i]));
            if (isfinite(lb[i]) && lower_dist <= fmin(upper_dist, lower_threshold)) active = -1;
            if (isfinite(ub[i]) && upper_dist <= fmin(lower_dist, upper_threshold)) active = 1;
        }
        double xn = x[i];
        if (active == -1)
            xn = (rstep == 0.0) ? nextafter(lb[i], ub[i]) : lb[i] + rstep * fmax(1.0, fabs(lb[i]));
        else if (active == 1)
            xn = (rstep == 0.0) ? nextafter(ub[i], lb[i]) : ub[i] - rstep * fmax(1.0, fabs(ub[i]));
        if (xn < lb[i] || xn > ub[i]) xn = 0.5 * (lb[i] + ub[i]);
        x[i] = xn;
    }
}

/* _numdiff.py: approx_derivative(method='2-point', bounds) -> dense J[NRES][n] */
static void approx_jacobian(const oracle_chain *ch, const double *x, const double *f0,
                            const double *target, double J[NRES][MAXN])
{
    const double RSTEP = 1.4901161193847656e-08; /* sqrt(eps) */
    int n = ch->n;
    double x1[MAXN];
    memcpy(x1, x, n * sizeof(double));
    for (int i = 0; i < n; ++i) {
        double sign = (x[i] >= 0.0) ? 1.0 : -1.0;
        double h = RSTEP * sign * fmax(1.0, fabs(x[i]));
        /* _adjust_scheme_to_bounds, '1-sided', num_steps = 1 */
        double lower_dist = x[i] - ch->lb[i];
        double upper_dist = ch->ub[i] - x[i];
        double xh = x[i] + h;
        int violated = (xh < ch->lb[i]) || (xh > ch->ub[i]);
        int fitting = fabs(h) <= fmax(lower_dist, upper_dist);
        if (violated && fitting) h = -h;
        else if (!fitting) h = (upper_dist >= lower_dist) ? upper_dist : -lower_dist;
        x1[i] = x[i] + h;
        double dx = x1[i] - x[i];
        double f1[NRES];
        residual(ch, x1, target, f1);
        double inv_dx = 1.0 / dx;
        for (int k = 0; k < NRES; ++k) J[k][i] = (f1[k] - f0[k]) * inv_dx;
        x1[i] = x[i];
    }
}

/* TEST HOOK (off): geometric (analytic) Jacobian of the end-effector position: column i = a_i x (p_ee - o_i) for a
 * revolute link whose axis a_i passes through o_i (both in the base frame), 0 for links without a rotation.
 * Measured and REJECTED as a replacement of the finite differences: on every well-posed frame of the fixtures the
 * angles and evaluation counts are the same to three digits, but the round-off of the 2-point differences is what
 * lets the solver leave the kinematic singularity of the anipose LF episode -- with the exact Jacobian it stays
 * stuck until frame 304 (reference: 287, finite differences: 286-288), 2 rad off on 16 more frames. */
static int g_analytic_jac = 0;
void oracle_set_analytic_jacobian(int on) { g_analytic_jac = on; }
static void analytic_jacobian(const oracle_chain *ch, const double *x, double J[NRES][MAXN])
{
    int n = ch->n;
    double frame[16], link[16], before[MAXN][16];
    mat4_identity(frame);
    for (int i = 0; i < n; ++i) {
        mat4_mul(frame, ch->base[i], before[i]);
        link_frame_matrix(ch, i, x[i], link);
        mat4_mul(frame, link, frame);
    }
    double pe[3] = {frame[3], frame[7], frame[11]};
    for (int i = 0; i < n; ++i) {
        if (ch->is_origin[i] || !ch->has_rot[i]) { for (int k = 0; k < NRES; ++k) J[k][i] = 0.0; continue; }
        const double *b = before[i];
        const double *ax = ch->axis[i];
        double aw[3], d[3];
        for (int r = 0; r < 3; ++r) aw[r] = FMA(b[4 * r + 2], ax[2], FMA(b[4 * r + 1], ax[1], b[4 * r] * ax[0]));
        d[0] = pe[0] - b[3]; d[1] = pe[1] - b[7]; d[2] = pe[2] - b[11];
        J[0][i] = FMA(aw[1], d[2], -(aw[2] * d[1]));
        J[1][i] = FMA(aw[2], d[0], -(aw[0] * d[2]));
        J[2][i] = FMA(aw[0], d[1], -(aw[1] * d[0]));
    }
}

/* _lsq/common.py:CL_scaling_vector */
static void cl_scaling_vector(const double *x, const double *g, const double *lb, const double *ub,
                              int n, double *v, double *dv)
{
    for (int i = 0; i < n; ++i) {
        v[i] = 1.0; dv[i] = 0.0;
        if (g[i] < 0 && isfinite(ub[i])) { v[i] = ub[i] - x[i]; dv[i] = -1.0; }
        if (g[i] > 0 && isfinite(lb[i])) { v[i] = x[i] - lb[i]; dv[i] = 1.0; }
    }
}

/* One-sided (Hestenes) Jacobi SVD of a[rows][n] (stand-in for LAPACK gesdd in
 * scipy.linalg.svd(J_augmented, full_matrices=False)).  Outputs singular values
 * sorted descending, U columns (rows x n), V (n x n, columns = right vectors). */
static void jacobi_svd(int rows, int n, double a[MAXROWS][MAXN], double *s,
                       double u[MAXROWS][MAXN], double v[MAXN][MAXN])
{
    const double TOL = 8.881784197001252e-16; /* 4 eps */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < rows; ++i) {
                    alpha = FMA(a[i][p], a[i][p], alpha);
                    beta = FMA(a[i][q], a[i][q], beta);
                    gamma = FMA(a[i][p], a[i][q], gamma);
                }
                if (gamma == 0.0) continue;
                if (fabs(gamma) <= TOL * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = 1.0 / (fabs(zeta) + sqrt(FMA(zeta, zeta, 1.0)));
                if (zeta < 0.0) t = -t;
                double c = 1.0 / sqrt(FMA(t, t, 1.0));
                double sn = c * t;
                for (int i = 0; i < rows; ++i) {
                    double ap = a[i][p], aq = a[i][q];
                    a[i][p] = FMA(c, ap, -(sn * aq));
                    a[i][q] = FMA(sn, ap, c * aq);
                }
                for (int i = 0; i < n; ++i) {
                    double vp = v[i][p], vq = v[i][q];
                    v[i][p] = FMA(c, vp, -(sn * vq));
                    v[i][q] = FMA(sn, vp, c * vq);
                }
            }
        if (!rotated) break;
    }
    double sv[MAXN];
    int order[MAXN];
    for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int i = 0; i < rows; ++i) acc = FMA(a[i][j], a[i][j], acc);
        sv[j] = sqrt(acc);
        order[j] = j;
    }
    /* stable insertion sort, descending */
    for (int i = 1; i < n; ++i) {
        int k = order[i];
        int j = i - 1;
        while (j >= 0 && sv[order[j]] < sv[k]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
    double vtmp[MAXN][MAXN];
    for (int jj = 0; jj < n; ++jj) {
        int j = order[jj];
        s[jj] = sv[j];
        double inv_sv = (sv[j] > 0.0) ? 1.0 / sv[j] : 0.0;
        for (int i = 0; i < rows; ++i) u[i][jj] = a[i][j] * inv_sv;
        for (int i = 0; i < n; ++i) vtmp[i][jj] = v[i][j];
    }
    memcpy(v, vtmp, sizeof(vtmp));
}

/* _lsq/common.py:phi_and_derivative (inner function of solve_lsq_trust_region):
 *   phi = ||suf / (s^2 + alpha)|| - Delta,  phi' = -sum(suf^2 / (s^2 + alpha)^3) / ||...||.
 * Returns phi and the Newton ratio phi / phi' = -phi * ||..|| / sum(..); the three quotients
 * by (s_i^2 + alpha) share one reciprocal. */
static void phi_and_ratio(double alpha, const double *suf, const double *s, int n, double Delta,
                          double *phi, double *ratio)
{
    double tmp[MAXN];
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        double r = 1.0 / FMA(s[i], s[i], alpha);
        tmp[i] = suf[i] * r;
        acc = FMA(tmp[i] * tmp[i], r, acc);
    }
    double p_norm = vnorm(tmp, n);
    *phi = p_norm - Delta;
    *ratio = -(*phi * p_norm) / acc;
}

/* _lsq/common.py:solve_lsq_trust_region(n, m, uf, s, V, Delta, initial_alpha), including the
 * full-rank Gauss-Newton shortcut.  It is called on the ACTIVE columns only (see the note in
 * oracle_least_squares); force_deficient = scipy's m < n branch (never full rank). */
static void solve_lsq_trust_region(int n, int m, const double *uf, const double *s, double v[MAXN][MAXN],
                                   double Delta, double *alpha_io, double *p, int force_deficient)
{
    const double EPS = 2.220446049250313e-16;
    double suf[MAXN], tmp[MAXN];
    for (int i = 0; i < n; ++i) suf[i] = s[i] * uf[i];
    int full_rank = 0;
    if (m >= n && !force_deficient) {
        double threshold = EPS * m * s[0];
        full_rank = s[n - 1] > threshold;
    }
    if (full_rank) {
        for (int i = 0; i < n; ++i) tmp[i] = uf[i] / s[i];
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = FMA(v[i][k], tmp[k], acc);
            p[i] = -acc;
        }
        if (vnorm(p, n) <= Delta) { *alpha_io = 0.0; return; }
    }
    const double inv_Delta = 1.0 / Delta;
    double alpha_upper = vnorm(suf, n) * inv_Delta;
    double alpha_lower = 0.0;
    if (full_rank) {
        double phi, ratio;
        phi_and_ratio(0.0, suf, s, n, Delta, &phi, &ratio);
        alpha_lower = -ratio;  /* -phi / phi' */
    }
    double alpha = *alpha_io;
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double phi, ratio;
        phi_and_ratio(alpha, suf, s, n, Delta, &phi, &ratio);
        if (phi < 0) alpha_upper = alpha;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio * inv_Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    for (int i = 0; i < n; ++i) tmp[i] = suf[i] / FMA(s[i], s[i], alpha);
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = FMA(v[i][k], tmp[k], acc);
        p[i] = -acc;
    }
    double scale = Delta / vnorm(p, n);
    for (int i = 0; i < n; ++i) p[i] = p[i] * scale;
    *alpha_io = alpha;
}

/* ------------------------------------------------------------------ */
/* Trust-region step without an SVD (generic chain: n = 7 unknowns, m = 3 residuals).
 *
 * scipy's solve_lsq_trust_region works on the SVD of A = [[J_h], [diag(sqrt(diag_h))]] ((3 + n) x n).  All it
 * needs from it is  p(alpha) = -(A^T A + alpha I)^-1 J_h^T f,  ||p||  and  phi'(alpha) = -p^T (A^T A + alpha I)^-1 p / ||p||.
 * With B = diag(diag_h + alpha) and A^T A = diag(diag_h) + J_h^T J_h the push-through identity gives
 *     (B + J_h^T J_h)^-1 r = W r - W J_h^T (I_3 + J_h W J_h^T)^-1 J_h W r,     W = B^-1,
 * i.e. one symmetric 3 x 3 inverse per alpha (by cofactors: no pivoting, valid for indefinite matrices too,
 * which the last, possibly negative alpha of scipy's m < n root search can produce).  Same root search, same
 * bracket updates, same final rescaling of p to the trust radius (_lsq/common.py:solve_lsq_trust_region,
 * m < n branch); alpha_upper = ||s * uf|| / Delta = ||J_h^T f|| / Delta.  The 10 x 7 one-sided Jacobi SVD this
 * replaces was > 95 % of the generic kernel's instructions. */
static void sym3_inverse(const double m[6] /* 00 01 02 11 12 22 */, double inv[6])
{
    double c00 = FMA(m[3], m[5], -(m[4] * m[4]));
    double c01 = FMA(m[2], m[4], -(m[1] * m[5]));
    double c02 = FMA(m[1], m[4], -(m[2] * m[3]));
    double c11 = FMA(m[0], m[5], -(m[2] * m[2]));
    double c12 = FMA(m[1], m[2], -(m[0] * m[4]));
    double c22 = FMA(m[0], m[3], -(m[1] * m[1]));
    double det = FMA(m[2], c02, FMA(m[1], c01, m[0] * c00));
    double r = 1.0 / det;
    inv[0] = c00 * r; inv[1] = c01 * r; inv[2] = c02 * r; inv[3] = c11 * r; inv[4] = c12 * r; inv[5] = c22 * r;
}

/* q = (B + J_h^T J_h)^-1 r  given W = 1 / (diag_h + alpha) and Minv = (I + J_h W J_h^T)^-1 */
static void woodbury_solve(int n, double Jh[NRES][MAXN], const double *W, const double Minv[6], const double *r, double *q)
{
    double wr[MAXN], t[3], y[3];
    for (int c = 0; c < n; ++c) wr[c] = W[c] * r[c];
    for (int k = 0; k < 3; ++k) {
        double acc = 0.0;
        for (int c = 0; c < n; ++c) acc = FMA(Jh[k][c], wr[c], acc);
        t[k] = acc;
    }
    y[0] = FMA(Minv[2], t[2], FMA(Minv[1], t[1], Minv[0] * t[0]));
    y[1] = FMA(Minv[4], t[2], FMA(Minv[3], t[1], Minv[1] * t[0]));
    y[2] = FMA(Minv[5], t[2], FMA(Minv[4], t[1], Minv[2] * t[0]));
    for (int c = 0; c < n; ++c) {
        double jy = FMA(Jh[2][c], y[2], FMA(Jh[1][c], y[1], Jh[0][c] * y[0]));
        q[c] = FMA(-W[c], jy, wr[c]);
    }
}

/* p(alpha) (un-negated: pp = (B + J_h^T J_h)^-1 J_h^T f), phi and the Newton ratio phi / phi'.
 * Two algebraically identical forms.  form 0 (round 3): two general solves with woodbury_solve, one for J_h^T f and one for
 * pp.  form 1 (default): the push-through identity applied once more,
 *     pp = W J_h^T y,  y = (I + J_h W J_h^T)^-1 f            (the right-hand side IS J_h^T f: no 7-vector product in front)
 *     pp^T (B + J_h^T J_h)^-1 pp = pp^T W pp - u^T (I + J_h W J_h^T)^-1 u,  u = J_h W pp
 * -- 37 + 54 multiply-adds instead of 65 + 72 per evaluation.  The angles of the generic chain are not pinned by the reference
 * (see chain_end_effector_rtl), so the form is this restatement's to choose; the kernel mirrors form 1 operation for
 * operation.  oracle_set_woodbury_form(0) restores form 0 (test hook). */
static int g_woodbury_form = 1;
void oracle_set_woodbury_form(int form) { g_woodbury_form = form; }
static void woodbury_phi(int n, double Jh[NRES][MAXN], const double *diag_h, const double *rhs /* J_h^T f */, const double *f,
                         double alpha, double Delta, double *pp, double *phi, double *ratio)
{
    double W[MAXN], M[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 1.0}, Minv[6], q[MAXN];
    for (int c = 0; c < n; ++c) W[c] = 1.0 / (diag_h[c] + alpha);
    for (int c = 0; c < n; ++c) {
        double w0 = W[c] * Jh[0][c], w1 = W[c] * Jh[1][c], w2 = W[c] * Jh[2][c];
        M[0] = FMA(w0, Jh[0][c], M[0]); M[1] = FMA(w0, Jh[1][c], M[1]); M[2] = FMA(w0, Jh[2][c], M[2]);
        M[3] = FMA(w1, Jh[1][c], M[3]); M[4] = FMA(w1, Jh[2][c], M[4]); M[5] = FMA(w2, Jh[2][c], M[5]);
    }
    sym3_inverse(M, Minv);
    if (g_woodbury_form == 0) {
        woodbury_solve(n, Jh, W, Minv, rhs, pp);
        if (phi) {
            double p_norm = vnorm(pp, n);
            woodbury_solve(n, Jh, W, Minv, pp, q);
            double acc = vdot(pp, q, n);
            *phi = p_norm - Delta;
            *ratio = -(*phi * p_norm) / acc;
        }
        return;
    }
    double y[3];
    y[0] = FMA(Minv[2], f[2], FMA(Minv[1], f[1], Minv[0] * f[0]));
    y[1] = FMA(Minv[4], f[2], FMA(Minv[3], f[1], Minv[1] * f[0]));
    y[2] = FMA(Minv[5], f[2], FMA(Minv[4], f[1], Minv[2] * f[0]));
    for (int c = 0; c < n; ++c) {
        double z = FMA(Jh[2][c], y[2], FMA(Jh[1][c], y[1], Jh[0][c] * y[0]));
        pp[c] = W[c] * z;
    }
    if (phi) {
        double p_norm = vnorm(pp, n);
        double wa[MAXN], u[3], t[3];
        for (int c = 0; c < n; ++c) wa[c] = W[c] * pp[c];
        double s1 = vdot(wa, pp, n);
        for (int k = 0; k < 3; ++k) {
            double acc = 0.0;
            for (int c = 0; c < n; ++c) acc = FMA(Jh[k][c], wa[c], acc);
            u[k] = acc;
        }
        t[0] = FMA(Minv[2], u[2], FMA(Minv[1], u[1], Minv[0] * u[0]));
        t[1] = FMA(Minv[4], u[2], FMA(Minv[3], u[1], Minv[1] * u[0]));
        t[2] = FMA(Minv[5], u[2], FMA(Minv[4], u[1], Minv[2] * u[0]));
        double s2 = FMA(u[2], t[2], FMA(u[1], t[1], u[0] * t[0]));
        double acc = s1 - s2;
        *phi = p_norm - Delta;
        *ratio = -(*phi * p_norm) / acc;
    }
}

static void solve_tr_woodbury(int n, double Jh[NRES][MAXN], const double *diag_h, const double *f, double Delta,
                              double *alpha_io, double *p)
{
    double rhs[MAXN], pp[MAXN];
    for (int c = 0; c < n; ++c) rhs[c] = FMA(Jh[2][c], f[2], FMA(Jh[1][c], f[1], Jh[0][c] * f[0]));
    const double inv_Delta = 1.0 / Delta;
    double alpha_upper = vnorm(rhs, n) * inv_Delta;
    double alpha_lower = 0.0;
    double alpha = *alpha_io;
    if (alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    if (g_tr2_shortcut) {  /* same shortcut as in solve_tr_2x2 (m < n: never full rank) */
        double au = alpha_upper, a_k = alpha;
        for (int it = 0; it < 10; ++it) {
            if (a_k < 0.0 || a_k > au) a_k = fmax(0.001 * au, 0.0);
            au = a_k;
            if (it < 9) a_k = -1.0;
        }
        double phi, ratio;
        woodbury_phi(n, Jh, diag_h, rhs, f, a_k, Delta, pp, &phi, &ratio);
        if (phi < 0 && !(fabs(phi) < 0.01 * Delta)) {
            alpha = a_k - (phi + Delta) * ratio * inv_Delta;
            goto final_step;
        }
    }
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double phi, ratio;
        woodbury_phi(n, Jh, diag_h, rhs, f, alpha, Delta, pp, &phi, &ratio);
        if (phi < 0) alpha_upper = alpha;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio * inv_Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
final_step:
    woodbury_phi(n, Jh, diag_h, rhs, f, alpha, Delta, pp, NULL, NULL);
    double scale = Delta / vnorm(pp, n);
    for (int c = 0; c < n; ++c) p[c] = -(pp[c] * scale);
    *alpha_io = alpha;
}

/* ------------------------------------------------------------------ */
/* Trust-region step of a 2-unknown problem in closed form (sequential stages 1-3).
 * A^T A = J_h^T J_h + diag(diag_h) is 2 x 2: (A^T A + alpha I)^-1 by cofactors, the extreme singular values of A
 * for scipy's rank test from the eigenvalues of A^T A (lambda_max = tr/2 + sqrt(((a-c)/2)^2 + b^2),
 * lambda_min = det / lambda_max).  Same root search as solve_lsq_trust_region. */
void oracle_set_tr2_shortcut(int on) { g_tr2_shortcut = on; }
static void tr2_apply(double a, double b, double c, double alpha, const double *r, double *q, double *inv_det_out)
{
    double aa = a + alpha, cc = c + alpha;
    double det = FMA(aa, cc, -(b * b));
    double inv = 1.0 / det;
    q[0] = FMA(cc, r[0], -(b * r[1])) * inv;
    q[1] = FMA(aa, r[1], -(b * r[0])) * inv;
    if (inv_det_out) *inv_det_out = inv;
}

static void tr2_phi(double a, double b, double c, double alpha, const double *r, double Delta, double *pp,
                    double *phi, double *ratio)
{
    double q[2];
    tr2_apply(a, b, c, alpha, r, pp, NULL);
    double p_norm = sqrt(FMA(pp[1], pp[1], pp[0] * pp[0]));
    tr2_apply(a, b, c, alpha, pp, q, NULL);
    double acc = FMA(pp[1], q[1], pp[0] * q[0]);
    *phi = p_norm - Delta;
    *ratio = -(*phi * p_norm) / acc;
}

static void solve_tr_2x2(double Jh[NRES][MAXN], const double *diag_h, const double *f, double Delta,
                         double *alpha_io, double *p, int force_deficient)
{
    double a = diag_h[0], b = 0.0, c = diag_h[1], r[2], pp[2];
    for (int k = 0; k < NRES; ++k) { a = FMA(Jh[k][0], Jh[k][0], a); b = FMA(Jh[k][0], Jh[k][1], b); c = FMA(Jh[k][1], Jh[k][1], c); }
    r[0] = FMA(Jh[2][0], f[2], FMA(Jh[1][0], f[1], Jh[0][0] * f[0]));
    r[1] = FMA(Jh[2][1], f[2], FMA(Jh[1][1], f[1], Jh[0][1] * f[0]));
    int full_rank = 0;
    STAT_KIND(force_deficient ? 0 : 1);
    STAT(0);
    if (!force_deficient) {
        double h = 0.5 * (a - c);
        double lmax = FMA(0.5, a + c, sqrt(FMA(h, h, b * b)));
        double lmin = FMA(a, c, -(b * b)) / lmax;
        full_rank = lmin > 4.437342591868191e-31 * lmax;  /* (3 eps)^2: s_min > eps * m * s_max */
    }
    if (full_rank) {
        tr2_apply(a, b, c, 0.0, r, pp, NULL);
        if (sqrt(FMA(pp[1], pp[1], pp[0] * pp[0])) <= Delta) { STAT(1); p[0] = -pp[0]; p[1] = -pp[1]; *alpha_io = 0.0; return; }
    }
    const double inv_Delta = 1.0 / Delta;
    double alpha_upper = sqrt(FMA(r[1], r[1], r[0] * r[0])) * inv_Delta;
    double alpha_lower = 0.0;
    if (full_rank) {
        double phi, ratio;
        tr2_phi(a, b, c, 0.0, r, Delta, pp, &phi, &ratio);
        alpha_lower = -ratio;
    }
    double alpha = *alpha_io;
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    if (g_tr2_shortcut && !full_rank) {
        /* SHORTCUT of the m < n root search.  When the Gauss-Newton step lies well inside the trust region
         * (||p(0+)|| <= 0.99 Delta: 89 % of the stage-1 solves) scipy's ten iterations are ten resets
         * alpha <- 0.001 alpha_upper followed by one Newton step from the last, tiny alpha: phi < 0 at every
         * alpha > 0 (||p(alpha)|| decreases in alpha), so alpha_upper <- alpha each time; 1 / ||p(alpha)|| is concave,
         * so the Newton step from the right of its (negative) root lands left of the root, i.e. below
         * alpha_lower = 0, which stays 0; and |phi| >= 0.01 Delta never stops the loop.  The reset sequence needs no
         * evaluation, so only the last alpha is evaluated.  This is exact in exact arithmetic; in binary64 it gave
         * the same bits as the verbatim loop on every one of 42 576 frames (447 071 uses; tests/test_oracle_golden.py
         * keeps checking).  Whenever the test below fails the verbatim loop runs. */
        double au = alpha_upper, a_k = alpha;
        for (int it = 0; it < 10; ++it) {
            if (a_k < 0.0 || a_k > au) a_k = fmax(0.001 * au, 0.0);
            au = a_k;
            if (it < 9) a_k = -1.0;  /* stands for "below alpha_lower": reset on the next iteration */
        }
        double phi, ratio;
        tr2_phi(a, b, c, a_k, r, Delta, pp, &phi, &ratio);
        if (phi < 0 && !(fabs(phi) < 0.01 * Delta)) {
            alpha = a_k - (phi + Delta) * ratio * inv_Delta;
            STAT(2);
            goto final_step;
        }
    }
    STAT(3);
    {
        int n_it = 0;
        for (int it = 0; it < 10; ++it) {
            if (alpha < alpha_lower || alpha > alpha_upper)
                alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
            double phi, ratio;
            tr2_phi(a, b, c, alpha, r, Delta, pp, &phi, &ratio);
            n_it += 1;
            if (phi < 0) alpha_upper = alpha;
            alpha_lower = fmax(alpha_lower, alpha - ratio);
            alpha -= (phi + Delta) * ratio * inv_Delta;
            if (fabs(phi) < 0.01 * Delta) break;
        }
        STAT_LOOP(n_it);
    }
final_step:
    tr2_apply(a, b, c, alpha, r, pp, NULL);
    double scale = Delta / sqrt(FMA(pp[1], pp[1], pp[0] * pp[0]));
    p[0] = -(pp[0] * scale); p[1] = -(pp[1] * scale);
    *alpha_io = alpha;
}

/* _lsq/common.py:step_size_to_bound */
static double step_size_to_bound(const double *x, const double *s, const double *lb, const double *ub,
                                 int n, int *hits)
{
    double steps[MAXN];
    double min_step = INFINITY;
    for (int i = 0; i < n; ++i) {
        if (s[i] != 0.0) {
            double inv_s = 1.0 / s[i];
            steps[i] = fmax((lb[i] - x[i]) * inv_s, (ub[i] - x[i]) * inv_s);
        }
        else steps[i] = INFINITY;
        if (steps[i] < min_step) min_step = steps[i];
    }
    if (hits)
        for (int i = 0; i < n; ++i) {
            int sg = (s[i] > 0) - (s[i] < 0);
            hits[i] = (steps[i] == min_step) ? sg : 0;
        }
    return min_step;
}

/* _lsq/common.py:intersect_trust_region -> positive root */
static double intersect_trust_region_pos(const double *x, const double *s, int n, double Delta)
{
    double a = vdot(s, s, n);
    double b = vdot(x, s, n);
    double c = FMA(-Delta, Delta, vdot(x, x, n));
    double d = sqrt(FMA(b, b, -(a * c)));
    double q = -(b + copysign(d, b));
    double t1 = q / a;
    double t2 = c / q;
    return (t1 < t2) ? t2 : t1;
}

static void mat_vec(double Jh[NRES][MAXN], const double *s, int n, double *out)
{
    for (int k = 0; k < NRES; ++k) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc = FMA(Jh[k][i], s[i], acc);
        out[k] = acc;
    }
}

/* _lsq/common.py:evaluate_quadratic (1-D s) */
static double evaluate_quadratic(double Jh[NRES][MAXN], const double *g, const double *s,
                                 const double *diag, int n)
{
    double Js[NRES];
    mat_vec(Jh, s, n, Js);
    double q = vdot(Js, Js, NRES);
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc = FMA(s[i] * diag[i], s[i], acc);
    q = q + acc;
    double l = vdot(s, g, n);
    return FMA(0.5, q, l);
}

/* _lsq/common.py:build_quadratic_1d */
static void build_quadratic_1d(double Jh[NRES][MAXN], const double *g, const double *s,
                               const double *diag, const double *s0, int n,
                               double *a_out, double *b_out, double *c_out)
{
    double v[NRES];
    mat_vec(Jh, s, n, v);
    double a = vdot(v, v, NRES);
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc = FMA(s[i] * diag[i], s[i], acc);
    a = a + acc;
    a = a * 0.5;
    double b = vdot(g, s, n);
    if (s0) {
        double u[NRES];
        mat_vec(Jh, s0, n, u);
        b = b + vdot(u, v, NRES);
        double c = FMA(0.5, vdot(u, u, NRES), vdot(g, s0, n));
        acc = 0.0;
        for (int i = 0; i < n; ++i) acc = FMA(s0[i] * diag[i], s[i], acc);
        b = b + acc;
        acc = 0.0;
        for (int i = 0; i < n; ++i) acc = FMA(s0[i] * diag[i], s0[i], acc);
        c = FMA(0.5, acc, c);
        *c_out = c;
    }
    *a_out = a;
    *b_out = b;
}

/* _lsq/common.py:minimize_quadratic_1d */
static double minimize_quadratic_1d(double a, double b, double lb, double ub, double c, double *y_out)
{
    double t[3];
    int nt = 2;
    t[0] = lb; t[1] = ub;
    if (a != 0) {
        double extremum = -0.5 * b / a;
        if (lb < extremum && extremum < ub) t[nt++] = extremum;
    }
    int best = 0;
    double ybest = FMA(t[0], FMA(a, t[0], b), c);
    for (int i = 1; i < nt; ++i) {
        double y = FMA(t[i], FMA(a, t[i], b), c);
        if (y < ybest) { ybest = y; best = i; }
    }
    *y_out = ybest;
    return t[best];
}

/* _lsq/trf.py:select_step */
static double select_step(const double *x, double Jh[NRES][MAXN], const double *diag_h, const double *g_h,
                          double *p, double *p_h, const double *d, double Delta,
                          const double *lb, const double *ub, double theta, int n,
                          double *step, double *step_h)
{
    double xp[MAXN];
    for (int i = 0; i < n; ++i) xp[i] = x[i] + p[i];
    STAT(5);
    if (!in_bounds(xp, lb, ub, n)) STAT(6);
    if (in_bounds(xp, lb, ub, n)) {
        double p_value = evaluate_quadratic(Jh, g_h, p_h, diag_h, n);
        memcpy(step, p, n * sizeof(double));
        memcpy(step_h, p_h, n * sizeof(double));
        return -p_value;
    }
    int hits[MAXN];
    double p_stride = step_size_to_bound(x, p, lb, ub, n, hits);

    double r_h[MAXN], r[MAXN];
    for (int i = 0; i < n; ++i) {
        r_h[i] = p_h[i];
        if (hits[i] != 0) r_h[i] = r_h[i] * -1.0;
        r[i] = d[i] * r_h[i];
    }
    double x_on_bound[MAXN];
    for (int i = 0; i < n; ++i) {
        p[i] = p[i] * p_stride;
        p_h[i] = p_h[i] * p_stride;
        x_on_bound[i] = x[i] + p[i];
    }
    double to_tr = intersect_trust_region_pos(p_h, r_h, n, Delta);
    double to_bound = step_size_to_bound(x_on_bound, r, lb, ub, n, NULL);

    double r_stride = fmin(to_bound, to_tr);
    double r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = (1 - theta) * p_stride / r_stride;
        r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        double a, b, c = 0.0;
        build_quadratic_1d(Jh, g_h, r_h, diag_h, p_h, n, &a, &b, &c);
        r_stride = minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c, &r_value);
        for (int i = 0; i < n; ++i) {
            r_h[i] = r_h[i] * r_stride;
            r_h[i] = r_h[i] + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = INFINITY;
    }

    for (int i = 0; i < n; ++i) { p[i] = p[i] * theta; p_h[i] = p_h[i] * theta; }
    double p_value = evaluate_quadratic(Jh, g_h, p_h, diag_h, n);

    double ag_h[MAXN], ag[MAXN];
    for (int i = 0; i < n; ++i) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    to_tr = Delta / vnorm(ag_h, n);
    to_bound = step_size_to_bound(x, ag, lb, ub, n, NULL);
    double ag_stride = (to_bound < to_tr) ? theta * to_bound : to_tr;
    double a, b, cdummy = 0.0, ag_value;
    build_quadratic_1d(Jh, g_h, ag_h, diag_h, NULL, n, &a, &b, &cdummy);
    ag_stride = minimize_quadratic_1d(a, b, 0.0, ag_stride, 0.0, &ag_value);
    for (int i = 0; i < n; ++i) { ag_h[i] = ag_h[i] * ag_stride; ag[i] = ag[i] * ag_stride; }

    if (p_value < r_value && p_value < ag_value) {
        memcpy(step, p, n * sizeof(double)); memcpy(step_h, p_h, n * sizeof(double));
        return -p_value;
    } else if (r_value < p_value && r_value < ag_value) {
        memcpy(step, r, n * sizeof(double)); memcpy(step_h, r_h, n * sizeof(double));
        return -r_value;
    }
    memcpy(step, ag, n * sizeof(double)); memcpy(step_h, ag_h, n * sizeof(double));
    return -ag_value;
}

/*
 * scipy.optimize.least_squares(fun, x0, bounds=(lb, ub)) exactly as IKPy calls
 * it: method='trf', jac='2-point', ftol=xtol=gtol=1e-8, x_scale=1, loss linear,
 * tr_solver='exact', max_nfev=100*n.  Returns 0 or a negative error code that
 * the host maps to the exception scipy would raise; *status_out gets the scipy
 * termination status (0..4), *nfev_out the trial-evaluation count.
 */
int oracle_least_squares(oracle_chain *ch, const double *target, const double *x0_in,
                         double *x_out, int *status_out, int *nfev_out, int null_mode)
{
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    int n = ch->n;
    const double *lb = ch->lb, *ub = ch->ub;
    for (int i = 0; i < n; ++i)
        if (!(lb[i] < ub[i])) return SEQIK_ERR_BAD_BOUNDS;
    if (!in_bounds(x0_in, lb, ub, n)) return SEQIK_ERR_X0_OUT_OF_BOUNDS;

    double x[MAXN], f[NRES], J[NRES][MAXN], g[MAXN] = {0};
    memcpy(x, x0_in, n * sizeof(double));
    make_strictly_feasible(x, lb, ub, n, 1e-10);
    double x0[MAXN];
    memcpy(x0, x, n * sizeof(double));

    residual(ch, x, target, f);
    if (g_analytic_jac) analytic_jacobian(ch, x, J); else approx_jacobian(ch, x, f, target, J);

    int nfev = 1;
    int max_nfev = 100 * n;
    double cost = 0.5 * vdot(f, f, NRES);
    for (int i = 0; i < n; ++i) {  /* compute_grad: J.T.dot(f) */
        double acc = 0.0;
        for (int k = 0; k < NRES; ++k) acc = FMA(J[k][i], f[k], acc);
        g[i] = acc;
    }

    double v[MAXN], dv[MAXN], tmp[MAXN];
    cl_scaling_vector(x, g, lb, ub, n, v, dv);
    for (int i = 0; i < n; ++i) tmp[i] = x0[i] * 1.0 / sqrt(v[i]);
    double Delta = vnorm(tmp, n);
    if (Delta == 0) Delta = 1.0;

    double alpha = 0.0;
    int termination_status = -99; /* None */

    /* ACTIVE SET.  IKPy optimises over one variable per link, but the base link,
     * "fixed" links and the last link cannot move the end-effector position, so
     * their Jacobian columns are exactly zero (n - 2 or n - 1 of the n columns).
     * In the reference those zero columns reach LAPACK gesdd, which returns
     * singular values ~1e-17 (not 0) and arbitrary null-space vectors; scipy's
     * root search on alpha then finds ||p|| = Delta by inflating exactly those
     * round-off directions (alpha ~ 1e-18).  The net effect is: Gauss-Newton step
     * on the variables that matter + an O(Delta) pseudo-random move of the inert
     * variables, which in turn triggers the reflective bound logic.  That part
     * of the reference is not reproducible between LAPACK builds (it is the
     * origin of the ~5e-5 rad run-to-run noise floor measured in SURVEY.md 7.4).
     * The restatement therefore runs scipy's algorithm with the zero columns
     * removed from the SVD / trust-region sub-problem, i.e. with the round-off
     * garbage replaced by exact zeros: inert variables never move, everything
     * else (scaling, Delta_0 = ||x0/sqrt(v)|| over ALL n entries, select_step,
     * radius update, ftol/xtol/gtol tests with ||x|| over all n) is verbatim. */
    int act[MAXN], na = 0;
    for (int i = 0; i < n; ++i)
        if (!ch->is_origin[i] && ch->has_rot[i] && i < n - 1) act[na++] = i;
    double x_new[MAXN], f_new[NRES];
    double cost_new = cost;

    while (1) {
        cl_scaling_vector(x, g, lb, ub, n, v, dv);
        double g_norm = 0.0;
        for (int i = 0; i < n; ++i) { double a = fabs(g[i] * v[i]); if (a > g_norm) g_norm = a; }
        if (g_norm < gtol) termination_status = 1;
        if (termination_status != -99 || nfev == max_nfev) break;

        double d[MAXN], diag_h[MAXN], g_h[MAXN];
        double Jh[NRES][MAXN];
        for (int i = 0; i < n; ++i) {
            d[i] = sqrt(v[i]) * 1.0;      /* v**0.5 * scale */
            diag_h[i] = g[i] * dv[i] * 1.0; /* g * dv * scale */
            g_h[i] = d[i] * g[i];
        }
        for (int k = 0; k < NRES; ++k)
            for (int i = 0; i < n; ++i) Jh[k][i] = J[k][i] * d[i];
        /* SVD of the augmented matrix [[J_h], [diag(sqrt(diag_h))]] restricted to the
         * ACTIVE columns (see note above): rows NRES.. carry the Coleman-Li diagonal. */
        double A[MAXROWS][MAXN];
        for (int c = 0; c < na; ++c) {
            int i = act[c];
            for (int k = 0; k < NRES; ++k) A[k][c] = Jh[k][i];
            for (int r = 0; r < na; ++r) A[NRES + r][c] = (r == c) ? sqrt(diag_h[i]) : 0.0;
        }
        double s[MAXN], U[MAXROWS][MAXN], V[MAXN][MAXN], uf[MAXN];
        double Jh_act[NRES][MAXN], diag_act[MAXN];
        int cf2 = g_closed_form_2x2 && na == 2 && null_mode != NULL_WOODBURY;
        if (null_mode == NULL_WOODBURY || cf2) {
            for (int c = 0; c < na; ++c) {
                for (int k = 0; k < NRES; ++k) Jh_act[k][c] = Jh[k][act[c]];
                diag_act[c] = diag_h[act[c]];
            }
        } else {
            jacobi_svd(NRES + na, na, A, s, U, V);
            for (int c = 0; c < na; ++c) {  /* uf = U.T.dot(f_augmented) */
                double acc = 0.0;
                for (int k = 0; k < NRES; ++k) acc = FMA(U[k][c], f[k], acc);
                uf[c] = acc;
            }
        }
        double theta = fmax(0.995, 1 - g_norm);

        double actual_reduction = -1;
        while (actual_reduction <= 0 && nfev < max_nfev) {
            double p_h[MAXN], p[MAXN], step[MAXN], step_h[MAXN];
            double p_act[MAXN];
            if (cf2) solve_tr_2x2(Jh_act, diag_act, f, Delta, &alpha, p_act, null_mode == NULL_EXACT_ZERO);
            else if (null_mode == NULL_WOODBURY) solve_tr_woodbury(na, Jh_act, diag_act, f, Delta, &alpha, p_act);
            else solve_lsq_trust_region(na, NRES, uf, s, V, Delta, &alpha, p_act, null_mode == NULL_EXACT_ZERO);
            for (int i = 0; i < n; ++i) p_h[i] = 0.0;
            for (int c = 0; c < na; ++c) p_h[act[c]] = p_act[c];
            for (int i = 0; i < n; ++i) p[i] = d[i] * p_h[i];
            double predicted_reduction =
                select_step(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, n, step, step_h);
            for (int i = 0; i < n; ++i) x_new[i] = x[i] + step[i];
            make_strictly_feasible(x_new, lb, ub, n, 0.0);
            residual(ch, x_new, target, f_new);
            nfev += 1;
            double step_h_norm = vnorm(step_h, n);
            if (!(isfinite(f_new[0]) && isfinite(f_new[1]) && isfinite(f_new[2]))) {
                Delta = 0.25 * step_h_norm;
                continue;
            }
            cost_new = 0.5 * vdot(f_new, f_new, NRES);
            actual_reduction = cost - cost_new;
            /* update_tr_radius */
            double ratio;
            if (predicted_reduction > 0) ratio = actual_reduction / predicted_reduction;
            else if (predicted_reduction == 0 && actual_reduction == 0) ratio = 1;
            else ratio = 0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
            else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            double step_norm = vnorm(step, n);
            /* check_termination */
            int ftol_ok = (actual_reduction < ftol * cost) && (ratio > 0.25);
            int xtol_ok = step_norm < xtol * (xtol + vnorm(x, n));
            if (ftol_ok && xtol_ok) termination_status = 4;
            else if (ftol_ok) termination_status = 2;
            else if (xtol_ok) termination_status = 3;
            if (termination_status != -99) break;
            alpha = alpha * (Delta / Delta_new);
            Delta = Delta_new;
        }
        if (actual_reduction > 0) {
            memcpy(x, x_new, n * sizeof(double));
            memcpy(f, f_new, sizeof(f));
            cost = cost_new;
            if (g_analytic_jac) analytic_jacobian(ch, x, J); else approx_jacobian(ch, x, f, target, J);
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int k = 0; k < NRES; ++k) acc = FMA(J[k][i], f[k], acc);
                g[i] = acc;
            }
        }
    }
    if (termination_status == -99) termination_status = 0;
    memcpy(x_out, x, n * sizeof(double));
    if (status_out) *status_out = termination_status;
    if (nfev_out) *nfev_out = nfev;
    return SEQIK_OK;
}

/* ------------------------------------------------------------------ */
/* seqikpy layer                                                        */
/* ------------------------------------------------------------------ */
/* DOF order used everywhere in this build:
 * 0 ThC_yaw(X) 1 ThC_pitch(Y) 2 ThC_roll(Z) 3 CTr_pitch(Y) 4 CTr_roll(Z) 5 FTi_pitch(Y) 6 TiTa_pitch(Y) */
enum { D_YAW = 0, D_PITCH, D_ROLL, D_CTR_PITCH, D_CTR_ROLL, D_FTI, D_TITA, NDOF };

static const double AX_X[3] = {1, 0, 0}, AX_Y[3] = {0, 1, 0}, AX_Z[3] = {0, 0, 1}, AX_0[3] = {0, 0, 0};

static void set_link(oracle_chain *ch, int i, double tz, const double *rpy, const double *axis,
                     double lb, double ub)
{
    ch->is_origin[i] = 0;
    ch->trans[i][0] = 0; ch->trans[i][1] = 0; ch->trans[i][2] = tz;
    ch->rpy[i][0] = rpy ? rpy[0] : 0; ch->rpy[i][1] = rpy ? rpy[1] : 0; ch->rpy[i][2] = rpy ? rpy[2] : 0;
    ch->has_rot[i] = axis != NULL;
    const double *a = axis ? axis : AX_0;
    ch->axis[i][0] = a[0]; ch->axis[i][1] = a[1]; ch->axis[i][2] = a[2];
    ch->lb[i] = lb; ch->ub[i] = ub;
}

static void set_origin(oracle_chain *ch)
{
    memset(ch, 0, sizeof(*ch));
    ch->is_origin[0] = 1;
    ch->lb[0] = -INFINITY; ch->ub[0] = INFINITY;
}

/* KinematicChainSeq.create_leg_chain_stage_k (kinematic_chain.py:152-421).
 * seg = {coxa, femur, tibia, tarsus}; bounds[dof][2]; ang[dof] = earlier-stage angles at frame t */
static void build_seq_chain(oracle_chain *ch, int stage, const double *seg, const double (*bnd)[2],
                            const double *ang)
{
    const double PI = 3.141592653589793;
    set_origin(ch);
    double r[3];
#define FIXED(i, tz, R0, R1, R2, dof) do { r[0] = R0; r[1] = R1; r[2] = R2; \
        set_link(ch, i, tz, r, NULL, bnd[dof][0], bnd[dof][1]); } while (0)
#define REV(i, tz, AX, dof) set_link(ch, i, tz, NULL, AX, bnd[dof][0], bnd[dof][1])
    if (stage == 1) {
        ch->n = 4;
        REV(1, 0.0, AX_X, D_YAW);
        REV(2, 0.0, AX_Y, D_PITCH);
        REV(3, -seg[0], AX_Y, D_CTR_PITCH);
    } else if (stage == 2) {
        ch->n = 6;
        FIXED(1, 0.0, ang[D_YAW], 0, 0, D_YAW);
        FIXED(2, 0.0, 0, ang[D_PITCH], 0, D_PITCH);
        REV(3, 0.0, AX_Z, D_ROLL);
        REV(4, -seg[0], AX_Y, D_CTR_PITCH);
        REV(5, -seg[1], AX_Y, D_FTI);
    } else if (stage == 3) {
        ch->n = 8;
        FIXED(1, 0.0, ang[D_YAW], 0, 0, D_YAW);
        FIXED(2, 0.0, 0, ang[D_PITCH], 0, D_PITCH);
        FIXED(3, 0.0, 0, 0, ang[D_ROLL], D_ROLL);
        FIXED(4, -seg[0], 0, ang[D_CTR_PITCH], 0, D_CTR_PITCH);
        REV(5, 0.0, AX_Z, D_CTR_ROLL);
        REV(6, -seg[1], AX_Y, D_FTI);
        REV(7, -seg[2], AX_Y, D_TITA);
    } else {
        ch->n = 9;
        FIXED(1, 0.0, ang[D_YAW], 0, 0, D_YAW);
        FIXED(2, 0.0, 0, ang[D_PITCH], 0, D_PITCH);
        FIXED(3, 0.0, 0, 0, ang[D_ROLL], D_ROLL);
        FIXED(4, -seg[0], 0, ang[D_CTR_PITCH], 0, D_CTR_PITCH);
        FIXED(5, 0.0, 0, 0, ang[D_CTR_ROLL], D_CTR_ROLL);
        FIXED(6, -seg[1], 0, ang[D_FTI], 0, D_FTI);
        REV(7, -seg[2], AX_Y, D_TITA);
        set_link(ch, 8, -seg[3], NULL, AX_0, -PI, PI);
    }
    chain_prepare(ch);
}

/* KinematicChainGeneric.create_leg_chain (kinematic_chain.py:464-530) */
static void build_generic_chain(oracle_chain *ch, const double *seg, const double (*bnd)[2])
{
    const double PI = 3.141592653589793;
    set_origin(ch);
    ch->n = 9;
    REV(1, 0.0, AX_Z, D_ROLL);
    REV(2, 0.0, AX_X, D_YAW);
    REV(3, 0.0, AX_Y, D_PITCH);
    REV(4, -seg[0], AX_Y, D_CTR_PITCH);
    REV(5, 0.0, AX_Z, D_CTR_ROLL);
    REV(6, -seg[1], AX_Y, D_FTI);
    REV(7, -seg[2], AX_Y, D_TITA);
    set_link(ch, 8, -seg[3], NULL, AX_0, -PI, PI);
    chain_prepare(ch);
    ch->rtl = g_generic_rtl;
#undef FIXED
#undef REV
}

static int g_null_mode_override = -1; /* test hook: force one mode for every stage */
void oracle_set_closed_form(int on) { g_closed_form_2x2 = on; }
void oracle_set_null_mode(int mode) { g_null_mode_override = mode; }
static int seq_null_mode(int stage)
{
    if (g_null_mode_override >= 0) return g_null_mode_override;
    return stage == 1 ? NULL_EXACT_ZERO : NULL_ACTIVE_ONLY;
}

/* link index of the DOFs stored by each stage (leg_inverse_kinematics.py:285-320) */
static const int STAGE_NLINK[5] = {0, 4, 6, 8, 9};
static const int STAGE_STORE_LINK[5][2] = {{0, 0}, {1, 2}, {3, 4}, {5, 6}, {7, -1}};
static const int STAGE_STORE_DOF[5][2] = {{0, 0}, {D_YAW, D_PITCH}, {D_ROLL, D_CTR_PITCH}, {D_CTR_ROLL, D_FTI}, {D_TITA, -1}};

/*
 * One leg of LegInvKinSeq.run_ik_and_fk, in the reference's own loop order
 * (stage-major, frames inner, frame t warm-started from frame t-1's full
 * solution vector; leg_inverse_kinematics.py:259-282, 373-385).
 *
 * pose      [N][5][3]   aligned key points (origin = row 0)
 * seeds     4+6+8+9 doubles, initial_angles["stage_1..4"]
 * angles    [N][7]  in/out: columns of stages < first_stage are read, stages
 *           first_stage..last_stage are written (DOF order above)
 * init_angles nullable [7]: frame 0 is warm-started from these joint angles (the solution of the frame
 *           preceding this piece of the recording) instead of from the seeds
 * fk        nullable [N][9][3], written when last_stage == 4
 * status/nfev nullable [N][4]
 * Returns 0, or a negative code; err_frame / err_stage locate the failure.
 */
int oracle_seq_leg(const double *pose, int64_t N, const double *seg, const double *bounds /*[7][2]*/,
                   const double *seeds, int first_stage, int last_stage,
                   double *angles, double *fk, int32_t *status, int32_t *nfev,
                   int64_t *err_frame, int32_t *err_stage, const double *init_angles /* nullable [7] */)
{
    if (first_stage < 1 || last_stage > 4 || first_stage > last_stage) return SEQIK_ERR_BAD_ARG;
    const double (*bnd)[2] = (const double (*)[2])bounds;
    const double *seed_ptr[5] = {NULL, seeds, seeds + 4, seeds + 10, seeds + 18};
    oracle_chain ch;
    for (int stage = first_stage; stage <= last_stage; ++stage) {
        int n = STAGE_NLINK[stage];
        double prev[MAXN], sol[MAXN];
        memcpy(prev, seed_ptr[stage], n * sizeof(double));
        if (init_angles)  /* continuation: frame 0 starts from the angles of the preceding frame */
            for (int k = 0; k < 2; ++k)
                if (STAGE_STORE_LINK[stage][k] >= 0) prev[STAGE_STORE_LINK[stage][k]] = init_angles[STAGE_STORE_DOF[stage][k]];
        if (stage == 1) build_seq_chain(&ch, 1, seg, bnd, NULL);
        for (int64_t t = 0; t < N; ++t) {
            const double *kp = pose + t * 15;
            double target[3] = {kp[3 * stage + 0] - kp[0], kp[3 * stage + 1] - kp[1], kp[3 * stage + 2] - kp[2]};
            if (stage > 1) build_seq_chain(&ch, stage, seg, bnd, angles + t * NDOF);
            int st = 0, nf = 0;
            int rc = oracle_least_squares(&ch, target, prev, sol, &st, &nf, seq_null_mode(stage));
            if (rc != SEQIK_OK) {
                if (err_frame) *err_frame = t;
                if (err_stage) *err_stage = stage;
                return rc;
            }
            for (int k = 0; k < 2; ++k)
                if (STAGE_STORE_LINK[stage][k] >= 0)
                    angles[t * NDOF + STAGE_STORE_DOF[stage][k]] = sol[STAGE_STORE_LINK[stage][k]];
            if (status) status[t * 4 + stage - 1] = st;
            if (nfev) nfev[t * 4 + stage - 1] = nf;
            if (stage == 4 && fk) {
                double all[MAXN * 16];
                chain_fk(&ch, sol, NULL, all);
                for (int i = 0; i < 9; ++i)
                    for (int a = 0; a < 3; ++a) fk[(t * 9 + i) * 3 + a] = all[16 * i + 4 * a + 3] + kp[a];
            }
            memcpy(prev, sol, n * sizeof(double));
        }
    }
    return SEQIK_OK;
}

/*
 * One leg of LegInvKinGeneric.run_ik_and_fk (leg_inverse_kinematics.py:474-499,
 * 581-594): single 9-link chain following the claw (pose row 4), seed =
 * initial_angles["stage_4"]; angles out [N][7] in this build's DOF order.
 */
/* The generic chain (7 unknowns, 3 residuals) uses the SVD-free trust-region step (solve_tr_woodbury); the
 * SVD-based variant (NULL_ACTIVE_ONLY) gives the same iteration counts and claw positions and stays selectable. */
static int g_generic_mode = NULL_WOODBURY;
void oracle_set_generic_mode(int mode) { g_generic_mode = mode; }

int oracle_generic_leg(const double *pose, int64_t N, const double *seg, const double *bounds,
                       const double *seed9, double *angles, double *fk, int32_t *status, int32_t *nfev,
                       int64_t *err_frame)
{
    const double (*bnd)[2] = (const double (*)[2])bounds;
    /* generic link order: Base, roll, yaw, pitch, CTr_pitch, CTr_roll, FTi, TiTa, Claw */
    static const int LINK_DOF[9] = {-1, D_ROLL, D_YAW, D_PITCH, D_CTR_PITCH, D_CTR_ROLL, D_FTI, D_TITA, -1};
    oracle_chain ch;
    build_generic_chain(&ch, seg, bnd);
    double prev[MAXN], sol[MAXN];
    memcpy(prev, seed9, 9 * sizeof(double));
    for (int64_t t = 0; t < N; ++t) {
        const double *kp = pose + t * 15;
        double target[3] = {kp[12] - kp[0], kp[13] - kp[1], kp[14] - kp[2]};
        int st = 0, nf = 0;
        int rc = oracle_least_squares(&ch, target, prev, sol, &st, &nf, g_generic_mode);
        if (rc != SEQIK_OK) { if (err_frame) *err_frame = t; return rc; }
        for (int i = 1; i < 8; ++i) angles[t * NDOF + LINK_DOF[i]] = sol[i];
        if (status) status[t] = st;
        if (nfev) nfev[t] = nf;
        if (fk) {
            double all[MAXN * 16];
            chain_fk(&ch, sol, NULL, all);
            for (int i = 0; i < 9; ++i)
                for (int a = 0; a < 3; ++a) fk[(t * 9 + i) * 3 + a] = all[16 * i + 4 * a + 3] + kp[a];
        }
        memcpy(prev, sol, 9 * sizeof(double));
    }
    return SEQIK_OK;
}

/* Single solve on an explicit seq-stage chain: the per-frame seam
 * LegInvKinBase.calculate_ik (leg_inverse_kinematics.py:62-69). */
int oracle_stage_solve(int stage, const double *seg, const double *bounds, const double *prior_angles,
                       const double *target, const double *x0, double *x_out, int32_t *status, int32_t *nfev)
{
    if (stage < 1 || stage > 4) return SEQIK_ERR_BAD_ARG;
    oracle_chain ch;
    build_seq_chain(&ch, stage, seg, (const double (*)[2])bounds, prior_angles);
    int st = 0, nf = 0;
    int rc = oracle_least_squares(&ch, target, x0, x_out, &st, &nf, seq_null_mode(stage));
    if (status) *status = st;
    if (nfev) *nfev = nf;
    return rc;
}

/* Chain.forward_kinematics(q, full_kinematics=True)[i][:3,3] on a seq-stage chain. */
int oracle_stage_fk(int stage, const double *seg, const double *bounds, const double *prior_angles,
                    const double *q, double *pos /*[n][3]*/)
{
    if (stage < 1 || stage > 4) return SEQIK_ERR_BAD_ARG;
    oracle_chain ch;
    build_seq_chain(&ch, stage, seg, (const double (*)[2])bounds, prior_angles);
    double all[MAXN * 16];
    chain_fk(&ch, q, NULL, all);
    for (int i = 0; i < ch.n; ++i)
        for (int a = 0; a < 3; ++a) pos[i * 3 + a] = all[16 * i + 4 * a + 3];
    return SEQIK_OK;
}

/*
 * Batch driver for the CPU baseline: [n_seq][n_legs] independent chains, each run exactly as
 * oracle_seq_leg does (one call per (sequence, leg), the task shape of the reference's
 * examples/example_leg_inv_kinematics_parallel.py:186-187).  Layouts as in include/seqik.h:
 * pose [n_seq][n_legs][N][5][3], angles [..][N][7], fk nullable [..][N][9][3];
 * seg [n_legs][4], bounds [n_legs][7][2], seeds [n_legs][27].
 */
int oracle_seq_batch(const double *pose, int64_t n_seq, int32_t n_legs, int64_t N, const double *seg,
                     const double *bounds, const double *seeds, double *angles, double *fk)
{
    for (int64_t s = 0; s < n_seq; ++s)
        for (int32_t l = 0; l < n_legs; ++l) {
            int64_t c = s * n_legs + l;
            int rc = oracle_seq_leg(pose + c * N * 15, N, seg + 4 * l, bounds + 14 * l, seeds + 27 * l, 1, 4,
                                    angles + c * N * 7, fk ? fk + c * N * 27 : NULL, NULL, NULL, NULL, NULL, NULL);
            if (rc != SEQIK_OK) return rc;
        }
    return SEQIK_OK;
}

int oracle_version(void) { return 1; }
