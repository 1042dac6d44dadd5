// seqik_head.hpp -- closed-form head / antenna joint angles, one frame per lane.
// Replaces HeadInverseKinematics.compute_head_angles and its helpers
// (seqikpy/head_inverse_kinematics.py:103-339): three head angles from the antenna bases and the
// neck, and per side antenna yaw / pitch after removing the head roll.  Purely elementwise:
// 96 B in, 56 B out per frame -> an HBM-bound streaming kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#ifndef SEQIK_HD
#define SEQIK_HD __host__ __device__ __forceinline__
#endif

namespace seqik {

// This file is built with -ffp-contract=off like the solver (one flag set for the library), but nothing here is pinned
// bit for bit to an oracle (the angles are compared with the reference at a tolerance), so products feeding sums are
// written as explicit fused multiply-adds: a third fewer vector instructions than the separate multiply + add.
SEQIK_HD double hfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1 / sqrt(v) and 1 / q without the IEEE corner-case handling of the compiler's division / square-root expansions (34
// and 12 instructions): hardware seed (v_rcp_f64 / v_rsq_f64) + two / three Newton steps (measured against the host
// formulas on 6000 frames, scripts/microbench/head_split.hip: 2 + 3 steps 4e-12 rad, 2 + 2 steps 1.5e-11; the seeds are
// good to ~2^-23 and the reciprocal's argument is in [0.6, 1], the square root's feeds an acos).  Host builds
// (tests/harness) use the plain expressions.
#ifndef SEQIK_HEAD_NEWTON
#define SEQIK_HEAD_NEWTON 2
#endif
SEQIK_HD double inv_sqrt(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(v);
    const double h = 0.5 * v;
#pragma unroll
    for (int i = 0; i < SEQIK_HEAD_NEWTON + 1; ++i) y = hfma(y, hfma(-(h * y), y, 0.5), y);
    return y;
#else
    return 1.0 / sqrt(v);
#endif
}

SEQIK_HD double inv(double q)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(q);
#pragma unroll
    for (int i = 0; i < SEQIK_HEAD_NEWTON; ++i) r = hfma(r, hfma(-q, r, 1.0), r);
    return r;
#else
    return 1.0 / q;
#endif
}

// acos for |x| <= 1 (the callers clamp).  The classic fdlibm scheme -- acos(x) = pi/2 - (x + x z R(z)) with z = x^2 for
// |x| <= 1/2, 2 (s + s z R(z)) with z = (1 - |x|) / 2, s = sqrt(z) otherwise, R = P / Q a (6, 4) rational -- evaluated
// branch-free: ONE rational for both ranges, selects at the end, without fdlibm's last-bit correction of the square
// root (<= 2 ulp instead of < 1 ulp; the angles are compared at 1e-9 rad).  57 vector instructions instead of the 93 of
// the device library's acos, seven times per frame: this kernel is meant to be bound by HBM, not by its arithmetic.
SEQIK_HD double acos_unit(double x)
{
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17,
                 PI = 3.14159265358979311600e+00;
    const double P0 = 1.66666666666666657415e-01, P1 = -3.25565818622400915405e-01, P2 = 2.01212532134862925881e-01,
                 P3 = -4.00555345006794114027e-02, P4 = 7.91534994289814532176e-04, P5 = 3.47933107596021167570e-05,
                 Q1 = -2.40339491173441421878e+00, Q2 = 2.02094576023350569471e+00, Q3 = -6.88283971605453293030e-01,
                 Q4 = 7.70381505559019352791e-02;
    const double a = fabs(x);
    const bool small = a <= 0.5;
    const double z = small ? x * x : hfma(a, -0.5, 0.5);
    const double p = z * hfma(z, hfma(z, hfma(z, hfma(z, hfma(z, P5, P4), P3), P2), P1), P0);
    const double q = hfma(z, hfma(z, hfma(z, hfma(z, Q4, Q3), Q2), Q1), 1.0);
    const double r = p * inv(q);          // q in [0.6, 1]
    const double s = z * inv_sqrt(z + 1e-300);  // sqrt(z), z in [0, 1/4] (the tiny offset keeps z = 0 finite)
    const double w = hfma(r, s, s);
    const double big = (x > 0.0) ? 2.0 * w : hfma(-2.0, w - PIO2_LO, PI);
    const double sm = PIO2_HI - (x - hfma(-x, r, PIO2_LO));
    return small ? sm : big;
}

// angle_between_segments (:163-178): acos of the normalised dot product, signed by
// det([rot_axis, v1, v2]) = rot_axis . (v1 x v2).  The reference normalises both vectors component by
// component (6 divisions, 2 square roots); here cos = (v1 . v2) / sqrt(|v1|^2 |v2|^2) -- one division,
// one square root, the same value up to ~1 ulp.  Every call of the reference works on vectors PROJECTED on a
// coordinate plane and rotates about that plane's normal, so the functions below take the two in-plane components
// (the third one is an exact zero whose products and sums change nothing: same bits as the 3-vector form).
//   planar_angle(a0, a1, b0, b1): vectors (a0, a1), (b0, b1) in the plane, det = a0 b1 - a1 b0
SEQIK_HD double planar_angle(double a0, double a1, double b0, double b1)
{
    const double n1 = hfma(a0, a0, a1 * a1);
    const double n2 = hfma(b0, b0, b1 * b1);
    double d = hfma(a0, b0, a1 * b1) * inv_sqrt(n1 * n2);
    const double det = hfma(a0, b1, -(a1 * b0));
    d = fmin(1.0, fmax(-1.0, d));  // guard the last-ulp overshoot of the fused normalisation
    const double ang = acos_unit(d);
    return (det > 0) ? ang : -ang;
}

//   axis_angle(b_along, b_across): first vector = the unit axis the plane's first coordinate runs along,
//   second = (b_along, b_across): cos = b_along / |b|, det = b_across.  Also hands back cos / sin of the SIGNED angle
//   (b_along / |b|, b_across / |b|), which is all the derotation by the head roll needs.
SEQIK_HD double axis_angle(double b_along, double b_across, double *cos_out = nullptr, double *sin_out = nullptr)
{
    const double n2 = hfma(b_along, b_along, b_across * b_across);
    const double rn = inv_sqrt(n2);
    const double d = fmin(1.0, fmax(-1.0, b_along * rn));
    if (cos_out) { *cos_out = d; *sin_out = b_across * rn; }
    const double ang = acos_unit(d);
    return (b_across > 0) ? ang : -ang;
}

// derotate_vector (:330-333): scipy Rotation.from_euler("x", -roll).apply(v), i.e. the rotation
// matrix of the unit quaternion (sin(-roll/2), 0, 0, cos(-roll/2)): m11 = cos(roll), two_xw = -sin(roll).  The roll is
// the signed angle between the Y axis and (hor_y, hor_z), so its cosine and sine are hor_y / |.| and hor_z / |.| -- no
// trigonometric function needed (the reference goes through sin / cos of roll / 2; same values to ~2e-16)
SEQIK_HD void derotate_x(double m11, double two_xw, const double *v, double *out)
{
    out[0] = v[0];
    out[1] = hfma(m11, v[1], -(two_xw * v[2]));
    out[2] = hfma(two_xw, v[1], m11 * v[2]);
}

// angle_between_segments (:163-178) for two general 3-vectors and a general axis: acos of the normalised dot product,
// negative unless det([axis, v1, v2]) = axis . (v1 x v2) > 0.
SEQIK_HD double signed_angle3(const double *v1, const double *v2, const double *axis)
{
    const double n1 = hfma(v1[0], v1[0], hfma(v1[1], v1[1], v1[2] * v1[2]));
    const double n2 = hfma(v2[0], v2[0], hfma(v2[1], v2[1], v2[2] * v2[2]));
    double d = hfma(v1[0], v2[0], hfma(v1[1], v2[1], v1[2] * v2[2])) * inv_sqrt(n1 * n2);
    d = fmin(1.0, fmax(-1.0, d));
    const double cx = hfma(v1[1], v2[2], -(v1[2] * v2[1])), cy = hfma(v1[2], v2[0], -(v1[0] * v2[2])),
                 cz = hfma(v1[0], v2[1], -(v1[1] * v2[0]));
    const double det = hfma(axis[0], cx, hfma(axis[1], cy, axis[2] * cz));
    const double ang = acos_unit(d);
    return (det > 0) ? ang : -ang;
}

struct HeadArgs {
    const double *r_head;   // [n][n_points][3]: antenna base, antenna tip (right); a frame's record is `rec` doubles
    const double *l_head;   // [n][n_points][3] (left)
    int64_t rec;            // 3 * n_points: 6 for (base, tip) records; 3 when only one head key point per side was
                            // tracked (no antenna angles then, head_inverse_kinematics.py:26)
    const double *roll_in;  // nullable [n]: the antenna vectors are derotated by THIS head roll instead of the frame's own
                            // (compute_antenna_pitch / compute_antenna_yaw take `head_roll` as an argument, :242, :278)
    const double *neck;     // [3], or [n][3] when neck_stride == 3
    int64_t neck_stride;    // 0 or 3
    double rest_head_pitch, rest_antenna_pitch;
    double *angles;         // [7][n]: head roll, pitch, yaw, antenna yaw L, pitch L, yaw R, pitch R
    int64_t n_frames;
    int32_t compute_ant;
};

// The seven angles of one frame from its key points: rb / lb = antenna base + tip (right / left, 6 doubles each),
// neck (3).  out: head roll, pitch, yaw, antenna yaw L, pitch L, yaw R, pitch R (the last four only with compute_ant).
// roll_given: nullable, the caller's head roll for the derotation (reference: Rotation.from_euler("x", -head_roll)).
SEQIK_HD void head_angles_compute(const double *rb, const double *lb, const double *neck, double rest_head_pitch,
                                  double rest_antenna_pitch, bool compute_ant, double *out,
                                  const double *roll_given = nullptr)
{
    const double PI = 3.141592653589793;
    double hor[3] = {lb[0] - rb[0], lb[1] - rb[1], lb[2] - rb[2]};            // R base -> L base
    double mid[3] = {hfma(rb[0] + lb[0], 0.5, -neck[0]), hfma(rb[1] + lb[1], 0.5, -neck[1]),
                     hfma(rb[2] + lb[2], 0.5, -neck[2])};                     // neck -> mid antenna base
    // head roll (:196-210): Y axis -> horizontal vector projected on the transverse (y, z) plane, about X:
    //   det([X, Y, v]) = v_z
    double cos_roll, sin_roll;
    out[0] = axis_angle(hor[1], hor[2], &cos_roll, &sin_roll);
    // head pitch (:180-194): X axis -> mid vector projected on the sagittal (z, x) plane, about Y: det([Y, X, v]) = -v_z
    out[1] = axis_angle(mid[0], -mid[2]) + rest_head_pitch;
    // head yaw (:212-226): Y axis -> horizontal vector projected on the frontal (x, y) plane, about Z: det([Z, Y, v]) = -v_x
    out[2] = axis_angle(hor[1], -hor[0]);
    if (!compute_ant) return;
    if (roll_given) { sin_roll = sin(*roll_given); cos_roll = cos(*roll_given); }
    const double m11 = cos_roll, two_xw = -sin_roll;
    double hor_d[3];
    derotate_x(m11, two_xw, hor, hor_d);
#pragma unroll
    for (int side = 0; side < 2; ++side) {  // 0 = L, 1 = R (the reference's dict order)
        const double *base = side == 0 ? lb : rb;
        double ant[3] = {base[3] - base[0], base[4] - base[1], base[5] - base[2]};
        double head[3] = {neck[0] - base[0], neck[1] - base[1], neck[2] - base[2]};
        double ant_d[3], head_d[3];
        derotate_x(m11, two_xw, ant, ant_d);
        derotate_x(m11, two_xw, head, head_d);
        // antenna yaw (:262-291): antenna vs horizontal head vector, both on the transverse (y, z) plane, about X
        double ayaw = planar_angle(ant_d[1], ant_d[2], hor_d[1], hor_d[2]);
        if (side == 1) ayaw = PI - ayaw;
        // antenna pitch (:228-260): head vector vs antenna, both on the sagittal plane, about Y:
        //   det([Y, h, a]) = h_z a_x - h_x a_z  (plane coordinates (z, x))
        out[3 + 2 * side] = ayaw;
        out[4 + 2 * side] = planar_angle(head_d[2], head_d[0], ant_d[2], ant_d[0]) - rest_antenna_pitch;
    }
}

// one frame straight from / to the caller's arrays (partial wavefronts at the end of a launch; tests/harness)
SEQIK_HD void head_angles_frame(const HeadArgs &a, int64_t t)
{
    double out[7];
    head_angles_compute(a.r_head + t * a.rec, a.l_head + t * a.rec, a.neck + t * a.neck_stride, a.rest_head_pitch,
                        a.rest_antenna_pitch, a.compute_ant != 0, out, a.roll_in ? a.roll_in + t : nullptr);
    const int n_out = a.compute_ant ? 7 : 3;
    for (int j = 0; j < n_out; ++j) a.angles[j * a.n_frames + t] = out[j];
}

}  // namespace seqik
