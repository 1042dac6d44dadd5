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

// angle_between_segments (:163-178): acos of the normalised dot product, signed by
// det([rot_axis, v1, v2]) = rot_axis . (v1 x v2).  The reference normalises both vectors component by
// component (6 divisions, 2 square roots); here cos = (v1 . v2) / sqrt(|v1|^2 |v2|^2) -- one division,
// one square root, the same value up to ~1 ulp -- because this kernel should be bound by HBM, not by
// FP64 division throughput.
SEQIK_HD double signed_angle(const double *v1, const double *v2, int rot_axis)
{
    double n1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    double n2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    double d = (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]) / sqrt(n1 * n2);
    double det;
    if (rot_axis == 0) det = v1[1] * v2[2] - v1[2] * v2[1];
    else if (rot_axis == 1) det = v1[2] * v2[0] - v1[0] * v2[2];
    else det = v1[0] * v2[1] - v1[1] * v2[0];
    d = fmin(1.0, fmax(-1.0, d));  // guard the last-ulp overshoot of the fused normalisation
    double a = acos(d);
    return (det > 0) ? a : -a;
}

// derotate_vector (:330-333): scipy Rotation.from_euler("x", -roll).apply(v), i.e. the rotation
// matrix of the unit quaternion (sin(-roll/2), 0, 0, cos(-roll/2)); m11 / two_xw are computed once per frame
SEQIK_HD void derotate_x(double m11, double two_xw, const double *v, double *out)
{
    out[0] = v[0];
    out[1] = m11 * v[1] - two_xw * v[2];
    out[2] = two_xw * v[1] + m11 * v[2];
}

struct HeadArgs {
    const double *r_head;   // [n][2][3]: antenna base, antenna tip (right)
    const double *l_head;   // [n][2][3] (left)
    const double *neck;     // [3], or [n][3] when neck_stride == 3
    int64_t neck_stride;    // 0 or 3
    double rest_head_pitch, rest_antenna_pitch;
    double *angles;         // [7][n]: head roll, pitch, yaw, antenna yaw L, pitch L, yaw R, pitch R
    int64_t n_frames;
    int32_t compute_ant;
};

SEQIK_HD void head_angles_frame(const HeadArgs &a, int64_t t)
{
    const double PI = 3.141592653589793;
    const double *rb = a.r_head + t * 6, *lb = a.l_head + t * 6;
    const double *neck = a.neck + t * a.neck_stride;
    double hor[3] = {lb[0] - rb[0], lb[1] - rb[1], lb[2] - rb[2]};            // R base -> L base
    double mid[3] = {(rb[0] + lb[0]) * 0.5 - neck[0], (rb[1] + lb[1]) * 0.5 - neck[1],
                     (rb[2] + lb[2]) * 0.5 - neck[2]};                        // neck -> mid antenna base
    const double X[3] = {1.0, 0.0, 0.0}, Y[3] = {0.0, 1.0, 0.0};
    double v[3];
    // head roll (:196-210): Y axis -> horizontal vector projected on the transverse plane, about X
    v[0] = 0.0; v[1] = hor[1]; v[2] = hor[2];
    double roll = signed_angle(Y, v, 0);
    // head pitch (:180-194): X axis -> mid vector projected on the sagittal plane, about Y
    v[0] = mid[0]; v[1] = 0.0; v[2] = mid[2];
    double pitch = signed_angle(X, v, 1) + a.rest_head_pitch;
    // head yaw (:212-226): Y axis -> horizontal vector projected on the frontal plane, about Z
    v[0] = hor[0]; v[1] = hor[1]; v[2] = 0.0;
    double yaw = signed_angle(Y, v, 2);
    const int64_t n = a.n_frames;
    a.angles[t] = roll;
    a.angles[n + t] = pitch;
    a.angles[2 * n + t] = yaw;
    if (!a.compute_ant) return;
    const double hh = -roll * 0.5;
    const double qx = sin(hh), qw = cos(hh);
    const double m11 = qw * qw - qx * qx, two_xw = 2.0 * (qx * qw);
    double hor_d[3];
    derotate_x(m11, two_xw, hor, hor_d);
    for (int side = 0; side < 2; ++side) {  // 0 = L, 1 = R (the reference's dict order)
        const double *base = side == 0 ? lb : rb;
        double ant[3] = {base[3] - base[0], base[4] - base[1], base[5] - base[2]};
        double head[3] = {neck[0] - base[0], neck[1] - base[1], neck[2] - base[2]};
        double ant_d[3], head_d[3];
        derotate_x(m11, two_xw, ant, ant_d);
        derotate_x(m11, two_xw, head, head_d);
        // antenna yaw (:262-291): antenna vs horizontal head vector, both on the transverse plane, about X
        double a1[3] = {0.0, ant_d[1], ant_d[2]};
        double h1[3] = {0.0, hor_d[1], hor_d[2]};
        double ayaw = signed_angle(a1, h1, 0);
        if (side == 1) ayaw = PI - ayaw;
        // antenna pitch (:228-260): head vector vs antenna, both on the sagittal plane, about Y
        double a2[3] = {ant_d[0], 0.0, ant_d[2]};
        double h2[3] = {head_d[0], 0.0, head_d[2]};
        double apitch = signed_angle(h2, a2, 1) - a.rest_antenna_pitch;
        a.angles[(3 + 2 * side) * n + t] = ayaw;
        a.angles[(4 + 2 * side) * n + t] = apitch;
    }
}

}  // namespace seqik
