// TEST INFRASTRUCTURE ONLY -- not part of the product, never loaded by seqikpy_amd.
//
// Runs the kernel's per-chain device function (csrc/seqik_core.hpp, `__host__ __device__`)
// on the HOST, one chain at a time, so that the CPU-only test tier can compare the
// kernel's arithmetic bit for bit with the generic restatement in oracle/seqik_oracle.c
// without a GPU.  Built by tests/conftest.py with `hipcc --offload-host-only`.
#include "../../sequential-inverse-kinematics_amd/csrc/seqik_core.hpp"
#include "../../sequential-inverse-kinematics_amd/csrc/seqik_consts.hpp"
#include "../../sequential-inverse-kinematics_amd/csrc/seqik_head.hpp"
#include <vector>

extern "C" int harness_run_chain(const double *pose, int64_t n_frames, const SeqikLegParams *leg,
                                 int32_t first_stage, int32_t last_stage, double *angles, double *fk,
                                 int32_t *status, int32_t *nfev, const SeqikAffine *affine, const double *init)
{
    int rc = seqik::validate_leg(*leg, first_stage, last_stage);
    if (rc != SEQIK_OK) return rc;
    seqik::LegConst lc;
    seqik::make_leg_consts(*leg, affine, lc);
    seqik::ChainIO io;
    io.pose = pose; io.pose_row = 3; io.pose_frame = 15;
    io.angles = angles; io.ang_dof = 1; io.ang_frame = 7;
    io.fk = (fk && last_stage == 4) ? fk : nullptr;
    io.status = status; io.nfev = nfev;
    io.init = init;
    io.n_frames = n_frames;
    const bool want_fk = io.fk != nullptr;
    const bool diag = status || nfev;
    std::vector<double> ws((size_t)n_frames * 12 + 1);
    io.frames = ws.data();
    // same launch sequence as seqik_hip.hip::launch(): one pass over all frames per stage; the first stage of a
    // run that starts after stage 1 rebuilds its prefix from the angles, later ones read the hand-off workspace
    for (int stage = first_stage; stage <= last_stage; ++stage) {
        const bool fa = (stage == first_stage) && stage > 1;
        const bool ho = stage < last_stage;
#define RUN4(S, FK, FA, HO)                                                   \
    do { if (diag) seqik::run_stage<S, FK, true, FA, HO>(lc, io);             \
         else seqik::run_stage<S, FK, false, FA, HO>(lc, io); } while (0)
#define RUNS(S)                                                               \
    do { if (want_fk) { if (fa && ho) RUN4(S, true, true, true); else if (fa) RUN4(S, true, true, false);      \
                        else if (ho) RUN4(S, true, false, true); else RUN4(S, true, false, false); }           \
         else { if (fa && ho) RUN4(S, false, true, true); else if (fa) RUN4(S, false, true, false);            \
                else if (ho) RUN4(S, false, false, true); else RUN4(S, false, false, false); } } while (0)
        if (stage == 1) { if (ho) RUN4(1, false, false, true); else RUN4(1, false, false, false); }
        else if (stage == 2) RUNS(2);
        else if (stage == 3) RUNS(3);
        else { if (want_fk) { if (fa) RUN4(4, true, true, false); else RUN4(4, true, false, false); }
               else { if (fa) RUN4(4, false, true, false); else RUN4(4, false, false, false); } }
#undef RUNS
#undef RUN4
    }
    return SEQIK_OK;
}

// Frame chunks (SeqikOptions.frame_chunk): the device code's CHUNKED instantiation of run_stage plus a serial
// re-enactment of the launch sequence of seqik_hip.hip (speculative pass, first verification with the automatic mode's
// per-chain guard, R x {scan, repair}, scan + sweep, serial walk of a chain the guard gave up on) for one chain.
// stats as SeqikOptions.chunk_stats (int32[16]), flags as SeqikOptions.chunk_flags (uint8[K]).
static void harness_run_stages(const seqik::LegConst &lc, seqik::ChainIO &io, bool fk)
{
    if (fk) {
        seqik::run_stage<1, false, false, false, true, true>(lc, io);
        seqik::run_stage<2, true, false, false, true, true>(lc, io);
        seqik::run_stage<3, true, false, false, true, true>(lc, io);
        seqik::run_stage<4, true, false, false, false, true>(lc, io);
    } else {
        seqik::run_stage<1, false, false, false, true, true>(lc, io);
        seqik::run_stage<2, false, false, false, true, true>(lc, io);
        seqik::run_stage<3, false, false, false, true, true>(lc, io);
        seqik::run_stage<4, false, false, false, false, true>(lc, io);
    }
}

static void harness_solve_chunk(const seqik::LegConst &lc, const double *pose, int64_t N, double *angles, double *fk,
                                int64_t k, int64_t C, int64_t h, int64_t lead, bool repair, const double *init0, double *ss,
                                double *ws)
{
    seqik::ChainIO io;
    io.pose = pose; io.pose_row = 3; io.pose_frame = 15;
    io.angles = angles; io.ang_dof = 1; io.ang_frame = 7;
    io.fk = fk; io.status = nullptr; io.nfev = nullptr;
    io.frames = ws;
    io.t_store = lead + k * C;
    io.n_frames = lead + (k + 1) * C < N ? lead + (k + 1) * C : N;
    if (!repair) {
        const bool run_in = k > 0 || lead > 0;
        io.t_begin = (k > 0 && io.t_store > h) ? io.t_store - h : 0;
        io.init = run_in ? nullptr : init0;
        io.init_stride = 1;
        io.start_state = run_in ? ss + k * 7 : nullptr;
    } else {
        io.t_begin = io.t_store;
        io.init = (k == 0) ? init0 : angles + (io.t_store - 1) * 7;
        io.init_stride = 1;
        io.start_state = nullptr;
        for (int d = 0; d < 7; ++d) ss[k * 7 + d] = io.init[d];
    }
    harness_run_stages(lc, io, fk != nullptr);
}

extern "C" int harness_run_chunked(const double *pose, int64_t N, const SeqikLegParams *leg, int32_t C, int32_t h,
                                   double tol, int32_t rounds, double *angles, double *fk, const double *init,
                                   int32_t *stats, int32_t guard, int32_t lead, uint8_t *flags)
{
    int rc = seqik::validate_leg(*leg, 1, 4);
    if (rc != SEQIK_OK) return rc;
    seqik::LegConst lc;
    seqik::make_leg_consts(*leg, nullptr, lc);
    const int64_t K = (N - lead + C - 1) / C;
    std::vector<double> ss((size_t)K * 7, 0.0), ws((size_t)(C + (h > lead ? h : lead)) * 12 + 1);
    std::vector<uint8_t> fl((size_t)K, 0);
    for (int i = 0; i < 16; ++i) stats[i] = 0;
    stats[0] = (int32_t)K; stats[1] = C; stats[2] = h;
    for (int64_t k = 0; k < K; ++k) harness_solve_chunk(lc, pose, N, angles, fk, k, C, h, lead, false, init, ss.data(), ws.data());
    const int64_t k_first = (lead > 0 && init) ? 0 : 1;
    auto inconsistent = [&](int64_t k) {
        bool bad = false;
        const double *truth = (k == 0) ? init : angles + (lead + k * C - 1) * 7;
        for (int d = 0; d < 7; ++d) bad |= !(fabs(ss[k * 7 + d] - truth[d]) <= tol);
        return bad;
    };
    int fails = 0;
    for (int64_t k = k_first; k < K; ++k)
        if (inconsistent(k)) { ++fails; fl[k] |= 1; }
    stats[7] = fails;
    if (guard && lead == 0 && (int64_t)fails * 8 > K) {  // the guard: this chain is walked serially
        stats[8] = 1; stats[9] = (int32_t)K;
        for (int64_t k = 0; k < K; ++k) fl[k] |= 8;
        seqik::ChainIO io;
        io.pose = pose; io.pose_row = 3; io.pose_frame = 15;
        io.angles = angles; io.ang_dof = 1; io.ang_frame = 7;
        io.fk = fk; io.status = nullptr; io.nfev = nullptr;
        std::vector<double> wsN((size_t)N * 12 + 1);
        io.frames = wsN.data();
        io.t_begin = 0; io.t_store = 0; io.n_frames = N; io.init = init; io.init_stride = 1; io.start_state = nullptr;
        harness_run_stages(lc, io, fk != nullptr);
        if (flags) for (int64_t k = 0; k < K; ++k) flags[k] = fl[k];
        return SEQIK_OK;
    }
    for (int r = 0; r <= rounds; ++r) {
        std::vector<int64_t> ready;
        int pending = 0;
        for (int64_t k = k_first; k < K; ++k)
            if (inconsistent(k)) {
                ++pending;
                if (!(k > k_first && inconsistent(k - 1))) ready.push_back(k);
            }
        if (pending == 0) break;
        if (r < rounds) {
            stats[3 + (r < 2 ? r : 2)] += (int32_t)ready.size();
            for (int64_t k : ready) {
                harness_solve_chunk(lc, pose, N, angles, fk, k, C, h, lead, true, init, ss.data(), ws.data());
                fl[k] |= 2;
            }
        } else {
            for (int64_t k = k_first; k < K; ++k)
                if (inconsistent(k)) {
                    harness_solve_chunk(lc, pose, N, angles, fk, k, C, h, lead, true, init, ss.data(), ws.data());
                    fl[k] |= 4;
                    stats[6] += 1;
                }
        }
    }
    if (flags) for (int64_t k = 0; k < K; ++k) flags[k] = fl[k];
    return SEQIK_OK;
}

extern "C" void harness_sincos(double x, double *s, double *c) { seqik::sincos_cw(x, *s, *c); }

// head / antenna angles: the kernel's per-frame device function, run on the host
extern "C" void harness_head_angles(const double *r_head, const double *l_head, int64_t n, const double *neck,
                                    int64_t neck_stride, double rest_head_pitch, double rest_antenna_pitch,
                                    int32_t compute_ant, double *angles, int32_t n_points, const double *head_roll)
{
    seqik::HeadArgs a;
    a.r_head = r_head; a.l_head = l_head; a.neck = neck; a.neck_stride = neck_stride;
    a.rec = 3 * (int64_t)n_points; a.roll_in = compute_ant ? head_roll : nullptr;
    a.rest_head_pitch = rest_head_pitch; a.rest_antenna_pitch = rest_antenna_pitch;
    a.angles = angles; a.n_frames = n; a.compute_ant = compute_ant;
    for (int64_t t = 0; t < n; ++t) seqik::head_angles_frame(a, t);
}

extern "C" void harness_signed_angles(const double *v1, int64_t s1, const double *v2, int64_t s2, const double *axis,
                                      int64_t n, double *out)
{
    for (int64_t t = 0; t < n; ++t) out[t] = seqik::signed_angle3(v1 + t * s1, v2 + t * s2, axis);
}

// generic (single-chain) IK: the kernel's per-chain device function, run on the host
extern "C" int harness_run_generic(const double *pose, int64_t n_frames, const SeqikLegParams *leg, double *angles,
                                   double *fk, int32_t *status, int32_t *nfev, const double *init)
{
    int rc = seqik::validate_leg_generic(*leg);
    if (rc != SEQIK_OK) return rc;
    seqik::GenericConst gc;
    seqik::make_generic_consts(*leg, gc);
    seqik::LegAffine aff;
    aff.enabled = 0;
    seqik::GenericIO io;
    io.pose = pose; io.pose_row = 3; io.pose_frame = 15;
    io.angles = angles; io.ang_dof = 1; io.ang_frame = 7;
    io.fk = fk; io.status = status; io.nfev = nfev; io.init = init; io.n_frames = n_frames;
    if (status || nfev) seqik::run_generic<true>(gc, aff, io);
    else seqik::run_generic<false>(gc, aff, io);
    return SEQIK_OK;
}

// the chain-queue instantiation of the same function (seqik_generic.hpp GenericQueue): ONE host "lane" takes the n_seq
// sequences of leg `leg` of a batch laid out [seq][n_legs][frames][...] one after the other from the counter
extern "C" int harness_run_generic_queue(const double *pose, int64_t n_seq, int32_t n_legs, int32_t leg_index, int64_t n_frames,
                                         const SeqikLegParams *leg, double *angles, double *fk, int32_t *status,
                                         int32_t *nfev, const double *init, int32_t *counter)
{
    int rc = seqik::validate_leg_generic(*leg);
    if (rc != SEQIK_OK) return rc;
    // a table of n_legs legs of which only `leg_index` has sequences left (the counters of the others start exhausted):
    // the lane walks over the exhausted legs to its own, as a lane of the kernel does at the end of a leg
    std::vector<seqik::GenericLeg> table((size_t)n_legs);
    std::vector<int32_t> counters((size_t)n_legs, (int32_t)n_seq);
    for (int l = 0; l < n_legs; ++l) { seqik::make_generic_consts(*leg, table[(size_t)l].gc); table[(size_t)l].aff.enabled = 0; }
    counters[(size_t)leg_index] = *counter;
    seqik::GenericIO io;
    io.pose = nullptr; io.pose_row = 3; io.pose_frame = 15;
    io.angles = nullptr; io.ang_dof = 1; io.ang_frame = 7;
    io.fk = nullptr; io.status = nullptr; io.nfev = nullptr; io.init = nullptr; io.n_frames = n_frames;
    seqik::GenericQueue q;
    q.counters = counters.data(); q.n_seq = n_seq; q.n_legs = n_legs; q.first = 0;
    for (int l = 0; l < 8; ++l) q.order[l] = (uint8_t)l;
    q.table = table.data();
    q.pose = pose; q.pose_chain = n_frames * 15; q.angles = angles; q.ang_chain = n_frames * 7;
    q.fk = fk; q.status = status; q.nfev = nfev; q.init = init;
    if (status || nfev) seqik::run_generic<true, false, true>(table[0].gc, table[0].aff, io, &q);
    else seqik::run_generic<false, false, true>(table[0].gc, table[0].aff, io, &q);
    *counter = counters[(size_t)leg_index];
    for (int l = 0; l < n_legs; ++l)   // every other leg was tried exactly once (found exhausted, never revisited)
        if (l != leg_index && counters[(size_t)l] != (int32_t)n_seq + 1) return -99;
    return SEQIK_OK;
}

// The reflective select_step in its two forms (seqik_core.hpp): compact (lane-per-chain kernels) and written for latency
// (run_stage<..., LAT>): out = {step[2], step_h[2], predicted_reduction}.  tests/test_core_bitexact_host.py feeds both the same
// made-up inputs and compares the bits.
extern "C" void harness_select_step(int32_t na, int32_t ilp, const double *x, const double *Jh6 /* [3][2] */, const double *diag_h,
                                    const double *g_h, const double *p_in, const double *p_h_in, const double *d, double Delta,
                                    const double *lb, const double *ub, double theta, double *out5)
{
    double Jh[3][2], p[2] = {p_in[0], p_in[1]}, p_h[2] = {p_h_in[0], p_h_in[1]}, step[2] = {0, 0}, step_h[2] = {0, 0};
    for (int k = 0; k < 3; ++k) { Jh[k][0] = Jh6[2 * k]; Jh[k][1] = Jh6[2 * k + 1]; }
    double pr;
    if (na == 2) pr = ilp ? seqik::select_step_reflective_ilp<2>(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h)
                          : seqik::select_step_reflective<2>(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h);
    else pr = ilp ? seqik::select_step_reflective_ilp<1>(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h)
                  : seqik::select_step_reflective<1>(x, Jh, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h);
    out5[0] = step[0]; out5[1] = step[1]; out5[2] = step_h[0]; out5[3] = step_h[1]; out5[4] = pr;
}
