#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: a synthetic recording set streamed from pinned host memory in slabs,
RAW (un-aligned) key points with AlignPose.align_leg fused into the kernel prologue (SeqikAffine).

    python scripts/stream_config5.py --frames 10000000 --slab-frames 500000

--frames frames x 6 legs in total, cut into sequences of --frames-per-seq frames; a slab holds
--slab-frames frames (x 6 legs).  To keep host memory and data generation bounded only --unique slabs of
distinct synthetic data are generated; the stream cycles through them (the kernels cannot tell).  The
timed region is first submit -> last result in host memory, i.e. PCIe-inclusive (H2D 120 B, D2H 56 B
+ 216 B FK per leg-frame).  One process per GPU does the same on its share of the slabs (no collective);
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def launch_ranks_if_needed(argv):
    """`--gpus N` without a rank environment: start `torch.distributed.run --nproc-per-node N <this script> ...` as a child
    (before torch / the GPU are touched), relay its JSON line, exit with its code (same scheme as bench.py)."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=float(os.environ.get("SEQIK_BENCH_TIMEOUT", "1500")))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    sys.exit(r.returncode if (r.returncode or lines) else 3)


if __name__ == "__main__":
    launch_ranks_if_needed(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (HIP runtime first, see _lib.load)

from seqikpy_amd import _lib, data, synthetic, utils  # noqa: E402
from seqikpy_amd.streaming import PinnedArray, SeqikStream  # noqa: E402


def one_recording(args):
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    L, T = len(legs), args.slab_frames
    n_slabs = max(1, args.frames // T)
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    rng = np.random.default_rng(5)
    scales = 1.0 + 0.4 * rng.random(L)
    fixed = rng.normal(0.0, 2.0, (L, 3))
    want_fk = not args.no_fk
    base = np.stack([z[f"{l}_pose"] for l in legs])                                    # (L, 1000, 5, 3) aligned
    al = np.tile(base, (1, -(-T // base.shape[1]), 1, 1))[:, :T]                        # one slab of T frames
    tcs = [base[i, 0, 0].copy() for i in range(L)]                                      # the (constant) aligned coxa
    affs = [_lib.make_affine(fixed[i], scales[i], tcs[i]) for i in range(L)]
    raw = np.stack([(al[i] - tcs[i]) / scales[i] + fixed[i] for i in range(L)])        # a made-up camera frame
    layout = _lib.planar_layout(T)
    p = PinnedArray((1, L, 5, T, 3))
    p.array[0] = raw.transpose(0, 2, 1, 3)
    outs = [(PinnedArray((1, L, 7, T)), PinnedArray((1, L, T, 9, 3)) if want_fk else None) for _ in range(args.slots)]
    with SeqikStream(params, 1, T, affine=affs, layout=layout, want_fk=want_fk, n_slots=args.slots, carry=True,
                     frame_chunk=-1) as st:
        for k in range(2):                                                              # warm-up
            a, f = outs[k % args.slots]
            st.submit(p.array, a.array, f.array if f else None)
        st.wait()
        st.reset_carry()
        t0 = time.perf_counter()
        for k in range(n_slabs):
            a, f = outs[k % args.slots]
            st.submit(p.array, a.array, f.array if f else None)
        st.wait()
        dt = time.perf_counter() - t0
        st.reset_carry()
        a, f = outs[0]
        st.submit(p.array, a.array, f.array if f else None)                             # slab 0 again, from the seeds
        st.wait()
    n = min(T, 3000)
    direct = _lib.solve_seq(np.ascontiguousarray(raw[None, :, :T]), params, want_fk=False, affine=affs, frame_chunk=-1)
    serial = _lib.solve_seq(np.ascontiguousarray(raw[None, :, :n]), params, want_fk=False, affine=affs)
    got = a.array[0].transpose(0, 2, 1).copy()                                          # (L, T, 7)
    units = n_slabs * L * T
    bytes_per = 120 + 56 + (216 if want_fk else 0)
    for arr in [p] + [x for pair in outs for x in pair if x is not None]:
        arr.free()
    return ({"metric": "leg-IK solves/s, ONE recording streamed from host memory in time slabs (PCIe-inclusive)",
                      "value": units / dt, "unit": "leg-frame solves/s", "n_gpus": 1, "seconds": dt, "leg_frames": units,
                      "frames_total": n_slabs * T, "legs": L, "slabs": n_slabs, "slab_frames": T, "slots": args.slots,
                      "data": "df3d locomotion recording (fixture) repeated, RAW key points through a made-up camera frame, "
                              "AlignPose.align_leg fused into the kernels",
                      "mode": "carried stream (frame 0 of a slab warm-started from the last frame of the slab before, on the "
                              "device), frame chunks inside every slab (automatic parameters)",
                      "outputs": "7 angles" + (" + 9x3 FK" if want_fk else ""),
                      "pcie_GBps_total": units * bytes_per / dt / 1e9,
                      "check": {"streamed_slab_equals_direct_chunked_call_bitwise": bool(np.array_equal(got, direct["angles"][0])),
                                "max_abs_vs_serial_walk_first_frames": float(np.abs(got[:, :n] - serial["angles"][0]).max()),
                                "frames_walked_serially": n}})


def one_recording_over_ranks(args):
    """Config 5 on N ranks: ONE recording of --frames frames x 6 legs, rank r streams its contiguous slabs from its own
    pinned buffers (seqikpy_amd.stream_sharding); pass 1 (alignment statistics from all RAW slabs) on every rank."""
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1))
    backend = "nccl" if n_dev >= world else "gloo"     # ranks that share a GPU (rehearsal) talk over gloo
    dist.init_process_group(backend, rank=rank, world_size=world)
    res = one_recording_over_ranks_core(args, dist, world, rank, backend)
    if rank == 0:
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def one_recording_over_ranks_core(args, dist, world, rank, backend):
    """The body of one_recording_over_ranks on an EXISTING process group (bench.py's N > 1 line calls it as a leg).  Collective:
    every rank calls it; rank 0 gets the result, the others None."""
    from seqikpy_amd import stream_sharding
    coll = "cuda" if backend == "nccl" else "cpu"
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    L, T = len(legs), args.slab_frames
    n_slabs = max(world, args.frames // T)
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    rng = np.random.default_rng(5)
    scales, fixed = 1.0 + 0.4 * rng.random(L), rng.normal(0.0, 2.0, (L, 3))
    want_fk = not args.no_fk
    base = np.stack([z[f"{l}_pose"] for l in legs])
    al = np.tile(base, (1, -(-T // base.shape[1]), 1, 1))[:, :T]
    tcs = [base[i, 0, 0].copy() for i in range(L)]
    affs = [_lib.make_affine(fixed[i], scales[i], tcs[i]) for i in range(L)]
    raw = np.stack([(al[i] - tcs[i]) / scales[i] + fixed[i] for i in range(L)])
    # every slab holds the same T frames (T is a multiple of the fixture's 1000 frames, so the recording is the fixture
    # repeated end to end): one pinned input slab per rank, output buffers cycled -- host memory stays bounded
    p = PinnedArray((1, L, 5, T, 3))
    p.array[0] = raw.transpose(0, 2, 1, 3)
    outs = [(PinnedArray((1, L, 7, T)), PinnedArray((1, L, T, 9, 3)) if want_fk else None) for _ in range(args.slots + 1)]

    def get_out(k):
        a, f = outs[k % len(outs)]
        return a.array, (f.array if f else None)

    stats_pass = None
    if args.gpu_stats:
        n_tot = n_slabs * T
        ranks_q = [r for q in (0.45, 0.55) for r in (int(np.floor((n_tot - 1) * q)), min(int(np.floor((n_tot - 1) * q)) + 1, n_tot - 1))]
        stream_sharding.align_stats_all_slabs(lambda k: p.array, 1, T, L, [0])      # warm-up
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        order = stream_sharding.align_stats_all_slabs(lambda k: p.array, n_slabs, T, L, ranks_q)
        stats_pass = {"seconds_this_rank": time.perf_counter() - t0, "frames_per_leg": n_tot,
                      "what": "every rank extracts and sorts the 7 series per leg of ALL RAW slabs over its own PCIe link (no "
                              "exchange): the constants are whole-recording order statistics", "median_coxa_x_RF": float(order[0, 0, 0])}
    st = {}
    stream_sharding.stream_recording_sharded(lambda k: p.array, get_out, min(n_slabs, 2 * world), T, params, affine=affs,
                                             want_fk=want_fk, n_slots=args.slots)   # warm-up
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    k0, k1 = stream_sharding.stream_recording_sharded(lambda k: p.array, get_out, n_slabs, T, params, affine=affs, want_fk=want_fk,
                                                      n_slots=args.slots, stats=st)
    torch.cuda.synchronize()
    dist.barrier()
    mine = time.perf_counter() - t0
    t = torch.tensor([mine], dtype=torch.float64, device=coll)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "slabs": [int(k0), int(k1)], "seconds": st.get("seconds", 0.0),
                                  "h2d_GBps": st.get("h2d_GBps", 0.0), "d2h_GBps": st.get("d2h_GBps", 0.0),
                                  "boundary_rounds": st.get("boundary_rounds"), "restreams": st.get("restreams")})
    res = None
    if rank == 0:
        c, h, _ = _lib.frame_chunk_plan(T)
        units = n_slabs * L * T
        res = {"metric": "leg-IK solves/s, ONE recording streamed from host memory, frames sharded over the ranks "
                         "(PCIe-inclusive)", "value": units / dt, "unit": "leg-frame solves/s", "n_gpus": world,
               "backend": backend, "seconds": dt, "leg_frames": units, "frames_total": n_slabs * T, "legs": L,
               "slabs": n_slabs, "slab_frames": T, "frames_per_chunk": c, "run_in_frames": h,
               "outputs": "7 angles" + (" + 9x3 FK" if want_fk else ""), "ranks": seen,
               "alignment_statistics_pass": stats_pass,
               "data": "df3d locomotion recording (fixture) repeated, RAW key points through a made-up camera frame, "
                       "AlignPose.align_leg fused into the kernels"}
    p.free()
    for a, f in outs:
        a.free()
        if f:
            f.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); > 1 launches them (--one-recording only)")
    ap.add_argument("--frames", type=int, default=10_000_000)
    ap.add_argument("--slab-frames", type=int, default=500_000)
    ap.add_argument("--frames-per-seq", type=int, default=64)
    ap.add_argument("--unique", type=int, default=3, help="distinct slabs of synthetic data (cycled)")
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--no-fk", action="store_true")
    ap.add_argument("--pageable", action="store_true", help="use ordinary numpy buffers instead of pinned ones")
    ap.add_argument("--check", action="store_true", help="compare one slab with a direct solve on aligned data")
    ap.add_argument("--gpu-stats", action="store_true",
                    help="also time pass 1: AlignPose's whole-recording statistics from the RAW slabs on the GPU")
    ap.add_argument("--one-recording", action="store_true",
                    help="config 5 read literally: ONE recording of --frames frames x 6 legs streamed in time slabs "
                         "(carried warm start), every slab cut into frame chunks on the device; real locomotion poses "
                         "(the df3d fixture repeated) instead of the synthetic sequences")
    args = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        if not args.one_recording:
            raise SystemExit("--gpus N > 1 needs --one-recording (independent sequences need no coordination: run one "
                             "process per GPU on its share of the slabs)")
        return one_recording_over_ranks(args)
    if args.one_recording:
        print(json.dumps(one_recording(args)))
        return
    print(json.dumps(synthetic_sequences(args)))


def synthetic_sequences(args):
    """Config 5 as BASELINE.json words it: --frames synthetic frames x 6 legs (independent sequences of --frames-per-seq
    frames, iid in-workspace poses) streamed from pinned slabs, RAW key points, alignment fused into the kernel prologue.
    Returns the result line as a dict (bench.py calls this for its `configs["5"]` entry)."""
    legs = data.LEGS
    L, T = len(legs), args.frames_per_seq
    S = args.slab_frames // T
    n_slabs = max(1, args.frames // (S * T))
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    params = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    # a made-up camera frame per leg: raw = (aligned - template_coxa) / scale + fixed_coxa
    rng = np.random.default_rng(5)
    scales = 1.0 + 0.4 * rng.random(L)
    fixed = rng.normal(0.0, 2.0, (L, 3))
    affs = [_lib.make_affine(fixed[i], scales[i], data.TEMPLATE_NMF_LOCOMOTION[f"{l}_Coxa"]) for i, l in enumerate(legs)]
    want_fk = not args.no_fk
    layout = _lib.planar_layout(T)

    def alloc(shape):
        return PinnedArray(shape) if not args.pageable else type("A", (), {"array": np.empty(shape), "free": lambda s: None})()

    t_gen = time.perf_counter()
    slabs, aligned0 = [], None
    for u in range(args.unique):
        al = synthetic.synthetic_pose(S, T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION,
                                      variant="iid", seed=synthetic.SEED_BASE + 77 * u)
        if u == 0 and args.check:
            aligned0 = al[:256].copy()
            raw0 = None
        raw = np.empty_like(al)
        for i, l in enumerate(legs):
            tc = data.TEMPLATE_NMF_LOCOMOTION[f"{l}_Coxa"]
            raw[:, i] = (al[:, i] - tc) / scales[i] + fixed[i]
        if u == 0 and args.check:
            raw0 = raw[:256].copy()
        p = alloc((S, L, 5, T, 3))
        p.array[...] = raw.transpose(0, 1, 3, 2, 4)
        slabs.append(p)
        del al, raw
    outs = [(alloc((S, L, 7, T)), alloc((S, L, T, 9, 3)) if want_fk else None) for _ in range(args.slots)]
    t_gen = time.perf_counter() - t_gen

    with SeqikStream(params, S, T, affine=affs, layout=layout, want_fk=want_fk, n_slots=args.slots) as st:
        for k in range(min(3, n_slabs)):  # warm-up: allocator pools, first-launch costs
            a, f = outs[k % args.slots]
            st.submit(slabs[k % args.unique].array, a.array, f.array if f else None)
        st.wait()
        t0 = time.perf_counter()
        for k in range(n_slabs):
            a, f = outs[k % args.slots]
            st.submit(slabs[k % args.unique].array, a.array, f.array if f else None)
        st.wait()
        dt = time.perf_counter() - t0
        check = None
        if args.check:
            # the slab that ended up in outs[(n_slabs - 1) % slots] is slab (n_slabs - 1) % unique; re-run slab 0
            a, f = outs[0]
            st.submit(slabs[0].array, a.array, f.array if f else None)
            st.wait()
            got = a.array[:256].transpose(0, 1, 3, 2)
            direct = _lib.solve_seq(raw0, params, want_fk=want_fk, affine=affs)  # same RAW data, one blocking call
            ref = _lib.solve_seq(aligned0, params, want_fk=want_fk)              # the data before the made-up camera
            # frame: (aligned - tc) / s + f followed by the fused (raw - f) * s + tc is the identity only up to
            # rounding, and on i.i.d. targets an ulp in the input can flip the reference's iteration path
            err = np.abs(got - ref["angles"]).max(-1)
            check = {"streamed_equals_direct_fused_solve_bitwise": bool(np.array_equal(got, direct["angles"])) and
                     (not want_fk or bool(np.array_equal(f.array[:256], direct["fk"]))),
                     "median_abs_diff_vs_solve_on_prealigned": float(np.median(err)),
                     "leg_frames_gt_1e-4_vs_prealigned": int((err > 1e-4).sum()),
                     "leg_frames_checked": int(err.size)}

    # ---- pass 1: the alignment constants from the RAW slabs (AlignPose's quantile reductions) on the GPU -- timed AFTER the
    # stream here (its multi-GB device buffers, allocated and freed in front of the stream, cost the stream 15 % on this box)
    stats = None
    if args.gpu_stats:
        n_tot = n_slabs * S * T
        qs = (0.45, 0.55)
        ranks = [r for q in qs for r in (int(np.floor((n_tot - 1) * q)), min(int(np.floor((n_tot - 1) * q)) + 1, n_tot - 1))]
        with _lib.AlignStats(L, n_tot) as ast:
            ast.add(slabs[0].array, n_seq=S, n_frames=T, layout=layout)     # warm-up (allocations, first launch)
            ast.reset()
            t0 = time.perf_counter()
            for k in range(n_slabs):
                ast.add(slabs[k % args.unique].array, n_seq=S, n_frames=T, layout=layout)
            order = ast.finish(ranks)
            dt_stats = time.perf_counter() - t0
        stats = {"seconds": dt_stats, "frames_per_leg": n_tot, "leg_frames_per_s": n_tot * L / dt_stats,
                 "what": "7 series per leg extracted from the RAW slabs, radix-sorted, 4 order statistics each "
                         "(np.quantile 0.45 / 0.55 neighbours) -> fixed_coxa, mean segment lengths, scale",
                 "median_coxa_x_RF": float(order[0, 0, 0])}

    units = n_slabs * S * L * T
    bytes_per = 120 + 56 + (216 if want_fk else 0)
    out = {"metric": "leg-IK solves/s, streamed from host memory (PCIe-inclusive)", "value": units / dt,
           "unit": "leg-frame solves/s", "n_gpus": 1, "seconds": dt, "leg_frames": units,
           "frames_total": n_slabs * S * T, "legs": L, "slabs": n_slabs, "slab_frames": S * T,
           "frames_per_sequence": T, "slots": args.slots, "unique_slabs": args.unique, "pinned": not args.pageable,
           "fused_alignment": True, "outputs": "7 angles" + (" + 9x3 FK" if want_fk else ""),
           "pcie_GBps_total": units * bytes_per / dt / 1e9, "h2d_GBps": units * 120 / dt / 1e9,
           "d2h_GBps": units * (bytes_per - 120) / dt / 1e9, "datagen_seconds": t_gen, "check": check,
           "alignment_statistics_pass": stats}
    for p in slabs:
        p.free()
    for a, f in outs:
        a.free()
        if f:
            f.free()
    return out


if __name__ == "__main__":
    main()
