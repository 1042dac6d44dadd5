#!/usr/bin/env python3
"""ONE-GPU rehearsal of what rank 0 sees in the multi-GPU benchmark: solver launches that keep every SIMD slot busy
plus, once per step, a receive-like kernel (few long-lived workgroups moving 7 x 336 MB inside HBM) on its own
stream.  Prints one JSON line per variant: step time, and how long the receive took / lagged behind its step.

    python scripts/contention/gather_contention.py [--steps 30]
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_support as bench  # noqa: E402
from seqikpy_amd import _lib, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    args = ap.parse_args()
    so = os.path.join(HERE, "libfakerecv.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "fake_recv.hip")):
        import subprocess
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so,
                               os.path.join(HERE, "fake_recv.hip")])
    fr = ctypes.CDLL(so)
    fr.fake_recv.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
    fr.masked_stream_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int32, ctypes.c_int32]

    T, S = 64, 15625
    legs, body, pose, params = bench.make_workload(S, T, "iid", synthetic.SEED_BASE)
    L = len(legs)
    layout = _lib.planar_layout(T)
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    d_fk = [torch.zeros((S, L, T, 9, 3), dtype=torch.float64, device="cuda") for _ in range(3)]
    d_ang = [torch.zeros((S, L, 7, T), dtype=torch.float64, device="cuda") for _ in range(3)]
    block = d_ang[0].numel() * 8
    src = torch.zeros(7 * block // 8, dtype=torch.float64, device="cuda")
    dst = torch.zeros_like(src)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count

    def run(name, recv_wg=0, high_priority=True, reserve_cus=0):
        if reserve_cus:
            words = (n_cu + 31) // 32
            mask = (ctypes.c_uint32 * words)(*([0xFFFFFFFF] * words))
            for b in range(reserve_cus):
                mask[b // 32] &= ~(1 << (b % 32))
            handles = []
            for _ in range(3):
                h = ctypes.c_void_p()
                rc = fr.masked_stream_create(ctypes.byref(h), mask, words, 0)
                assert rc == 0, rc
                handles.append(h)
            streams = [torch.cuda.ExternalStream(h.value) for h in handles]
        else:
            streams = [torch.cuda.Stream() for _ in range(3)]
        hp = torch.cuda.Stream(priority=-1 if high_priority else 0)
        n = args.steps + args.warmup
        ev_solved = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        ev_r0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        ev_r1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        torch.cuda.synchronize()
        t0 = None
        for i in range(n):
            if i == args.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            s = streams[i % 3]
            _lib.solve_seq_device(d_pose.data_ptr(), S, L, T, params, d_ang[i % 3].data_ptr(), d_fk[i % 3].data_ptr(),
                                  stream=s.cuda_stream, layout=layout)
            ev_solved[i].record(s)
            if recv_wg:
                hp.wait_event(ev_solved[i])
                ev_r0[i].record(hp)
                fr.fake_recv(dst.data_ptr(), src.data_ptr(), 7 * block, recv_wg, ctypes.c_void_p(hp.cuda_stream))
                ev_r1[i].record(hp)
                if i >= 5:   # five angle buffers in the real pipeline: launch i must not overtake gather i - 5
                    streams[(i + 1) % 3].wait_event(ev_r1[i - 4])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        out = {"variant": name, "ms_per_step": round(ms, 2), "solves_per_s": round(S * L * T / ms * 1e3 / 1e8, 3) * 1e8,
               "recv_workgroups": recv_wg, "high_priority": high_priority, "reserved_cus": reserve_cus}
        if recv_wg:
            dur = [ev_r0[i].elapsed_time(ev_r1[i]) for i in range(args.warmup, n)]
            lag = [ev_solved[i].elapsed_time(ev_r1[i]) for i in range(args.warmup, n)]
            out.update(recv_ms_mean=round(float(np.mean(dur)), 2), recv_ms_max=round(float(np.max(dur)), 2),
                       lag_after_step_ms_mean=round(float(np.mean(lag)), 2), lag_after_step_ms_max=round(float(np.max(lag)), 2),
                       recv_GBps=round(7 * block / (np.mean(dur) * 1e-3) / 1e9, 1))
        print(json.dumps(out), flush=True)
        if reserve_cus:
            torch.cuda.synchronize()
            for h in handles:
                fr.masked_stream_destroy(h)

    run("solver only")
    run("recv alone baseline is below; recv 28 wg, high priority", recv_wg=28)
    run("recv 28 wg, normal priority", recv_wg=28, high_priority=False)
    run("recv 56 wg, high priority", recv_wg=56)
    run("solver on all but 8 CUs, no recv", reserve_cus=8)
    run("solver on all but 8 CUs, recv 28 wg", recv_wg=28, reserve_cus=8)
    run("solver on all but 16 CUs, recv 28 wg", recv_wg=28, reserve_cus=16)
    # the receive kernel with the GPU to itself
    hp = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for wg in (28, 56):
        torch.cuda.synchronize()
        e0.record(hp)
        fr.fake_recv(dst.data_ptr(), src.data_ptr(), 7 * block, wg, ctypes.c_void_p(hp.cuda_stream))
        e1.record(hp)
        torch.cuda.synchronize()
        print(json.dumps({"variant": f"recv {wg} wg alone", "recv_ms": round(e0.elapsed_time(e1), 2),
                          "recv_GBps": round(7 * block / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
