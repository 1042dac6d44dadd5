"""ORACLE / TEST INFRASTRUCTURE ONLY -- the reference's frame loop over REAL scipy, without the reference.

The C restatement (``oracle/seqik_oracle.c``) re-implements scipy's ``least_squares`` (TRF).  This module
is the other half of the pin: it runs the *actual* ``scipy.optimize.least_squares`` exactly the way
IKPy 3.3.4 calls it, in the reference's loop order

    LegInvKinSeq.run_ik_and_fk          seqikpy/leg_inverse_kinematics.py:324-403  (legs x stages)
    LegInvKinSeq.calculate_ik_stage     seqikpy/leg_inverse_kinematics.py:200-322  (frames, warm start :272)
    KinematicChainSeq.create_leg_chain  seqikpy/kinematic_chain.py:99-421          (chain per stage / frame)
    LegInvKinBase.calculate_ik / _fk    seqikpy/leg_inverse_kinematics.py:62-77

with chains described by the build's own host mirror (``seqikpy_amd.kinematic_chain`` -- plain link
tables) turned into the build-owned IKPy stand-in of ``oracle/shim/ikpy`` (frame matrices +
``Chain.inverse_kinematics`` -> scipy).  It needs neither ``/root/reference`` nor a GPU, so it also runs
on the GPU box, where it is (i) a second check of the C oracle on *new* inputs (tests/test_scipy_oracle.py)
and (ii) the "IKPy-equivalent CPU path" of the benchmark's ``cpu_baseline`` (process pool,
one task per (sequence, leg), the shape of examples/example_leg_inv_kinematics_parallel.py:186-187).

Pinned by ``tests/golden/anipose_scipy_cut.npz`` / ``df3d_100.npz`` (the reference's unmodified source run
in the build container): it reproduces those runs bit for bit, status and nfev included.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for _p in (os.path.join(_HERE, "shim"), os.path.join(_ROOT, "sequential-inverse-kinematics_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
SEGMENTS = ["Coxa", "Femur", "Tibia", "Tarsus"]
# joint-angle columns a stage stores, by link index of its chain (leg_inverse_kinematics.py:285-320)
STORED = {1: ((1, 0), (2, 1)), 2: ((3, 2), (4, 3)), 3: ((5, 4), (6, 5)), 4: ((7, 6),)}


def _to_ikpy(chain):
    """seqikpy_amd.kinematic_chain.Chain (description) -> shim ikpy Chain (frame matrices + scipy)."""
    from ikpy.chain import Chain
    from ikpy.link import OriginLink, URDFLink
    links = []
    for l in chain.links:
        if l.name == "Base link":
            links.append(OriginLink())
        else:
            links.append(URDFLink(name=l.name, origin_translation=l.origin_translation,
                                  origin_orientation=l.origin_orientation, rotation=l.rotation,
                                  joint_type=l.joint_type, bounds=l.bounds))
    return Chain(name=chain.name, links=links)


def seq_leg(pose, leg, bounds_dof, body_size, initial_angles, stages=(1, 2, 3, 4)):
    """One leg, all frames: returns dict(angles (N, 7), fk (N, 9, 3), status (N, 4), nfev (N, 4)).

    ``pose`` (N, 5, 3) aligned key points; dict-shaped parameters as in the reference."""
    from ikpy.chain import Chain
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    pose = np.asarray(pose, dtype=np.float64)
    n = pose.shape[0]
    factory = KinematicChainSeq(bounds_dof, [leg], body_size)
    angles = np.zeros((n, 7))
    fk = np.zeros((n, 9, 3))
    status = np.full((n, 4), -1, dtype=np.int32)
    nfev = np.zeros((n, 4), dtype=np.int32)
    origin = pose[:, 0]
    adict = {}
    for stage in stages:
        target = pose[:, stage] - origin
        seed = np.asarray(initial_angles[leg][f"stage_{stage}"], dtype=np.float64)
        sol = np.empty((n, len(seed)))
        chain = _to_ikpy(factory.create_leg_chain(leg, stage=1)) if stage == 1 else None
        for t in range(n):
            if stage > 1:
                chain = _to_ikpy(factory.create_leg_chain(leg, stage=stage, angles=adict, t=t))
            x0 = seed if t == 0 else sol[t - 1]
            Chain.solve_log = []
            sol[t] = chain.inverse_kinematics(target_position=target[t], initial_position=x0)
            status[t, stage - 1], nfev[t, stage - 1] = Chain.solve_log[-1]
            if stage == 4:
                frames = chain.forward_kinematics(sol[t], full_kinematics=True)
                fk[t] = np.array([m[:3, 3] for m in frames]) + origin[t]
        Chain.solve_log = None
        for link, dof in STORED[stage]:
            angles[:, dof] = sol[:, link]
            adict[f"Angle_{leg}_{DOFS[dof]}"] = angles[:, dof]
    return dict(angles=angles, fk=fk, status=status, nfev=nfev)


def generic_leg(pose, leg, bounds_dof, body_size, initial_angles):
    """LegInvKinGeneric.calculate_ik_stage (seqikpy/leg_inverse_kinematics.py:474-499) over real scipy: ONE 9-link chain
    (KinematicChainGeneric, kinematic_chain.py:464-530) follows the claw, frame t warm-started from frame t - 1, frame 0
    from initial_angles[leg]["stage_4"].  Returns dict(angles (N, 7) in DOFS order, fk (N, 9, 3), status (N,), nfev (N,))."""
    from ikpy.chain import Chain
    from seqikpy_amd.kinematic_chain import KinematicChainGeneric
    pose = np.asarray(pose, dtype=np.float64)
    n = pose.shape[0]
    chain = _to_ikpy(KinematicChainGeneric(bounds_dof, [leg], body_size).create_leg_chain(leg))
    names = [l.name for l in chain.links]
    cols = [names.index(f"{leg}_{d}") for d in DOFS]
    origin = pose[:, 0]
    target = pose[:, 4] - origin
    sol = np.empty((n, 9))
    fk = np.zeros((n, 9, 3))
    status = np.full(n, -1, dtype=np.int32)
    nfev = np.zeros(n, dtype=np.int32)
    x0 = np.asarray(initial_angles[leg]["stage_4"], dtype=np.float64)
    for t in range(n):
        Chain.solve_log = []
        sol[t] = chain.inverse_kinematics(target_position=target[t], initial_position=x0 if t == 0 else sol[t - 1])
        status[t], nfev[t] = Chain.solve_log[-1]
        fk[t] = np.array([m[:3, 3] for m in chain.forward_kinematics(sol[t], full_kinematics=True)]) + origin[t]
    Chain.solve_log = None
    return dict(angles=sol[:, cols], fk=fk, status=status, nfev=nfev)


def generic_leg_arrays(pose, seg, bounds, seeds, leg="RF"):
    """Same, with the array-shaped parameters of the C oracle / the fixtures (seg[4], bounds[7,2], seeds[27])."""
    body = {f"{leg}_{s}": float(seg[i]) for i, s in enumerate(SEGMENTS)}
    bd = {f"{leg}_{d}": (float(bounds[i][0]), float(bounds[i][1])) for i, d in enumerate(DOFS)}
    return generic_leg(pose, leg, bd, body, {leg: {"stage_4": np.asarray(seeds[18:27], dtype=np.float64)}})


def seq_leg_arrays(pose, seg, bounds, seeds, leg="RF"):
    """Same, with the array-shaped parameters of the C oracle / the fixtures (seg[4], bounds[7,2], seeds[27])."""
    body = {f"{leg}_{s}": float(seg[i]) for i, s in enumerate(SEGMENTS)}
    bd = {f"{leg}_{d}": (float(bounds[i][0]), float(bounds[i][1])) for i, d in enumerate(DOFS)}
    cuts = np.cumsum([0, 4, 6, 8, 9])
    init = {leg: {f"stage_{k + 1}": np.asarray(seeds[cuts[k]:cuts[k + 1]], dtype=np.float64) for k in range(4)}}
    return seq_leg(pose, leg, bd, body, init)


def _pool_task(args):
    pose, leg, bounds_dof, body_size, initial_angles = args
    import warnings
    warnings.filterwarnings("ignore")
    return seq_leg(pose, leg, bounds_dof, body_size, initial_angles)["angles"]


def pool_run(pose, legs, bounds_dof, body_size, initial_angles, processes, context="fork"):
    """``pose`` (S, L, N, 5, 3): one task per (sequence, leg) on a process pool -- the reference's
    parallel example (examples/example_leg_inv_kinematics_parallel.py:186-187).  Returns (S, L, N, 7).
    Only call this with ``context="fork"`` from a process that has not touched the GPU (see
    ``pool_run_subprocess``)."""
    import multiprocessing as mp
    S, L = pose.shape[:2]
    tasks = [(pose[s, li], leg, bounds_dof, body_size, initial_angles) for s in range(S) for li, leg in enumerate(legs)]
    with mp.get_context(context).Pool(processes=processes) as pool:
        out = pool.map(_pool_task, tasks, chunksize=1)
    return np.stack(out).reshape(S, L, pose.shape[2], 7)


def pool_run_subprocess(pose, legs, bounds_dof, body_size, initial_angles, processes, timeout=600):
    """``pool_run`` in a fresh interpreter (a process that holds a HIP context must not fork, and its
    children should not have to import torch).  Returns (angles (S, L, N, 7), seconds spent in the pool)."""
    import pickle
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        job, res = os.path.join(tmp, "job.pkl"), os.path.join(tmp, "res.pkl")
        with open(job, "wb") as f:
            pickle.dump(dict(pose=np.ascontiguousarray(pose), legs=list(legs), bounds_dof=dict(bounds_dof),
                             body_size=dict(body_size), initial_angles=initial_angles, processes=int(processes)), f)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), job, res], check=True, timeout=timeout, env=env)
        with open(res, "rb") as f:
            out = pickle.load(f)
    return out["angles"], out["seconds"]


if __name__ == "__main__":
    import pickle
    import time
    import warnings
    warnings.filterwarnings("ignore")
    with open(sys.argv[1], "rb") as f:
        job = pickle.load(f)
    t0 = time.perf_counter()
    ang = pool_run(job["pose"], job["legs"], job["bounds_dof"], job["body_size"], job["initial_angles"], job["processes"])
    dt = time.perf_counter() - t0
    with open(sys.argv[2], "wb") as f:
        pickle.dump(dict(angles=ang, seconds=dt), f)
