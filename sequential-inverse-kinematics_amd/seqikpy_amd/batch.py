"""Many recordings in one launch.

The reference processes one recording per ``LegInvKinSeq`` object (``seqikpy/leg_inverse_kinematics.py:324``) and
its parallel example forks one process per leg (``examples/example_leg_inv_kinematics_parallel.py:186``).  On
the MI355X the unit of parallelism is the chain = (recording, leg), so a set of recordings -- flies, trials --
is best handed over at once: ``run_ik_and_fk_many`` stacks them along the sequence axis of the C ABI and returns
what ``LegInvKinSeq.run_ik_and_fk`` would have returned for each of them, bit for bit.

Recordings may differ in length: they are bucketed by length and every bucket is one ``seqik_solve_seq`` call
(optionally the shorter ones of a bucket are padded by repeating their last frame -- a repeated frame converges
at once from its own warm start -- and the padding is cut off again).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .data import DOFS, INITIAL_ANGLES
from .kinematic_chain import KinematicChainSeq


def run_ik_and_fk_many(recordings: Sequence[Dict[str, np.ndarray]], kinematic_chain_class: KinematicChainSeq,
                       initial_angles: Optional[Dict[str, Dict[str, np.ndarray]]] = None,
                       pad_to_multiple: int = 0, device: int = -1,
                       leg_affine: Optional[Dict[str, tuple]] = None, frame_parallel=None,
                       reports: Optional[list] = None
                       ) -> List[Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]]:
    """``LegInvKinSeq(rec, kinematic_chain_class, initial_angles).run_ik_and_fk(frame_parallel=...)`` for every ``rec``.

    ``frame_parallel``: as ``LegInvKinSeq.run_ik_and_fk`` (None = its default, ``"auto"``; ``False`` = the serial walk, which is
    also the faster choice once a call carries enough recordings to fill the GPU by itself -- some 40 000 chains: the chunks'
    run-in frames are extra work).  With ``"auto"`` the chunk geometry depends on a recording's length only, so every recording gets the bits it would get alone (with
    ``pad_to_multiple`` it is the PADDED length that counts).  ``reports``: a list that receives one
    ``frame_chunk_report`` dict per recording.

    recordings: dicts ``"<leg>_leg" -> (N_i, 5, 3)`` holding the same leg keys (other keys are ignored);
    all legs of one recording must have the same number of frames.  ``pad_to_multiple`` > 0 rounds the
    lengths up to that multiple so that recordings of similar length share a launch.
    Returns a list of ``(joint_angles_dict, forward_kinematics_dict)`` in input order."""
    from .leg_inverse_kinematics import chunk_report, default_frame_parallel
    if initial_angles is None:
        initial_angles = INITIAL_ANGLES
    if frame_parallel is None:
        frame_parallel = default_frame_parallel()
    chunk_opts = dict(frame_chunk=0)
    if frame_parallel:
        fp = frame_parallel if isinstance(frame_parallel, dict) else {}
        chunk_opts = dict(frame_chunk=int(fp.get("chunk", -1)), frame_halo=int(fp.get("halo", 0)),
                          chunk_tol=float(fp.get("tol", 0.0)), chunk_rounds=int(fp.get("rounds", 0)), want_chunk_flags=True)
    kc = kinematic_chain_class
    if not recordings:
        return []
    segs = [(name, name.split("_")[0]) for name in recordings[0]
            if "leg" in name.lower() and f"{name.split('_')[0]}_Coxa" in kc.body_size]
    if not segs:
        raise ValueError("no leg of the recordings is covered by the kinematic chain's body_size")
    legs = [_lib.make_leg_params(leg, kc.bounds_dof, kc.body_size, initial_angles) for _, leg in segs]
    affine = [_lib.make_affine(*leg_affine[leg]) for _, leg in segs] if leg_affine is not None else None
    lengths = []
    for r in recordings:
        n = {np.asarray(r[name]).shape[0] for name, _ in segs}
        if len(n) != 1:
            raise ValueError("all legs of a recording must have the same number of frames")
        lengths.append(n.pop())
    buckets: Dict[int, List[int]] = {}
    for i, n in enumerate(lengths):
        key = n if pad_to_multiple <= 0 else -(-n // pad_to_multiple) * pad_to_multiple
        buckets.setdefault(key, []).append(i)
    results: List[Optional[Tuple[dict, dict]]] = [None] * len(recordings)
    for n_pad, idx in buckets.items():
        pose = np.empty((len(idx), len(segs), n_pad, 5, 3))
        for s, i in enumerate(idx):
            for li, (name, _) in enumerate(segs):
                a = np.asarray(recordings[i][name], dtype=np.float64)[:, :5, :]
                pose[s, li, :lengths[i]] = a
                pose[s, li, lengths[i]:] = a[-1] if lengths[i] else 0.0
        out = _lib.solve_seq(pose, legs, want_fk=True, device=device, affine=affine, **chunk_opts) if n_pad else None
        reps = chunk_report(out, [leg for _, leg in segs], n_pad) if n_pad else [{} for _ in idx]
        for s, i in enumerate(idx):
            n = lengths[i]
            ang, fk = {}, {}
            for li, (name, leg) in enumerate(segs):
                for d, dof in enumerate(DOFS):
                    ang[f"Angle_{leg}_{dof}"] = out["angles"][s, li, :n, d].copy() if n_pad else np.zeros(0)
                fk[name] = out["fk"][s, li, :n].copy() if n_pad else np.zeros((0, 9, 3))
            results[i] = (ang, fk)
            if reports is not None:
                while len(reports) < len(recordings):
                    reports.append({})
                reports[i] = reps[s]
    return results
