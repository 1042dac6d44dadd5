"""On-disk format for results that are produced slab by slab (``SeqikStream``, BASELINE config 5) -- SURVEY.md 8(f-4).

The reference writes one pickle per recording at the very end (``leg_joint_angles.pkl`` = ``{"Angle_<leg>_<dof>":
(N,)}``, ``forward_kinematics.pkl`` = ``{"<leg>_leg": (N, 9, 3)}``; ``seqikpy/leg_inverse_kinematics.py:392-401``,
loaded again by ``seqikpy/visualization.py:191-213`` through ``seqikpy/utils.py:235-245``).  A 10 M-frame run does not
want to hold 4 GB of results in one Python dict before anything reaches the disk, so a streamed run writes a
DIRECTORY:

    manifest.json          legs, DOF order, frames per slab, slabs written so far, what each file holds
    slab_000000.npz ...    one uncompressed ``.npz`` per slab with EXACTLY the reference's dict layout for the
                           frames of that slab: ``Angle_<leg>_<dof>`` -> (T,) float64 and ``<leg>_leg`` -> (T, 9, 3)
                           (for S > 1 sequences per slab the arrays carry a leading sequence axis: (S, T), (S, T, 9, 3))

so that every slab on its own is a valid (short) recording for the reference's consumers, slabs can be written as soon
as ``SeqikStream`` hands them back, and ``SlabReader.load_all()`` / ``to_pickles()`` give back the reference's
whole-recording dictionaries / pickle files.
"""
import json
import os
import pickle
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from .data import DOFS

FORMAT = "seqikpy_amd.slabs/1"


def results_to_dicts(legs: List[str], angles: np.ndarray, fk: Optional[np.ndarray]):
    """``angles (S, L, T, 7)``, ``fk (S, L, T, 9, 3)`` -> the reference's two dictionaries (leading axis dropped for
    S == 1)."""
    S = angles.shape[0]
    ja, fkd = {}, {}
    for li, leg in enumerate(legs):
        for d, dof in enumerate(DOFS):
            a = angles[:, li, :, d]
            ja[f"Angle_{leg}_{dof}"] = np.ascontiguousarray(a[0] if S == 1 else a)
        if fk is not None:
            f = fk[:, li]
            fkd[f"{leg}_leg"] = np.ascontiguousarray(f[0] if S == 1 else f)
    return ja, fkd


class SlabWriter:
    """``w = SlabWriter(directory, legs, frames_per_slab); w.write(angles, fk); ...; w.close()``."""

    def __init__(self, directory, legs: List[str], frames_per_slab: int, n_seq: int = 1, in_time: bool = True,
                 overwrite: bool = False):
        self.dir = str(directory)
        os.makedirs(self.dir, exist_ok=True)
        if not overwrite and os.path.exists(os.path.join(self.dir, "manifest.json")):
            raise FileExistsError(f"{self.dir} already holds a slab set (pass overwrite=True)")
        self.manifest = {"format": FORMAT, "legs": list(legs), "dofs": list(DOFS), "frames_per_slab": int(frames_per_slab),
                         "sequences_per_slab": int(n_seq),
                         # in_time: consecutive slabs are consecutive pieces of the same recording(s) (carried stream);
                         # otherwise every slab holds other recordings
                         "in_time": bool(in_time), "slabs": [],
                         "keys": {"Angle_<leg>_<dof>": "(T,) float64 joint angle in rad [(S, T) when sequences_per_slab > 1]",
                                  "<leg>_leg": "(T, 9, 3) float64 joint positions of the stage-4 chain + origin, as "
                                               "forward_kinematics.pkl [(S, T, 9, 3)]"}}
        self._flush()

    def _flush(self):
        tmp = os.path.join(self.dir, "manifest.json.tmp")
        with open(tmp, "w") as f:
            json.dump(self.manifest, f, indent=1)
        os.replace(tmp, os.path.join(self.dir, "manifest.json"))

    def write(self, angles: np.ndarray, fk: Optional[np.ndarray] = None) -> str:
        """One slab: ``angles (S, L, T, 7)`` (+ ``fk (S, L, T, 9, 3)``) as ``SeqikStream`` returns them."""
        legs = self.manifest["legs"]
        if angles.ndim != 4 or angles.shape[1] != len(legs) or angles.shape[3] != 7:
            raise ValueError(f"angles must have shape (S, {len(legs)}, T, 7), got {angles.shape}")
        if angles.shape[0] != self.manifest["sequences_per_slab"]:
            raise ValueError("number of sequences differs from the manifest's sequences_per_slab")
        if fk is not None and fk.shape != angles.shape[:3] + (9, 3):
            raise ValueError(f"fk must have shape {angles.shape[:3] + (9, 3)}, got {fk.shape}")
        ja, fkd = results_to_dicts(legs, angles, fk)
        k = len(self.manifest["slabs"])
        name = f"slab_{k:06d}.npz"
        np.savez(os.path.join(self.dir, name), **ja, **fkd)
        self.manifest["slabs"].append({"file": name, "frames": int(angles.shape[2]), "has_fk": fk is not None})
        self._flush()
        return name

    def close(self):
        self._flush()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class SlabReader:
    def __init__(self, directory):
        self.dir = str(directory)
        with open(os.path.join(self.dir, "manifest.json")) as f:
            self.manifest = json.load(f)
        if self.manifest.get("format") != FORMAT:
            raise ValueError(f"{self.dir}: not a {FORMAT} slab set")

    def __len__(self):
        return len(self.manifest["slabs"])

    def load_slab(self, k: int) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
        """``(joint_angles_dict, forward_kinematics_dict)`` of slab ``k``, the reference's layout."""
        with np.load(os.path.join(self.dir, self.manifest["slabs"][k]["file"])) as z:
            ja = {key: z[key] for key in z.files if key.startswith("Angle_")}
            fk = {key: z[key] for key in z.files if key.endswith("_leg")}
        return ja, fk

    def __iter__(self) -> Iterator[Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]]:
        for k in range(len(self)):
            yield self.load_slab(k)

    def load_all(self) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
        """Whole-recording dictionaries: slabs joined along the frame axis (``in_time`` sets) or, for sets whose
        slabs hold different recordings, along a leading sequence axis."""
        parts = list(self)
        if not parts:
            return {}, {}
        multi = self.manifest["sequences_per_slab"] > 1
        if self.manifest["in_time"]:
            axis = 1 if multi else 0
            join = lambda arrs: np.concatenate(arrs, axis=axis)  # noqa: E731
        else:
            join = lambda arrs: (np.concatenate(arrs, axis=0) if multi else np.stack(arrs))  # noqa: E731
        ja = {key: join([p[0][key] for p in parts]) for key in parts[0][0]}
        fk = {key: join([p[1][key] for p in parts]) for key in parts[0][1]}
        return ja, fk

    def to_pickles(self, export_path) -> None:
        """Writes ``leg_joint_angles.pkl`` / ``forward_kinematics.pkl`` the way ``run_ik_and_fk(export_path=...)``
        does (``seqikpy/leg_inverse_kinematics.py:392-401``)."""
        ja, fk = self.load_all()
        os.makedirs(str(export_path), exist_ok=True)
        with open(os.path.join(str(export_path), "leg_joint_angles.pkl"), "wb") as f:
            pickle.dump(ja, f)
        if fk:
            with open(os.path.join(str(export_path), "forward_kinematics.pkl"), "wb") as f:
                pickle.dump(fk, f)


def stream_recording_to_slabs(pose: np.ndarray, legs_params, leg_names: List[str], directory, slab_frames: int,
                              affine=None, want_fk: bool = True, n_slots: int = 3, device: int = -1,
                              overwrite: bool = False) -> "SlabReader":
    """``pose (S, L, N, 5, 3)`` pushed through a carried ``SeqikStream`` in slabs of ``slab_frames`` frames; every slab
    is written to ``directory`` as soon as its results are on the host (the recordings advance in lock step, frame 0
    of a slab is warm-started on the device from the last frame of the slab before).  Joined over the slabs the
    result equals ``_lib.solve_seq(pose, ...)`` bit for bit."""
    from . import _lib
    from .streaming import SeqikStream
    pose = np.asarray(pose, dtype=np.float64)
    _lib._check_finite(pose)
    S, L, N = pose.shape[:3]
    T = int(slab_frames)
    if N % T:
        raise ValueError("the number of frames must be a multiple of slab_frames (pad or use a divisor)")
    with SlabWriter(directory, leg_names, T, n_seq=S, in_time=True, overwrite=overwrite) as w, \
            SeqikStream(legs_params, S, T, affine=affine, want_fk=want_fk, n_slots=n_slots, carry=True, device=device) as st:
        pending = []  # slabs in flight: (host buffers); a slot is reused after n_slots further submits

        def drain(keep):
            while len(pending) > keep:
                _, a, f = pending.pop(0)
                w.write(a, f)
        for k in range(N // T):
            p = np.ascontiguousarray(pose[:, :, k * T:(k + 1) * T])
            a = np.empty((S, L, T, 7))
            f = np.empty((S, L, T, 9, 3)) if want_fk else None
            st.submit(p, a, f)  # blocks only while all slots are busy: slab k - n_slots has then reached the host
            pending.append((p, a, f))
            drain(n_slots)      # everything older than the slots in flight is complete
        st.wait()
        drain(0)
    return SlabReader(directory)
