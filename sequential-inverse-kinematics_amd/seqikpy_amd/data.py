"""Constants of the leg-IK path, re-declared as data.

Values are those of the reference's ``seqikpy/data.py:4-41,139-167`` (default
``INITIAL_ANGLES``, ``BOUNDS``, ``NMF_TEMPLATE`` for the RF/LF grooming setup) and of
``examples/example_leg_inv_kinematics_parallel.py:21-140`` (six-leg locomotion setup).
``tests/test_data_constants.py`` checks them against the reference when it is present.
"""
import numpy as np

LEGS = ["RF", "LF", "RM", "LM", "RH", "LH"]

#: Joint order used by the C ABI (include/seqik.h) and every (.., 7) angle array.
DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]

#: Leg segments whose lengths parametrise the chain (body_size["<leg>_<segment>"]).
SEGMENTS = ["Coxa", "Femur", "Tibia", "Tarsus"]


def _seeds(yaw, pitch, roll, ctr_roll):
    # link order: Base, ThC yaw, ThC pitch, [ThC roll], CTr pitch, [CTr roll, FTi pitch, TiTa pitch, Claw]
    return {
        "stage_1": np.array([0.0, yaw, pitch, -2.14]),
        "stage_2": np.array([0.0, yaw, pitch, roll, -2.14, 1.4]),
        "stage_3": np.array([0.0, yaw, pitch, roll, -2.14, ctr_roll, 1.48, 0.0]),
        "stage_4": np.array([0.0, yaw, pitch, roll, -2.14, ctr_roll, 1.48, 0.0, 0.0]),
    }


INITIAL_ANGLES = {
    "RF": _seeds(0.45, -0.07, -0.32, -1.25),
    "LF": _seeds(-0.45, -0.07, 0.32, 1.25),
}

# Lower bound of a DOF should be strictly lower than the initial angle,
# upper bound strictly bigger (scipy rejects a seed outside its bounds).
BOUNDS = {
    "RF_ThC_roll": (np.deg2rad(-130), np.deg2rad(50)),
    "RF_ThC_yaw": (np.deg2rad(-50), np.deg2rad(50)),
    "RF_ThC_pitch": (np.deg2rad(-40), np.deg2rad(60)),
    "RF_CTr_pitch": (np.deg2rad(-180), np.deg2rad(0)),
    "RF_CTr_roll": (np.deg2rad(-150), np.deg2rad(0)),
    "RF_FTi_pitch": (np.deg2rad(0), np.deg2rad(170)),
    "RF_TiTa_pitch": (np.deg2rad(-150), np.deg2rad(0)),
    "LF_ThC_roll": (np.deg2rad(-50), np.deg2rad(130)),
    "LF_ThC_yaw": (np.deg2rad(-50), np.deg2rad(50)),
    "LF_ThC_pitch": (np.deg2rad(-40), np.deg2rad(60)),
    "LF_CTr_pitch": (np.deg2rad(-180), np.deg2rad(0)),
    "LF_CTr_roll": (np.deg2rad(0), np.deg2rad(150)),
    "LF_FTi_pitch": (np.deg2rad(0), np.deg2rad(170)),
    "LF_TiTa_pitch": (np.deg2rad(-150), np.deg2rad(0)),
}

# Pose of each body landmark in the NeuroMechFly v0.0.6 model; each leg segment
# name denotes the joint at its proximal end (RF_Coxa = Thorax-Coxa joint).
NMF_TEMPLATE = {
    "RF_Coxa": np.array([0.33, -0.17, 1.07]),
    "RF_Femur": np.array([0.33, -0.17, 0.67]),
    "RF_Tibia": np.array([0.33, -0.17, -0.02]),
    "RF_Tarsus": np.array([0.33, -0.17, -0.56]),
    "RF_Claw": np.array([0.33, -0.17, -1.19]),
    "LF_Coxa": np.array([0.33, 0.17, 1.07]),
    "LF_Femur": np.array([0.33, 0.17, 0.67]),
    "LF_Tibia": np.array([0.33, 0.17, -0.02]),
    "LF_Tarsus": np.array([0.33, 0.17, -0.56]),
    "LF_Claw": np.array([0.33, 0.17, -1.19]),
    "R_Antenna_base": np.array([1.01, -0.10, 1.41]),
    "L_Antenna_base": np.array([1.01, 0.10, 1.41]),
    "R_Antenna_edge": np.array([1.06, -0.10, 1.14]),
    "L_Antenna_edge": np.array([1.06, 0.10, 1.14]),
    "R_post_vertical": np.array([0.7, -0.2, 1.59]),
    "L_post_vertical": np.array([0.7, 0.2, 1.59]),
    "R_wing": np.array([0.08, -0.4, 1.43]),
    "L_wing": np.array([0.08, 0.4, 1.43]),
    "Neck": np.array([0.53, 0.0, 1.3]),
    "Thorax_mid": np.array([0.08, 0.0, 1.43]),
    "L_dorsal_hum": np.array([0.41, 0.37, 1.32]),
    "R_dorsal_hum": np.array([0.41, -0.37, 1.32]),
}

# Key points to align per segment (anipose naming; reference ``seqikpy/data.py:80-101``)
PTS2ALIGN = {
    "R_head": ["base_anten_R", "tip_anten_R"],
    "RF_leg": ["thorax_coxa_R", "coxa_femur_R", "femur_tibia_R", "tibia_tarsus_R", "claw_R"],
    "Thorax": ["thorax_wing_R", "thorax_midpoint_tether", "thorax_wing_L"],
    "L_head": ["base_anten_L", "tip_anten_L"],
    "LF_leg": ["thorax_coxa_L", "coxa_femur_L", "femur_tibia_L", "tibia_tarsus_L", "claw_L"],
}



def get_pts2align(path: str):
    """``PTS2ALIGN`` without the leg(s) the recording's path says were not tracked (reference ``seqikpy/data.py:101-113``:
    ``_RF`` / ``_LF`` / ``_RLF`` | ``_LRF`` in the path)."""
    pts = PTS2ALIGN.copy()
    if "_RF" in path:
        del pts["RF_leg"]
    elif "_LF" in path:
        del pts["LF_leg"]
    elif "_RLF" in path or "_LRF" in path:
        del pts["LF_leg"]
        del pts["RF_leg"]
    return pts


# every key point the alignment uses, in PTS2ALIGN's order (reference ``seqikpy/data.py:116-134``)
SKELETON = [name for names in PTS2ALIGN.values() for name in names]

# ---------------------------------------------------------------------------
# Six-leg locomotion setup (df3d recording)
# ---------------------------------------------------------------------------
_PI = 3.141592653589793


def _leg_template(x, y, zs):
    names = ["Coxa", "Femur", "Tibia", "Tarsus", "Claw"]
    return {n: np.array([x, y, z]) for n, z in zip(names, zs)}


TEMPLATE_NMF_LOCOMOTION = {}
for _leg, (_x, _y, _zs) in {
    "RF": (0.35, -0.27, [0.400, -0.025, -0.731, -1.249, -1.912]),
    "LF": (0.35, 0.27, [0.400, -0.025, -0.731, -1.249, -1.912]),
    "RM": (0, -0.125, [0, -0.182, -0.965, -1.633, -2.328]),
    "LM": (0, 0.125, [0, -0.182, -0.965, -1.633, -2.328]),
    "RH": (-0.215, -0.087, [-0.073, -0.272, -1.108, -1.793, -2.588]),
    "LH": (-0.215, 0.087, [-0.073, -0.272, -1.108, -1.793, -2.588]),
}.items():
    for _seg, _pos in _leg_template(_x, _y, _zs).items():
        TEMPLATE_NMF_LOCOMOTION[f"{_leg}_{_seg}"] = _pos

INITIAL_ANGLES_LOCOMOTION = {
    "RF": _seeds(0.45, -0.07, -0.32, -1.25),
    "LF": _seeds(-0.45, -0.07, 0.32, 1.25),
    "RM": _seeds(0.45, 0.37, -0.32, -1.25),
    "LM": _seeds(-0.45, 0.37, 0.32, 1.25),
    "RH": _seeds(0.45, 0.07, -0.32, -1.25),
    "LH": _seeds(-0.45, 0.07, 0.32, 1.25),
}


def _loco_bounds(leg, yaw, pitch, roll, ctr_pitch):
    full = (-_PI, _PI)
    return {
        f"{leg}_ThC_yaw": yaw,
        f"{leg}_ThC_pitch": pitch,
        f"{leg}_ThC_roll": roll,
        f"{leg}_CTr_pitch": ctr_pitch,
        f"{leg}_FTi_pitch": full,
        f"{leg}_CTr_roll": full,
        f"{leg}_TiTa_pitch": (-_PI, np.deg2rad(0)),
    }


_D50 = (np.deg2rad(-50), np.deg2rad(50))
_D90 = (np.deg2rad(-90), np.deg2rad(90))
_FULL = (-_PI, _PI)
BOUNDS_LOCOMOTION = {}
BOUNDS_LOCOMOTION.update(_loco_bounds("RF", _FULL, _D90, _FULL, _FULL))
BOUNDS_LOCOMOTION.update(_loco_bounds("RM", _D50, _FULL, (-_PI, 0), _FULL))
BOUNDS_LOCOMOTION.update(_loco_bounds("RH", _D50, _D50, (-_PI, 0), (np.deg2rad(-180), np.deg2rad(0))))
BOUNDS_LOCOMOTION.update(_loco_bounds("LF", _FULL, _D90, _FULL, _FULL))
BOUNDS_LOCOMOTION.update(_loco_bounds("LM", _D50, _FULL, (0, _PI), _FULL))
BOUNDS_LOCOMOTION.update(_loco_bounds("LH", _D50, _D50, (0, _PI), (np.deg2rad(-180), np.deg2rad(0))))

# Sizes of the template's body segments (reference ``seqikpy/data.py:44-77``; the parallel example imports it,
# ``examples/example_leg_inv_kinematics_parallel.py:18``): the front legs' rounded lengths, the middle and hind legs' from
# the locomotion template above, the antenna's from NMF_TEMPLATE -- the same numbers the reference lists.
NMF_SIZE = {}
for _seg, _front in (("Coxa", 0.40), ("Femur", 0.69), ("Tibia", 0.54), ("Tarsus", 0.63)):
    _nxt = {"Coxa": "Femur", "Femur": "Tibia", "Tibia": "Tarsus", "Tarsus": "Claw"}[_seg]
    for _leg in ("RF", "RM", "RH", "LF", "LM", "LH"):
        NMF_SIZE[f"{_leg}_{_seg}"] = _front if _leg[1] == "F" else float(np.linalg.norm(
            TEMPLATE_NMF_LOCOMOTION[f"{_leg}_{_seg}"] - TEMPLATE_NMF_LOCOMOTION[f"{_leg}_{_nxt}"]))
for _leg, _total in (("RF", 2.26), ("RM", 2.328), ("RH", 2.515), ("LF", 2.26), ("LM", 2.328), ("LH", 2.515)):
    NMF_SIZE[_leg] = _total
NMF_SIZE["Antenna"] = float(np.linalg.norm(NMF_TEMPLATE["R_Antenna_base"] - NMF_TEMPLATE["R_Antenna_edge"]))
NMF_SIZE["Antenna_mid_thorax"] = float(np.linalg.norm(NMF_TEMPLATE["R_Antenna_base"] - NMF_TEMPLATE["Thorax_mid"]))
