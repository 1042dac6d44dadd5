"""GPU tier: the example scripts that mirror the reference's callers (examples/example_alignment.py,
example_head_kinematics.py, example_leg_inv_kinematics_parallel.py; the entire pipeline has tests/test_pipeline.py),
run on the committed fixtures and checked against the reference outputs those hold."""
import importlib
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import DOFS, ROOT, load_golden

pytestmark = pytest.mark.gpu


def example(name):
    # imported by name from examples/ (not from a file location): the process-pool example pickles its worker function,
    # and the spawned workers -- which inherit sys.path -- must be able to import the module it lives in
    if os.path.join(ROOT, "examples") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
    return importlib.import_module(name)


def test_alignment_example(tmp_path, hiplib):
    z = load_golden("anipose_raw_cut")
    mod = example("alignment")
    res = mod.main(["--gpu-statistics"])
    assert list(res) == ["from memory"]
    for k, v in res["from memory"].items():
        assert np.array_equal(v, z[f"aligned_{k}"]), k
    with open(tmp_path / "converted_dict.pkl", "wb") as f:
        pickle.dump({str(k): z[f"raw_{k}"] for k in z["segments"]}, f)
    res = mod.main(["-p", str(tmp_path), "--export"])
    assert list(res) == ["from the converted dictionary", "from memory"]
    assert all(np.array_equal(res["from the converted dictionary"][k], z[f"aligned_{k}"]) for k in res["from memory"])
    assert os.path.exists(tmp_path / "pose3d_aligned.pkl")


def test_head_kinematics_example(tmp_path, hiplib):
    z = load_golden("anipose_head")
    with open(tmp_path / "pose3d_aligned.pkl", "wb") as f:
        pickle.dump({k: z[k] for k in ("R_head", "L_head", "Neck")}, f)
    mod = example("head_kinematics")
    for argv in ([], ["-p", str(tmp_path), "--export"]):
        ang = mod.main(argv)
        assert [str(n) for n in z["names"]] == list(ang)
        assert np.abs(np.stack(list(ang.values()), 1) - z["shipped"]).max() < 1e-6
    assert os.path.exists(tmp_path / "head_joint_angles.pkl")


def test_six_leg_example_one_launch_and_process_pool(hiplib):
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    mod = example("leg_inv_kinematics_parallel")
    ang, fk = mod.main([])
    assert sorted(fk) == sorted(f"{l}_leg" for l in legs) and len(ang) == 7 * len(legs)
    for leg in legs:
        got = np.stack([ang[f"Angle_{leg}_{d}"] for d in DOFS], 1)
        assert np.abs(got - z[f"{leg}_angles"]).max() < 1e-4, leg          # the reference-source run
        assert np.abs(fk[f"{leg}_leg"] - z[f"{leg}_fk"]).max() < 1e-4, leg
    # the reference's shape: one task per leg in a process pool (two workers here: the GPU box allows few processes)
    ang_p, fk_p = mod.main(["--pool", "--processes", "2"])
    assert sorted(ang_p) == sorted(ang) and all(np.array_equal(ang_p[k], ang[k]) for k in ang)
    assert all(np.array_equal(fk_p[k], fk[k]) for k in fk)
    serial, _ = mod.main(["--serial"])          # the reference's own order; the default above is verified frame chunks
    assert max(np.abs(serial[k] - ang[k]).max() for k in ang) < 2e-5
    same, _ = mod.main(["--frame-chunks"])      # (older command lines: the flag is the default now)
    assert all(np.array_equal(same[k], ang[k]) for k in ang)
