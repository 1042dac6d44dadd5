"""Reference-held pin of all six legs: the joint-angle plot the reference's notebook stores (tests/notebook_pin.py)."""
import numpy as np
import pytest

pytest.importorskip("matplotlib")
pytest.importorskip("PIL")

import notebook_pin as pin  # noqa: E402
from conftest import load_golden  # noqa: E402

BUDGET = 40          # stray pixels over all seven colours (blends where two curves cross): 16 of 160 000 on the build container


def total(v):
    return sum(a + b for a, b in v)


def oracle_angles(oracle, z):
    return {str(l): oracle.seq_leg(z[f"{l}_pose"], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"])["angles"] for l in z["legs"]}


def test_stored_image_is_the_notebook_plot():
    img = pin.stored_image()
    assert img.shape == (1382, 1772, 3)
    sizes = [int(m.sum()) for m in pin.colour_masks(img)]
    assert min(sizes) > 10000 and max(sizes) < 40000      # seven curves x six axes (+ the legend's samples)


def test_oracle_reproduces_the_notebook_plot_of_all_six_legs(oracle):
    """The C restatement on the df3d_100 inputs, drawn as cell 16 draws it, against the image real IKPy left in the notebook:
    same canvas, and every curve pixel of either image has one of the same colour within ONE pixel (~1 degree = 0.017 rad) in
    the other -- for RM / RH / LM / LH too, with BOUNDS_LOCOMOTION and the locomotion template."""
    z = load_golden("df3d_100")
    ang = oracle_angles(oracle, z)
    ref, mine = pin.stored_image(), pin.render(ang)
    assert mine.shape == ref.shape
    v = pin.violations(ref, mine, radius=1)
    assert total(v) <= BUDGET, v
    assert total(pin.violations(ref, mine, radius=2)) <= BUDGET
    # the fixture's own reference-source run (over the IKPy stand-in) draws the same picture
    fix = {str(l): z[f"{l}_angles"] for l in z["legs"]}
    assert total(pin.violations(ref, pin.render(fix), radius=1)) <= BUDGET


def test_the_pin_has_teeth(oracle):
    """What the comparison resolves: a whole joint series of a middle or hind leg off by 2 degrees (0.035 rad), ten frames off by 3
    degrees, and mirror-symmetric mistakes (the sign of a roll on BOTH middle legs; two joints swapped on both hind legs) --
    none of which tests/test_mirror.py or the shipped RF / LF files can see -- all fail it by a wide margin."""
    z = load_golden("df3d_100")
    ang = oracle_angles(oracle, z)
    ref = pin.stored_image()

    def moved(fn):
        a2 = {k: v.copy() for k, v in ang.items()}
        fn(a2)
        return total(pin.violations(ref, pin.render(a2), radius=1))

    def shift(leg, joint, deg, frames=slice(None)):
        def fn(a):
            a[leg][frames, joint] += np.deg2rad(deg)
        return fn

    assert moved(shift("RM", 6, 2.0)) > 400 and moved(shift("RH", 0, 2.0)) > 200 and moved(shift("LH", 2, 2.0)) > 100
    assert moved(shift("RM", 4, 3.0, slice(40, 50))) > 100

    def flip(a):
        a["RM"][:, 2] *= -1
        a["LM"][:, 2] *= -1
    assert moved(flip) > 5000

    def swap(a):
        for leg in ("RH", "LH"):
            a[leg][:, [0, 4]] = a[leg][:, [4, 0]]
    assert moved(swap) > 1000


@pytest.mark.gpu
def test_hip_reproduces_the_notebook_plot_of_all_six_legs(hiplib):
    """The same comparison on the HIP path's own output (HIP == oracle bit for bit, so this is the CPU test once more -- run on
    the GPU box so that the pin is seen to hold for the product, not only for its checker)."""
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    out = hiplib.solve_seq(np.stack([z[f"{l}_pose"] for l in legs])[None], params, want_fk=False)
    ang = {l: out["angles"][0, i] for i, l in enumerate(legs)}
    ref, mine = pin.stored_image(), pin.render(ang)
    assert mine.shape == ref.shape and total(pin.violations(ref, mine, radius=1)) <= BUDGET
