"""Test helper: the reference-held pin of ALL SIX legs (round-5 review, item 2).

`tests/golden/df3d_notebook_cell16.png` is the image the reference's notebook keeps as the output of its cell 16
(examples/seqikpy_locomotion.ipynb: `leg_joint_angles` of `LegInvKinSeq.run_ik_and_fk` over REAL ikpy 3.3.4 on frames 300:400
of the df3d recording -- the inputs of tests/golden/df3d_100.npz --, 7 joints x 6 legs x 100 frames in degrees; extracted by
oracle/gen_golden.py, image data only).  It is the only output of real IKPy for RM / RH / LM / LH and for BOUNDS_LOCOMOTION in
reach.  `render()` draws a set of joint angles the way that cell does (3 x 2 axes in the order RF LF RM LM RH LH, figsize (9, 7),
dpi 200, lw 2, matplotlib's default colour cycle = one colour per joint, tight layout, saved with a tight bounding box as the
notebook's inline backend does) and `violations()` compares two such images per colour mask: a curve pixel of one image with no
pixel of the same colour within `radius` pixels in the other image is a violation.

Resolution: the axes are ~345 pixels high and span 270-350 degrees, so one pixel is 0.8-1.0 degree = 0.014-0.017 rad; a whole
curve moved by 2 degrees produces hundreds of violations at radius 1, ten frames moved by 3 degrees about 150 (measured:
tests/test_notebook_pin.py).  Coarse next to the 1e-4 rad of the shipped RF / LF files -- but it catches what they and the mirror
test cannot: a mistake on the middle and hind legs that is itself mirror-symmetric (a wrong limit table, a swapped joint, a sign)."""
import io
import os
import warnings

import numpy as np

DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
PANELS = ["RF", "LF", "RM", "LM", "RH", "LH"]
COLOUR_CYCLE = ["#1f77b4", "#ff7f0e", "#2ca02c", "#d62728", "#9467bd", "#8c564b", "#e377c2"]   # matplotlib's C0..C6: one per joint
PNG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "df3d_notebook_cell16.png")


def stored_image():
    from PIL import Image
    return np.asarray(Image.open(PNG).convert("RGB")).astype(np.int16)


def render(angles):
    """`angles`: {leg: (100, 7) radians in DOFS order} -> RGB image (int16) drawn as the notebook's cell 16 draws it."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from PIL import Image
    with warnings.catch_warnings(), matplotlib.rc_context(matplotlib.rcParamsDefault):
        warnings.simplefilter("ignore")          # (set_xticklabels without fixed ticks, as in the cell)
        fig, axs = plt.subplots(3, 2, figsize=(9, 7), dpi=200)
        axs = axs.flatten()
        for j, name in enumerate(DOFS):
            for i, leg in enumerate(PANELS):
                axs[i].plot(np.rad2deg(angles[leg][:, j]), label=name, lw=2)
                axs[i].set_ylabel(leg)
        for ax in axs:
            ax.set_xticklabels(np.array(ax.get_xticks() * 1e-2, dtype="f"))
        axs[-1].set_xlabel("Time (sec)")
        axs[-2].set_xlabel("Time (sec)")
        axs[1].legend(bbox_to_anchor=(1.1, 1), frameon=False)
        plt.suptitle("Leg joint angles (deg)")
        plt.tight_layout()
        buf = io.BytesIO()
        fig.savefig(buf, format="png", bbox_inches="tight")
        plt.close(fig)
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(np.int16)


def colour_masks(img, threshold=40):
    """One boolean mask per joint colour: pixels within `threshold` (sum of |RGB differences|) of the pure line colour -- the
    core of a line, not its anti-aliased rim."""
    return [np.abs(img - np.array([int(c[i:i + 2], 16) for i in (1, 3, 5)])).sum(-1) < threshold for c in COLOUR_CYCLE]


def violations(img_a, img_b, radius=1):
    """Per joint colour: (pixels of a with no pixel of that colour within `radius` in b, the same the other way round)."""
    from scipy.ndimage import binary_dilation
    box = np.ones((2 * radius + 1, 2 * radius + 1), bool)
    return [(int((a & ~binary_dilation(b, structure=box)).sum()), int((b & ~binary_dilation(a, structure=box)).sum()))
            for a, b in zip(colour_masks(img_a), colour_masks(img_b))]
