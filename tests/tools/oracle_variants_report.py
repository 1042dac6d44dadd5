#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the C restatement's switchable variants over ALL committed fixtures.

The restatement (oracle/seqik_oracle.c) departs from "scipy verbatim" in three equivalent-form choices that exist for
the kernel's sake (closed-form 2 x 2 trust-region step, short-cut root search, zero Jacobian columns removed in stages
2-3).  Every one stays selectable through an oracle_set_* hook.  This tool runs each variant over every fixture, full
length, and records its distance to the reference outputs the fixture holds (shipped pickles / reference-source run over
real scipy) and to the default variant.

    python tests/tools/oracle_variants_report.py > profiles/r02_oracle_variants.json        (CPU only, ~1 minute)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from oracle import c_oracle  # noqa: E402

LF_WINDOW = (284, 302)  # tests/conftest.py::LF_DEGENERATE
VARIANTS = {
    "default (closed-form 2x2 step, short-cut root search, zero columns removed in stages 2-3)": {},
    "root search verbatim (ten iterations)": dict(tr2_shortcut=False),
    "one-sided Jacobi SVD instead of the closed-form 2x2 step": dict(closed_form_2x2=False),
    "scipy's loop verbatim: Jacobi SVD + ten-iteration root search": dict(tr2_shortcut=False, closed_form_2x2=False),
    "zero Jacobian columns kept in stages 2-3 (exact-zero singular values, never full rank)": dict(null_mode=1),
    "analytic (geometric) Jacobian instead of scipy's 2-point differences": dict(analytic_jacobian=True),
}


def run_variant(kw, cases):
    c_oracle.reset_variants()
    c_oracle.lib().oracle_set_null_mode(-1)
    kw = dict(kw)
    if "null_mode" in kw:
        c_oracle.lib().oracle_set_null_mode(kw.pop("null_mode"))
    c_oracle.set_variant(**kw)
    try:
        return [c_oracle.seq_leg(pose, seg, b, seeds) for (_, _, pose, seg, b, seeds, _, _) in cases]
    finally:
        c_oracle.reset_variants()
        c_oracle.lib().oracle_set_null_mode(-1)


def main():
    cases = []
    for name in ("anipose_shipped", "anipose_scipy_cut", "df3d_100", "df3d_1000"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        for leg in [str(l) for l in z["legs"]]:
            ok = np.ones(z[f"{leg}_pose"].shape[0], bool)
            if name.startswith("anipose") and leg == "LF":
                frames = z["frames"] if "frames" in z.files else np.arange(ok.size)
                ok[(frames >= LF_WINDOW[0]) & (frames < LF_WINDOW[1])] = False
            cases.append((name, leg, z[f"{leg}_pose"], z[f"{leg}_seg"], z[f"{leg}_bounds"], z[f"{leg}_seeds"], z[f"{leg}_angles"], ok))
    base = None
    rep = {"lf_window_excluded": list(LF_WINDOW), "fixtures": sorted({c[0] for c in cases}),
           "leg_frames": int(sum(c[2].shape[0] for c in cases)), "variants": {}}
    for vname, kw in VARIANTS.items():
        res = run_variant(kw, cases)
        if base is None:
            base = res
        worst, over, per_fixture = 0.0, 0, {}
        same_bits, max_vs_default, nfev_equal = True, 0.0, []
        for (name, leg, _, _, _, _, ref, ok), r, b in zip(cases, res, base):
            err = np.abs(r["angles"] - ref)
            e = float(err[ok].max())
            per_fixture[f"{name}/{leg}"] = e
            worst = max(worst, e)
            over += int((err[ok].max(1) > 1e-4).sum())
            same_bits &= bool(np.array_equal(r["angles"], b["angles"]) and np.array_equal(r["fk"], b["fk"]) and
                              np.array_equal(r["nfev"], b["nfev"]) and np.array_equal(r["status"], b["status"]))
            max_vs_default = max(max_vs_default, float(np.abs(r["angles"] - b["angles"])[ok].max()))
            nfev_equal.append((r["nfev"] == b["nfev"]).mean())
        rep["variants"][vname] = {"max_abs_vs_fixture_outside_lf_window": worst, "leg_frames_over_1e-4_outside_lf_window": over,
                                  "max_abs_vs_fixture_per_leg": per_fixture,
                                  "bit_identical_to_default_incl_status_nfev": same_bits,
                                  "max_abs_vs_default_outside_lf_window": max_vs_default,
                                  "share_of_solves_with_default_nfev": float(np.mean(nfev_equal))}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
