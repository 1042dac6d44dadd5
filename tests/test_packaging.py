"""Drop-in packaging: the opt-in `seqikpy` alias (seqikpy_amd/as_seqikpy.py) and the installable distribution
(pyproject.toml / setup.py).  Reference: /root/reference/setup.py, the import lines of /root/reference/examples/*.py."""
import ast
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

PKG_PARENT = os.path.join(ROOT, "sequential-inverse-kinematics_amd")

# What the reference's five example scripts import from the package (names only, collected from their import lines:
# examples/example_alignment.py, example_entire_pipeline.py, example_head_inv_kinematics.py,
# example_leg_inv_kinematics.py, example_leg_inv_kinematics_parallel.py); `test_reference_example_imports_resolve`
# re-derives the list from the scripts where the reference checkout is present.
EXAMPLE_IMPORTS = {
    "seqikpy.alignment": ["AlignPose", "convert_from_anipose_to_dict"],
    "seqikpy.data": ["BOUNDS", "INITIAL_ANGLES", "NMF_TEMPLATE", "PTS2ALIGN", "NMF_SIZE"],
    "seqikpy.utils": ["load_file", "save_file", "calculate_body_size"],
    "seqikpy.kinematic_chain": ["KinematicChainSeq", "KinematicChainGeneric"],
    "seqikpy.leg_inverse_kinematics": ["LegInvKinSeq", "LegInvKinGeneric"],
    "seqikpy.head_inverse_kinematics": ["HeadInverseKinematics"],
}


def run_py(code, cwd="/tmp"):
    env = dict(os.environ, PYTHONPATH=PKG_PARENT)
    return subprocess.run([sys.executable, "-c", code], cwd=cwd, env=env, capture_output=True, text=True, timeout=300)


def resolve_script(imports):
    lines = ["import seqikpy_amd.as_seqikpy as a", "assert a.installed", "import importlib"]
    for mod, names in imports.items():
        lines.append(f"m = importlib.import_module({mod!r})")
        lines.append(f"assert m.__name__ == {mod.replace('seqikpy', 'seqikpy_amd', 1)!r}, m.__name__")
        for n in names:
            lines.append(f"assert hasattr(m, {n!r}), {mod + '.' + n!r}")
        lines.append(f"exec('from {mod} import {', '.join(names)}')")
    lines.append("print('resolved')")
    return "\n".join(lines)


def test_alias_makes_reference_import_lines_run():
    r = run_py(resolve_script(EXAMPLE_IMPORTS))
    assert r.returncode == 0 and "resolved" in r.stdout, r.stderr
    # the part of the reference that is NOT the path stays an ImportError, it is not silently something else
    r = run_py("import seqikpy_amd.as_seqikpy\ntry:\n    import seqikpy.visualization\nexcept ImportError:\n    print('absent')")
    assert "absent" in r.stdout, r.stderr


def test_alias_leaves_a_real_seqikpy_alone_and_can_be_removed():
    code = ("import sys, types\n"
            "real = types.ModuleType('seqikpy'); sys.modules['seqikpy'] = real\n"
            "import seqikpy_amd.as_seqikpy as a\n"
            "assert a.installed is False and sys.modules['seqikpy'] is real\n"
            "assert a.install(force=True) and sys.modules['seqikpy'].__name__ == 'seqikpy_amd'\n"
            "a.uninstall(); assert not [k for k in sys.modules if k == 'seqikpy' or k.startswith('seqikpy.')]\n"
            "print('ok')")
    r = run_py(code)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="reference checkout not present")
def test_reference_example_imports_resolve():
    """The import lines of the reference's example scripts, read as data (ast of the import statements only), resolve
    against this package under the alias -- everything but the out-of-scope seqikpy.visualization."""
    found = {}
    scripts = sorted(glob.glob("/root/reference/examples/*.py"))
    assert len(scripts) >= 5
    for path in scripts:
        with open(path) as fh:
            tree = ast.parse(fh.read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "seqikpy":
                found.setdefault(node.module, set()).update(a.name for a in node.names)
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name.split(".")[0] == "seqikpy":
                        found.setdefault(a.name, set())
    found.pop("seqikpy.visualization", None)
    assert found, "no seqikpy imports found in the reference examples"
    r = run_py(resolve_script({m: sorted(n) for m, n in found.items()}))
    assert r.returncode == 0 and "resolved" in r.stdout, r.stderr
    for mod, names in found.items():        # the static list above is what the examples really import
        assert set(names) <= set(EXAMPLE_IMPORTS.get(mod, [])), (mod, names)


def test_distribution_installs_and_loads_its_own_library(tmp_path):
    """`pip install .` into a scratch directory: the installed package carries libseqik_hip.so + seqik.h and loads THAT
    library (no GPU call).  Needs the library to be built in the tree (build()) -- the install step reuses it."""
    from seqikpy_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    target = tmp_path / "site"
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index", "--quiet",
                        "--target", str(target), ROOT], capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and "No module named pip" in r.stderr:
        pytest.skip("pip not available")
    assert r.returncode == 0, r.stderr[-2000:]
    assert (target / "seqikpy_amd" / "_native" / "libseqik_hip.so").exists()
    assert (target / "seqikpy_amd" / "_native" / "seqik.h").exists()
    code = ("import seqikpy_amd, seqikpy_amd._lib as L\n"
            "assert '_native' in L.LIB_PATH, L.LIB_PATH\n"
            "assert L.load().seqik_abi_version() == L.ABI_VERSION\n"
            "print(seqikpy_amd.__file__)")
    env = dict(os.environ, PYTHONPATH=str(target))
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and str(target) in r.stdout, r.stderr
    for junk in ("build", os.path.join("sequential-inverse-kinematics_amd", "seqikpy_amd.egg-info"),
                 os.path.join("sequential-inverse-kinematics_amd", "seqikpy_amd.egg-info".replace("_", "-"))):
        import shutil
        shutil.rmtree(os.path.join(ROOT, junk), ignore_errors=True)
