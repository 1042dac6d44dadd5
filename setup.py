"""Distribution `seqikpy-amd`: the host layer (package `seqikpy_amd`) + the HIP library, built in one step.

The build step compiles the HIP library with hipcc (gfx950) through the package's own `_lib.build()` and places it, with
the C header, inside the built package (`seqikpy_amd/_native/`), where `_lib` looks when it is not running from the
source tree.  Reference counterpart: /root/reference/setup.py (pure Python; this one has a native step)."""
import os
import re
import shutil
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "sequential-inverse-kinematics_amd")


def version():
    with open(os.path.join(SRC, "seqikpy_amd", "__init__.py")) as fh:
        return re.search(r'^__version__ = "([^"]+)"', fh.read(), re.M).group(1)


class build_py_with_hip(build_py):
    def run(self):
        super().run()
        if getattr(self, "editable_mode", False):
            return                      # editable install: the tree's csrc/ is used as it is
        sys.path.insert(0, SRC)
        try:
            from seqikpy_amd import _lib
            lib = _lib.build()
        finally:
            sys.path.remove(SRC)
        dst = os.path.join(self.build_lib, "seqikpy_amd", "_native")
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(lib, os.path.join(dst, os.path.basename(lib)))
        shutil.copy2(os.path.join(ROOT, "include", "seqik.h"), os.path.join(dst, "seqik.h"))


setup(
    name="seqikpy-amd",
    version=version(),
    description="MI355X-native drop-in for the leg inverse-kinematics path of seqikpy "
                "(hand-written HIP for gfx950 behind a C ABI)",
    long_description=open(os.path.join(ROOT, "README.md")).read(),
    long_description_content_type="text/markdown",
    python_requires=">=3.8",
    install_requires=["numpy"],
    extras_require={"multi-gpu": ["torch"], "dev": ["pytest", "scipy"]},
    package_dir={"": "sequential-inverse-kinematics_amd"},
    packages=["seqikpy_amd"],
    zip_safe=False,
    cmdclass={"build_py": build_py_with_hip},
)
