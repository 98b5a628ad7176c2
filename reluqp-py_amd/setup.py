"""Packaging of the MI355X ReLU-QP drop-in: same distribution / import name as the reference (`reluqp`, reference
ReLU-QP-py/setup.py:3-7), so `import reluqp.reluqpth as reluqpth` keeps working after `pip install -e .` (or with this
directory on PYTHONPATH).  The HIP library is built in-tree by `make -C csrc` (hipcc, gfx950) into reluqp/lib/ and shipped
as package data; there is no CPU fallback (reluqp/_cabi.py raises RqpUnavailable without it)."""
import os
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class build_py_with_hip(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")])
        super().run()


setup(
    name="reluqp",
    version="1.0+mi355x",
    description="ReLU-QP (OSQP-style ADMM) on AMD MI355X: hand-written HIP kernels behind the reference's Python API",
    packages=find_packages(include=["reluqp", "reluqp.*"]),
    package_data={"reluqp": ["lib/librqp_hip.so"]},
    python_requires=">=3.8",
    install_requires=[],            # torch (ROCm build) and numpy come with the platform image
    cmdclass={"build_py": build_py_with_hip},
)
