"""setuptools glue: the native libraries (C++ consensus core, gfx950 HIP kernels) are built
in-tree by build_native.py before the Python packages are collected, so a wheel or an
editable install carries ``llm_consensus_amd/_native/*.so``. Metadata lives in pyproject.toml."""
import sys
from pathlib import Path

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = Path(__file__).resolve().parent


class BuildWithNative(build_py):
    def run(self):
        sys.path.insert(0, str(ROOT))
        import build_native

        build_native.build_core()
        try:
            build_native.build_kernels()
        except (RuntimeError, FileNotFoundError) as e:       # no hipcc: CPU-only install (plumbing, tests)
            print(f"warning: gfx950 kernels not built ({e}); GPU execution will be unavailable", file=sys.stderr)
        super().run()


def legacy_metadata() -> dict:
    """setuptools before 61 ignores pyproject's [project] table: hand the same data to setup()."""
    import setuptools

    if int(setuptools.__version__.split(".")[0]) >= 61:
        return {}
    try:
        import tomllib as toml
    except ImportError:
        import tomli as toml
    from setuptools import find_packages

    p = toml.loads((ROOT / "pyproject.toml").read_text())["project"]
    return {
        "name": p["name"], "version": p["version"], "description": p["description"],
        "python_requires": p["requires-python"], "install_requires": p["dependencies"],
        "packages": find_packages(include=["llm_consensus_amd*"]),
        "package_data": {"llm_consensus_amd._native": ["*.so"]},
        "entry_points": {"console_scripts": [f"{k} = {v}" for k, v in p["scripts"].items()]},
    }


setup(cmdclass={"build_py": BuildWithNative}, **legacy_metadata())
