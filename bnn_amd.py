"""Import shim: ``import bnn_amd`` loads the package in ``bayesian-neural-nets_amd/``.

The package directory keeps the project's hyphenated name, which is not a Python identifier, so
this module replaces itself in ``sys.modules`` with the real package (submodules resolve
normally: ``bnn_amd.mnf``, ``bnn_amd.ops`` ...).
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "bayesian-neural-nets_amd")
_spec = importlib.util.spec_from_file_location(
    "bnn_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bnn_amd"] = _mod
_spec.loader.exec_module(_mod)
