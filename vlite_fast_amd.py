"""Importable alias of the package directory `vlite-fast_amd/` (hyphen in the name)."""
import importlib
import sys

_pkg = importlib.import_module("vlite-fast_amd")
sys.modules[__name__] = _pkg
