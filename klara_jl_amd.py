"""Loader: makes the package directory `klara.jl_amd/` importable as `klara_jl_amd`.

`import klara.jl_amd` cannot work in Python (the dot is the submodule separator), so this module loads
`klara.jl_amd/__init__.py` under the name `klara_jl_amd` and re-exports it.  Put the repository root on
sys.path (pytest's rootdir / bench.py / __graft_entry__.py already do) and `import klara_jl_amd`.
"""
import importlib.util as _ilu
import sys as _sys
from pathlib import Path as _Path

_pkg_dir = _Path(__file__).resolve().parent / "klara.jl_amd"
_spec = _ilu.spec_from_file_location("klara_jl_amd", _pkg_dir / "__init__.py",
                                     submodule_search_locations=[str(_pkg_dir)])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["klara_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
