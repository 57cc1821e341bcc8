"""Import shim: the package directory is named `multigridbarrier.jl_amd` (with a dot, as
the project layout requires), which Python's import statement cannot spell.  Load it
under the importable name `mgb_amd`."""
import importlib.util as _u
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg = _os.path.join(_here, "multigridbarrier.jl_amd")
_spec = _u.spec_from_file_location("mgb_amd", _os.path.join(_pkg, "__init__.py"),
                                   submodule_search_locations=[_pkg])
_mod = _u.module_from_spec(_spec)
_sys.modules["mgb_amd"] = _mod
_spec.loader.exec_module(_mod)
