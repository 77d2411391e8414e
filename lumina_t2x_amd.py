"""Import shim: the package directory is ``lumina-t2x_amd/`` (hyphen, as the project layout prescribes), which
Python cannot import by name.  Importing this module registers that directory as the package
``lumina_t2x_amd`` (proper spec + submodule search path) and replaces itself with it in ``sys.modules``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lumina-t2x_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
