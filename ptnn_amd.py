"""Import shim: makes the package in `parallel-tempering-neural-net_amd/` (a directory name Python cannot import
directly) available as `ptnn_amd`.  `import ptnn_amd` then `from ptnn_amd.pt_timeseries_regression import
ParallelTempering`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parallel-tempering-neural-net_amd")
_spec = importlib.util.spec_from_file_location("ptnn_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ptnn_amd"] = _mod
_spec.loader.exec_module(_mod)
