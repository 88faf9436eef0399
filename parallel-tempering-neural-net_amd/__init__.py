"""ptnn_amd -- MI355X-native parallel-tempering sampler for Bayesian feed-forward networks.

Drop-in for the one hot path of sydney-machine-learning/parallel-tempering-neural-net:
`ParallelTempering(...).run_chains()` of pt_timeseries_regression.py / pt_classification.py,
computed by hand-written gfx950 kernels behind the C ABI of include/ptnn.h (libptnn.so, bound
with ctypes).  There is no CPU fallback: without the built library or without a gfx950 device
every compute entry point raises.

The directory name contains hyphens, so it is imported through the `ptnn_amd` loader at the
repository root:   import ptnn_amd;  from ptnn_amd.pt_timeseries_regression import ParallelTempering
"""
from . import ladder, philox  # noqa: F401
from ._lib import PtnnError, library_path, load_library  # noqa: F401

__all__ = ["ladder", "philox", "PtnnError", "library_path", "load_library"]
