"""Drop-in for `ParallelTempering` of multicore-pt-regression/pt_timeseries_regression.py (REG:487-875).

Same constructor (REG:489), same methods, same return tuple and files; the sampling itself runs on one MI355X
through libptnn.so.  Gaussian likelihood with sampled eta = log tau^2; hand-off after step i when
i % swap_interval == 0 and i != 0 (REG:427); posted scalar = likelihood * temperature (REG:430).
"""
from . import _lib
from .parallel_tempering import ParallelTemperingBase


class ParallelTempering(ParallelTemperingBase):
    task = _lib.TASK_REG
    rmse_fmt = '%1.8f'

    def __init__(self, use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp,
                 NumSample, swap_interval, langevin_prob, path, **kw):
        super().__init__(use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp,
                         NumSample, swap_interval, langevin_prob, path, **kw)

    def _likelihood_rows(self, burnin):
        return 1                      # likelihood_rep[i, :] = dat[1:] (REG:801): all rows but the first
