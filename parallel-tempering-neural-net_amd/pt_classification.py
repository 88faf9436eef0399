"""Drop-in for `ParallelTempering` of multicore-pt-classification/pt_classification.py (CLS:497-897).

Same constructor (CLS:499: no langevin_prob, the replica fixes l_prob = 0.5, CLS:192), same methods, return tuple
and files.  Multinomial likelihood on softmax-of-sigmoid outputs; hand-off when (i+1) % swap_interval == 0
(CLS:438); posted scalar = tempered likelihood (CLS:439); per-chain rmse files use '%1.2f' (CLS:473-475).
"""
from . import _lib
from .parallel_tempering import ParallelTemperingBase


class ParallelTempering(ParallelTemperingBase):
    task = _lib.TASK_CLS
    rmse_fmt = '%1.2f'

    def __init__(self, use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp,
                 NumSample, swap_interval, path, **kw):
        super().__init__(use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp,
                         NumSample, swap_interval, 0.5, path, **kw)

    def _likelihood_rows(self, burnin):
        return burnin                 # likelihood_rep[i, :] = dat[burnin:] (CLS:806)
