"""Temperature ladder of the reference (pt_timeseries_regression.py:529-636, pt_classification.py:537-645).

`assign_temperatures` always calls `default_beta_ladder(2, ntemps=num_chains, Tmax=maxtemp)`; with both arguments
given the routine reduces to betas = logspace(0, -log10(Tmax), ntemps) and T_i = 1/betas[i].  The argument checks
and the integer-Tmax requirement (`range(maxtemp)`, REG:576) are kept because callers can observe them.
"""
import numpy as np


def default_beta_ladder(ndim, ntemps=None, Tmax=None):
    if type(ndim) != int or ndim < 1:
        raise ValueError('Invalid number of dimensions specified.')
    if ntemps is None and Tmax is None:
        raise ValueError('Must specify one of ``ntemps`` and ``Tmax``.')
    if Tmax is not None and Tmax <= 1:
        raise ValueError('``Tmax`` must be greater than 1.')
    if ntemps is not None and (type(ntemps) != int or ntemps < 1):
        raise ValueError('Invalid number of temperatures specified.')
    range(Tmax)                       # the reference iterates range(maxtemp): a float Tmax is a TypeError there too
    if ntemps < 2:
        raise ZeroDivisionError('float division by zero')    # numchain**(-1/(numchain-1)), REG:577
    return np.logspace(0, -np.log10(Tmax), ntemps)


def temperatures(num_chains, maxtemp):
    betas = default_beta_ladder(2, ntemps=num_chains, Tmax=maxtemp)
    return [float(1.0 / b) for b in betas]
