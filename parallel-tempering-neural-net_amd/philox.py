"""Host-side Philox4x32-10 tape (numpy), same specification as the device code in csrc/ptnn_device.hpp.

Used by the host only for the initial weights w0 (stream 3), which the reference draws in the parent process
with np.random.randn (pt_timeseries_regression.py:649).  Streams: 0 step scalars, 1 proposal noise, 2 swap
uniforms, 3 initial weights; counter = (index, step|round, global replica, stream); key = (seed lo, seed hi);
uniform u = ((x >> 9) + 0.5) * 2^-23; normals by Box-Muller on (x0,x1) and (x2,x3).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
STREAM_STEP, STREAM_WNOISE, STREAM_SWAP, STREAM_INIT = 0, 1, 2, 3


def philox4x32(c0, c1, c2, c3, seed):
    c = [np.array(x, dtype=np.uint64, copy=True) for x in np.broadcast_arrays(
        *(np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3)))]
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & _MASK,
             (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & _MASK]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def uniform23(x):
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(9)).astype(np.float64) + 0.5) * (1.0 / 8388608.0)


def normals(n, c1, c2, stream, seed):
    """n standard normals of counter block (., c1, c2, stream)."""
    nq = (n + 3) // 4
    x = philox4x32(np.arange(nq), c1, c2, stream, seed)
    r0 = np.sqrt(-2.0 * np.log(uniform23(x[0])))
    t0 = 2.0 * np.pi * uniform23(x[1])
    r1 = np.sqrt(-2.0 * np.log(uniform23(x[2])))
    t1 = 2.0 * np.pi * uniform23(x[3])
    return np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], axis=-1).reshape(-1)[:n]


def initial_weights(seed, replica, n):
    """w0 of global replica `replica` (stands in for np.random.randn(num_param), REG:649)."""
    return normals(n, 0, replica, STREAM_INIT, seed)
