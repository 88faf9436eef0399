"""Sharded ladder: one process per GPU, each owning a contiguous block of the temperature ladder.

Replaces the reference's star topology (every replica ships its whole parameter vector to the parent through a
multiprocessing.Queue each swap round, REG:427-437 <-> 719-752).  Replicas are independent for a swap interval, so
the data path needs exactly one exchange step per interval.  Two modes:

  "gather" (default): ONE collective per round.  Every rank packs, per local replica, an exchange row {state, cached
     gradient, posted L}; the rows are all-gathered in place (R x (8 P + 16) bytes: 18 KB per rank for the Sunspot net --
     nothing next to an interval of MH steps, and a single large collective is what xGMI rings like); every rank then runs
     the identical cascade kernel on the gathered L (uniforms are Philox(seed; round, pair)) and copies each local slot's
     source row out of the buffer, wherever that replica ran.  The host never looks at the permutation, so the whole round
     is queued without a host synchronisation: the segment kernel, the pack kernel, the collective (on torch's stream,
     chained to libptnn's stream by two events) and the swap kernel.
  "p2p": 1. all-gather of the R posted scalars L (4 R bytes); 2. cascade kernel, permutation to the host; 3. only rows whose
     source lives on another rank travel point-to-point into the destination row of the receiver's next-state buffer;
     4. local rows are copied by the swap kernel.  Least bytes (for nets whose rows are megabytes), three host waits.

`torch.distributed` is the plumbing (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests); the
tensors it moves are views of libptnn's own device buffers.  The driver is written against a small shard protocol so
that the CPU tests can run the same routing code over the oracle.
"""
import numpy as np


class DeviceShard:
    """Adapter: a `_lib.Sampler` handle seen through the shard protocol, device buffers as torch tensors."""

    def __init__(self, sampler, device_index):
        import torch
        self.torch = torch
        self.s = sampler
        self.dev = torch.device("cuda", device_index)
        self.R_local = sampler.cfg.n_replicas_local
        self.R_global = sampler.cfg.n_replicas_global
        self.first = sampler.cfg.first_global_replica
        self.S = sampler.S
        self.PS = sampler.state_row_floats()
        # stream-ordered mode: collectives are issued on the library's own HIP stream (as a torch ExternalStream), so the
        # segment kernel -> all-gather -> cascade -> row exchange -> apply chain is ordered on the device and the host only
        # waits once per round for the 4 R-byte permutation.  Opt-in (PTNN_DIST_STREAM=1) until measured on several GPUs.
        import os
        self.stream_ordered = os.environ.get("PTNN_DIST_STREAM", "0") == "1"
        self._views = {}
        # "gather" mode: event-chained by default; PTNN_DIST_SYNC=host falls back to host synchronisation around the collective
        self.host_sync = os.environ.get("PTNN_DIST_SYNC", "event") == "host"
        self.ext_stream = torch.cuda.ExternalStream(sampler.stream_ptr(), device=self.dev)
        self._ev_seg, self._ev_coll = torch.cuda.Event(), torch.cuda.Event()
        base, xs = sampler.xchg_ptr()
        self.XS = xs
        self._xchg = self._view(base, self.R_global * xs)

    def _view(self, ptr, n):
        class _Arr:            # __cuda_array_interface__ v2: zero-copy view of library-owned HBM
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(a, device=self.dev)

    def run_segment(self):
        return self.s.run_segment()

    def sync(self):
        if not self.stream_ordered:
            self.s.sync()

    def fence_collectives(self):
        if not self.stream_ordered:
            self.torch.cuda.synchronize(self.dev)

    def collective_context(self):
        if self.stream_ordered:
            return self.torch.cuda.stream(self.ext_stream)
        import contextlib
        return contextlib.nullcontext()

    def steps_done(self):
        return self.s.steps_done()

    def L_tensor(self, phantom):
        ptr = self.s.swap_L_ptr(phantom)                   # views are cached per device address (two L buffers, 2 Rl rows)
        t = self._views.get(ptr)
        if t is None:
            t = self._views[ptr] = self._view(ptr, self.R_global)
        return t

    def row_tensors(self, local):
        out = []
        for ptr in self.s.swap_row_ptr(local):
            t = self._views.get(ptr)
            if t is None:
                t = self._views[ptr] = self._view(ptr, self.PS)
            out.append(t)
        return tuple(out)

    def swap_cascade(self, phantom):
        return self.s.swap_cascade(phantom)

    # ---- gather mode ----
    def xchg_tensor(self):
        return self._xchg

    def pack(self, phantom):
        self.s.swap_pack(phantom)

    def before_collective(self):
        """Order torch's current stream (where the collective runs) after everything queued on libptnn's stream."""
        if self.host_sync:
            self.s.sync()
        else:
            self._ev_seg.record(self.ext_stream)
            self.torch.cuda.current_stream(self.dev).wait_event(self._ev_seg)

    def after_collective(self):
        """Order libptnn's stream (where the swap kernel runs) after the collective."""
        if self.host_sync:
            self.torch.cuda.synchronize(self.dev)
        else:
            self._ev_coll.record(self.torch.cuda.current_stream(self.dev))
            self.ext_stream.wait_event(self._ev_coll)

    def apply_gathered(self, phantom):
        self.s.swap_apply_gathered(phantom)

    def swap_apply(self, src, phantom):
        self.s.swap_apply(src, phantom)


def route(src, rank, world, R_local):
    """Which rows this rank receives and sends in one round.  Pure function of the permutation.
    Returns (recvs, sends): recvs = [(local_dest, peer)], sends = [(local_source, peer)], both in ascending order of
    the GLOBAL destination slot so that the two ends of every pair enumerate their messages in the same order."""
    first = rank * R_local
    src = np.asarray(src, dtype=np.int64)
    dst_owner = np.arange(world * R_local, dtype=np.int64) // R_local
    src_owner = src // R_local
    cross = np.nonzero(dst_owner != src_owner)[0]          # ascending global destination slot; usually a handful of rows
    recvs = [(int(kg) - first, int(src_owner[kg])) for kg in cross if dst_owner[kg] == rank]
    sends = [(int(src[kg]) - first, int(dst_owner[kg])) for kg in cross if src_owner[kg] == rank]
    return recvs, sends


class ShardedLadder:
    def __init__(self, shard, rank, world, dist=None, mode=None):
        if dist is None:
            import torch.distributed as dist
        import os
        self.dist = dist
        self.shard, self.rank, self.world = shard, rank, world
        self.mode = mode or os.environ.get("PTNN_DIST_MODE", "gather")
        if self.mode not in ("gather", "p2p"):
            raise ValueError("mode must be 'gather' or 'p2p'")
        self.rounds = 0
        self.bytes_moved = 0
        try:
            self._gather_in_place = hasattr(dist, "get_backend") and str(dist.get_backend()) == "nccl"
        except Exception:
            self._gather_in_place = False

    def swap_round(self, phantom):
        if self.mode == "gather":
            return self._swap_round_gather(phantom)
        return self._swap_round_p2p(phantom)

    def _swap_round_gather(self, phantom):
        sh, dist = self.shard, self.dist
        sh.pack(phantom)                                    # exchange rows of the local replicas (queued behind the segment)
        X = sh.xchg_tensor()
        n = sh.R_local * sh.XS
        sh.before_collective()
        if self._gather_in_place:                           # RCCL: in place, the input is this rank's block of the buffer
            dist.all_gather_into_tensor(X, X[self.rank * n:(self.rank + 1) * n])
        else:
            mine = X[self.rank * n:(self.rank + 1) * n].clone()
            dist.all_gather(list(X.split(n)), mine)
        sh.after_collective()
        sh.apply_gathered(phantom)                          # identical cascade on every rank + copy of the source rows
        self.bytes_moved += 4 * n * (self.world - 1)
        self.rounds += 1
        return None

    def _swap_round_p2p(self, phantom):
        import contextlib
        sh, dist = self.shard, self.dist
        ctx = sh.collective_context() if hasattr(sh, "collective_context") else contextlib.nullcontext()
        sh.sync()                                           # L of the local block is written
        L = sh.L_tensor(phantom)
        Rl = sh.R_local
        with ctx:
            if self._gather_in_place:                       # RCCL: in-place all-gather, the input is this rank's slice of L
                dist.all_gather_into_tensor(L, L[self.rank * Rl:(self.rank + 1) * Rl])
            else:
                mine = L[self.rank * Rl:(self.rank + 1) * Rl].clone()
                dist.all_gather(list(L.split(Rl)), mine)    # 4 R bytes, latency-bound
        if hasattr(sh, "fence_collectives"):
            sh.fence_collectives()
        src = sh.swap_cascade(phantom)                      # identical on every rank
        if not phantom:
            recvs, sends = route(src, self.rank, self.world, Rl)
            ops = []
            for local, peer in recvs:
                ops.append(dist.P2POp(dist.irecv, sh.row_tensors(local)[1], peer))
            for local, peer in sends:
                ops.append(dist.P2POp(dist.isend, sh.row_tensors(local)[0], peer))
            if ops:
                with ctx:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                if hasattr(sh, "fence_collectives"):
                    sh.fence_collectives()
                self.bytes_moved += 4 * sh.PS * len(ops)
        sh.swap_apply(src, phantom)
        self.rounds += 1
        return src

    def run_intervals(self, n_intervals=None):
        """Advance by n swap intervals (None = to the chain end).  Returns the number of intervals done."""
        done = 0
        while n_intervals is None or done < n_intervals:
            ho = self.shard.run_segment()
            if ho:
                self.swap_round(phantom=(ho == 2))
                done += 1
            if self.shard.steps_done() >= self.shard.S - 1 and ho != 1:
                break
        return done
