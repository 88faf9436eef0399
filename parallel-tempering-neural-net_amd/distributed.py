"""Sharded ladder on the host side: several GPUs, each owning a contiguous block of the temperature ladder.

The exchange itself lives in libptnn (include/ptnn.h, "sharded ladder"): ptnn_run performs the swap rounds through the
communicator attached to the handle -- RCCL over xGMI (ptnn_comm_init) or a host-staged transport behind two callbacks
(ptnn_comm_init_host).  It replaces the reference's star topology, where every replica ships its whole parameter vector to
the parent through a multiprocessing.Queue each swap round and blocks on an Event (REG:427-437 <-> 694-759).

This module holds what the Python host needs around that:

  LadderGroup      one process driving several handles (one host thread per GPU, the GIL is released inside libptnn): what
                   `ParallelTempering(..., devices=[0, 1, ...])` runs on.  Distinct devices talk over RCCL; when a device
                   appears twice (a one-GPU box rehearsing the N > 1 path) the blocks talk through ThreadTransport.
  ThreadTransport  host-staged transport between the threads of one process (barrier + mailboxes).
  gloo_transport   host-staged transport between processes over torch.distributed's gloo backend (tests, and bench.py's
                   --transport host rehearsal); torch is imported only if this is called.

One process per GPU (bench.py under torch.distributed.run) needs none of it: each rank builds its `_lib.Sampler`, calls
`comm_init(unique_id, rank, nranks)` and `run()`.
"""
import atexit
import json
import os
import queue
import struct
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAULT_STAGES = ("ncclGetUniqueId", "ncclCommInitRank")      # what $PTNN_COMM_FAULT can name (ptnn.hip parses it the same way)


def injected_fault():
    """The stages $PTNN_COMM_FAULT makes libptnn fail at (test hook), () when it names none the library knows."""
    f = os.environ.get("PTNN_COMM_FAULT", "")
    return tuple(st for st in FAULT_STAGES if st in f)


def single_node_rccl_env(environ=None):
    """RCCL defaults for a ladder sharded inside ONE node, set only where unset and only by callers that own their process
    (LadderGroup before its first thread exists, bench.py at start-up; libptnn itself never touches the environment): bootstrap
    over loopback, no InfiniBand probe -- two stages of ncclGetUniqueId / ncclCommInitRank whose duration otherwise depends on
    the box's network set-up -- and dmabuf IPC, all this pool's driver offers.  $PTNN_COMM_KEEP_ENV=1 leaves everything alone."""
    env = os.environ if environ is None else environ
    if env.get("PTNN_COMM_KEEP_ENV", "0") not in ("", "0"):
        return env
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    env.setdefault("NCCL_IB_DISABLE", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


_PROBE_CHILD = ("import json, os, sys; sys.path.insert(0, {root!r}); import ptnn_amd; from ptnn_amd import _lib\n"
                "try:\n"
                "    out = dict(ok=True, seconds=_lib.comm_probe({devices!r}))\n"
                "except Exception as e:\n"
                "    out = dict(ok=False, why=str(e), last_stage=_lib.comm_last_stage())\n"
                "sys.stdout.write('PTNN_PROBE ' + json.dumps(out) + '\\n'); sys.stdout.flush()\n"
                "os._exit(0 if out['ok'] else 3)\n")      # no interpreter teardown: a half-initialised RCCL must not hold the child


def rccl_probe(devices, timeout_s=None):
    """Does RCCL come up among `devices`?  Asked in a FRESH CHILD process (a new interpreter started with subprocess, never a
    re-exec of this one), so that this process has not touched RCCL when the answer is no: a unique id that no ncclCommInitRank
    follows, or init helpers abandoned inside RCCL, can keep the process they live in from exiting.  The child runs
    ptnn_comm_probe (unique id, ncclCommInitRank per device, one all-gather, destroy) under libptnn's own bounds; this side
    waits `timeout_s` (default $PTNN_COMM_TIMEOUT_S + 30, 150 s) and then kills that one pid.  -> (ok, text)."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("PTNN_COMM_TIMEOUT_S", "120")) + 30.0
    env = single_node_rccl_env(dict(os.environ))
    code = _PROBE_CHILD.format(root=_ROOT, devices=[int(d) for d in devices])
    try:
        child = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
    except OSError as e:
        return False, f"could not start the probe process: {e}"
    try:
        out, err = child.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        child.kill()                                         # this pid only
        child.communicate()
        return False, f"the RCCL probe process did not finish within {timeout_s:g} s and was killed"
    for line in out.splitlines():
        if line.startswith("PTNN_PROBE "):
            res = json.loads(line[len("PTNN_PROBE "):])
            if res["ok"]:
                return True, f"RCCL probe over devices {list(devices)}: {res['seconds']:.2f} s"
            return False, f"{res['why']} (last stage: {res['last_stage']})"
    return False, f"the RCCL probe process ended with code {child.returncode} and no verdict: {err.strip()[-300:]}"


_hard_exit_armed = False


def arm_hard_exit():
    """An RCCL bring-up failed INSIDE this process (after the child's probe had passed, or with the probe switched off): native
    threads RCCL started may never end, and the C runtime would wait for them in its own exit handlers for ever (seen on the
    test box: an ncclGetUniqueId that no ncclCommInitRank follows).  From here on the process leaves through os._exit: an
    atexit hook runs the other Python exit handlers, flushes the standard streams and ends the process with the status
    sys.exit() or an uncaught exception had asked for."""
    global _hard_exit_armed
    if _hard_exit_armed:
        return
    _hard_exit_armed = True
    status = [0]
    orig_exit, orig_hook = sys.exit, sys.excepthook

    def _exit(code=0):
        status[0] = code if isinstance(code, int) else (0 if code is None else 1)
        orig_exit(code)

    def _hook(*a):
        status[0] = 1
        orig_hook(*a)
    sys.exit, sys.excepthook = _exit, _hook
    leaving = [False]
    orig_register = atexit.register

    def register_once(fn, *a, **k):                          # handlers registered from now on run before _leave AND would be run again
        ran = [False]                                        # by its _run_exitfuncs(): make them idempotent

        def once(*aa, **kk):
            if not ran[0]:
                ran[0] = True
                return fn(*aa, **kk)
        orig_register(once, *a, **k)
        return fn
    atexit.register = register_once

    def _leave():
        if leaving[0]:
            return
        leaving[0] = True
        try:
            atexit._run_exitfuncs()                          # the handlers registered before this one
        finally:
            try:
                sys.stdout.flush()
                sys.stderr.flush()
            finally:
                os._exit(status[0])
    orig_register(_leave)


class TransportAborted(RuntimeError):
    """A block of the ladder failed: the collective this block was waiting in can never complete."""


class ThreadTransport:
    """Host-staged transport between n handles of one process.  all_gather is a collective (every rank calls it each round);
    send_recv is called only by ranks that have messages, so rows travel through per-pair mailboxes.

    No wait is unbounded: when one block fails (its ptnn_run returns an error, or an exception leaves its callback) it calls
    `abort()`, which breaks the barrier and poisons the mailboxes, so the blocks waiting for it raise `TransportAborted` at once
    instead of hanging; independently every wait gives up after `timeout` seconds (the reference's parent would block for ever in
    queue.get() when a replica dies, REG:661 -- SURVEY section 5)."""

    def __init__(self, n, timeout=120.0):
        self.n = n
        self.timeout = float(timeout)
        self._barrier = threading.Barrier(n)
        self._blocks = [None] * n
        self._mail = {(a, b): queue.Queue() for a in range(n) for b in range(n) if a != b}
        self.failed = None                                   # why the transport was aborted

    def abort(self, why="a block of the ladder failed"):
        if self.failed is None:
            self.failed = str(why)
        self._barrier.abort()
        for q in self._mail.values():
            q.put(None)                                      # wakes a receiver; None = aborted

    def _wait(self):
        try:
            self._barrier.wait(self.timeout)
        except threading.BrokenBarrierError:
            if self.failed is None:
                self.failed = f"a block did not reach the all-gather of this swap round within {self.timeout:g} s"
                self.abort(self.failed)
            raise TransportAborted(self.failed) from None

    def callbacks(self, rank):
        def all_gather(buf):                                 # buf: uint8 [n, bytes_per_rank], own block filled
            if self.failed is not None:
                raise TransportAborted(self.failed)
            self._blocks[rank] = buf[rank].copy()
            self._wait()
            for r in range(self.n):
                if r != rank:
                    buf[r] = self._blocks[r]
            self._wait()                                     # nobody overwrites its block before everyone has read it

        def send_recv(msgs):                                 # [(peer, is_send, uint8 row)]
            if self.failed is not None:
                raise TransportAborted(self.failed)
            for peer, is_send, row in msgs:
                if is_send:
                    self._mail[(rank, peer)].put(row.copy())
            for peer, is_send, row in msgs:
                if not is_send:
                    try:
                        got = self._mail[(peer, rank)].get(timeout=self.timeout)
                    except queue.Empty:
                        self.abort(f"block {rank} waited {self.timeout:g} s for a row from block {peer}")
                        raise TransportAborted(self.failed) from None
                    if got is None:
                        raise TransportAborted(self.failed)
                    row[:] = got
        return all_gather, send_recv


def gloo_transport(dist):
    """(all_gather, send_recv) over an initialised torch.distributed process group with CPU tensors (gloo)."""
    import torch
    rank = dist.get_rank()

    def all_gather(buf):
        blocks = [torch.from_numpy(buf[r]) for r in range(buf.shape[0])]
        dist.all_gather(blocks, blocks[rank].clone())

    def send_recv(msgs):
        works = []
        for peer, is_send, row in msgs:
            t = torch.from_numpy(row)
            works.append(dist.isend(t, peer) if is_send else dist.irecv(t, peer))
        for w in works:
            w.wait()
    return all_gather, send_recv


_GROUP_MAGIC = b"PTNG"


class LadderGroup:
    """The handles of one ladder cut into len(devices) equal contiguous blocks, driven by one host thread each.  Offers the
    part of the `_lib.Sampler` interface `ParallelTempering` uses, over the whole ladder."""

    def __init__(self, devices, exchange=_lib.XCHG_AUTO, transport=None, fallback=None, **config):
        self.devices = [int(d) for d in devices]
        n = len(self.devices)
        R = int(config["n_replicas_global"])
        if n < 1 or R % n != 0:
            raise ValueError(f"{R} replicas cannot be cut into {n} equal blocks (one per device)")
        self.n, self.R, self.Rl = n, R, R // n
        distinct = len(set(self.devices)) == n
        if fallback is None:
            fallback = transport in (None, "auto")           # an RCCL the caller asked for by name fails loudly
        if transport in (None, "auto"):
            transport = "rccl" if distinct else "host"
        if transport not in ("rccl", "host"):
            raise ValueError("transport must be None, 'auto', 'rccl' or 'host'")
        if transport == "rccl" and not distinct and not injected_fault():
            # never attempted: two ranks of ONE process initialising RCCL on one device did not come back on the test box.  The
            # only way past this check is a $PTNN_COMM_FAULT that names a stage libptnn really fails at BEFORE ncclCommInitRank
            # runs (tests of the fall-back on a one-GPU box); any other value of that variable leaves the check in force.
            raise ValueError("RCCL needs one distinct device per block; use transport='host' to rehearse on one GPU")
        self.transport = transport
        self.transport_note = None
        self.rccl_probe_result = None
        if transport == "rccl" and n > 1:
            single_node_rccl_env()                           # before the first thread of this group exists
        self._pool = ThreadPoolExecutor(max_workers=n)
        self.shards = [None] * n
        self._tt = None                                      # ThreadTransport of a host-staged group

        shared_gpu = len(set(self.devices)) < n
        def create(k):
            cfg = dict(config, device_id=self.devices[k], n_replicas_local=self.Rl, first_global_replica=k * self.Rl)
            # blocks that share one GPU (a rehearsal of the N > 1 path): no schedule whose work-groups wait for each other
            if shared_gpu:
                cfg["shared_device"] = 1
            self.shards[k] = _lib.Sampler(**cfg)
        self._each(create)
        s0 = self.shards[0]
        self.cfg, self.P, self.S = s0.cfg, s0.P, s0.S
        if n > 1:
            if transport == "rccl":
                # ncclCommInitRank is a collective: all threads at once, all with the same id.  libptnn bounds it (a rank that
                # fails before joining makes the others return error -7 after $PTNN_COMM_TIMEOUT_S instead of blocking), and
                # _each() collects every thread before the first error is raised.
                # RCCL has never run on more than one GPU here: with `fallback` a failed bring-up (every stage bounded) moves the
                # whole group to the host-staged transport, with a warning and `transport_note`, instead of ending the run.
                # Before THIS process touches RCCL the same bring-up is rehearsed in a fresh child process (rccl_probe): when RCCL
                # does not come up among these devices, the answer arrives without a half-initialised RCCL in the process that
                # has to go on (and, one day, exit).  $PTNN_RCCL_PROBE=0 skips the rehearsal.
                why = None
                if os.environ.get("PTNN_RCCL_PROBE", "1") not in ("0", ""):
                    ok, text = rccl_probe(self.devices)
                    self.rccl_probe_result = (ok, text)
                    if not ok:
                        why = f"the probe in a child process failed, RCCL was not touched in this process: {text}"
                if why is None:
                    try:
                        uid = _lib.comm_unique_id()
                        self._each(lambda k: self.shards[k].comm_init(uid, k, n))
                    except _lib.PtnnError as e:
                        why = f"{e} (last stage: {_lib.comm_last_stage()})"
                        arm_hard_exit()                     # RCCL is partly up in this process: it may never let it exit
                        if not fallback:
                            raise
                if why is not None:
                    if not fallback:
                        raise _lib.PtnnError(why)
                    import warnings
                    self.transport_note = f"host-staged after an RCCL bring-up failure: {why}"
                    warnings.warn("LadderGroup: " + self.transport_note, RuntimeWarning, stacklevel=2)
                    for sh in self.shards:                   # a block whose own bring-up had returned: drop its communicator
                        try:
                            sh.comm_finalize()
                        except _lib.PtnnError:
                            pass
                    self.failed = False
                    self.transport = transport = "host"
            if transport == "host":
                self._tt = ThreadTransport(n)
                self._each(lambda k: self.shards[k].comm_init_host(k, n, *self._tt.callbacks(k)))
            self._each(lambda k: self.shards[k].comm_set_mode(exchange))

    def _each(self, fn):
        """fn(k) on every block's own thread.  One failing block must not leave the others inside a collective: its error
        aborts the host-staged transport at once (RCCL collectives are bounded inside libptnn), every thread is waited for, and
        only then the first error is raised -- the pool is never left with stuck workers, so close() can always run."""
        def guarded(k):
            try:
                return fn(k)
            except BaseException as e:                       # noqa: BLE001
                if self._tt is not None:
                    self._tt.abort(f"block {k}: {e}")
                raise
        futures = [self._pool.submit(guarded, k) for k in range(self.n)]
        results, first = [], None
        for f in futures:
            try:
                results.append(f.result())
            except BaseException as e:                       # noqa: BLE001
                results.append(None)
                if first is None or (isinstance(first, TransportAborted) and not isinstance(e, TransportAborted)):
                    first = e                                # the cause, not the blocks that were woken by the abort
        if first is not None:
            self.failed = True
            raise first
        return results

    # ---- configuration
    def set_data(self, train, test):
        self._each(lambda k: self.shards[k].set_data(train, test))

    def set_state(self, w0, temperatures):
        w0, t = np.asarray(w0), np.asarray(temperatures)
        Rl = self.Rl
        self._each(lambda k: self.shards[k].set_state(w0[k * Rl:(k + 1) * Rl], t[k * Rl:(k + 1) * Rl]))

    def set_ladder(self, temperatures_global):
        self._each(lambda k: self.shards[k].set_ladder(temperatures_global))

    # ---- running: every block advances by the same number of steps; the swap rounds inside exchange through the communicator
    def run(self, n_steps=-1):
        self._each(lambda k: self.shards[k].run(n_steps))

    def sync(self):
        self._each(lambda k: self.shards[k].sync())

    def steps_done(self):
        return self.shards[0].steps_done()

    # ---- results, in ladder order
    def traces(self, step0=0, nsteps=None, pos_w=True):
        parts = self._each(lambda k: self.shards[k].traces(step0, nsteps, pos_w))
        return {key: (None if parts[0][key] is None else np.concatenate([p[key] for p in parts], axis=0)) for key in parts[0]}

    def state(self):
        parts = self._each(lambda k: self.shards[k].state())
        return {key: np.concatenate([p[key] for p in parts], axis=0) for key in parts[0]}

    def swap_stats(self):
        stats = self._each(lambda k: self.shards[k].swap_stats())
        if any(st != stats[0] for st in stats):              # every rank counts the whole (replicated) cascade
            raise _lib.PtnnError(f"the blocks disagree about the swap rounds: {stats}")
        return stats[0]

    def swap_log(self, max_rounds=None):
        return self.shards[0].swap_log(max_rounds)

    def kernel_time(self, reset=False):
        t = self._each(lambda k: self.shards[k].kernel_time(reset))
        return max(a for a, _ in t), max(b for _, b in t)

    def comm_stats(self):
        return self._each(lambda k: self.shards[k].comm_stats()) if self.n > 1 else []

    def comm_info(self):
        return self._each(lambda k: self.shards[k].comm_info())

    def describe(self):
        return self.shards[0].describe()

    # ---- checkpoint: the blocks' blobs behind a small index
    def checkpoint(self):
        blobs = self._each(lambda k: self.shards[k].checkpoint())
        return _GROUP_MAGIC + struct.pack("<i", self.n) + b"".join(struct.pack("<q", len(b)) for b in blobs) + b"".join(blobs)

    def restore(self, blob):
        if blob[:4] != _GROUP_MAGIC or struct.unpack("<i", blob[4:8])[0] != self.n:
            raise _lib.PtnnError(f"not a checkpoint of a ladder cut into {self.n} blocks")
        lens = struct.unpack(f"<{self.n}q", blob[8:8 + 8 * self.n])
        offs = np.concatenate([[8 + 8 * self.n], 8 + 8 * self.n + np.cumsum(lens)]).astype(int)
        self._each(lambda k: self.shards[k].restore(blob[offs[k]:offs[k + 1]]))

    def close(self):
        if self.shards:
            shards, self.shards = self.shards, []
            try:
                self._each(lambda k: shards[k].close() if shards[k] is not None else None)
            finally:
                self._pool.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:                                    # noqa: BLE001
            pass
