"""N > 1 path on CPU (no GPU in this container).  What can run here is everything of the sharded ladder that is host logic:

* `ptnn_route` -- the C routing function of libptnn (which rows cross which GPU boundary) -- against the properties SURVEY 8e
  states, for permutations the cascade can produce;
* the two host-staged transports of distributed.py (gloo between processes at world size 2 and 4, ThreadTransport between
  the threads of one process) moving real bytes in the order the library asks for them;
* the round protocol of `comm_swap_round` (csrc/ptnn.hip), restated here over the ORACLE as compute engine (`OracleRank`):
  gathered exchange, boundary exchange (all-gather of L, replicated cascade, routed rows) and the gathered exchange under
  swap_rule 1 must reproduce the single-process oracle run bit for bit -- results identical for every GPU count at a fixed seed.

The device side of the same protocol is tested on the GPU box (tests/dist_device_check*.py, test_gpu_dropin.py)."""
import os
import socket
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ptnn_oracle as orc  # noqa: E402


def _lib():
    import __graft_entry__
    __graft_entry__.build()
    import ptnn_amd  # noqa: F401
    from ptnn_amd import _lib
    return _lib


def test_route_is_consistent_between_ranks():
    lib = _lib()
    rng = np.random.default_rng(0)
    for world, Rl in ((2, 2), (4, 4), (8, 32), (8, 128), (1, 8)):
        R = world * Rl
        for _ in range(20):
            # permutations the cascade can produce: bubble pass with random swap decisions
            u = rng.uniform(size=R - 1)
            src, _ = orc.swap_cascade(list(rng.normal(0, 2, R)), u)
            plans = [lib.route(src, Rl, r) for r in range(world)]
            sent = sorted((r, peer, row + r * Rl, dst) for r, (rc, sd) in enumerate(plans) for row, peer, dst in sd)
            recvd = sorted((peer, r, src[dst], dst) for r, (rc, sd) in enumerate(plans) for row, peer, dst in rc)
            assert sent == recvd                       # every receive has its matching send (same source row, same slot)
            for r, (rc, sd) in enumerate(plans):
                assert all(row + r * Rl == dst for row, _, dst in rc)
                below = [p for _, p, _ in rc if p < r]
                above = [p for _, p, _ in rc if p > r]
                assert len(below) <= 1 and len(above) <= 1      # at most one row from below, one from above
                assert all(p == r + 1 for p in above)
                assert len([p for _, p, _ in sd if p > r]) <= 1 and len([p for _, p, _ in sd if p < r]) <= 1
            # per ordered pair, both ends list their messages in the same (global destination) order
            for a in range(world):
                for b in range(world):
                    if a != b:
                        assert [d for _, p, d in plans[a][1] if p == b] == [d for _, p, d in plans[b][0] if p == a]
    with pytest.raises(lib.PtnnError):
        lib.route([0, 1, 2, 7], 2, 0)


class OracleRank:
    """One rank of a sharded ladder: the round protocol of comm_swap_round (csrc/ptnn.hip) over oracle replicas (float64)."""

    def __init__(self, pt_args, rank, world, transport, mode, lib):
        full = orc.PTOracle(*pt_args["args"], **pt_args["kw"])
        self.lib, self.rank, self.world, self.mode = lib, rank, world, mode
        self.all_gather, self.send_recv = transport
        self.task, self.si, self.S, self.tape = full.task, full.si, full.S, full.tape
        self.R = full.R
        self.Rl = full.R // world
        self.first = rank * self.Rl
        self.reps = full.replicas[self.first:self.first + self.Rl]
        self.P = full.P
        self.PS = self.P + 1
        self.swap_rule = int(pt_args["kw"].get("swap_rule", 0))
        self.temps = list(full.temperatures)
        self.rounds_done = self.num_swap = 0
        self.bytes_rows = 0
        self.XS = self.PS + 3                       # exchange row: {w, eta | posted L | untempered L | prior}

    def _gather(self, arr):
        """arr: float64 [R, width], this rank's rows filled -> all rows filled."""
        width = arr.shape[1]
        buf = arr.reshape(self.world, self.Rl * width).view(np.uint8)
        self.all_gather(buf)

    def swap_round(self, phantom):
        R, Rl, P, PS = self.R, self.Rl, self.P, self.PS
        posted = [rep.likelihood if phantom else rep.posted_L() for rep in self.reps]
        u = self.tape.swap_uniforms(self.rounds_done, R - 1)
        if self.mode == "gather" or self.swap_rule == 1:
            X = np.zeros((R, self.XS))
            for k, rep in enumerate(self.reps):
                row = X[self.first + k]
                row[:P], row[P], row[PS] = rep.w, rep.eta, posted[k]
                row[PS + 1], row[PS + 2] = rep.likelihood * rep.adapttemp, rep.prior_current
            self._gather(X)
            if self.swap_rule == 1:
                import math
                src = list(range(R))
                for k in range(self.rounds_done & 1, R - 1, 2):
                    dd = (1.0 / self.temps[k] - 1.0 / self.temps[k + 1]) * (X[k + 1, PS + 1] - X[k, PS + 1])
                    pr = 1.0 if dd != dd else min(1.0, math.exp(min(dd, 80.0)))
                    if u[k] < pr:
                        src[k], src[k + 1] = k + 1, k
                        self.num_swap += 1
                for k, rep in enumerate(self.reps):
                    sg = src[self.first + k]
                    if sg != self.first + k:
                        rep.w, rep.eta = X[sg, :P].copy(), float(X[sg, P])
                        rep.prior_current = float(X[sg, PS + 2])
                        rep.likelihood = float(X[sg, PS + 1]) / rep.adapttemp
            else:
                src, nsw = orc.swap_cascade(X[:, PS].tolist(), u)
                if not phantom:
                    for k, rep in enumerate(self.reps):
                        sg = int(src[self.first + k])
                        rep.w, rep.eta = X[sg, :P].copy(), float(X[sg, P])
                self.num_swap += nsw
        else:                                               # boundary exchange
            L = np.zeros((R, 1))
            L[self.first:self.first + Rl, 0] = posted
            self._gather(L)
            src, nsw = orc.swap_cascade(L[:, 0].tolist(), u)
            if not phantom:
                cur = np.stack([np.concatenate([rep.w, [rep.eta]]) for rep in self.reps])
                nxt = np.zeros_like(cur)
                recvs, sends = self.lib.route(src, Rl, self.rank)
                msgs = sorted([(dst, peer, False, nxt[row].view(np.uint8)) for row, peer, dst in recvs] +
                              [(dst, peer, True, cur[row].view(np.uint8)) for row, peer, dst in sends])
                if msgs:
                    self.send_recv([(peer, is_send, buf) for _, peer, is_send, buf in msgs])
                    self.bytes_rows += 8 * PS * len(msgs)
                for k in range(Rl):
                    sl = int(src[self.first + k]) - self.first
                    if 0 <= sl < Rl:
                        nxt[k] = cur[sl]
                for k, rep in enumerate(self.reps):
                    rep.w, rep.eta = nxt[k, :P].copy(), float(nxt[k, P])
            self.num_swap += nsw
        self.rounds_done += 1

    def run(self):                                          # the loop of ptnn_run
        for i in range(self.S - 1):
            for rep in self.reps:
                rep.step(i)
            if orc.swap_trigger(self.task, i, self.si):
                self.swap_round(False)
        if self.swap_rule == 0 and int(self.S / self.si) > self.rounds_done:
            self.swap_round(True)
        return self

    def result(self):
        return dict(pos_w=np.stack([r.pos_w for r in self.reps]), accept=np.stack([r.accept_list for r in self.reps]),
                    likeh=np.stack([r.likeh for r in self.reps]), rmse=np.stack([r.rmse_train for r in self.reps]),
                    num_swap=self.num_swap, rounds=self.rounds_done, bytes_rows=self.bytes_rows)


def _case(datasets, task, rule):
    if task == orc.TASK_REG:
        args = (task, (4, 5, 1), datasets["sunspot_train"], datasets["sunspot_test"], 8, 2, 8 * 43, 5)
        kw = dict(use_lg=True, l_prob=0.5, lr=0.1, seed=31)        # S = 43: no phantom; hand-offs at i = 5..40
    else:
        args = (task, (4, 12, 3), datasets["iris_train"], datasets["iris_test"], 8, 10, 8 * 40, 5)
        kw = dict(use_lg=False, l_prob=0.5, lr=0.01, seed=32)      # S = 40: S % si == 0 -> phantom round
    if rule:
        kw["swap_rule"] = 1
    return dict(args=args, kw=kw)


def _check(results, ref, world, mode):
    Rl = 8 // world
    rows = 0
    for rank, z in enumerate(results):
        for k in range(Rl):
            rep = ref.replicas[rank * Rl + k]
            assert (z["pos_w"][k] == rep.pos_w).all()
            assert (z["accept"][k] == rep.accept_list).all()
            assert (z["likeh"][k] == rep.likeh).all()
            assert (z["rmse"][k] == rep.rmse_train).all()
        assert int(z["num_swap"]) == ref.num_swap and int(z["rounds"]) == ref.rounds_done
        rows += int(z["bytes_rows"])
    assert ref.num_swap > 0
    if mode == "boundary":
        # something crossed a shard boundary, and never more than two rows in + two rows out per rank per round
        assert 0 < rows <= ref.rounds_done * world * 4 * 8 * (ref.P + 1)


def _worker(rank, world, port, pt_args, outdir, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib()
    from ptnn_amd import distributed as dm
    r = OracleRank(pt_args, rank, world, dm.gloo_transport(dist), mode, lib).run()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **r.result())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [(2, orc.TASK_REG, "gather", 0), (4, orc.TASK_CLS, "gather", 0), (2, orc.TASK_REG, "boundary", 0),
         (4, orc.TASK_CLS, "boundary", 0), (4, orc.TASK_REG, "gather", 1)]


@pytest.mark.parametrize("world,task,mode,rule", CASES)
def test_sharded_ladder_over_gloo_matches_single_process(tmp_path, datasets, world, task, mode, rule):
    import torch.multiprocessing as mp
    pt_args = _case(datasets, task, rule)
    ref = orc.PTOracle(*pt_args["args"], **pt_args["kw"]).run()
    mp.spawn(_worker, args=(world, _free_port(), pt_args, str(tmp_path), mode), nprocs=world, join=True)
    _check([np.load(tmp_path / f"rank{rank}.npz") for rank in range(world)], ref, world, mode)


@pytest.mark.parametrize("world,task,mode,rule", CASES + [(8, orc.TASK_REG, "boundary", 0)])
def test_sharded_ladder_over_thread_transport_matches_single_process(datasets, world, task, mode, rule):
    """The same protocol between the threads of one process: the transport `ParallelTempering(devices=[...])` falls back to
    when a device is listed twice."""
    lib = _lib()
    from ptnn_amd import distributed as dm
    pt_args = _case(datasets, task, rule)
    ref = orc.PTOracle(*pt_args["args"], **pt_args["kw"]).run()
    tt = dm.ThreadTransport(world)
    ranks = [OracleRank(pt_args, r, world, tt.callbacks(r), mode, lib) for r in range(world)]
    errs = []

    def go(r):
        try:
            r.run()
        except Exception as e:                               # noqa: BLE001
            errs.append(e)
            tt._barrier.abort()
    th = [threading.Thread(target=go, args=(r,)) for r in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    _check([r.result() for r in ranks], ref, world, mode)


def test_rccl_call_pattern_with_mock_library(tmp_path):
    """The RCCL code path of csrc/ptnn_comm.hpp (Comm::all_gather, Comm::exchange_rows, route_rows) against an in-process fake of
    librccl (tests/native/comm_mock.cpp): 8 ranks as threads, R = 256 with Ionosphere's rows and R = 1024 with config 5's, hundreds
    of random cascades -- in-place all-gathers with equal counts, sends / receives only inside one group per rank and round and
    matched FIFO per pair, at most one row in from below and one from above, the ladder equal to the permutation applied directly,
    and an injected collective failure reported by its rank without hanging the others.  Never run on N > 1 GPUs; this is the
    rehearsal of that call sequence a one-GPU pool allows."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "comm_mock")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "comm_mock.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK all" in r.stdout and r.stdout.count("\nOK R=") + r.stdout.startswith("OK R=") == 5 and "OK failure-injection" in r.stdout
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_one_failing_block_aborts_the_group_instead_of_hanging():
    """distributed.ThreadTransport / LadderGroup._each: when one block of a ladder fails inside a swap round, the blocks
    waiting for it in the all-gather or for one of its rows are woken with TransportAborted, the cause (not the wake-ups) is what
    the caller sees, the worker pool is left without stuck threads, and nothing waits longer than the transport's timeout."""
    import time
    import ptnn_amd  # noqa: F401
    from ptnn_amd import distributed as dm
    n = 4
    tt = dm.ThreadTransport(n, timeout=30.0)
    cbs = [tt.callbacks(k) for k in range(n)]

    class Boom(RuntimeError):
        pass

    grp = dm.LadderGroup.__new__(dm.LadderGroup)            # the part of LadderGroup under test needs no GPU: pool + transport
    from concurrent.futures import ThreadPoolExecutor
    grp.n, grp._pool, grp._tt, grp.shards = n, ThreadPoolExecutor(max_workers=n), tt, []

    def round_(k):
        buf = np.zeros((n, 8), dtype=np.uint8)
        buf[k] = k + 1
        cbs[k][0](buf)                                       # round 0: everybody arrives
        assert [int(buf[r, 0]) for r in range(n)] == [1, 2, 3, 4]
        if k == 2:
            raise Boom("device error on block 2")            # e.g. ptnn_run returned -2 on this rank
        if k == 3:
            cbs[k][1]([(2, 0, np.zeros(4, dtype=np.uint8))])  # waits for a row block 2 will never send
        else:
            cbs[k][0](buf)                                   # next all-gather: block 2 never comes
        return k
    t0 = time.time()
    with pytest.raises(Boom, match="block 2"):
        grp._each(round_)
    assert time.time() - t0 < 10.0                           # woken by the abort, not by the 30 s timeout
    assert tt.failed and "block 2" in tt.failed
    with pytest.raises(dm.TransportAborted):                 # the transport stays failed: a later round cannot half-run
        cbs[0][0](np.zeros((n, 8), dtype=np.uint8))
    assert grp._each(lambda k: k * k) == [0, 1, 4, 9]        # the pool has no stuck workers
    grp._pool.shutdown(wait=True)
    # a block that silently never arrives: bounded by the timeout, reported
    tt2 = dm.ThreadTransport(2, timeout=0.5)
    ag0 = tt2.callbacks(0)[0]
    t0 = time.time()
    with pytest.raises(dm.TransportAborted, match="did not reach the all-gather"):
        ag0(np.zeros((2, 4), dtype=np.uint8))
    assert time.time() - t0 < 5.0
