"""Child process of test_rccl_communicator_world_size_one: the RCCL transport of libptnn (ptnn_comm_init -> dlopen librccl,
ncclCommInitRank, in-place ncclAllGather on the handle's own stream) at world size 1 -- the only RCCL world a one-GPU box
allows -- in both exchange modes and under swap_rule 1, against the plain single-handle run.  No torch in this process."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("PTNN_COMM_TRACE", "1")           # libptnn stamps every communicator stage on stderr
os.environ.setdefault("PTNN_COMM_TIMEOUT_S", "60")      # ... and gives up on a stage after this long (error -7 naming it)
T0 = time.time()


def stamp(what):
    """Flushed progress line: if the parent's time limit kills this process, the last one says where it was."""
    print(f"[dist_device_check {time.time() - T0:8.3f} s] {what}", file=sys.stderr, flush=True)


def timeout_case():
    """A communicator of two ranks of which only this one exists: ncclCommInitRank can never complete, and ptnn_comm_init must
    come back with error -7 naming the stage instead of blocking (PTNN_COMM_TIMEOUT_S)."""
    import parity
    from parity import orc
    from ptnn_amd import _lib, ladder, philox
    d = parity.datasets()
    topo, R = (4, 5, 1), 8
    s_ = parity.make_sampler(orc.TASK_REG, topo, d["sunspot_train"], d["sunspot_test"], R_local=R // 2, R_global=R, first=0, S=40, si=10,
                             use_lg=True, lr=0.1, seed=5)
    stamp("timeout case: comm_unique_id")
    uid = _lib.comm_unique_id()
    stamp("timeout case: comm_init(rank 0 of 2), the peer never joins")
    t = time.time()
    try:
        s_.comm_init(uid, 0, 2)
    except _lib.PtnnError as e:
        took = time.time() - t
        stamp(f"timeout case: error after {took:.1f} s: {e}")
        assert "ncclCommInitRank(rank 0 of 2" in str(e) and "did not return" in str(e), str(e)
        assert "ncclCommInitRank" in _lib.comm_last_stage()
        assert took < float(os.environ["PTNN_COMM_TIMEOUT_S"]) + 10
        print("OK bounded init", flush=True)
        os._exit(0)          # the abandoned helper thread is still inside ncclCommInitRank: leave without waiting for it
    raise AssertionError("comm_init of a world that never completes returned without an error")


def main():
    stamp("start")
    import parity
    from parity import orc
    from ptnn_amd import _lib, ladder, philox
    stamp("imports done")
    d = parity.datasets()
    train, test = d["sunspot_train"], d["sunspot_test"]
    topo, R, S, si = (4, 5, 1), 16, 8 * 12 + 3, 12
    Pw = topo[0] * topo[1] + topo[1] * topo[2] + topo[1] + topo[2]

    def make(rule=0):
        s_ = parity.make_sampler(orc.TASK_REG, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=si, use_lg=True,
                                 lr=0.1, seed=77, swap_rule=rule)
        s_.set_state(np.stack([philox.initial_weights(77, r, Pw) for r in range(R)]), ladder.temperatures(R, 2))
        if rule:
            s_.set_ladder(ladder.temperatures(R, 2))
        return s_
    assert "torch" not in sys.modules
    for mode, rule in ((_lib.XCHG_GATHER, 0), (_lib.XCHG_BOUNDARY, 0), (_lib.XCHG_GATHER, 1)):
        stamp(f"mode {mode} rule {rule}: plain run")
        ref = make(rule)
        ref.run(-1)
        ref.sync()
        want, want_log, want_stats = ref.traces(), ref.swap_log(), ref.swap_stats()
        ref.close()
        assert want_stats[0] > 0
        s = make(rule)
        stamp("comm_unique_id")
        uid = _lib.comm_unique_id()
        stamp("comm_init")
        s.comm_init(uid, 0, 1)
        s.comm_set_mode(mode)
        stamp("run(40)")
        s.run(40)                                            # in chunks, like a caller that drains a trace ring
        stamp("run(-1)")
        s.run(-1)
        stamp("sync")
        s.sync()
        stamp("synced")
        got = s.traces()
        for k in want:
            assert np.array_equal(got[k], want[k]), (mode, k)
        assert np.array_equal(s.swap_log(), want_log), mode
        assert s.swap_stats() == want_stats, mode
        st = s.comm_stats()
        assert st["rounds"] == want_stats[2] and st["mode"] == {1: "gather", 2: "boundary"}[mode], st
        assert s.describe()["exchange"] == st["mode"]
        s.close()
        print("OK", st["mode"], "rule", rule, flush=True)
    assert "torch" not in sys.modules
    print("OK no torch", flush=True)


if __name__ == "__main__":
    if "--timeout-case" in sys.argv:
        os.environ["PTNN_COMM_TIMEOUT_S"] = "8"
        timeout_case()
    else:
        main()
