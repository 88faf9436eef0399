"""Child process of test_rccl_communicator_world_size_one: the RCCL transport of libptnn (ptnn_comm_init -> dlopen librccl,
ncclCommInitRank, in-place ncclAllGather on the handle's own stream) at world size 1 -- the only RCCL world a one-GPU box
allows -- in both exchange modes and under swap_rule 1, against the plain single-handle run.  No torch in this process."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import parity
    from parity import orc
    from ptnn_amd import _lib, ladder, philox
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
        ref = make(rule)
        ref.run(-1)
        ref.sync()
        want, want_log, want_stats = ref.traces(), ref.swap_log(), ref.swap_stats()
        ref.close()
        assert want_stats[0] > 0
        s = make(rule)
        s.comm_init(_lib.comm_unique_id(), 0, 1)
        s.comm_set_mode(mode)
        s.run(40)                                            # in chunks, like a caller that drains a trace ring
        s.run(-1)
        s.sync()
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
    main()
