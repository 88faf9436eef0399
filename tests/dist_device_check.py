"""Child process of test_sharded_ladder_driver_on_device_matches_plain_run: world-size-1 rehearsal of distributed.py on the GPU."""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import parity
    from parity import orc
    from ptnn_amd import distributed as dm, ladder, philox
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
    try:
        for mode, rule in (("gather", 0), ("p2p", 0), ("gather", 1)):
            ref = make(rule)
            ref.run(-1)
            ref.sync()
            want, want_log, want_stats = ref.traces(), ref.swap_log(), ref.swap_stats()
            ref.close()
            assert want_stats[0] > 0
            s = make(rule)
            lad = dm.ShardedLadder(dm.DeviceShard(s, 0), 0, 1, dist, mode=mode)
            lad.run_intervals(None)
            s.sync()
            got = s.traces()
            for k in want:
                assert np.array_equal(got[k], want[k]), (mode, k)
            assert np.array_equal(s.swap_log(), want_log), mode
            assert s.swap_stats() == want_stats, mode
            s.close()
            print("OK", mode, "rule", rule, flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
