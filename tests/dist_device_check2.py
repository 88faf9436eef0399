"""Two ranks sharing ONE GPU over gloo: the sharded-ladder driver on real device buffers with a real cross-process exchange
(RCCL refuses two ranks on one device, gloo does not care).  Run by test_sharded_ladder_two_ranks_on_one_gpu."""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

TOPO, R, S, SI, SEED = (4, 5, 1), 16, 8 * 12 + 3, 12, 77


def make(rule, Rl, first):
    import parity
    from parity import orc
    from ptnn_amd import ladder, philox
    d = parity.datasets()
    Pw = TOPO[0] * TOPO[1] + TOPO[1] * TOPO[2] + TOPO[1] + TOPO[2]
    s = parity.make_sampler(orc.TASK_REG, TOPO, d["sunspot_train"], d["sunspot_test"], R_local=Rl, R_global=R, first=first, S=S,
                            si=SI, use_lg=True, lr=0.1, seed=SEED, swap_rule=rule)
    T = ladder.temperatures(R, 2)
    s.set_state(np.stack([philox.initial_weights(SEED, r, Pw) for r in range(first, first + Rl)]), T[first:first + Rl])
    if rule:
        s.set_ladder(T)
    return s


def worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ptnn_amd import distributed as dm
    Rl = R // world
    for mode, rule in (("gather", 0), ("p2p", 0), ("gather", 1)):
        s = make(rule, Rl, rank * Rl)
        lad = dm.ShardedLadder(dm.DeviceShard(s, 0), rank, world, dist, mode=mode)
        lad.run_intervals(None)
        s.sync()
        tr = s.traces()
        np.savez(os.path.join(outdir, f"{mode}_{rule}_rank{rank}.npz"), log=s.swap_log(), stats=np.array(s.swap_stats()), **tr)
        s.close()
        dist.barrier()
    dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    world = 2
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
        for mode, rule in (("gather", 0), ("p2p", 0), ("gather", 1)):
            ref = make(rule, R, 0)
            ref.run(-1)
            ref.sync()
            want, want_log, want_stats = ref.traces(), ref.swap_log(), ref.swap_stats()
            ref.close()
            Rl = R // world
            for rank in range(world):
                z = np.load(os.path.join(out, f"{mode}_{rule}_rank{rank}.npz"))
                assert np.array_equal(z["log"], want_log), (mode, rule, "swap log")
                assert tuple(int(v) for v in z["stats"]) == tuple(want_stats), (mode, rule, "stats")
                for k in want:
                    assert np.array_equal(z[k], want[k][rank * Rl:(rank + 1) * Rl]), (mode, rule, rank, k)
            print("OK", mode, "rule", rule, flush=True)


if __name__ == "__main__":
    main()
