"""Two PROCESSES sharing the one GPU of the box, each owning half of a 16-temperature ladder through the C ABI: ptnn_run with
the host-staged transport (ptnn_comm_init_host) over gloo -- RCCL refuses two ranks on one device.  Real device buffers, a
real cross-process exchange, the library's own routing.  Gathered exchange, boundary exchange and the gathered exchange under
swap_rule 1 must reproduce the single-handle run bit for bit.  Run by test_sharded_ladder_two_ranks_on_one_gpu."""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

TOPO, R, S, SI, SEED = (4, 5, 1), 16, 8 * 12 + 3, 12, 77
CASES = (("gather", 0), ("boundary", 0), ("gather", 1))


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
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ptnn_amd import _lib, distributed as dm
    Rl = R // world
    for mode, rule in CASES:
        s = make(rule, Rl, rank * Rl)
        s.comm_init_host(rank, world, *dm.gloo_transport(dist))
        s.comm_set_mode({"gather": _lib.XCHG_GATHER, "boundary": _lib.XCHG_BOUNDARY}[mode])
        s.run(-1)
        s.sync()
        tr = s.traces()
        st = s.comm_stats()
        np.savez(os.path.join(outdir, f"{mode}_{rule}_rank{rank}.npz"), log=s.swap_log(), stats=np.array(s.swap_stats()),
                 moved=np.array([st["bytes_sent"], st["bytes_received"], st["rounds"]]), **tr)
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
    P = TOPO[0] * TOPO[1] + TOPO[1] * TOPO[2] + TOPO[1] + TOPO[2]
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
        for mode, rule in CASES:
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
                sent, recvd, rounds = (int(v) for v in z["moved"])
                assert rounds == want_stats[2] and sent > 0
                if mode == "boundary":
                    # L (4 Rl bytes per round) + at most one (w, eta) row each way across the one boundary
                    PS = (P + 1 + 3) & ~3
                    assert sent <= rounds * (4 * Rl + 4 * PS) and recvd <= rounds * (4 * Rl + 4 * PS), (sent, recvd)
            print("OK", mode, "rule", rule, flush=True)


if __name__ == "__main__":
    main()
