"""Where a swap round of the sharded-ladder driver spends host time (world size 1 rehearsal on one GPU)."""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import bench
from ptnn_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
train, test, _ = bench.load_sunspot()
S = 111 * 100 + 2
s = bench.make_sampler(train, test, 64, 64, 0, S, 0, True, 0, 0, 0)
sh = D.DeviceShard(s, 0)
lad = D.ShardedLadder(sh, 0, 1, dist)
T = {}
def timed(name, f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
lad.run_intervals(10)
n = 100
t0 = time.perf_counter()
for _ in range(n):
    ho = timed("run_segment (launch)", sh.run_segment)
    timed("sync (kernel)", sh.sync)
    L = timed("L view", sh.L_tensor, False)
    timed("all_gather (in place)", dist.all_gather_into_tensor, L, L[0:64])
    timed("fence", sh.fence_collectives)
    src = timed("cascade (+D2H)", sh.swap_cascade, False)
    rs = timed("route", D.route, src, 0, 1, 64)
    timed("apply", sh.swap_apply, src, False)
tot = time.perf_counter() - t0
print(f"stream_ordered={sh.stream_ordered}: {tot/n*1e3:.3f} ms per interval")
for k, v in T.items():
    print(f"   {k:24s} {v/n*1e6:8.1f} us")
dist.destroy_process_group(); s.close()
