import sys, os, time, tempfile, shutil, numpy as np
R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import parity, ptnn_amd
from ptnn_amd.pt_timeseries_regression import ParallelTempering
d = parity.datasets()
SUB = ["predictions","posterior","results","surrogate","surrogate/learnsurrogate_data","posterior/pos_w","posterior/pos_likelihood","posterior/surg_likelihood","posterior/accept_list"]
for R, S in ((64, 10000),):
    for io in (16, 64):
        tmp = tempfile.mkdtemp(dir="/tmp")
        pt = ParallelTempering(True, 0.1, d["sunspot_train"], d["sunspot_test"], [4,5,1], R, 2, R*S, 100, 0.5, tmp, seed=1, io_threads=io)
        for s_ in SUB: pt.make_directory(os.path.join(tmp, s_))
        t0=time.perf_counter(); pt.initialize_chains(0.5); t1=time.perf_counter()
        res = pt.run_chains(); t2=time.perf_counter()
        sz = sum(os.path.getsize(os.path.join(dp,f)) for dp,_,fs in os.walk(tmp) for f in fs)
        print(f"R={R} S={S} io_threads={io}: init {t1-t0:.2f}s run_chains {t2-t1:.2f}s  files {sz/1e6:.0f} MB  swap% {res[8]:.1f}", {k: round(v,3) if isinstance(v,float) else v for k,v in pt.timings.items()}, flush=True)
        shutil.rmtree(tmp)
