"""run_chains() end to end for the headline workload by number of launches of the overlapped path.   e2e_chunks.py [chunks ...]"""
import os, sys, time, shutil, tempfile
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import bench
import ptnn_amd
from ptnn_amd.pt_timeseries_regression import ParallelTempering
wl = bench.WORKLOADS["sunspot64"]
train, test, _ = bench.load_data(wl["data"])
R, S, si = wl["R"], wl["S"], wl["si"]
for oc in [int(v) for v in sys.argv[1:]] or [0, 4, 8, 16, 32]:
    best = None
    for rep in range(3):
        tmp = tempfile.mkdtemp(prefix="ptnn_e2e_")
        pt = ParallelTempering(True, wl["lr"], train, test, list(wl["topo"]), R, wl["maxtemp"], R * S, si, 0.5, tmp, seed=bench.SEED, overlap_chunks=oc)
        for sub in ("predictions", "posterior", "posterior/pos_w", "posterior/pos_likelihood", "posterior/accept_list"):
            pt.make_directory(os.path.join(tmp, sub))
        pt.initialize_chains(0.5)
        t0 = time.perf_counter(); pt.run_chains(); dt = time.perf_counter() - t0
        shutil.rmtree(tmp, ignore_errors=True)
        tm = pt.timings
        if best is None or dt < best[0]:
            best = (dt, tm)
        pt._sampler.close()
    dt, tm = best
    print(f"overlap_chunks {oc:2d}: run_chains {dt*1e3:6.1f} ms (best of 3); sampling {tm['sampling_s']*1e3:.1f} fetch {tm['fetch_s']*1e3:.1f} files+results {tm['files_and_results_s']*1e3:.1f} "
          f"(show_results {tm['show_results_s']*1e3:.1f}, drain {tm['files_drain_s']*1e3:.1f}) launches {tm.get('launches_per_run', 1)}", flush=True)
