"""The drop-in classes end to end on the GPU: same files, same return tuple as the reference's run_chains()."""
import json
import os

import numpy as np
import pytest

import parity
from parity import orc

pytestmark = pytest.mark.gpu

SUBDIRS = ["predictions", "posterior", "results", "surrogate", "surrogate/learnsurrogate_data", "posterior/pos_w",
           "posterior/pos_likelihood", "posterior/surg_likelihood", "posterior/accept_list"]


def _run(key, tmp_path, **kw):
    import ptnn_amd
    g = parity.golden(f"swap_trajectory_{key}.npz")
    d = parity.datasets()
    dname = str(g["dataset"])
    topo = [int(v) for v in g["topology"]]
    args = (bool(g["use_lg"]), float(g["lr"]), d[dname + "_train"], d[dname + "_test"], topo, int(g["R"]), int(g["maxtemp"]),
            int(g["NumSample"]), int(g["si"]))
    path = str(tmp_path)
    kw.setdefault("shared_noise", False)     # the F6 fixtures were made with one Philox tape PER chain (make_fixtures.py: gen_swap_trajectories)
    if int(g["task"]) == 0:
        from ptnn_amd.pt_timeseries_regression import ParallelTempering
        pt = ParallelTempering(*args, 0.5, path, seed=int(g["seed"]), **kw)
    else:
        from ptnn_amd.pt_classification import ParallelTempering
        pt = ParallelTempering(*args, path, seed=int(g["seed"]), **kw)
    for sdir in SUBDIRS:
        pt.make_directory(os.path.join(path, sdir))
    pt.initialize_chains(0.5)
    res = pt.run_chains(**_RUN_KW)
    return g, pt, res


_RUN_KW = {}


@pytest.mark.parametrize("key", ["reg", "reg_nophantom", "cls", "cls_nophantom", "reg_sunspot5"])
def test_run_chains_return_tuple_matches_reference(key, tmp_path):
    g, pt, res = _run(key, tmp_path)
    names = ["pos_w", "fx_train", "fx_test", "rmse_train", "rmse_test", "acc_train", "acc_test", "likelihood_vec",
             "swap_perc", "accept_vec", "accept"]
    assert len(res) == 11
    out = dict(zip(names, res))
    for nm in ("pos_w", "rmse_train", "rmse_test", "acc_train", "acc_test", "likelihood_vec", "accept_vec"):
        assert out[nm].shape == g["ret_" + nm].shape, nm
        assert out[nm].dtype == np.float64
    assert out["fx_train"].shape == tuple(g["ret_fx_train_shape"]) and not out["fx_train"].any()
    assert out["fx_test"].shape == tuple(g["ret_fx_test_shape"]) and not out["fx_test"].any()
    assert out["accept"] == 0.0
    assert pt.total_swap_proposals == int(g["total_swap_proposals"])
    assert pt.temperatures == list(g["temperatures"])
    # the reference started from float64 w0, the device from its fp32 rounding: decisions agree until a coin toss lands
    # inside the fp32 noise; up to there everything the reference returned must be reproduced
    same = out["accept_vec"] == g["ret_accept_vec"]
    if same.all():
        assert pt.num_swap == int(g["num_swap"]) and out["swap_perc"] == pytest.approx(float(g["swap_perc"]))
        np.testing.assert_allclose(out["pos_w"], g["ret_pos_w"], rtol=1e-4, atol=5e-5)
        np.testing.assert_allclose(out["rmse_train"], g["ret_rmse_train"], rtol=1e-4, atol=2e-2 if key.startswith("cls") else 1e-6)
        np.testing.assert_allclose(out["likelihood_vec"], g["ret_likelihood_vec"], rtol=1e-4, atol=5e-3)
        np.testing.assert_allclose(out["acc_train"], g["ret_acc_train"], atol=0.011)
    else:
        first = int(np.min(np.argmax(~same, axis=1)[~same.all(axis=1)]))
        assert first > 10, f"diverged from the reference after only {first} samples"


def test_result_files_layout(tmp_path):
    g, pt, res = _run("reg", tmp_path)
    want = json.load(open(os.path.join(parity.GOLDEN, "layout_tree.json")))
    got = {}
    for dp, dn, fn in os.walk(tmp_path):
        rel = os.path.relpath(dp, tmp_path)
        for d_ in dn:
            got[os.path.normpath(os.path.join(rel, d_)) + "/"] = None
        for f in fn:
            lines = open(os.path.join(dp, f)).read().splitlines()
            got[os.path.normpath(os.path.join(rel, f))] = dict(nlines=len(lines), ncols=len(lines[0].split()) if lines else 0,
                                                               first=lines[0] if lines else "")
    assert set(got) == set(want), set(got) ^ set(want)
    for k, v in want.items():
        if v is None:
            continue
        assert got[k]["nlines"] == v["nlines"] and got[k]["ncols"] == v["ncols"], k
        # same printf format: same number of characters per field on the first line (row 0 is fixed by Q7)
        if "accept.txt" not in k and "likelihood.txt" != k and k != "accept_list.txt":
            assert got[k]["first"] == v["first"], k
        else:
            assert [len(t.split(".")[-1]) for t in got[k]["first"].split()] == [len(t.split(".")[-1]) for t in v["first"].split()], k


def test_timings_and_no_files_mode(tmp_path):
    g, pt, res = _run("cls", tmp_path, write_files=False)
    assert not os.path.exists(os.path.join(tmp_path, "posterior/pos_w", f"chain_{pt.temperatures[0]}.txt"))
    assert pt.timings["samples_per_s"] > 0 and pt.timings["segment_launches"] >= 1


def test_zero_rounds_raises_like_the_reference(tmp_path):
    import ptnn_amd
    from ptnn_amd.pt_timeseries_regression import ParallelTempering
    d = parity.datasets()
    pt = ParallelTempering(False, 0.1, d["sunspot_train"], d["sunspot_test"], [4, 5, 1], 4, 2, 200, 1000, 0.5, str(tmp_path), seed=3)
    for sdir in SUBDIRS:
        pt.make_directory(os.path.join(tmp_path, sdir))
    pt.initialize_chains(0.5)
    with pytest.raises(ZeroDivisionError):          # swap_interval > NumSamples: no round, 0/0 (REG:769)
        pt.run_chains()


def _expected_driver_row(task, topo, train, test, R, maxtemp, NumSample, si, use_lg, lr, seed, burn_in, problem, col5, rmse_fmt):
    """Columns 0..13 of the row main() writes (REG:1004-1052 / CLS:1090-1138), computed from an ORACLE run of the same experiment
    on the same tape (the drop-in's defaults: shared noise tape, Philox stream-3 initial weights rounded to float32): the
    statistics are taken over what show_results returns, i.e. the per-chain files after burn-in, read back through their text
    format (REG:795-846)."""
    import ptnn_oracle_c as orc_c
    from ptnn_amd import _lib
    pt = orc.PTOracle(task, topo, train, test, R, maxtemp, NumSample, si, use_lg=use_lg, l_prob=0.5, lr=lr, seed=seed, shared_noise=True)
    w0 = np.stack([rep.w for rep in pt.replicas]).astype(np.float32).astype(np.float64)
    orc_c.adopt(pt, w0=w0)
    pt.run()
    S = pt.S
    b = int(S * burn_in)
    metric, best = ("rmse", np.amin) if task == orc.TASK_REG else ("acc", np.amax)
    stats = []
    for part in ("train", "test"):
        v = _lib.text_round(np.stack([getattr(rep, f"{metric}_{part}")[b:] for rep in pt.replicas]), rmse_fmt if metric == "rmse" else "%1.2f")
        stats += [np.mean(v), np.std(v), best(v)]
    accept_per = 100.0 * float(np.mean([rep.accept_list[S - 1] / S for rep in pt.replicas]))
    return [problem, NumSample, maxtemp, si, col5, lr, *stats, pt.swap_perc, accept_per], pt


def test_experiment_drivers(tmp_path):
    """drivers.run_regression / run_classification: run-dir numbering, result.txt, master_result_file.txt (REG:1044-1061) -- and the
    VALUES of the row: columns 0..13 against the row computed from an oracle run of the same experiment (column 14 is minutes)."""
    import ptnn_amd
    from ptnn_amd import drivers
    d = parity.datasets()
    base, db = str(tmp_path / "work") + "/", str(tmp_path / "db") + "/"
    os.makedirs(base); os.makedirs(db)
    for k in range(2):
        out = drivers.run_regression("Sunspot", d["sunspot_train"], d["sunspot_test"], hidden=5, NumSample=4000, num_chains=4,
                                     problemfolder=base, problemfolder_db=db, seed=5 + k, plots=(k == 0))
        assert out["run_name"] == f"Sunspot_{k}" and os.path.isdir(base + f"Sunspot_{k}/posterior/pos_w")
        row = open(out["path"] + "/result.txt").read().split()
        assert len(row) == 15 and row[0] == "2.0000" and row[1] == "4000.0000" and row[3] == "10.0000"
        assert all(len(t.split(".")[1]) == 4 for t in row)
        want, opt = _expected_driver_row(orc.TASK_REG, (4, 5, 1), d["sunspot_train"], d["sunspot_test"], 4, 2, 4000, 10, True, 0.1, 5 + k, 0.5,
                                         2, 0.5, "%1.8f")
        got = [float(t) for t in row[:14]]
        # what the row prints: 4 decimals.  The chains follow the oracle's decisions (1000 steps per chain: a coin flip inside the
        # fp32 error of log alpha is possible, not expected), so the RMSE statistics agree to the print resolution and the two
        # percentages exactly (they are ratios of decision counts)
        np.testing.assert_allclose(got[:6], want[:6], atol=5.1e-5)
        same_decisions = out["pt"].num_swap == opt.num_swap and int(round(got[13] * 1000 * 4 / 100)) == int(sum(rep.accept_list[-1] for rep in opt.replicas))
        np.testing.assert_allclose(got[6:12], want[6:12], atol=1.01e-4 if same_decisions else 5e-3, err_msg=f"RMSE statistics of the row, run {k}")
        np.testing.assert_allclose(got[12:14], want[12:14], atol=5.1e-5 if same_decisions else 1.0, err_msg=f"swap % / accept % of the row, run {k}")
        assert same_decisions, "the device took a decision the oracle did not within 4 x 1000 steps (possible, but look at it)"
    master = open(base + "master_result_file.txt").read().strip().splitlines()
    assert len(master) == 2 and master[0].split()[-1] == "Sunspot_0" and master[1].split()[-1] == "Sunspot_1"
    assert len(master[0].split()) == 16
    assert os.path.exists(db + "Sunspot_0/rmse_samples.pdf") and os.path.exists(base + "Sunspot_0/likelihood.pdf")
    out = drivers.run_classification("iris", d["iris_train"], d["iris_test"], NumSample=4000, num_chains=4, problemfolder=base,
                                     problemfolder_db=db, seed=7, plots=False)
    row = open(out["path_db"] + "/result.txt").read().split()
    assert len(row) == 15 and row[0] == "3.00" and row[2] == "10.00" and row[3] == "20.00"
    assert 0.0 <= float(row[6]) <= 100.0 and 0.0 <= float(row[12]) <= 100.0
    want, opt = _expected_driver_row(orc.TASK_CLS, (4, 12, 3), d["iris_train"], d["iris_test"], 4, 10, 4000, 20, False, 0.01, 7, 0.5, 3, 0.0, None)
    got = [float(t) for t in row[:14]]
    np.testing.assert_allclose(got[:6], want[:6], atol=5.1e-3)
    # CLS prints 2 decimals; accuracies are multiples of 100 / rows, so one data row predicted differently on some recorded steps
    # moves a mean by well under a point
    np.testing.assert_allclose(got[6:12], want[6:12], atol=0.5, err_msg="accuracy statistics of the row")
    np.testing.assert_allclose(got[12:14], want[12:14], atol=1.0, err_msg="swap % / accept % of the row")


def test_streaming_run_chains_equals_resident(tmp_path):
    a = _run("reg", tmp_path / "a")[2] if os.makedirs(tmp_path / "a") is None else None
    b = _run("reg", tmp_path / "b", trace_capacity=17)[2] if os.makedirs(tmp_path / "b") is None else None
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def _same_files(a, b):
    n = 0
    for root, _, files in os.walk(a):
        for f in files:
            pa = os.path.join(root, f)
            assert open(pa, "rb").read() == open(pa.replace(str(a), str(b)), "rb").read(), f
            n += 1
    return n


@pytest.mark.parametrize("key,chunks", [("reg", 8), ("cls", 8), ("reg_nophantom", 3), ("reg_sunspot5", 2)])
def test_overlapped_run_chains_equals_resident(key, chunks, tmp_path):
    """run_chains() by default cuts the run into `overlap_chunks` launches and lets the trace rows of each leave for the host (pinned
    images, second stream) and into the per-chain files (append mode) while the next launch samples.  Same return tuple, same bytes in
    every file, same swap statistics as the resident path (overlap_chunks=0: one launch, download and files after the last step)."""
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    a = _run(key, tmp_path / "a", overlap_chunks=0)
    b = _run(key, tmp_path / "b", overlap_chunks=chunks)
    assert a[1].timings["overlapped"] is False and b[1].timings["overlapped"] is True
    assert b[1].timings["launches_per_run"] > 1
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert (a[1].num_swap, a[1].total_swap_proposals, a[1].rounds) == (b[1].num_swap, b[1].total_swap_proposals, b[1].rounds)
    assert _same_files(tmp_path / "a", tmp_path / "b") >= 8 * int(a[0]["R"]) + 3


@pytest.mark.parametrize("dname,topo,kernel", [("sunspot", [4, 5, 1], "segment_pack_kernel"), ("mackey", [4, 10, 1], "segment_packm_kernel")])
def test_overlapped_run_at_the_headline_shape(dname, topo, kernel, tmp_path):
    """The same at a size where the pieces really overlap: Sunspot 4-5-1 (packed schedule on one CU) and Mackey-Glass 4-10-1 (packed
    over 4 CUs, its swap rounds inside every launch), 16 chains x 3000 samples, Langevin, shared noise; and further run_chains() on
    fresh objects in the same process (the images are per handle)."""
    from ptnn_amd.pt_timeseries_regression import ParallelTempering
    d = parity.datasets()
    res = {}
    for name, oc in (("a", 0), ("b", 8), ("c", 5)):
        path = str(tmp_path / name)
        pt = ParallelTempering(True, 0.1, d[dname + "_train"], d[dname + "_test"], topo, 16, 2, 16 * 3000, 100, 0.5, path, seed=11, overlap_chunks=oc)
        for sdir in SUBDIRS:
            pt.make_directory(os.path.join(path, sdir))
        pt.initialize_chains(0.5)
        assert kernel in pt._sampler.describe()["kernel"]
        res[name] = (pt.run_chains(), pt.num_swap, pt.timings)
    assert res["b"][2]["overlapped"] and res["b"][2]["launches_per_run"] == 8 and res["c"][2]["launches_per_run"] == 5
    for other in "bc":
        for x, y in zip(res["a"][0], res[other][0]):
            assert np.array_equal(np.asarray(x), np.asarray(y))
        assert res["a"][1] == res[other][1]
        assert _same_files(tmp_path / "a", tmp_path / other) == 8 * 16 + 4


@pytest.mark.parametrize("key", ["reg", "cls"])
def test_run_chains_checkpoint_and_resume(key, tmp_path):
    """a run that is cut off after max_steps and resumed from its checkpoint file (in a new object: new
    handle, same data) returns the same 11-tuple and writes the same files as the uninterrupted run."""
    global _RUN_KW
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    a = _run(key, tmp_path / "a")[2]
    ck = str(tmp_path / "ck.npz")
    try:
        _RUN_KW = dict(checkpoint_path=ck, checkpoint_every=7, max_steps=23)
        g, pt, res = _run(key, tmp_path / "b")
        assert res is None and os.path.exists(ck)
        _RUN_KW = dict(checkpoint_path=ck, checkpoint_every=9, resume_from=ck)
        b = _run(key, tmp_path / "b")[2]
    finally:
        _RUN_KW = {}
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    for root, _, files in os.walk(tmp_path / "a"):
        for f in files:
            pa = os.path.join(root, f)
            pb = pa.replace(str(tmp_path / "a"), str(tmp_path / "b"))
            assert open(pa, "rb").read() == open(pb, "rb").read(), f


@pytest.mark.parametrize("key,devices,exchange", [("reg", [0, 0], "auto"), ("reg", [0, 0, 0, 0], "boundary"), ("cls", [0, 0], "boundary"),
                                                  ("cls", [0, 0, 0, 0], "gather"), ("reg_nophantom", [0, 0], "boundary")])
def test_run_chains_on_a_sharded_ladder_equals_one_device(key, devices, exchange, tmp_path):
    """`ParallelTempering(..., devices=[...])`: run_chains() itself cuts the ladder into one block per listed device and the
    swap rounds exchange inside libptnn (REG:694-771 replaced end to end).  The box has one GPU, so it is listed several
    times: one handle and one host thread per block, host-staged ThreadTransport between them (RCCL refuses a device twice) --
    the same ptnn_run / comm_swap_round code the RCCL path runs.  Return tuple and every result file must equal the one-device
    run bit for bit."""
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    a = _run(key, tmp_path / "a")
    b = _run(key, tmp_path / "b", devices=devices, exchange=exchange)
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert (a[1].num_swap, a[1].total_swap_proposals) == (b[1].num_swap, b[1].total_swap_proposals)
    stats = b[1]._sampler.comm_stats()
    assert len(stats) == len(devices) and all(st["rounds"] == b[1].rounds for st in stats)
    want_mode = "gather" if exchange in ("auto", "gather") else "boundary"
    assert all(st["mode"] == want_mode for st in stats)
    assert sum(st["bytes_sent"] for st in stats) == sum(st["bytes_received"] for st in stats) > 0
    for root, _, files in os.walk(tmp_path / "a"):
        for f in files:
            pa = os.path.join(root, f)
            assert open(pa, "rb").read() == open(pa.replace(str(tmp_path / "a"), str(tmp_path / "b")), "rb").read(), f


def test_failed_rccl_bring_up_falls_back_to_the_host_transport(tmp_path, monkeypatch):
    """RCCL inside libptnn has never run on more than one GPU.  `transport=None / "auto"` try it, and a bring-up that fails (here
    injected by $PTNN_COMM_FAULT before RCCL is loaded: the box has one GPU, RCCL is never asked to put two ranks of one process
    on one device, and an ncclGetUniqueId that no ncclCommInitRank follows keeps the process from exiting) moves the whole group to the host-staged transport with a warning; the run equals the one-device run.
    `transport="rccl"` fails loudly instead, and without the injection a repeated device is refused before RCCL is touched."""
    from ptnn_amd import _lib
    for sub in "abcd":
        os.makedirs(tmp_path / sub)
    a = _run("reg", tmp_path / "a")
    with pytest.raises(ValueError, match="one distinct device per block"):
        _run("reg", tmp_path / "d", devices=[0, 0], exchange="boundary", transport="rccl")
    monkeypatch.setenv("PTNN_COMM_FAULT", "ncclGetUniqueId,ncclCommInitRank")
    with pytest.warns(RuntimeWarning, match="host-staged after an RCCL bring-up failure"):
        from ptnn_amd import distributed
        grp_probe = {}
        orig = distributed.LadderGroup.__init__

        def forced(self, devices, **kw):                    # the drop-in's "auto" on a repeated device picks host: force the RCCL attempt
            kw["transport"], kw["fallback"] = "rccl", True
            orig(self, devices, **kw)
            grp_probe["g"] = self
        monkeypatch.setattr(distributed.LadderGroup, "__init__", forced)
        b = _run("reg", tmp_path / "b", devices=[0, 0], exchange="boundary")
        monkeypatch.setattr(distributed.LadderGroup, "__init__", orig)
    assert grp_probe["g"].transport == "host" and "ncclGetUniqueId" in grp_probe["g"].transport_note
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    with pytest.raises(_lib.PtnnError, match="ncclGetUniqueId"):
        _run("reg", tmp_path / "c", devices=[0, 0], exchange="boundary", transport="rccl")


def test_sharded_run_chains_checkpoint_and_resume(tmp_path):
    """Checkpoint / resume of a sharded ladder: one blob per block behind an index, resumed in a new object."""
    global _RUN_KW
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    a = _run("reg", tmp_path / "a")[2]
    ck = str(tmp_path / "ck.npz")
    try:
        _RUN_KW = dict(checkpoint_path=ck, checkpoint_every=7, max_steps=23)
        g, pt, res = _run("reg", tmp_path / "b", devices=[0, 0], exchange="boundary")
        assert res is None and os.path.exists(ck)
        _RUN_KW = dict(checkpoint_path=ck, checkpoint_every=9, resume_from=ck)
        b = _run("reg", tmp_path / "b", devices=[0, 0], exchange="boundary")[2]
    finally:
        _RUN_KW = {}
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def test_sharded_handle_without_communicator_is_refused():
    import ptnn_amd  # noqa: F401
    from ptnn_amd import _lib
    d = parity.datasets()
    s = parity.make_sampler(0, (4, 5, 1), d["sunspot_train"], d["sunspot_test"], R_local=2, R_global=4, first=2, S=20, si=5,
                            use_lg=False, lr=0.1, seed=1)
    s.set_state(np.zeros((2, 31), np.float32), np.ones(2, np.float32))
    with pytest.raises(_lib.PtnnError, match="communicator"):
        s.run(-1)
    with pytest.raises(_lib.PtnnError, match="equal contiguous blocks"):
        s.comm_init_host(0, 2, lambda b: None, lambda m: None)      # this handle is rank 1 of 2, not rank 0
    s.close()


def test_bench_run_bare_starts_its_ranks_and_reports_what_the_communicator_saw(tmp_path):
    """`python3 bench.py --gpus 2` WITHOUT a launcher (what a driver that runs the N = 1 line as `python3 bench.py --gpus 1` would
    do for N > 1): the script starts its own two ranks before anything touches the GPU, relays rank 0's one JSON line and ends
    with status 0.  On this one-GPU box both ranks share GPU 0, so RCCL is not attempted (it refuses two ranks on a device) and the
    line NAMES the host-staged fall-back; the ranks, devices, bytes and rounds in `comm` are what the communicators report."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(parity.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["config"]["replicas"] == 128
    c = out["comm"]
    assert c["nranks_seen"] == 2 and c["transport"] == "host" and c["device_ids"] == [0, 0] and c["launcher"] == "self"
    assert "RCCL needs one device per rank" in out["config"]["transport_note"]
    assert c["rounds"] == 100 * (1 + 2) and sum(c["bytes_sent"]) == sum(c["bytes_received"]) > 0      # cumulative: 1 warm-up + 2 timed runs
    # strong scaling: the workload's 64 replicas in total, 32 per rank
    r = subprocess.run([sys.executable, os.path.join(parity.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--scaling", "strong"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["replicas"] == 64 and out["config"]["replicas_per_gpu"] == 32 and out["comm"]["nranks_seen"] == 2


def test_bench_line_survives_an_rccl_bring_up_that_fails_half_way(tmp_path):
    """bench.py's own ranks: rank 0 really creates the unique id (RCCL loaded, bootstrap listener up), then ncclCommInitRank fails
    ($PTNN_COMM_FAULT).  Every rank moves to the host-staged transport, the line says so, and the process EXITS with status 0 (it
    arms the hard exit: a half-initialised RCCL kept such a process alive before)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PTNN_COMM_FAULT="ncclCommInitRank", PTNN_COMM_TIMEOUT_S="60", MASTER_PORT="29641")
    r = subprocess.run([sys.executable, os.path.join(parity.ROOT, "bench.py"), "--gpus", "1", "--force-comm", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=400, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["comm"]["transport"] == "host" and out["comm"]["nranks_seen"] == 1
    assert "fallback after an RCCL failure" in out["config"]["transport_note"] and "ncclCommInitRank" in out["config"]["transport_note"]
    assert out["value"] > 1e6


_EXIT_CHILD = """
import os, sys, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
import ptnn_amd, parity
from ptnn_amd import distributed, _lib
from ptnn_amd.pt_timeseries_regression import ParallelTempering
orig = distributed.LadderGroup.__init__
def forced(self, devices, **kw):
    kw["transport"], kw["fallback"] = "rccl", True
    orig(self, devices, **kw)
    sys.stdout.write("TRANSPORT %s | %s\\n" % (self.transport, self.transport_note)); sys.stdout.flush()
distributed.LadderGroup.__init__ = forced
d = parity.datasets()
pt = ParallelTempering(True, 0.1, d["sunspot_train"], d["sunspot_test"], [4, 5, 1], 4, 2, 400, 10, 0.5, {path!r}, seed=3, devices=[0, 0], write_files=False)
warnings.simplefilter("ignore")
pt.initialize_chains(0.5)
res = pt.run_chains()
print("DONE swap_perc %.3f armed %s" % (res[8], distributed._hard_exit_armed), flush=True)
"""


@pytest.mark.parametrize("probe", ["1", "0"])
def test_process_exits_after_an_rccl_bring_up_that_fails_half_way(tmp_path, probe):
    """The case the fall-back exists for, end to end in a process of its own: ncclGetUniqueId REALLY runs (RCCL loaded, its bootstrap
    listener started), then ncclCommInitRank fails ($PTNN_COMM_FAULT=ncclCommInitRank).  With the probe (default) that happens in a
    throw-away child and the drop-in's process never touches RCCL; with $PTNN_RCCL_PROBE=0 it happens in the drop-in's own process,
    which then leaves through the hard exit LadderGroup arms.  Either way run_chains() completes on the host-staged transport and
    THE PROCESS EXITS (a half-initialised RCCL kept it alive before)."""
    import subprocess
    import sys
    env = dict(os.environ, PTNN_COMM_FAULT="ncclCommInitRank", PTNN_RCCL_PROBE=probe, PTNN_COMM_TIMEOUT_S="60")
    code = _EXIT_CHILD.format(root=parity.ROOT, tests=os.path.join(parity.ROOT, "tests"), path=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)      # TimeoutExpired = it did not exit
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "TRANSPORT host | host-staged after an RCCL bring-up failure" in r.stdout and "ncclCommInitRank" in r.stdout, r.stdout
    assert ("RCCL was not touched in this process" in r.stdout) == (probe == "1")
    assert f"armed {probe == '0'}" in r.stdout, r.stdout
