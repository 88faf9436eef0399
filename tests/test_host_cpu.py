"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/ptnn.h declares, the host-only
helpers agree with numpy, the host logic mirrors the reference's observable behaviour, and the product path refuses to
run without a device instead of falling back to anything."""
import json
import os
import re

import numpy as np
import pytest

import ptnn_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pt():
    import __graft_entry__
    __graft_entry__.build()
    import ptnn_amd
    return ptnn_amd


def test_header_and_library_agree(pt):
    hdr = open(os.path.join(ROOT, "include", "ptnn.h")).read()
    declared = set(re.findall(r"\b(ptnn_[a-z_A-Z0-9]+)\s*\(", hdr))
    from ptnn_amd import _lib
    lib = pt.load_library()
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ptnn_abi_version() == 4


def test_supports_table(pt):
    lib = pt.load_library()
    for task, I, H, O in [(0, 4, 5, 1), (0, 4, 10, 1), (0, 5, 5, 1), (1, 4, 12, 3), (1, 34, 50, 2), (1, 9, 12, 2),
                          (1, 16, 30, 10), (1, 11, 50, 10), (1, 20, 50, 2), (1, 6, 25, 18), (0, 32, 64, 1)]:
        assert lib.ptnn_supports(task, I, H, O) == 1, (task, I, H, O)
    assert lib.ptnn_supports(0, 32, 512, 1) == 1 and lib.ptnn_supports(0, 32, 513, 1) == 0
    assert lib.ptnn_supports(0, 3, 5, 1) == 0


def test_no_device_no_fallback(pt):
    """Without a GPU the compute entry points must fail loudly (this container has none)."""
    from ptnn_amd import _lib
    from ptnn_amd.pt_timeseries_regression import ParallelTempering
    try:
        import ctypes
        ctypes.CDLL("libamdhip64.so")
    except OSError:
        pytest.skip("no HIP runtime")
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "datasets.npz")))
    p = ParallelTempering(False, 0.1, d["sunspot_train"], d["sunspot_test"], [4, 5, 1], 4, 2, 400, 10, 0.5, "/tmp", seed=1)
    with pytest.raises(_lib.PtnnError):
        p.initialize_chains(0.5)


@pytest.mark.parametrize("fmt", ["%.18e", "%1.8f", "%1.2f", "%1.4f", "%1.5f"])
def test_savetxt_matches_numpy(pt, tmp_path, fmt):
    from ptnn_amd import _lib
    rng = np.random.default_rng(3)
    for arr in (rng.normal(size=(57, 31)) * 10.0 ** rng.integers(-8, 8, size=(57, 31)), rng.normal(size=13),
                np.array([0.0, -0.0, 1.0, -100.0, 1e-300, 123456789.125, 0.005, 0.015, 2.675]), np.ones((3, 1))):
        a, b = tmp_path / "a.txt", tmp_path / "b.txt"
        np.savetxt(a, arr, fmt=fmt)
        _lib.savetxt(str(b), arr, fmt)
        assert a.read_bytes() == b.read_bytes()
    with pytest.raises(_lib.PtnnError):
        _lib.savetxt(str(tmp_path / "c.txt"), np.ones(3), "%s")


def test_host_philox_is_the_oracle_tape(pt):
    from ptnn_amd import philox
    for seed in (0, 1, 0xDEADBEEFCAFEF00D):
        t = orc.PhiloxTape(seed)
        for rep in (0, 3, 1023):
            for n in (31, 99, 1852):
                assert (philox.initial_weights(seed, rep, n) == t.w_init(rep, n)).all()
    x = philox.philox4x32(0, 0, 0, 0, 0)
    assert [int(v) for v in x] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_ladder_and_errors(pt, golden_dir):
    from ptnn_amd import ladder
    for c in json.load(open(os.path.join(golden_dir, "ladder.json"))):
        T = ladder.temperatures(c["R"], c["Tmax"])
        assert [float(t).hex() for t in T] == c["T"] and [str(t) for t in T] == c["s"]
    with pytest.raises(ValueError):
        ladder.default_beta_ladder(2, ntemps=4, Tmax=1)
    with pytest.raises(ValueError):
        ladder.default_beta_ladder(2, ntemps=None, Tmax=None)
    with pytest.raises(TypeError):
        ladder.default_beta_ladder(2, ntemps=4, Tmax=2.5)      # range(maxtemp) in the reference (REG:576)
    with pytest.raises(ZeroDivisionError):
        ladder.default_beta_ladder(2, ntemps=1, Tmax=2)


def test_host_class_surface(pt):
    from ptnn_amd.pt_classification import ParallelTempering as CLS
    from ptnn_amd.pt_timeseries_regression import ParallelTempering as REG
    import inspect
    reg_args = list(inspect.signature(REG.__init__).parameters)[1:12]
    assert reg_args == ["use_langevin_gradients", "learn_rate", "traindata", "testdata", "topology", "num_chains",
                        "maxtemp", "NumSample", "swap_interval", "langevin_prob", "path"]
    cls_args = list(inspect.signature(CLS.__init__).parameters)[1:11]
    assert cls_args == reg_args[:9] + ["path"]
    z = np.zeros((3, 5))
    p = REG(True, 0.1, z, z, [4, 5, 1], 10, 2, 100000, 100, 0.5, "/tmp", seed=7)
    assert p.NumSamples == 10000 and p.num_param == 31 and p.num_swap == 0 and p.total_swap_proposals == 0
    assert p._pt_switch_step() == 6000
    assert REG(True, 0.1, z, z, [4, 5, 1], 7, 2, 399, 100, 0.5, "/tmp")._pt_switch_step() == -1     # S = 57
    p.assign_temperatures()
    assert p.temperatures == orc.temperature_ladder(10, 2)
    c = CLS(False, 0.01, z, z, [4, 12, 3], 10, 10, 50000, 100, "/tmp")
    assert c.langevin_prob == 0.5 and c.num_param == 99 and c.task == 1 and c.rmse_fmt == '%1.2f'
    for meth in ("make_directory", "initialize_chains", "run_chains", "show_results", "assign_temperatures",
                 "default_beta_ladder"):
        assert callable(getattr(p, meth))


def test_text_round_is_the_file_round_trip(pt, tmp_path):
    from ptnn_amd.parallel_tempering import _text_round
    a = np.random.default_rng(0).normal(size=(4, 5000)) * 3
    a[0, :9] = [0.005, 0.015, 0.025, 2.675, -0.125, 1e-9, -1e-9, 123456.789, 0.0]
    for fmt in ("%1.8f", "%1.2f", "%1.4f"):
        f = tmp_path / "x.txt"
        np.savetxt(f, a, fmt=fmt)
        assert (np.loadtxt(f) == _text_round(a, fmt)).all()


def test_takens_embedding_reproduces_shipped_files(pt, datasets):
    from ptnn_amd import drivers
    for name in ("sunspot", "mackey", "lazer"):
        tr, te = drivers.takens_embedding(datasets[name + "_scaled"], window=5, stride=2)
        assert tr.shape == datasets[name + "_train"].shape and te.shape == datasets[name + "_test"].shape, name
        np.testing.assert_allclose(tr, datasets[name + "_train"], atol=1e-12)
        np.testing.assert_allclose(te, datasets[name + "_test"], atol=1e-12)


def test_split_and_normalise(pt):
    from ptnn_amd import drivers
    rng = np.random.default_rng(0)
    f = rng.normal(3, 2, (150, 4))
    c = rng.integers(0, 3, (150, 1)).astype(float)
    tr, te = drivers.split_and_normalise(f, c, 4, rng=np.random.default_rng(1))
    assert tr.shape == (105, 5) and te.shape == (45, 5)
    allx = np.vstack([tr, te])[:, :4]
    np.testing.assert_allclose(allx.mean(0), 0, atol=1e-12)
    np.testing.assert_allclose(allx.std(0), 1, atol=1e-12)


def test_create_validates_before_touching_the_gpu(pt):
    """Every malformed configuration is refused with a message (negative code + ptnn_last_error), no device needed."""
    import ctypes as C
    from ptnn_amd import _lib
    lib = pt.load_library()

    def create(**over):
        cfg = _lib.Config()
        base = dict(struct_bytes=C.sizeof(_lib.Config), device_id=0, task=0, n_in=4, n_hidden=5, n_out=1, n_replicas_local=4,
                    n_replicas_global=4, first_global_replica=0, n_samples=100, swap_interval=10, pt_switch_step=60,
                    use_langevin=1, l_prob=0.5, learn_rate=0.1, step_w=0.025, step_eta=0.2, sigma_squared=25.0, seed=1)
        base.update(over)
        for k, v in base.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.ptnn_create(C.byref(cfg), C.byref(h))
        msg = lib.ptnn_last_error().decode()
        if rc == 0:
            lib.ptnn_destroy(h)
        return rc, msg

    for over, needle in [(dict(struct_bytes=8), "size mismatch"), (dict(task=7), "unknown task"), (dict(n_out=2), "n_out == 1"),
                         (dict(n_in=7), "no gfx950 kernel compiled"), (dict(n_hidden=513), "n_hidden=513"),
                         (dict(n_replicas_global=1, n_replicas_local=1), "replica partition"),
                         (dict(first_global_replica=2), "replica partition"), (dict(n_samples=1), "n_samples"),
                         (dict(swap_interval=0), "swap_interval"), (dict(swap_rule=3), "swap_rule"),
                         (dict(n_in=0), "bad topology")]:
        rc, msg = create(**over)
        assert rc < 0 and needle in msg, (over, rc, msg)
    assert lib.ptnn_create(None, None) < 0
    assert lib.ptnn_destroy(None) == 0
    assert lib.ptnn_sync(None) < 0 and lib.ptnn_steps_done(None) == -1


REF_REG = "/root/reference/multicore-pt-regression"
REF_CLS = "/root/reference/multicore-pt-classification"


@pytest.mark.skipif(not os.path.isdir(REF_CLS), reason="the reference's data directories are only present in the build container")
def test_problem_loaders_follow_the_reference_rules():
    """drivers.load_*_problem: the loading rules of REG:881-917 / CLS:909-1012 as tables.  Checked against the
    fixtures the reference-importing script wrote (same files, same rules) and against properties of the rules."""
    import ptnn_amd
    from ptnn_amd import drivers
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datasets.npz"))
    for name, key in (("Sunspot", "sunspot"), ("Mackey", "mackey"), ("Lazer", "lazer")):
        tr, te = drivers.load_regression_problem(name, REF_REG)
        assert np.array_equal(tr, d[key + "_train"]) and np.array_equal(te, d[key + "_test"])
    tr, te, topo = drivers.load_classification_problem("Ionosphere", REF_CLS)
    assert topo == (34, 50, 2) and np.array_equal(tr, d["ions_train"]) and np.array_equal(te, d["ions_test"])
    tr, te, topo = drivers.load_classification_problem("Cancer", REF_CLS)
    assert topo == (9, 12, 2) and np.array_equal(tr, d["cancer_train"]) and np.array_equal(te, d["cancer_test"])
    tr, te, topo = drivers.load_classification_problem("iris", REF_CLS, rng=np.random.default_rng(2024))
    assert topo == (4, 12, 3) and np.array_equal(tr, d["iris_train"]) and np.array_equal(te, d["iris_test"])
    # wine: header row dropped, 11 z-scored features, quality label untouched, 70/30 split
    tr, te, topo = drivers.load_classification_problem("winequality-red", REF_CLS, rng=np.random.default_rng(1))
    assert topo == (11, 50, 10) and tr.shape[1] == 12 and tr.shape[0] == int(0.7 * (tr.shape[0] + te.shape[0]))
    both = np.vstack([tr, te])
    assert np.allclose(both[:, :11].mean(axis=0), 0, atol=1e-9) and np.allclose(both[:, :11].std(axis=0), 1)
    assert set(np.unique(both[:, 11])) <= set(range(10))
    # PenDigit: train and test z-scored with their own moments (CLS:975-981), labels in the last column
    tr, te, topo = drivers.load_classification_problem("PenDigit", REF_CLS)
    assert topo == (16, 30, 10) and tr.shape[1] == 17
    for a in (tr, te):
        assert np.allclose(a[:, :16].mean(axis=0), 0, atol=1e-9) and np.allclose(a[:, :16].std(axis=0), 1)
        assert set(np.unique(a[:, 16])) <= set(range(10))
    with pytest.raises(KeyError):
        drivers.load_classification_problem("mnist", REF_CLS)


def test_bench_starts_its_own_ranks_when_run_bare(capfd):
    """`python3 bench.py --gpus N` without a launcher (WORLD_SIZE unset): the plan is N fresh processes of the same script with the
    same arguments and the launcher's environment variables; rank 0's stdout is relayed as is, the others' to stderr, and the
    exit status is the worst child's.  (The GPU side of it runs on the box: tests/test_gpu_dropin.py.)"""
    import sys
    import bench
    plans = bench.launch_plan(["--gpus", "4", "--steps", "3", "--workload", "iris16"], 4, master_port=29999, environ={"FOO": "1", "WORLD_SIZE": "9"},
                              python="py", script="bench.py")
    assert len(plans) == 4
    for r, (cmd, env) in enumerate(plans):
        assert cmd == ["py", "bench.py", "--gpus", "4", "--steps", "3", "--workload", "iris16"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == (str(r), str(r), "4", "127.0.0.1", "29999")
        assert env["FOO"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PTNN_BENCH_SELF_LAUNCHED"] == "1"
    auto = bench.launch_plan([], 2, environ={})
    assert auto[0][1]["MASTER_PORT"] == auto[1][1]["MASTER_PORT"] and 1024 < int(auto[0][1]["MASTER_PORT"]) < 65536
    child = ("import os, sys; r = int(os.environ['RANK']); print('[lib] chatter of rank', r); "
             "print('{\"line of rank\": %d, \"of\": %s}' % (r, os.environ['WORLD_SIZE'])); sys.exit(3 if r == 1 else 0)")
    fake = [([sys.executable, "-c", child], dict(os.environ, RANK=str(r), WORLD_SIZE="3")) for r in range(3)]
    rc = bench.self_launch([], 3, plans=fake)
    out, err = capfd.readouterr()
    assert rc == 3 and out == '{"line of rank": 0, "of": 3}\n'         # rank 0's JSON line, nothing else
    assert '[rank 1] {"line of rank": 1, "of": 3}' in err and '[rank 2] {"line of rank": 2' in err and "[rank 0] [lib] chatter of rank 0" in err


@pytest.mark.parametrize("fmt", ["%.18e", "%1.8f", "%1.2f", "%1.4f", "%1.5f", "%.6e", "%10.3f", "%+.3e", "%.0f", "%1.18f"])
def test_fast_text_path_is_numpy_byte_for_byte(pt, tmp_path, fmt):
    """The result files are formatted by exact integer arithmetic instead of printf (csrc/ptnn_text.hpp): the bytes must be
    np.savetxt's for float32 traces (also strided views, repeated rows, append mode) and for arbitrary doubles -- ties (decimal
    halves that ARE binary fractions round to even), carries (9.99..5 -> 1.0e+01), zeros of both signs, denormals, huge and tiny
    magnitudes (those take the printf fall-back), non-finite values -- and the read-back value must be np.loadtxt's."""
    from ptnn_amd import _lib
    rng = np.random.default_rng(11)
    mags = 10.0 ** rng.integers(-12, 12, size=(400, 31))
    f32 = (rng.normal(size=(400, 31)) * mags).astype(np.float32)
    f32[5:9] = f32[4]                                        # repeated rows (rejected MH steps)
    f32[100] = [0.0, -0.0, 0.125, 0.375, 2.5, 3.5, -0.5, 0.5, 1.5, 9.5, 99.5, 0.001953125, 999999.5, 9.9999995, 99999.996, 1e-45, 3.4e38,
                -3.4e38, 1.17549435e-38, 0.1, 0.2, 0.3, 1.0, -1.0, 10.0, 100.0, 1e10, 1e-10, 123456.789, 7.0, 0.999999]
    special = np.array([0.005, 0.015, 0.025, 2.675, 1e300, -1e300, 5e-324, 2.2250738585072014e-308, np.nan, np.inf, -np.inf, 0.5, 1.5, 2.5,
                        9.9999999999999995e22, 999999999999999868928.0, 0.99999999999999989, 9.995, 9.9995, 99.99995, 1e22, 1e23, 123456789.125])
    f64 = np.concatenate([rng.normal(size=4000) * 10.0 ** rng.integers(-25, 25, size=4000), special, f32[100].astype(np.float64)])
    for arr in (f32, f32[:, :5], f32[::3, 2:9], f32[:, 0], f64, f64.reshape(-1, 2)[:1000]):
        a, b = tmp_path / "a.txt", tmp_path / "b.txt"
        np.savetxt(a, arr, fmt=fmt)
        _lib.savetxt(str(b), arr, fmt)
        assert a.read_bytes() == b.read_bytes(), (fmt, arr.dtype, arr.shape)
    # append mode = one file written in windows
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    np.savetxt(a, f32, fmt=fmt)
    _lib.savetxt(str(b), f32[:150], fmt)
    _lib.savetxt(str(b), f32[150:151], fmt, append=True)
    _lib.savetxt(str(b), f32[151:], fmt, append=True)
    assert a.read_bytes() == b.read_bytes()
    # a window of many files in one call (what run_chains() queues per launch of an overlapped run), in pieces
    jobs = [(f32, "m"), (f32[:, :5], "s5"), (f32[::3, 2:9], "v"), (f32[:, 0], "c"), (f32[:, 3:4], "c1")]
    for lo, hi in ((0, 77), (77, 78), (78, None)):
        _lib.savetxt_batch([(str(tmp_path / f"batch_{n}.txt"), arr[lo:hi], fmt) for arr, n in jobs], append=lo > 0, threads=3)
    for arr, n in jobs:
        np.savetxt(a, arr, fmt=fmt)
        assert a.read_bytes() == (tmp_path / f"batch_{n}.txt").read_bytes(), (fmt, n)
    with pytest.raises(_lib.PtnnError, match="cannot open"):
        _lib.savetxt_batch([(str(tmp_path / "ok.txt"), f32[:3], fmt), (str(tmp_path / "no_such_dir" / "x.txt"), f32[:3], fmt)], threads=2)
    # the value np.loadtxt reads back
    fin = f64[np.isfinite(f64)]
    for arr in (f32, fin):
        np.savetxt(a, arr, fmt=fmt)
        back = np.loadtxt(a).reshape(arr.shape)
        got = _lib.text_round(arr, fmt, threads=3)
        assert got.dtype == np.float64 and np.array_equal(back, got), fmt


def test_overlapped_run_is_cut_at_whole_swap_intervals(pt):
    """parallel_tempering.overlap_cuts: the launches of an overlapped run_chains() end after whole swap intervals, never more of them
    than asked for or than there are intervals, and the last one ends the run."""
    from ptnn_amd.parallel_tempering import overlap_cuts
    assert overlap_cuts(10000, 100, 8) == [1200, 2500, 3700, 5000, 6200, 7400, 8700, 9999]
    rng = np.random.default_rng(3)
    for _ in range(2000):
        S, si, K = int(rng.integers(2, 5000)), int(rng.integers(1, 600)), int(rng.integers(1, 20))
        ends = overlap_cuts(S, si, K)
        assert ends[-1] == S - 1 and ends == sorted(set(ends)) and ends[0] >= 1
        assert len(ends) <= max(1, min(K, max(1, (S - 1) // si)))
        assert all(e % si == 0 for e in ends[:-1])
    assert overlap_cuts(50, 100, 8) == [49] and overlap_cuts(2, 1, 8) == [1]


def test_posterior_matrix_is_the_transposed_burn_in_cut(pt):
    from ptnn_amd import _lib
    rng = np.random.default_rng(2)
    for R, S, P, b in ((3, 50, 7, 25), (5, 1000, 31, 500), (2, 9, 4, 0), (1, 5, 3, 5)):
        pos_w = rng.normal(size=(R, S, P)).astype(np.float32)
        want = pos_w[:, b:, :].astype(np.float64).transpose(2, 0, 1).reshape(P, -1)
        padded = np.zeros((R, S, P + 5), np.float32)         # the padded rows of a trace image are read in place
        padded[:, :, :P] = pos_w
        for threads in (1, 4):
            assert np.array_equal(_lib.posterior_matrix(pos_w, b, threads), want)
            assert np.array_equal(_lib.posterior_matrix(padded[:, :, :P], b, threads), want)
        assert np.array_equal(_lib.posterior_matrix(pos_w.astype(np.float64), b, 2), want)


def test_recorded_bf16_study_meets_its_stated_bounds():
    """profiles/r04_bf16_study.json (BASELINE config 5's fp32 vs bf16 tolerance study, recorded on the MI355X by
    profiles/tools/bf16_study.py: 128 chains x 200 steps per forward mode, every accepted step's log-likelihood against the float64
    oracle at identical inputs) against the bounds the GPU test holds a live, smaller run to."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_parity_bounds", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    src = open(spec.origin).read()
    ns = {}
    a = src.index("BF16_STUDY_BOUNDS = {")
    b = src.index("def test_config5_bf16_forward_tolerance_study")
    exec(src[a:b], ns)                                       # the bounds table and its checker: plain Python, no GPU import
    study = json.load(open(os.path.join(ROOT, "profiles", "r04_bf16_study.json")))
    ns["check_bf16_study"](study, decisions_min=128 * 200)
    m = study["modes"]
    assert m["split"]["flips"] == 0 and m["exact"]["flips"] == 0 and m["split"]["chains_identical_to_exact"] == 128
    assert m["bf16"]["chains_identical_to_exact"] < 128 and m["bf16"]["flips"] > 0
