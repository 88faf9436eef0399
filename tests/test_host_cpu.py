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
    assert lib.ptnn_abi_version() == 3


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
