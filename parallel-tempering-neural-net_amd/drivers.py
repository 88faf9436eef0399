"""Experiment drivers: what the reference's `main()` functions do around `ParallelTempering`.

Reference: multicore-pt-regression/pt_timeseries_regression.py:877-1154 (REG) and
multicore-pt-classification/pt_classification.py:899-1222 (CLS).  Kept: run-directory numbering
(`<folder><name>_<n>`), the nine sub-directories, the summary statistics, `result.txt` and
`master_result_file.txt` (15 numbers + run name; REG '%1.4f', CLS '%1.2f'; column meaning REG:1052 / CLS:1138)
in both the working folder and the "db" folder, and the plots (rmse_samples.pdf / acc_samples.png, likelihood,
accept.png) when matplotlib is importable.  The hyper-parameters that are literals inside the reference's `main()`
are keyword arguments here with the same defaults.

    python -m ... is not needed:  from ptnn_amd import drivers; drivers.run_regression("Sunspot", train, test, ...)
"""
import os
import time

import numpy as np

SUBDIRS = ['/predictions/', '/posterior', '/results', '/surrogate', '/surrogate/learnsurrogate_data',
           '/posterior/pos_w', '/posterior/pos_likelihood', '/posterior/surg_likelihood', '/posterior/accept_list']

# the reference's problem tables: name -> (problem number, ip, hidden, output) (REG:882-917, CLS:909-995)
REG_PROBLEMS = {"Lazer": 1, "Sunspot": 2, "Mackey": 3, "Lorenz": 4, "Rossler": 5, "Henon": 6, "ACFinance": 7}
CLS_PROBLEMS = {"winequality-red": (1, 11, 50, 10), "winequality-white": (2, 11, 50, 10), "iris": (3, 4, 12, 3),
                "Ionosphere": (4, 34, 50, 2), "Cancer": (5, 9, 12, 2), "bank-additional": (6, 20, 50, 2),
                "PenDigit": (7, 16, 30, 10), "chess": (8, 6, 25, 18)}


def _next_run_dir(folder, name):
    """`<folder><name>_<n>` with the first n that does not exist yet (REG:964-975)."""
    n = 0
    while os.path.exists(f"{folder}{name}_{n}"):
        n += 1
    os.makedirs(f"{folder}{name}_{n}")
    return f"{folder}{name}_{n}", n


def _scatter_pair(plt, first, second, targets, fname, title=None, xlabel=None, ylabel=None):
    """Two point clouds over the sample index, saved once per target directory.  The reference labels the TRAIN series 'Test'
    and the TEST series 'Train' (REG:1079-1080, CLS:1152-1153); the files are compared by eye with its own, so that is kept."""
    for target in targets:
        plt.plot(first, '.', label='Test')
        plt.plot(second, '.', label='Train')
        plt.legend(loc='upper right')
        if title:
            plt.title(title)
        if xlabel:
            plt.xlabel(xlabel, fontsize=12)
            plt.ylabel(ylabel, fontsize=12)
        plt.savefig(os.path.join(target, fname))
        plt.clf()


def _lines(plt, curves, targets, fname, ylabel):
    for target in targets:
        plt.plot(curves)
        plt.xlabel('Samples', fontsize=12)
        plt.ylabel(ylabel, fontsize=12)
        plt.savefig(os.path.join(target, fname))
        plt.clf()


def _plots(kind, path, path_db, res, num_chains):
    """The figures of a run (REG:1077-1131, CLS:1150-1199): metric over samples, proposed log-likelihood per chain, accepted
    proposals per chain.  Skipped when matplotlib is not importable."""
    try:
        import matplotlib
        matplotlib.use('agg')
        import matplotlib.pyplot as plt
    except Exception:                                        # noqa: BLE001
        return False
    if kind == "reg":
        _scatter_pair(plt, res["rmse_train"], res["rmse_test"], (path_db, path), 'rmse_samples.pdf', xlabel='Samples', ylabel='RMSE')
    else:
        _scatter_pair(plt, res["acc_train"], res["acc_test"], (path, path_db), 'acc_samples.png',
                      title="Plot of Classification Acc. over time")
    per_chain = res["likelihood"][:, 0].reshape(num_chains, -1)
    _lines(plt, per_chain.T, (path, path_db), 'likelihood.' + ('pdf' if kind == "reg" else 'png'), ' Log-Likelihood')
    _lines(plt, res["accept_vec"].T, (path_db,), 'accept.png', ' Number accepted proposals')
    return True


_RESULT_NAMES = ("pos_w", "fx_train", "fx_test", "rmse_train", "rmse_test", "acc_train", "acc_test", "likelihood", "swap_perc",
                 "accept_vec", "accept")
# what differs between the two main() functions: result format, the metric that is summarised and which extreme is "best"
_KINDS = {"reg": dict(fmt='%1.4f', metric="rmse", best=np.amin), "cls": dict(fmt='%1.2f', metric="acc", best=np.amax)}


def _experiment(kind, pt_class, pt_args, name, problem, col5, NumSample, maxtemp, swap_interval, learn_rate, num_chains, burn_in,
                problemfolder, problemfolder_db, plots):
    """Run one experiment and leave what the reference's main() leaves: the run directories, result.txt and the master file in
    both trees (15 numbers + run name: problem, NumSample, maxtemp, swap interval, Langevin probability | flag, learning rate,
    train mean / std / best, test mean / std / best, swap %, accept %, minutes; REG:1052 / CLS:1138) and the figures."""
    spec = _KINDS[kind]
    path, run_nb = _next_run_dir(problemfolder, name)
    path_db, _ = _next_run_dir(problemfolder_db, name)
    t0 = time.time()
    pt = pt_class(*pt_args(path))
    for d in SUBDIRS:
        pt.make_directory(path + d)
    pt.initialize_chains(burn_in)
    res = dict(zip(_RESULT_NAMES, pt.run_chains()))
    minutes = (time.time() - t0) / 60
    S = res["accept_vec"].shape[1]
    accept_per = 100.0 * float(np.mean(res["accept_vec"][:, -1] / S))                # last cumulative count over samples (REG:1013-1016)
    stats = [f(res[f"{spec['metric']}_{part}"]) for part in ("train", "test") for f in (np.mean, np.std, spec["best"])]
    row = np.asarray([problem, NumSample, maxtemp, swap_interval, col5, learn_rate, *stats, res["swap_perc"], accept_per, minutes])
    run_name = f"{name}_{run_nb}"
    for run_dir, master_dir in ((path_db, problemfolder_db), (path, problemfolder)):
        with open(run_dir + '/result.txt', 'a+') as f:
            np.savetxt(f, row, fmt=spec["fmt"], newline=' ')
        with open(master_dir + '/master_result_file.txt', 'a+') as f:
            np.savetxt(f, row, fmt=spec["fmt"], newline=' ')
            np.savetxt(f, [run_name], fmt="%s", newline=' \n')
    if plots:
        _plots(kind, path, path_db, res, num_chains)
    return dict(allres=row, path=path, path_db=path_db, run_name=run_name, pt=pt)


def run_regression(name, traindata, testdata, *, problem=None, hidden=10, ip=4, output=1, NumSample=100000, maxtemp=2,
                   swap_ratio=0.01, num_chains=10, burn_in=0.5, learn_rate=0.1, use_langevin_gradients=True,
                   langevin_prob=0.5, problemfolder='Res_PT/', problemfolder_db='Res_PT_db/', plots=True, **pt_kwargs):
    """One pass of the loop body of REG main() (REG:881-1150): defaults are its literals.  Returns the summary row and paths."""
    from .pt_timeseries_regression import ParallelTempering
    swap_interval = int(swap_ratio * NumSample / num_chains)                 # REG:949
    topology = [ip, hidden, output]

    def pt_args(path):
        return (use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp, NumSample, swap_interval,
                langevin_prob, path)
    pt_class = (lambda *a: ParallelTempering(*a, **pt_kwargs))
    return _experiment("reg", pt_class, pt_args, name, REG_PROBLEMS.get(name, 0) if problem is None else problem, langevin_prob,
                       NumSample, maxtemp, swap_interval, learn_rate, num_chains, burn_in, problemfolder, problemfolder_db, plots)


def run_classification(name, traindata, testdata, *, problem=None, hidden=None, ip=None, output=None, NumSample=50000,
                       maxtemp=10, swap_ratio=0.02, num_chains=10, burn_in=0.5, learn_rate=0.01, use_langevin_gradients=False,
                       problemfolder='PT_Eval/', problemfolder_db='PT_Eval_db/', plots=True, **pt_kwargs):
    """One pass of the loop body of CLS main() (CLS:903-1199); the topology of a known problem comes from its table."""
    from .pt_classification import ParallelTempering
    if name in CLS_PROBLEMS:
        pnum, pip, phid, pout = CLS_PROBLEMS[name]
        problem = pnum if problem is None else problem
        ip, hidden, output = ip or pip, hidden or phid, output or pout
    swap_interval = int(swap_ratio * NumSample / num_chains)                 # CLS:1040
    topology = [ip, hidden, output]

    def pt_args(path):
        return (use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp, NumSample, swap_interval, path)
    pt_class = (lambda *a: ParallelTempering(*a, **pt_kwargs))
    return _experiment("cls", pt_class, pt_args, name, problem or 0, use_langevin_gradients, NumSample, maxtemp, swap_interval,
                       learn_rate, num_chains, burn_in, problemfolder, problemfolder_db, plots)


def split_and_normalise(features, classes, ip, train_ratio=0.7, rng=None):
    """The `separate_flag` branch of CLS main() (CLS:1001-1012): z-score every feature, random 70/30 split.
    The reference permutes with the unseeded global generator; pass `rng` for a reproducible split."""
    features = np.array(features, dtype=np.float64, copy=True)
    for k in range(ip):
        features[:, k] = (features[:, k] - np.mean(features[:, k])) / np.std(features[:, k])
    n = features.shape[0]
    indices = (rng or np.random).permutation(n)
    cut = int(train_ratio * n)
    traindata = np.hstack([features[indices[:cut], :], classes[indices[:cut], :]])
    testdata = np.hstack([features[indices[cut:], :], classes[indices[cut:], :]])
    return traindata, testdata


def takens_embedding(series, window=5, stride=2):
    """Replacement for the source-less `process`/`fnn` binaries of Data_OneStepAhead: rows are
    `series[stride*k : stride*k + window]` (window-1 lag inputs + 1 target).  The shipped split keeps the first
    int(0.6 n) rows for training and the LAST int(0.4 n) - 1 rows for testing (Sunspot: 498 rows -> 298 / 198, two rows
    in between unused); checked against the shipped Sunspot, Mackey and Lazer files."""
    series = np.asarray(series, dtype=np.float64).reshape(-1)
    nrows = (series.shape[0] - window) // stride + 1
    rows = np.stack([series[stride * k:stride * k + window] for k in range(nrows)])
    n_train, n_test = int(0.6 * nrows), int(0.4 * nrows) - 1
    return rows[:n_train], rows[nrows - n_test:]


# ---------------------------------------------------------------------------------------------------------------------
# Data ingestion: the per-problem loading rules of the reference's main() functions as tables.
# REG:881-917: `Data_OneStepAhead/<name>/{train,test}.txt`, whitespace separated, 4 lag columns + target.
# CLS:909-1012: one rule per problem -- file(s), delimiter, header row, label column and its offset, whether train/test ship
# separately (a trailing column is dropped there) or are split 70/30 after z-scoring (`separate_flag`), and PenDigit's
# z-scoring of train and test with their OWN moments.
# ---------------------------------------------------------------------------------------------------------------------
REG_FILES = {name: ("Data_OneStepAhead/%s/train.txt" % name, "Data_OneStepAhead/%s/test.txt" % name) for name in REG_PROBLEMS}

# name: (mode, files, delimiter, header rows, label column, label offset)
#   mode "split":  one file, features = columns [0, ip), z-score + random 70/30 split (split_and_normalise)
#   mode "pair":   train and test files, last column dropped
#   mode "pair_z": train and test files kept whole, the first ip columns z-scored per file
CLS_FILES = {
    "winequality-red": ("split", ("DATA/winequality-red.csv",), ";", 1, 11, 0.0),
    "winequality-white": ("split", ("DATA/winequality-white.csv",), ";", 1, 11, 0.0),
    "iris": ("split", ("DATA/iris.csv",), ";", 0, 4, -1.0),
    "Ionosphere": ("pair", ("DATA/Ions/Ions/ftrain.csv", "DATA/Ions/Ions/ftest.csv"), ",", 0, None, 0.0),
    "Cancer": ("pair", ("DATA/Cancer/ftrain.txt", "DATA/Cancer/ftest.txt"), " ", 0, None, 0.0),
    "bank-additional": ("split", ("DATA/Bank/bank-processed.csv",), ";", 0, 20, 0.0),
    "PenDigit": ("pair_z", ("DATA/PenDigit/train.csv", "DATA/PenDigit/test.csv"), ",", 0, None, 0.0),
    "chess": ("split", ("DATA/chess.csv",), ";", 0, 6, 0.0),
}


def load_regression_problem(name, data_root):
    """(traindata, testdata) of a time-series problem of REG main(), read from `data_root` (the directory that holds
    `Data_OneStepAhead/`)."""
    if name not in REG_FILES:
        raise KeyError(f"unknown regression problem {name!r}; known: {sorted(REG_FILES)}")
    tr, te = (os.path.join(data_root, f) for f in REG_FILES[name])
    return np.loadtxt(tr), np.loadtxt(te)


def load_classification_problem(name, data_root, rng=None, train_ratio=0.7):
    """(traindata, testdata, (ip, hidden, output)) of a classification problem of CLS main(), read from `data_root` (the
    directory that holds `DATA/`).  Problems the reference splits at random take `rng` (a numpy Generator) for a
    reproducible split; the reference itself uses the unseeded global generator (CLS:1010)."""
    if name not in CLS_FILES:
        raise KeyError(f"unknown classification problem {name!r}; known: {sorted(CLS_FILES)}")
    mode, files, delim, header, label_col, label_off = CLS_FILES[name]
    _, ip, hidden, output = CLS_PROBLEMS[name]
    arrs = [np.genfromtxt(os.path.join(data_root, f), delimiter=delim)[header:] for f in files]
    if mode == "split":
        data = arrs[0]
        classes = data[:, label_col].reshape(-1, 1) + label_off
        train, test = split_and_normalise(data[:, :ip], classes, ip, train_ratio=train_ratio, rng=rng)
    elif mode == "pair":
        train, test = arrs[0][:, :-1], arrs[1][:, :-1]
    else:
        train, test = (np.array(a, dtype=np.float64, copy=True) for a in arrs)
        for a in (train, test):
            a[:, :ip] = (a[:, :ip] - a[:, :ip].mean(axis=0)) / a[:, :ip].std(axis=0)
    return train, test, (ip, hidden, output)
