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
    run_nb = 0
    while os.path.exists(folder + name + '_%s' % (run_nb)):
        run_nb += 1
    os.makedirs(folder + name + '_%s' % (run_nb))
    return folder + name + '_%s' % (run_nb), run_nb


def _append_row(path, allres, fmt, xv=None):
    with open(path, 'a+') as f:
        np.savetxt(f, allres, fmt=fmt, newline=' ')
        if xv is not None:
            np.savetxt(f, [xv], fmt="%s", newline=' \n')


def _plots(task, path, path_db, rmse_train, rmse_test, acc_train, acc_test, likelihood_rep, accept_vec, num_chains):
    try:
        import matplotlib
        matplotlib.use('agg')
        import matplotlib.pyplot as plt
    except Exception:
        return False
    if task == "reg":                                     # REG:1077-1097 (labels are swapped in the reference, kept)
        for p in (path_db, path):
            plt.plot(rmse_train, '.', label='Test')
            plt.plot(rmse_test, '.', label='Train')
            plt.legend(loc='upper right')
            plt.xlabel('Samples', fontsize=12)
            plt.ylabel('RMSE', fontsize=12)
            plt.savefig(p + '/rmse_samples.pdf')
            plt.clf()
        ext = 'pdf'
    else:                                                 # CLS:1150-1170
        x = np.linspace(0, acc_train.shape[0], num=acc_train.shape[0])
        for p in (path, path_db):
            plt.plot(x, acc_train, '.', label='Test')
            plt.plot(x, acc_test, '.', label='Train')
            plt.legend(loc='upper right')
            plt.title("Plot of Classification Acc. over time")
            plt.savefig(p + '/acc_samples.png')
            plt.clf()
        ext = 'png'
    likelihood = np.asarray(np.split(likelihood_rep[:, 0], num_chains))      # proposed likelihood per chain
    for p in (path, path_db):
        plt.plot(likelihood.T)
        plt.xlabel('Samples', fontsize=12)
        plt.ylabel(' Log-Likelihood', fontsize=12)
        plt.savefig(p + '/likelihood.' + ext)
        plt.clf()
    plt.plot(accept_vec.T)
    plt.xlabel('Samples', fontsize=12)
    plt.ylabel(' Number accepted proposals', fontsize=12)
    plt.savefig(path_db + '/accept.png')
    plt.clf()
    return True


def run_regression(name, traindata, testdata, *, problem=None, hidden=10, ip=4, output=1, NumSample=100000, maxtemp=2,
                   swap_ratio=0.01, num_chains=10, burn_in=0.5, learn_rate=0.1, use_langevin_gradients=True,
                   langevin_prob=0.5, problemfolder='Res_PT/', problemfolder_db='Res_PT_db/', plots=True, **pt_kwargs):
    """One pass of the loop body of REG main() (REG:881-1150).  Returns the 15-number summary row and the run paths."""
    from .pt_timeseries_regression import ParallelTempering
    problem = REG_PROBLEMS.get(name, 0) if problem is None else problem
    topology = [ip, hidden, output]
    swap_interval = int(swap_ratio * NumSample / num_chains)                 # REG:949
    path, run_nb = _next_run_dir(problemfolder, name)
    path_db, _ = _next_run_dir(problemfolder_db, name)
    timer = time.time()
    pt = ParallelTempering(use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp, NumSample,
                           swap_interval, langevin_prob, path, **pt_kwargs)
    for d in SUBDIRS:
        pt.make_directory(path + d)
    pt.initialize_chains(burn_in)
    (pos_w, fx_train, fx_test, rmse_train, rmse_test, acc_train, acc_test, likelihood_rep, swap_perc, accept_vec,
     accept) = pt.run_chains()
    list_end = accept_vec.shape[1]
    accept_ratio = accept_vec[:, list_end - 1:list_end] / list_end
    accept_per = np.mean(accept_ratio) * 100
    timetotal = (time.time() - timer) / 60
    rmse_tr, rmsetr_std, rmsetr_max = np.mean(rmse_train[:]), np.std(rmse_train[:]), np.amin(rmse_train[:])
    rmse_tes, rmsetest_std, rmsetes_max = np.mean(rmse_test[:]), np.std(rmse_test[:]), np.amin(rmse_test[:])
    xv = name + '_' + str(run_nb)
    allres = np.asarray([problem, NumSample, maxtemp, swap_interval, langevin_prob, learn_rate, rmse_tr, rmsetr_std, rmsetr_max,
                         rmse_tes, rmsetest_std, rmsetes_max, swap_perc, accept_per, timetotal])
    for p_run, p_master in ((path_db, problemfolder_db), (path, problemfolder)):
        _append_row(p_run + '/result.txt', allres, '%1.4f')
        _append_row(p_master + '/master_result_file.txt', allres, '%1.4f', xv)
    if plots:
        _plots("reg", path, path_db, rmse_train, rmse_test, acc_train, acc_test, likelihood_rep, accept_vec, num_chains)
    return dict(allres=allres, path=path, path_db=path_db, run_name=xv, pt=pt)


def run_classification(name, traindata, testdata, *, problem=None, hidden=None, ip=None, output=None, NumSample=50000,
                       maxtemp=10, swap_ratio=0.02, num_chains=10, burn_in=0.5, learn_rate=0.01, use_langevin_gradients=False,
                       problemfolder='PT_Eval/', problemfolder_db='PT_Eval_db/', plots=True, **pt_kwargs):
    """One pass of the loop body of CLS main() (CLS:903-1199)."""
    from .pt_classification import ParallelTempering
    if name in CLS_PROBLEMS:
        pnum, pip, phid, pout = CLS_PROBLEMS[name]
        problem = pnum if problem is None else problem
        ip, hidden, output = ip or pip, hidden or phid, output or pout
    topology = [ip, hidden, output]
    swap_interval = int(swap_ratio * NumSample / num_chains)                 # CLS:1040
    path, run_nb = _next_run_dir(problemfolder, name)
    path_db, _ = _next_run_dir(problemfolder_db, name)
    timer = time.time()
    pt = ParallelTempering(use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp, NumSample,
                           swap_interval, path, **pt_kwargs)
    for d in SUBDIRS:
        pt.make_directory(path + d)
    pt.initialize_chains(burn_in)
    (pos_w, fx_train, fx_test, rmse_train, rmse_test, acc_train, acc_test, likelihood_rep, swap_perc, accept_vec,
     accept) = pt.run_chains()
    timer2 = time.time()
    list_end = accept_vec.shape[1]
    accept_ratio = accept_vec[:, list_end - 1:list_end] / list_end
    accept_per = np.mean(accept_ratio) * 100
    timetotal = (timer2 - timer) / 60
    acc_tr, acctr_std, acctr_max = np.mean(acc_train[:]), np.std(acc_train[:]), np.amax(acc_train[:])
    acc_tes, acctest_std, acctes_max = np.mean(acc_test[:]), np.std(acc_test[:]), np.amax(acc_test[:])
    xv = name + '_' + str(run_nb)
    allres = np.asarray([problem or 0, NumSample, maxtemp, swap_interval, use_langevin_gradients, learn_rate, acc_tr, acctr_std,
                         acctr_max, acc_tes, acctest_std, acctes_max, swap_perc, accept_per, timetotal])
    for p_run, p_master in ((path_db, problemfolder_db), (path, problemfolder)):
        _append_row(p_run + '/result.txt', allres, '%1.2f')
        _append_row(p_master + '/master_result_file.txt', allres, '%1.2f', xv)
    if plots:
        _plots("cls", path, path_db, rmse_train, rmse_test, acc_train, acc_test, likelihood_rep, accept_vec, num_chains)
    return dict(allres=allres, path=path, path_db=path_db, run_name=xv, pt=pt)


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
