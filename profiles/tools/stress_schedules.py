"""Randomised differential test: for random nets / data subsets / ladders, every schedule must commit the same chain bit for bit
(traces, swap log, counters).  A development aid (the fixed cases live in tests/test_gpu_parity.py)."""
import os, sys, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import parity
from parity import orc


def run(seed=0, ncase=30, verbose=True, shapes="timeseries+iris", oracle=False):
  d = parity.datasets()
  rng = np.random.default_rng(seed)
  bad = 0
  for case in range(ncase):
      if shapes in ("all", "wide"):
          # every compiled (task, I, O) on synthetic data of that shape, hidden layers up to 64 units (mid-sized ones take the
          # matrix-core forward pass under the cooperative schedule)
          task, I, O = [(0, 4, 1), (0, 5, 1), (0, 32, 1), (1, 4, 3), (1, 34, 2), (1, 9, 2), (1, 11, 10), (1, 20, 2), (1, 16, 10),
                        (1, 6, 18)][int(rng.integers(0, 10))]
          H = int(rng.choice([int(rng.integers(1, 17)), int(rng.integers(17, 65))]))
          if shapes == "wide":                        # one thread per hidden unit, vectors in HBM, MFMA forward pass
              H = int(rng.choice([int(rng.integers(65, 300)), 32 * int(rng.integers(3, 9))]))
          ntr, nte = int(rng.integers(3, 100)), int(rng.integers(2, 40))
          X = rng.standard_normal((ntr + nte, I)) if task else rng.uniform(0, 1, (ntr + nte, I))
          y = (np.argmax(X @ rng.standard_normal((I, O)), axis=1).astype(np.float64) if task
               else np.clip(0.5 + 0.3 * np.sin(X.sum(axis=1)) + 0.02 * rng.standard_normal(ntr + nte), 0, 1))
          data = np.hstack([X, y[:, None]])
          train, test = data[:ntr], data[ntr:]
          name, mt, lr = "synthetic", (10 if task else 2), (0.01 if task else 0.1)
      else:
          task = int(rng.integers(0, 2))
          if task == 0:
              name, I, O, mt, lr = str(rng.choice(["sunspot", "mackey", "lazer"])), 4, 1, 2, 0.1
          else:
              name, I, O, mt, lr = "iris", 4, 3, 10, 0.01
          H = int(rng.integers(1, 17))
          ntr = int(rng.integers(3, min(120, d[name + "_train"].shape[0])))
          nte = int(rng.integers(2, 40))
          train, test = d[name + "_train"][:ntr], d[name + "_test"][:nte]
      R = int(rng.choice([2, 3, 4, 6, 8])); si = int(rng.integers(4, 15)); S = int(rng.integers(3, 7)) * si + int(rng.integers(0, 4)) + 2
      lg = bool(rng.integers(0, 2)) or task == 0
      seed = int(rng.integers(1, 10**6))
      topo = (I, H, O); P = orc.num_param(topo)
      tape = orc.PhiloxTape(seed)
      w0 = (float(rng.choice([0.3, 1.0])) * np.stack([tape.w_init(r, P) for r in range(R)])).astype(np.float32)
      T = np.array(orc.temperature_ladder(R, mt), dtype=np.float32)
      # every speculative layout and the packed schedule sum in the same (wave-local) order: bit-identical (and so is the one-wave
      # cooperative schedule for small nets).  The cooperative schedule on several waves adds the row likelihoods in another
      # order, and for mid-sized nets takes the matrix-core forward pass: compared within round-off.
      variants = [dict(schedule=2, waves=1, groups=1), dict(schedule=2, waves=4, groups=2), dict(schedule=2, waves=8, groups=1),
                  dict(schedule=2, waves=4, groups=4), dict(schedule=2, waves=2, groups=8)]
      if H > 64:                                      # one kernel, one to four work-groups per replica: bit-identical
          variants = [dict(schedule=0, groups=1), dict(schedule=0, groups=2), dict(schedule=0, groups=4), dict(schedule=0), dict(schedule=1)]
      if H <= 16:
          variants.append(dict(schedule=3))
          if H > 8 and lg:                            # 16-lane groups: the packed round on one, two, four CUs per replica
              variants += [dict(schedule=3, groups=1), dict(schedule=3, groups=2), dict(schedule=3, groups=4)]
      if H < 24 or I < 6:
          variants.append(dict(schedule=1, waves=1))
      if H <= 64:
          variants += [dict(schedule=1, waves=4), dict(schedule=1, waves=2), dict(schedule=0)]
      if task == 1 and not lg and H <= 64:            # prefetching tree: random-walk classification only
          if H < 24 or I < 6:
              variants += [dict(schedule=4, waves=1, groups=3), dict(schedule=4, waves=1, groups=15)]   # one wave: bit-identical
          variants.append(dict(schedule=4, groups=7))                                                 # default block: round-off
      ref = None
      for v in variants:
          try:
              s = parity.make_sampler(task, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=si, use_lg=lg, lr=lr, seed=seed, **v)
          except Exception as e:
              print("   skip", v, str(e)[:80]); continue
          s.set_state(w0, T); s.run(-1); s.sync()
          got = (s.traces(), s.swap_stats(), s.swap_log().copy())
          s.close()
          if not all(np.isfinite(got[0][k]).all() for k in got[0]):
              bad += 1
              print(f"NON-FINITE case {case}: task={task} {name} topo={topo} ntr={ntr} R={R} S={S} si={si} lg={lg} seed={seed} {v}", flush=True)
          if ref is None:
              ref, refv = got, v
              if oracle:                                  # the reference variant against the float64 oracle on the same tape
                  pt = orc.PTOracle(task, topo, train, test, R, mt, R * S, si, use_lg=lg, l_prob=0.5, lr=lr, seed=seed)
                  for rep, w in zip(pt.replicas, w0):
                      rep.__init__(task, topo, pt.train, pt.test, w.astype(np.float64), rep.T, S, lg, 0.5, lr, pt.tape, rep.gid)
                  o = parity.OracleRun(pt).run()
                  try:
                      assert got[1][2] == pt.rounds_done and got[1][1] == pt.total_swap_proposals
                      # a swap decided the other way (fp32 vs float64 L at a near tie) legitimately sends the chains apart:
                      # compare the rows written before that round only
                      olog = np.array(pt.src_log, dtype=np.int32).reshape(-1, R)
                      nr = min(len(olog), len(got[2]))
                      dr = [q for q in range(nr) if not np.array_equal(olog[q], got[2][q])]
                      tr_cmp, limit = got[0], S
                      if dr:
                          cut = (dr[0] + 1) * si - (0 if task == 0 else 1)
                          if verbose:
                              print(f"   info: swap round {dr[0]} decided differently (near tie), comparing rows < {cut}", flush=True)
                          limit = cut
                      # ... and so does an MH decision at a near tie in ANY replica (its state reaches the others through
                      # the next exchange): compare up to the earliest one
                      firsts = []
                      for r in range(R):
                          dd = np.nonzero(tr_cmp["accept"][r].astype(np.int64)[:limit] !=
                                          pt.replicas[r].accept_list.astype(np.int64)[:limit])[0]
                          if dd.size:
                              firsts.append((int(dd[0]), r))
                      if firsts:
                          f0, r0 = min(firsts)
                          i_ = f0 - 2
                          assert abs(o.logalpha[r0, i_] - o.logu[r0, i_]) < 0.05, (r0, i_, o.logalpha[r0, i_], o.logu[r0, i_])
                          limit = f0 - 1
                      for r in range(R):
                          first = parity.compare_replica_trace(tr_cmp, r, pt.replicas[r], f"case {case} r{r} ", limit=limit)
                          if first is not None:
                              i_ = first - 2
                              assert abs(o.logalpha[r, i_] - o.logu[r, i_]) < 0.05, (r, i_, o.logalpha[r, i_], o.logu[r, i_])
                  except AssertionError as e:
                      msg = str(e)
                      if task == 1 and any(t_ in msg for t_ in ("rmse_train", "rmse_test", "acc_train", "acc_test")) and "pos_w" not in msg and "likeh" not in msg:
                          # argmax is discontinuous: a row whose two largest outputs tie within fp32 round-off may classify
                          # differently (class-id RMSE and accuracy move by one row's worth); weights and likelihood agreed
                          if verbose:
                              print(f"   info: case {case}: class prediction of a near-tie row differs (rmse/acc only)", flush=True)
                          continue
                      bad += 1
                      if os.environ.get("STRESS_DEBUG"):
                          for r in range(R):
                              rep = pt.replicas[r]
                              n_ = min(rep.pos_w.shape[0], got[0]["pos_w"].shape[1])
                              dif = np.abs(got[0]["pos_w"][r][:n_] - rep.pos_w[:n_]).max(axis=1)
                              print("   r", r, "accept gpu", got[0]["accept"][r][:n_].tolist(), flush=True)
                              print("   r", r, "accept orc", rep.accept_list[:n_].astype(int).tolist(), flush=True)
                              print("   r", r, "row diff", np.round(dif, 3).tolist(), flush=True)
                              print("   r", r, "la-lu", np.round(o.logalpha[r, :n_] - o.logu[r, :n_], 3).tolist(), flush=True)
                      print(f"ORACLE MISMATCH case {case}: task={task} {name} topo={topo} ntr={ntr} nte={nte} R={R} S={S} si={si} lg={lg} seed={seed}: {str(e)[:300]}", flush=True)
              continue
          loose = H <= 64 and v.get("schedule") in (0, 1, 4) and (v.get("waves", 0) != 1 or (H >= 24 and I >= 6))
          if loose:
              same_dec = np.array_equal(got[0]["accept"], ref[0]["accept"])
              ok = (not same_dec) or (got[1] == ref[1] and np.array_equal(got[2], ref[2]) and
                                      all(np.allclose(got[0][k], ref[0][k], rtol=2e-5, atol=2e-4, equal_nan=True) for k in ref[0]))
          else:
              ok = got[1] == ref[1] and np.array_equal(got[2], ref[2]) and all(np.array_equal(got[0][k], ref[0][k], equal_nan=True) for k in ref[0])
          if not ok:
              bad += 1
              diffs = [k for k in ref[0] if not np.array_equal(got[0][k], ref[0][k], equal_nan=True)]
              print(f"MISMATCH case {case}: task={task} {name} H={H} ntr={ntr} nte={nte} R={R} S={S} si={si} lg={lg} seed={seed} {refv} vs {v}: {diffs} stats {ref[1]} {got[1]}", flush=True)
              for k in diffs[:3]:                          # where and by how much (first few positions)
                  a_, b_ = np.asarray(ref[0][k]), np.asarray(got[0][k])
                  idx = np.argwhere(~((a_ == b_) | (np.isnan(a_) & np.isnan(b_))))[:6]
                  print("   ", k, [(tuple(int(t) for t in ix), float(a_[tuple(ix)]), float(b_[tuple(ix)])) for ix in idx], flush=True)
      if verbose:
          print(f"case {case}: task={task} {name} H={H} ntr={ntr} R={R} S={S} si={si} lg={lg} ok", flush=True)
  if verbose:
      print("mismatching variants:", bad)
  return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30,
        shapes=sys.argv[3] if len(sys.argv) > 3 else "timeseries+iris", oracle=len(sys.argv) > 4)
