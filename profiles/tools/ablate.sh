# Timing experiment (diagnostic builds, results of the runs are garbage): Ionosphere 256 with one phase of the cooperative step
# removed at a time -- what each phase costs in the PRODUCT build (the stamped build is distorted by its own scratch).
# Builds: for k in 0..6: NOSTAMPS=1 EXTRA=-DPTNN_ABLATE=$k bash profiles/tools/build_stamps.sh 1 34 2; cp ... libptnn_abl$k.so
names=("nothing removed" "random tape of the next step" "trace row" "row scoring" "work-group reduction" "matrix products + epilogues" "weight split")
for k in 0 1 2 3 4 5 6; do
  PTNN_LIBRARY=$PWD/profiles/tools/libptnn_abl$k.so timeout -k 10 120 python3 bench.py --workload ionosphere256 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('removed: %-30s %8.2f ms per run  %6.0f cycles per MH step at 2.39 GHz' % ('${names[$k]}', j['ms_per_step'], j['ms_per_step'] * 1e-3 / 9999 * 2.39e9))"
done
