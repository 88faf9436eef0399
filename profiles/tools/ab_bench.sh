# A/B of two builds on ONE box: ab_bench.sh <old.so> [workload] [reps] [extra bench flags]   (new = the in-tree library)
set -e
OLD=$1; WL=${2:-sunspot64}; REPS=${3:-3}; EXTRA=${4:-}
for i in $(seq $REPS); do
  for v in old new; do
    if [ $v = old ]; then export PTNN_LIBRARY=$PWD/$OLD; else unset PTNN_LIBRARY; fi
    timeout -k 10 300 python3 bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extras $EXTRA 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', '$WL', '$EXTRA', round(j['value']), j['ms_per_step'], j['roofline']['avg_launch_ms'])" | tee -a gpurun_out/ab_bench.log
  done
done
