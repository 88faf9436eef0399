# A/B of tree variants on ONE box, interleaved: ab_iris.sh OUT variant...
OUT=$1; shift
for i in 1 2 3; do
  for v in "$@"; do
    PTNN_LIBRARY=$PWD/profiles/tools/libptnn_$v.so timeout -k 10 200 python3 bench.py --workload iris16 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', round(j['value']), round(j['ms_per_step'],3), round(j['roofline']['avg_launch_ms'],5))" | tee -a $OUT
  done
done
