#!/bin/bash
# Three rocprofv3 passes of ONE bench.py command on the GPU box (gpurun), then profiles/summarize.py.
#   profiles/collect.sh <workload> <tag> [extra bench.py arguments]
# The command is the driver's own (`python3 bench.py --gpus 1 --steps 20 --warmup 5`) plus --no-cpu-baseline --no-extras, so
# that every kernel launch of the process belongs to a warm-up run or to a timed run.  Counters never share a pass with --stats,
# FETCH_SIZE and WRITE_SIZE never share a pass (MI355X_MICROARCH.md, HBM section).
set -o pipefail
W=$1; TAG=$2; shift 2
export TMPDIR=/tmp
P=gpurun_out/prof_$W
rm -rf $P; mkdir -p $P
CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5 --workload $W --no-cpu-baseline --no-extras $*"
echo "$CMD" > $P/command.txt
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- $CMD > $P/kt.log 2>&1 && \
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $CMD > $P/fetch.log 2>&1 && \
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $CMD > $P/write.log 2>&1 && \
python3 profiles/summarize.py $P $TAG --current $W > $P/sum.log 2>&1
rc=$?
tail -3 $P/sum.log | cut -c1-300
# the condensed files travel back through gpurun_out/ (profiles/ on the box is a scratch copy)
mkdir -p gpurun_out/profiles_new && cp profiles/${TAG}_* profiles/current_pmc_$W.json gpurun_out/profiles_new/ 2>/dev/null
rm -rf $P/kt/*/*_kernel_trace.csv $P/fetch $P/write 2>/dev/null   # keep the pull small
exit $rc
