#!/bin/bash
# Runs the steps of one gpurun call in order, each under its own time limit.  An ordinary failure (a red test, a non-zero exit)
# is recorded and the next step runs; a step that had to be KILLED at its limit ends the call (no further GPU step is started
# after a timeout).  Usage: gpu_steps.sh OUTDIR  <<< "limit_seconds|name|command" lines on stdin
out=${1:?outdir}; mkdir -p "$out"
while IFS='|' read -r limit name cmd; do
    [ -z "$name" ] && continue
    echo "=== $name (limit ${limit}s): $cmd" | tee -a "$out/steps.log"
    t0=$(date +%s)
    timeout -k 10 "$limit" bash -c "$cmd" > "$out/$name.out" 2> "$out/$name.err"
    rc=$?
    echo "=== $name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$out/steps.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name was killed at its limit: stopping" | tee -a "$out/steps.log"; exit 1; fi
done
exit 0
