export TMPDIR=/tmp
for W in ${WORKLOADS:-mackey64 ionosphere256 iris16}; do
P=gpurun_out/prof_$W; rm -rf $P; mkdir -p $P
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python3 bench.py --workload $W --steps 20 --warmup 5 > $P/kt.log 2>&1 && \
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- python3 bench.py --workload $W --steps 20 --warmup 5 > $P/fetch.log 2>&1 && \
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- python3 bench.py --workload $W --steps 20 --warmup 5 > $P/write.log 2>&1 && \
python profiles/summarize.py $P r01g_$W > $P/sum.log 2>&1; tail -1 $P/kt.log | cut -c1-150
done
cp profiles/r01g_*64* profiles/r01g_*256* profiles/r01g_iris16* gpurun_out/ 2>/dev/null || true
