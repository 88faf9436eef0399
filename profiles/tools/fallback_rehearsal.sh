set -e
export PTNN_COMM_TIMEOUT_S=25 PTNN_COMM_TRACE=1
echo "== N=2 on one GPU, rccl requested (expected: fallback to host) =="
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/fb_n2.json 2> gpurun_out/fb_n2.err
tail -c 900 gpurun_out/fb_n2.json; grep -a "bench\]" gpurun_out/fb_n2.err | cut -c1-600
echo "== N=1 force-comm rccl =="
timeout -k 10 300 python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-comm --no-cpu-baseline --no-extras > gpurun_out/fb_n1.json 2> gpurun_out/fb_n1.err
tail -c 500 gpurun_out/fb_n1.json
echo "== N=2 host transport requested =="
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --transport host > gpurun_out/fb_n2h.json 2> gpurun_out/fb_n2h.err
tail -c 500 gpurun_out/fb_n2h.json
