#!/bin/bash
# two ranks on ONE GPU through gloo: the N > 1 code path (gathered loss, overlapped gradient buckets behind the weight-gradient stream,
# VQ statistics all-reduce) on a real device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ddp; mkdir -p $O
export CTCLIP_BENCH_BACKEND=gloo CTCLIP_BENCH_SINGLE_DEVICE=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --no-attn-block --profile-steps 0 > $O/ddp2.json 2> $O/ddp2.err; echo "rc=$?"
tail -n 2 $O/ddp2.json | cut -c1-600; tail -n 5 $O/ddp2.err
unset CTCLIP_BENCH_BACKEND CTCLIP_BENCH_SINGLE_DEVICE
timeout 300 python bench.py --steps 3 --warmup 1 --batch 8 --no-attn-block --profile-steps 0 --no-cpu-baseline --no-pmc | cut -c1-400
