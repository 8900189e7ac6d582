#!/bin/bash
# round-2 GPU session A: new kernels' parity, full-size fixture, op timings, a short step bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn2 or segment or vq or bert_embed" > gpurun_out/r2a/t_new_kernels.log 2>&1
echo "new kernels rc=$?" >> gpurun_out/r2a/summary.log
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -s > gpurun_out/r2a/t_full.log 2>&1
echo "full size rc=$?" >> gpurun_out/r2a/summary.log
timeout 300 python tools/bench_ops.py attn2 10 > gpurun_out/r2a/ops_attn2.json 2> gpurun_out/r2a/ops_attn2.err
timeout 300 python tools/bench_ops.py attn 10 > gpurun_out/r2a/ops_attn.json 2> gpurun_out/r2a/ops_attn.err
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_full_size_gpu.py > gpurun_out/r2a/t_all.log 2>&1
echo "all gpu tests rc=$?" >> gpurun_out/r2a/summary.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
echo "bench rc=$?" >> gpurun_out/r2a/summary.log
tail -5 gpurun_out/r2a/t_new_kernels.log gpurun_out/r2a/t_full.log gpurun_out/r2a/t_all.log
cat gpurun_out/r2a/summary.log gpurun_out/r2a/ops_attn2.json gpurun_out/r2a/bench.json
