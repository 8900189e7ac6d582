#!/bin/bash
# round-5 session K: record run on the final library (Adam clears gradients, BERT fan-in adds in the dX epilogues) + determinism trace + soak
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_run.sh tests smoke bench prof
cp gpurun_out/run/summary.log gpurun_out/run/summary_k.log
RUNS=60 BENCH_RUNS=10 SOAK_REPS=40 bash tools/gpu_determinism.sh > gpurun_out/run/determinism_k.log 2>&1
cat gpurun_out/run/summary_k.log | grep -v "Gloo\|W926\|attn2_bwd_fused"
tail -n 60 gpurun_out/run/determinism_k.log
