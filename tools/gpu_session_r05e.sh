#!/bin/bash
# round-5 session E: XCD-contiguous tile ownership of gemm_nt (product) against the round-4 map (variant oldmap): parity subset, A/B, PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_properties_bench_size_gpu.py -q -m gpu -x -k "gemm or geglu or headnorm or residual_comp or argmax or identity or tile" > $O/tests_e.log 2>&1
echo "tests rc=$? $(tail -n 1 $O/tests_e.log)" >> $O/summary.log
VARIANTS="oldmap" bash tools/gpu_ab_variants.sh > /dev/null 2>&1
cat gpurun_out/abv/ab.log >> $O/summary.log
timeout 300 python tools/probe_text_cost.py > $O/probe_text.log 2>&1; tail -n 5 $O/probe_text.log >> $O/summary.log
cat $O/summary.log
