#!/bin/bash
# GPU session: new kernel tests + full-size parity with the compensated residual stream, same-box A/B of the step (comp on / off),
# determinism trace, fused VocabFine bench -> gpurun_out/s3
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "residual_comp or peg_fwd_comp or peg_fwd_bwd" > $O/t_kernels.log 2>&1; echo "kernel tests rc=$? $(tail -n 1 $O/t_kernels.log)" >> $O/summary.log
timeout 1200 python -m pytest tests/test_full_size_gpu.py -q -s > $O/t_full.log 2>&1; echo "full-size tests rc=$? $(tail -n 1 $O/t_full.log)" >> $O/summary.log
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('A compensated stream', r['ms_per_step'], r['loss'])" >> $O/ab.log
  CTCLIP_RESIDUAL_COMP=0 timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B plain bf16 stream ', r['ms_per_step'], r['loss'])" >> $O/ab.log
done
timeout 600 python tools/trace_determinism.py --runs 200 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
timeout 600 python bench.py --workload vocabfine > $O/bench_vocabfine.json 2> $O/bench_vocabfine.err; echo "vocabfine (fused) bench rc=$?" >> $O/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
cat $O/summary.log $O/ab.log; grep -h "FAILED\|^E  " $O/t_kernels.log $O/t_full.log | head -20; grep -E "\[full|   [st][0-9]+_in|VQ code agreement:" $O/t_full.log | cut -c1-300; tail -n 4 $O/smoke.log; cut -c1-300 $O/bench_vocabfine.json
