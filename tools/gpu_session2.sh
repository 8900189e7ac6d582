#!/bin/bash
# GPU session: whole GPU suite, smoke (bf16 and f32 text tower), same-box A/B of the step time (text tower dtype; counted epilogue wait
# build), the fine-tuning workloads of bench.py -> gpurun_out/s2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > $O/t_all.log 2>&1; echo "all gpu tests rc=$? $(tail -n 1 $O/t_all.log)" >> $O/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
CTCLIP_TEXT_DTYPE=f32 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_textf32.log 2>&1; echo "smoke(text f32) rc=$?" >> $O/summary.log
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('A default        ', r['ms_per_step'], r['loss'])" >> $O/ab.log
  CTCLIP_TEXT_DTYPE=f32 timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B text tower f32 ', r['ms_per_step'], r['loss'])" >> $O/ab.log
  if [ -f ct_clip_amd/libctclip_epi.so ]; then
    CTCLIP_LIB=ct_clip_amd/libctclip_epi.so timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C counted epilogue', r['ms_per_step'], r['loss'])" >> $O/ab.log
  fi
done
if [ -f ct_clip_amd/libctclip_epi.so ]; then
  CTCLIP_LIB=ct_clip_amd/libctclip_epi.so timeout 600 python tools/trace_determinism.py --runs 300 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
  CTCLIP_LIB=ct_clip_amd/libctclip_epi.so timeout 600 python tools/trace_determinism.py --runs 30 --config bench 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
fi
timeout 600 python bench.py --workload lipro > $O/bench_lipro.json 2> $O/bench_lipro.err; echo "lipro bench rc=$?" >> $O/summary.log
timeout 900 python bench.py --workload vocabfine > $O/bench_vocabfine.json 2> $O/bench_vocabfine.err; echo "vocabfine bench rc=$?" >> $O/summary.log
cat $O/summary.log $O/ab.log; grep -h "FAILED\|^E  " $O/t_all.log | head -20; tail -n 3 $O/smoke.log $O/smoke_textf32.log; cut -c1-500 $O/bench_lipro.json $O/bench_vocabfine.json; tail -n 3 $O/bench_lipro.err $O/bench_vocabfine.err
