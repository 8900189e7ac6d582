#!/bin/bash
# closing session: PEG kernel tests (D3 = 8 routing), per-kernel bit-reproducibility soak incl. the round's new kernels, a long in-situ
# determinism trace, one more default bench line on this box -> gpurun_out/soak
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/soak; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "peg" > $O/t_k.log 2>&1; echo "peg kernel tests rc=$? $(tail -n 1 $O/t_k.log)" >> $O/summary.log
timeout 900 python tools/soak_kernels.py 200 > $O/soak.log 2>&1; echo "soak rc=$?" >> $O/summary.log
timeout 900 python tools/trace_determinism.py --runs 400 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?" >> $O/summary.log
cat $O/summary.log; cat $O/soak.log | tail -n 40; tail -n 1 $O/bench_default.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['value'], b['roofline']['frac'], b['attn_block']['fwd_us'], b['attn_block']['fwd_bwd_us'])"
