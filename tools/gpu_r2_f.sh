#!/bin/bash
# round-2 GPU session F: PEG marching kernels (parity, timing against the first generation), similarity head / VocabFine, step time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "peg or latent_similarity" > $O/t_peg.log 2>&1; echo "peg tests rc=$?" >> $O/summary.log
timeout 300 python tools/bench_ops.py peg 10 > $O/ops_peg_lds.json 2> $O/ops_peg.err
CTCLIP_PEG_LDS=0 timeout 300 python tools/bench_ops.py peg 10 > $O/ops_peg_old.json 2>> $O/ops_peg.err
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all gpu tests rc=$?" >> $O/summary.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
tail -n 6 $O/t_peg.log; tail -n 8 $O/t_all.log; cat $O/summary.log; cat $O/ops_peg_lds.json $O/ops_peg_old.json; tail -n 2 $O/ops_peg.err
python -c "
import json
b=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(b['ms_per_step'],b['value'],b.get('attn_block'))"
