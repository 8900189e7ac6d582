#!/bin/bash
# round-2 GPU session D: slab-resident forward / dQ / dKV kernels (8 waves), fine-tuning heads, golden zero-shot; timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attn2 or finetune or adam or segment or vq" > $O/t_kernels.log 2>&1; echo "kernels rc=$?" >> $O/summary.log
timeout 600 python -m pytest tests/test_finetune_gpu.py tests/test_zz_zero_shot_gpu.py -q -s > $O/t_ft.log 2>&1; echo "finetune+zeroshot rc=$?" >> $O/summary.log
timeout 120 python tools/bench_attn2_fwd.py 20 all > $O/abl_product.json 2>> $O/abl.err
for m in 1 3 12 15 32; do
  CTCLIP_LIB=ct_clip_amd/libctclip_attn2_slab_abl$m.so timeout 120 python tools/bench_attn2_fwd.py 20 fwd >> $O/abl.jsonl 2>> $O/abl.err
done
timeout 200 python tools/bench_ops.py attn2 10 > $O/ops_attn2.json 2> $O/ops_attn2.err
timeout 1200 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all gpu tests rc=$?" >> $O/summary.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
for f in t_kernels t_ft t_all; do echo "== $f"; tail -n 8 $O/$f.log; done
cat $O/summary.log $O/abl_product.json $O/abl.jsonl; python -c "
import json;d=json.load(open('$O/ops_attn2.json'));print({k:v['avg_us'] for k,v in d.items()})
b=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(b['ms_per_step'],b['value'],b['attn_block'])"
tail -n 3 $O/abl.err $O/bench.err
