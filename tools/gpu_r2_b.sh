#!/bin/bash
# round-2 GPU session B: attn2 after the fragment-wait fix + slab-resident forward; full-size parity; op timings; short bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attn2 or segment or vq or bert_embed" > $O/t_new_kernels.log 2>&1; echo "new kernels rc=$?" >> $O/summary.log
CTCLIP_ATTN_SLAB=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attn2" > $O/t_attn2_ring.log 2>&1; echo "attn2 ring rc=$?" >> $O/summary.log
timeout 300 python tools/bench_ops.py attn2 10 > $O/ops_attn2_slab.json 2> $O/ops_attn2_slab.err
CTCLIP_ATTN_SLAB=0 timeout 300 python tools/bench_ops.py attn2 10 > $O/ops_attn2_ring.json 2> $O/ops_attn2_ring.err
timeout 1200 python -m pytest tests/test_full_size_gpu.py -q -s > $O/t_full.log 2>&1; echo "full size rc=$?" >> $O/summary.log
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_full_size_gpu.py > $O/t_all.log 2>&1; echo "all gpu tests rc=$?" >> $O/summary.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
for f in t_new_kernels t_attn2_ring t_full t_all; do echo "== $f"; tail -n 6 $O/$f.log; done
cat $O/summary.log $O/ops_attn2_slab.json; grep -A3 "attn2_fwd" $O/ops_attn2_ring.json; cat $O/bench.json
