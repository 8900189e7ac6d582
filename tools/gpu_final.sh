#!/bin/bash
# The round's record session: everything the round's profiles/ files are copied from, in one lease.
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_run.sh tests smoke soak=30 bench prof
O=gpurun_out/run
timeout 600 python bench.py --workload lipro --batch 16 --spatial-depth 4 --temporal-depth 4 > $O/lipro.json 2> $O/lipro.err
timeout 600 python bench.py --workload vocabfine --spatial-depth 4 --temporal-depth 4 > $O/vocab.json 2> $O/vocab.err
CTCLIP_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --no-attn-block --profile-steps 0 > $O/single_rank.json 2> $O/single_rank.err
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" > $O/full_size.log
tail -3 $O/full_size.log
