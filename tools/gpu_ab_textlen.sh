#!/bin/bash
# Same-box A/B of the BERT attention kernels (CTCLIP_ATTN_LDS=1: workgroup-shared LDS tiles, 0: the register-only first generation) at
# T = 128 and T = 512, interleaved twice.  -> gpurun_out/ab/textlen.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab; mkdir -p $O; : > $O/textlen.txt
for r in 1 2; do
  for T in 128 512; do
    for L in 1 0; do
      CTCLIP_ATTN_LDS=$L timeout 600 python bench.py --steps 20 --warmup 3 --text-len $T --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --no-attn-block --profile-steps 0 > $O/t.json 2> $O/t.err
      python - <<PY | tee -a $O/textlen.txt
import json
try:
    b=json.loads(open("$O/t.json").read().strip().splitlines()[-1]); print("rep $r T=$T CTCLIP_ATTN_LDS=$L", b["ms_per_step"], "ms/step", b["value"], "volumes/s loss", b["loss"])
except Exception as e:
    print("failed", e); print(open("$O/t.err").read()[-800:])
PY
    done
  done
done
