cd "$GRAFT_REPO_ROOT"; O=gpurun_out/abg; mkdir -p $O
for r in 1 2; do for m in eager graph; do
  fl=""; [ $m = graph ] && fl="--graph"
  timeout 900 python bench.py $fl --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-text512 --no-attn-block --profile-steps 0 > $O/$m$r.json 2> $O/$m$r.err
  python - <<PY
import json
try:
    b=json.loads(open("$O/$m$r.json").read().strip().splitlines()[-1]); print("$m$r", b["ms_per_step"], b["value"], b["config"].get("launch","")[:20], "| 4+4", b["reference_depth_4+4"]["ms_per_step"], b["reference_depth_4+4"].get("launch"))
except Exception as e: print("$m$r failed", e); print(open("$O/$m$r.err").read()[-800:])
PY
done; done
