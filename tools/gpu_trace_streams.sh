#!/bin/bash
# Kernel traces (rocprofv3 --kernel-trace, timestamps + HSA queue + stream per dispatch) of the training step under several environments, one
# tools/trace_timeline.py summary each:   tools/gpu_trace_streams.sh "ENV_A" "ENV_B" ...      ("-" = no extra environment)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/trace; mkdir -p $O; rm -f $O/summary.md
export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i + 1)); L="$E"; [ "$E" = "-" ] && E="X_NONE=1"
  (cd /tmp && env $E timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t$i -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 \
     --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth --no-text512 > $O/t$i.json 2> $O/t$i.err)
  python tools/trace_timeline.py $O/t$i "[$L] $(python -c "import json,sys; print(json.loads(open('$O/t$i.json').read().strip().splitlines()[-1])['ms_per_step'], 'ms/step under the tracer')" 2>/dev/null)" $O/t$i.csv.gz >> $O/summary.md 2>&1
  rm -rf $O/t$i
done
cat $O/summary.md | cut -c1-400 | head -120
