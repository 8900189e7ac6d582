cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p512; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && CTCLIP_WGRAD_STREAM=0 CTCLIP_TEXT_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --text-len 512 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth --no-text512 > $GRAFT_REPO_ROOT/$O/b.json 2> $GRAFT_REPO_ROOT/$O/b.err)
python - <<'PY'
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/p512/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:60]:
    if any(k in n for k in ("gemm_sm", "attn_fwd", "attn_bwd", "attn64", "attn_delta", "head_transpose", "layernorm_fwd_kernel<float", "layernorm_bwd_kernel<float", "dropout", "gelu", "convert_pad", "colsum", "tn_reduce", "transpose2d", "accumulate", "gemm_kernel", "bert", "seg_")):
        print(f"{n[:90]:90s} calls {len(v):5d} total {sum(v)/1e3:8.2f} ms avg {sum(v)/len(v):8.1f} us max {max(v):8.1f}")
print("total", tot/1e3)
PY
rm -rf $O/prof
