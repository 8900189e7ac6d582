#!/bin/bash
# PMC passes (own runs, --kernel-trace only next to --pmc) over tools/bench_gemm_shapes.py (SHAPES_ONLY=ff: the feed-forward in-projection shapes --
# NT forward N = 2816 / K = 512, NT grad-input N = 512 / K = 2816, TN weight gradient K = 110592 -- and the fused GEGLU launches) and the text tower's
# shapes (tools/bench_gemm_sm.py).  -> gpurun_out/pmc_gemm/summary.md
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_gemm; mkdir -p $O
export TMPDIR=/tmp SHAPES_ONLY=ff
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_gemm_shapes.py 4"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- $CMD > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- $CMD > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc3 -- $CMD > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc3.err
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc*/*/*.db
python - <<'PY' | tee gpurun_out/pmc_gemm/summary.md
import csv, glob, collections, re
tot = collections.defaultdict(dict)
for d in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"gpurun_out/pmc_gemm/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            n = re.sub(r"^void ", "", n).split("(")[0]
            if not n.startswith(("gemm_nt", "gemm_tn", "gemm_sm", "tn_reduce", "geglu")):
                continue
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            key = f"{n} ~{round(us / 20) * 20} us"          # (the same kernel serves several shapes: bucket by duration)
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["us"].append(us)
    for n, c in acc.items():
        for k, v in c.items():
            tot[n][k if k != "us" else f"us_{d}"] = sum(v) / len(v)
            if k == "us":
                tot[n]["launches"] = len(v)
print("| kernel (bucketed by duration) | us | MFMA busy % | VALU busy % of wave cycles | LDS busy % | VMEM busy % | wave parked (WAIT_ANY) % | issue stall (WAIT_INST_ANY) % | LDS bank-conflict % of LDS cycles | HBM MB read (x2 corrected) + written |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for n, c in sorted(tot.items()):
    g = lambda k: c.get(k)
    mf = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE") else float("nan")
    wc = g("SQ_WAVE_CYCLES") or float("nan")
    pct = lambda k: 100 * g(k) / wc if g(k) is not None else float("nan")
    bc = 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan")
    mb = (g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024) / 1e6 if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None else float("nan")
    vm = 100 * g("SQ_ACTIVE_INST_VMEM") / (g("SQ_ACTIVE_INST_ANY") + 1) if g("SQ_ACTIVE_INST_VMEM") is not None and g("SQ_ACTIVE_INST_ANY") else float("nan")
    print(f"| `{n}` | {c.get('us_pmc2', c.get('us_pmc1', 0)):.1f} | {mf:.1f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | {pct('SQ_ACTIVE_INST_LDS'):.1f} | {vm:.1f} | {pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | {bc:.1f} | {mb:.0f} |")
PY
