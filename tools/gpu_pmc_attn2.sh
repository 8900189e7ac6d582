#!/bin/bash
# PMC passes (own runs, --kernel-trace only next to --pmc) over tools/bench_attn2_bwd.py: the one-pass attention backward (bwd1_kernel) next to
# the three-pass kernels it replaces and the forward, at the bench shape.  -> gpurun_out/pmc_attn2/summary.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_attn2; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_attn2_bwd.py 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_attn2_bwd.py 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc3 -- python $GRAFT_REPO_ROOT/tools/bench_attn2_bwd.py 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc3.err
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc*/*/*.db
python - <<'PY' | tee gpurun_out/pmc_attn2/summary.txt
import csv, glob, collections, re
tot = collections.defaultdict(dict)
for d in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"gpurun_out/pmc_attn2/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(bwd1_kernel|bwd2_kernel|\w+_slab_kernel|dbias_\w+|attn_unprep_kernel|bwd1_\w+_kernel)", r["Kernel_Name"])
            if not m: continue
            n = m.group(1)
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[n]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, c in acc.items():
        for k, v in c.items():
            tot[n][k if k != "us" else f"us_{d}"] = sum(v) / len(v)
print("| kernel | us | MFMA busy % (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)) | VALU busy % of wave cycles | LDS busy % | wait_any % | wait_inst_lds % | LDS bank-conflict % of LDS cycles | HBM MB read (x2 corrected) + written |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for n, c in sorted(tot.items()):
    g = lambda k: c.get(k)
    mf = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE") else float("nan")
    wc = g("SQ_WAVE_CYCLES") or float("nan")
    pct = lambda k: 100 * g(k) / wc if g(k) is not None else float("nan")
    bc = 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan")
    mb = (g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024) / 1e6 if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None else float("nan")
    wil = 100 * g("SQ_WAIT_INST_LDS") / (g("SQ_ACTIVE_INST_ANY") + 1) if g("SQ_WAIT_INST_LDS") is not None and g("SQ_ACTIVE_INST_ANY") else float("nan")
    print(f"| `{n}` | {c.get('us_pmc2', c.get('us_pmc1', 0)):.1f} | {mf:.1f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | {pct('SQ_ACTIVE_INST_LDS'):.1f} | {pct('SQ_WAIT_ANY'):.1f} | {wil:.1f} | {bc:.1f} | {mb:.0f} |")
PY
