"""Per-function instruction statistics of a gfx950 assembly dump (hipcc -S --cuda-device-only): scratch traffic, MFMA / exp / AGPR-copy counts."""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r'\n(?=_Z[^\n]*:\s)', s)
R = dict(scratch=r'scratch_(load|store)', mfma=r'v_mfma', accvgpr=r'v_accvgpr', exp=r'v_exp_f32', ds=r'\bds_', waitcnt=r's_waitcnt', nop=r's_nop',
         vmem=r'\b(global|buffer|flat)_', valu=r'^\tv_(?!mfma)')
for f in funcs:
    name = f.split('\n', 1)[0][:100]
    if not name.startswith('_Z') or pat not in name:
        continue
    body = f.split('.Lfunc_end')[0]
    ins = [l for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    print(name)
    print("   instructions", len(ins), " ".join(f"{k} {len([l for l in ins if re.search(r, l)])}" for k, r in R.items()))
    out = []
    for k in ("NumVgprs", "NumAgprs", "TotalNumVgprs", "ScratchSize", "Occupancy"):
        m = re.search(r'; ' + k + r': (\d+)', f)
        if m:
            out.append(f"{k} {m.group(1)}")
    print("   " + " ".join(out))
