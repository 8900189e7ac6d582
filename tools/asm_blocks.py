"""Per-basic-block instruction mix of one function of a gfx950 assembly dump: python tools/asm_blocks.py file.s <function substring> [min instrs] [dump block label]"""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dump = sys.argv[4] if len(sys.argv) > 4 else None
f = [x for x in re.split(r'\n(?=_Z[^\n]*:\s)', s) if x.startswith('_Z') and pat in x.split('\n', 1)[0]][0]
body = f.split('.Lfunc_end')[0]
blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', body)
R = dict(mfma=r'v_mfma', acc=r'v_accvgpr', exp=r'v_exp_f32', ds=r'\bds_', wait=r's_waitcnt', nop=r's_nop', vmem=r'\b(global|buffer|flat|scratch)_',
         valu=r'^\tv_(?!mfma|accvgpr)', salu=r'^\ts_(?!waitcnt|nop)')
for b in blocks:
    name = b.split('\n', 1)[0][:30]
    ins = [l for l in b.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    if dump and name.startswith(dump):
        for l in ins:
            print(l.split(';')[0].rstrip()[:120])
        continue
    if len(ins) < minn or dump:
        continue
    print(name, len(ins), {k: len([l for l in ins if re.search(r, l)]) for k, r in R.items()})
