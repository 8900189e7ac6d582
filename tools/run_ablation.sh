# usage: bash tools/run_ablation.sh hip abl1 abl4s5 ...   (libraries built by tools/build_ablation.py)
for m in "$@"; do
  echo "== $m"; GEMM_CASES=NT CTCLIP_LIB=ct_clip_amd/libctclip_$m.so python tools/bench_gemm.py 10 2>&1 | grep -E "avg_us|rror" 
done
