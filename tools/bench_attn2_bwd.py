"""Event-timed backward of the CTViT spatial attention at the bench shape (B = 8: 192 sequences x 8 heads x 576 tokens): the one-pass kernel
(ctclip_attn2_bwd_fused) against the path it replaces (ctclip_attn2_bwd_tok = query pass + key pass + dBias pass + folds + q un-prep).
python tools/bench_attn2_bwd.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ct_clip_amd import backend


def main():
    stamps = "--stamps" in sys.argv
    if stamps:
        os.environ["CTCLIP_BWD1_STAMPS"] = "1"
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(args[0]) if args else 20
    be = backend.get()
    dev = torch.device("cuda:0")
    nseq, H, gh, gw, D = 192, 8, 24, 24, 32
    L, M, HD = gh * gw, 192 * 576, 256
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    q, kv = rn(M, HD).bfloat16(), rn(M, 2 * HD).bfloat16()
    qs, ks = 1.0 + 0.2 * rn(D), 1.0 + 0.2 * rn(D)
    tab = rn((2 * gh - 1) * (2 * gw - 1), H, sc=0.5)
    qh, kh, vh, qinv, kinv = be.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    o, lse2 = be.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L)
    do = rn(M, HD, sc=1e-3).bfloat16()
    dq, dkv = torch.empty(M, HD, dtype=torch.bfloat16, device=dev), torch.empty(M, 2 * HD, dtype=torch.bfloat16, device=dev)
    dqs, dks = torch.zeros(D, device=dev), torch.zeros(D, device=dev)

    def old():
        return be.attn2_bwd_tok(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, qinv, kinv, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks, nseq, L, os.environ.get("BENCH_NO_DTAB") is None)

    def new():
        return be.attn2_bwd_fused(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, qinv, kinv, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks, nseq, L, os.environ.get("BENCH_NO_DTAB") is None)

    out = {}
    for name, fn in (("three_pass_us", old), ("one_pass_us", new)):
        if stamps and name == "one_pass_us":       # the per-step sums accumulate: start from zero
            nws = be.lib.ctclip_attn2_bwd_fused_workspace(nseq, H, L, gh, gw)
            be.workspace(dev, nws)[nws - 4096:nws].zero_()
        for _ in range(3):
            r = fn()
        if r is None:
            out[name] = None
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) * 1e3 / reps, 1)
    if stamps:      # phase clocks (100 MHz) of workgroup 0, wave 0: see BWD1_STAMP in csrc/attn2_bwd1.hip
        n = be.lib.ctclip_attn2_bwd_fused_workspace(nseq, H, L, gh, gw)
        st = be.workspace(dev, n)[n - 4096:n].view(torch.int64).cpu().reshape(-1, 16)
        if os.environ.get("CTCLIP_ATTN_BWD2", "1") != "0":      # the four-wave form (csrc/attn2_bwd2.hip): its own phase boundaries
            spans2 = [("loads consumed + barrier", 0, 1), ("triples + barrier (load phase end)", 1, 2), ("tile loop (wave 0)", 2, 3), ("barrier", 3, 4),
                      ("dQ hand-over + barrier", 4, 5), ("q un-prep", 5, 6), ("table flush + barrier", 6, 7)]
            rows = []
            for it in range(6):
                t = st[it].tolist()
                row = {nm: round((t[b] - t[a]) / 100.0, 2) for nm, a, b in spans2}
                bs = st.reshape(-1)[128 + it * 32:128 + it * 32 + 27].tolist()
                row["sweep_us_per_block"] = [round((bs[3 * k + 1] - bs[3 * k]) / 100.0, 2) for k in range(9)]
                row["block_end_us"] = [round((bs[3 * k + 2] - bs[3 * k + 1]) / 100.0, 2) for k in range(9)]
                rows.append(row)
            out["phases_us_per_item"] = rows
            ts = st.reshape(-1)[128 + 192:128 + 192 + 27].tolist()      # BWD2_TSTAMPS builds: shader clocks at tile start / after phase B / after phase C
            if any(ts):
                out["tile_phase_cycles_block1"] = [[ts[3 * k + 1] - ts[3 * k], ts[3 * k + 2] - ts[3 * k + 1], (ts[3 * k + 3] - ts[3 * k + 2]) if k < 8 else 0] for k in range(9)]
            print(json.dumps(out))
            return
        spans = [("loads issued+consumed", 0, 1), ("dO'' written (load phase end)", 1, 2), ("tile steps", 2, 3), ("parked stores drained + barrier", 3, 4),
                 ("dq un-prep + next item's L2 touches", 4, 7), ("table flush + barrier", 7, 8)]
        rows = []
        for it in range(6):
            t = st[it].tolist()
            rows.append({nm: round((t[b] - t[a]) / 100.0, 2) for nm, a, b in spans} | {"tile-counter wait in steps (wave 0)": round(t[9] / 100.0, 2)})
        out["phases_us_per_item"] = rows
        flat = be.workspace(dev, n)[n - 4096:n].view(torch.int64).cpu()
        per = flat[128:128 + 8 * 48].reshape(8, 48).double() / 100.0 / (reps + 3) / 6       # us per item (sums over every launch since the buffer was zeroed)
        out["tile_wait_us_per_step"] = {f"wave{w}": [round(float(v), 2) for v in per[w][:41]] for w in range(7)}
        out["step_us_wave0"] = [round(float(v), 2) for v in per[7][:41]]
    flops = 2.0 * nseq * H * L * L * D * 5          # five matrix products per score tile
    if out.get("one_pass_us"):
        out["one_pass_mfma_frac"] = round(flops / (out["one_pass_us"] * 1e-6) / 2.5e15, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
