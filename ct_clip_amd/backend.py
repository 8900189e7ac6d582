"""Primitive ops of the CT-CLIP hot path, as thin wrappers over the C-ABI HIP library.

Every function takes/returns torch tensors that already live on the GPU; PyTorch is used only for memory
(caching allocator), streams and autograd bookkeeping.  Kernels are launched on torch's current stream.
``get()`` returns the active backend.  The only implementation shipped is the HIP one -- importing it
without the built library raises ImportError.  ``tests/ref_backend.py`` holds a pure-torch *checker* with
the same interface that the CPU tests swap in (via ``use()``) to validate the host-side composition and
the hand-derived backward formulas against the oracle; it is test infrastructure, never a fallback.
"""
import torch

from . import _lib

F32, BF16 = 0, 1


def dcode(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rowmajor(t, what):
    assert t.dim() == 2 and t.stride(1) == 1, f"{what}: need a 2-D row-major view, got {tuple(t.shape)} {t.stride()}"
    return t.stride(0)


def price_gemm_group(key, ms, peak_tflops=2500.0, peak_gbps=8000.0):
    """One GEMM launch group of the per-launch timing (key = (layout, dtype, M, N, K, side stream[, fused variant]), ms = event times)
    against both floors: algorithmic FLOPs / dense MFMA peak and algorithmic bytes / HBM peak; the larger floor names the bound."""
    layout, dt, M, N, K, side = key[:6]
    variant = key[6] if len(key) > 6 else ""       # fused GEGLU launches: forward (u and / or g written), backward
    flops = 2.0 * M * N * K
    esz = 2 if dt == "bf16" else 4
    nbytes = float((M * K + N * K) * esz + M * N * (4 if layout == "TN" else esz))
    if variant == "+geglu":
        nbytes += M * (N // 2) * esz                       # u and g
    elif variant == "+geglu(g only)":
        nbytes = float((M * K + N * K + M * (N // 2)) * esz)
    elif variant == "geglu-bwd recompute":
        nbytes += M * (N // 2) * esz                       # dg read, du written
    elif variant == "+geglu-bwd":
        nbytes = float((M * K + N * K + 4 * M * N) * esz)  # dy, W_out^T, u = [x | gate] read, du written (N = hp)
    avg_us = sum(ms) / len(ms) * 1e3
    peak_tf = peak_tflops if dt == "bf16" else 157.3
    t_mfma, t_hbm = flops / (peak_tf * 1e12) * 1e6, nbytes / (peak_gbps * 1e9) * 1e6
    return dict(kernel=f"gemm_kernel<{dt},{layout}> M={M} N={N} K={K}" + (f" [{variant}]" if variant else ""), launches=len(ms),
                side_stream=side, avg_us=avg_us, total_ms=sum(ms), flops_per_launch=flops, bytes_per_launch=nbytes,
                tflops=flops / (avg_us * 1e-6) / 1e12, gbps=nbytes / (avg_us * 1e-6) / 1e9, peak_tflops=peak_tf,
                bound="mfma" if t_mfma >= t_hbm else "hbm", frac=max(t_mfma, t_hbm) / avg_us,
                spec=f"{layout} {M} {N} {K} {variant.replace(' ', '_') or '-'}")


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._ws = {}
        self._gemm_events = None
        self._shadow_tables = {}

    # ------------------------------------------------------------------ per-launch timing (bench.py roofline)
    def start_gemm_timing(self):
        """Record an event pair on the launch stream around every ctclip_gemm launch until stop_gemm_timing()."""
        self._gemm_events = {}

    def stop_gemm_timing(self, peak_tflops=2500.0, peak_gbps=8000.0):
        """-> the roofline record of the GEMM launch group with the largest total time on the main stream.  Each group is priced
        against BOTH floors (algorithmic FLOPs / dense MFMA peak, algorithmic bytes / HBM peak); the larger floor names its bound:
        the fused launches (GEGLU epilogues) move 1.0-1.4 GB per launch and are HBM-bound, the plain feed-forward GEMMs MFMA-bound."""
        ev, self._gemm_events = self._gemm_events, None
        torch.cuda.synchronize()
        groups = [price_gemm_group(key, [a.elapsed_time(b) for a, b in pairs], peak_tflops, peak_gbps) for key, pairs in ev.items()]
        # launches on the text tower's side stream wait for the image tower's kernels between their two events: their event time is
        # not kernel time, so the dominant kernel is picked among the launches of the main stream
        groups.sort(key=lambda g: (g["side_stream"], -g["total_ms"]))
        top = groups[0]
        mf = top["bound"] == "mfma"
        return dict(bound=top["bound"], kernel=top["kernel"], achieved=round(top["tflops"] if mf else top["gbps"], 1),
                    peak=top["peak_tflops"] if mf else peak_gbps, unit="TFLOP/s" if mf else "GB/s", frac=round(top["frac"], 4), traffic=None,
                    launches=top["launches"], avg_us=round(top["avg_us"], 1), tflops=round(top["tflops"], 1), algorithmic_GBps=round(top["gbps"], 1),
                    algorithmic_flops_per_launch=top["flops_per_launch"], algorithmic_bytes_per_launch=top["bytes_per_launch"],
                    probe_spec=top["spec"], gemm_total_ms=round(sum(g["total_ms"] for g in groups if not g["side_stream"]), 2),
                    top5=[dict(kernel=g["kernel"], launches=g["launches"], avg_us=round(g["avg_us"], 1), tflops=round(g["tflops"], 1),
                               GBps=round(g["gbps"], 1), bound=g["bound"], frac=round(g["frac"], 3)) for g in groups[:5]])

    # ------------------------------------------------------------------ helpers
    def workspace(self, device, nbytes):
        key = (device.index, _stream())
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._ws[key] = buf
        return buf

    # ------------------------------------------------------------------ GEMM
    def gemm(self, a, b, *, a_kc=True, b_kc=True, bias=None, residual=None, out=None, out_dtype=None,
             accumulate=False, alpha=1.0, split_k=1, M=None, N=None, K=None):
        """out[m,n] = alpha * sum_k A(m,k) B(n,k) + bias[n] + residual[m,n] (+ out).  A = a if a_kc else a^T, same for b."""
        lda, ldb = _rowmajor(a, "gemm a"), _rowmajor(b, "gemm b")
        m_, k_ = (a.shape if a_kc else (a.shape[1], a.shape[0]))
        n_, k2 = (b.shape if b_kc else (b.shape[1], b.shape[0]))
        M = m_ if M is None else M
        N = n_ if N is None else N
        K = min(k_, k2) if K is None else K
        assert a.dtype == b.dtype
        if out is None:
            out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
            assert not accumulate
        ldc = _rowmajor(out, "gemm out")
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() >= N
        ldr = _rowmajor(residual, "gemm residual") if residual is not None else 0
        timing = self._gemm_events
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ws, wsn = None, 0
        if split_k != 1 and out.dtype == torch.float32:
            wsn = self.lib.ctclip_gemm_workspace(M, N, K, dcode(a.dtype), int(split_k))
            if wsn:
                ws = self.workspace(a.device, wsn)
        rc = self.lib.ctclip_gemm(_p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, lda, ldb, ldc, ldr,
                                  int(a_kc), int(b_kc), dcode(a.dtype), dcode(out.dtype),
                                  dcode(residual.dtype) if residual is not None else 0, int(accumulate), int(split_k),
                                  float(alpha), _p(ws), ws.numel() if ws is not None else 0, _stream())
        if timing is not None:
            e1.record()
            key = ("NT" if a_kc and b_kc else "NN" if a_kc else "TN", "bf16" if a.dtype == torch.bfloat16 else "f32", M, N, K,
                   torch.cuda.current_stream() != torch.cuda.default_stream())
            timing.setdefault(key, []).append((e0, e1))
        _lib.check(rc, "ctclip_gemm")
        return out

    def gemm_nt2_select(self, mask):
        """Which epilogue families of the big bf16 NT GEMMs run on the two-workgroups-per-CU kernel (csrc/gemm_nt2.hip): bit 0 plain /
        residual, bit 1 GEGLU forward, bit 2 GEGLU backward; -1 = environment / built-in default.  Returns the previous mask."""
        return int(self.lib.ctclip_gemm_nt2_select(int(mask)))

    def gemm_dw_db(self, dy, x, dw, db, accumulate=True):
        """dw (N_out, K_in) f32 (+)= dy^T x and db (N_out) f32 (+)= column sums of dy in ONE launch (csrc/gemm_sm.hip); dy (T, N_out), x (T, K_in) bf16,
        possibly column views.  False when the shape is not served (the caller composes gemm + colsum)."""
        T, n_out = dy.shape
        k_in = x.shape[1]
        if (dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or dw.dtype != torch.float32 or db.dtype != torch.float32 or not db.is_contiguous()
                or dy.stride(1) != 1 or x.stride(1) != 1 or dw.stride(1) != 1 or tuple(dw.shape) != (n_out, k_in) or db.numel() != n_out):
            return False
        rc = self.lib.ctclip_gemm_dw_db(_p(dy), _p(x), _p(dw), _p(db), T, n_out, k_in, dy.stride(0), x.stride(0), dw.stride(0), int(accumulate), _stream())
        if rc == -2:
            return False
        _lib.check(rc, "ctclip_gemm_dw_db")
        return True

    def gemm_argmax(self, a, b):
        M, K = a.shape
        N = b.shape[0]
        idx = torch.empty(M, dtype=torch.int64, device=a.device)
        val = torch.empty(M, dtype=torch.float32, device=a.device)
        nbytes = self.lib.ctclip_gemm_argmax_workspace(M, N)
        ws = self.workspace(a.device, nbytes)
        rc = self.lib.ctclip_gemm_argmax(_p(a), _p(b), _p(idx), _p(val), M, N, K, _rowmajor(a, "a"), _rowmajor(b, "b"),
                                         dcode(a.dtype), _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_gemm_argmax")
        return idx, val

    @staticmethod
    def gemm_argmax_hilo_ok(a, ncodes):
        """shapes ctclip_gemm_argmax_hilo serves (the persistent NT kernel): bf16 row-major tokens, K a multiple of 64, at least 160 tiles of 256 x 256"""
        M, d = a.shape
        return (a.dtype == torch.bfloat16 and d % 64 == 0 and a.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0
                and ((M + 255) // 256) * ((ncodes + 255) // 256) >= 160)

    def gemm_argmax_hilo(self, a, b2):
        """arg-max over codes c of a[m] . (b2[c, :K] + b2[c, K:]): RAW bf16 tokens (M, K) against the unit codebook's [hi | lo] rows (C, 2 K)
        (l2norm_split3 order 2).  -> (idx int64 (M,), val f32 (M,) = the winning dot product, not normalised by |a[m]|)"""
        M, K = a.shape
        C = b2.shape[0]
        assert b2.shape[1] == 2 * K and b2.dtype == torch.bfloat16 and a.dtype == torch.bfloat16
        idx = torch.empty(M, dtype=torch.int64, device=a.device)
        val = torch.empty(M, dtype=torch.float32, device=a.device)
        ws = self.workspace(a.device, self.lib.ctclip_gemm_argmax_workspace(M, C))
        rc = self.lib.ctclip_gemm_argmax_hilo(_p(a), _p(b2), _p(idx), _p(val), M, C, K, _rowmajor(a, "a"), _rowmajor(b2, "b2"), _p(ws), ws.numel(),
                                              _stream())
        _lib.check(rc, "ctclip_gemm_argmax_hilo")
        return idx, val

    # ------------------------------------------------------------------ norms
    def layernorm_fwd(self, x, gamma, beta, eps, want_stats=True):
        rows, cols = x.shape
        assert x.is_contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
        rc = self.lib.ctclip_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols, float(eps),
                                           dcode(x.dtype), _stream())
        _lib.check(rc, "ctclip_layernorm_fwd")
        return y, mean, rstd

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dgamma=None, dbeta=None, add1=None, add2=None):
        rows, cols = x.shape
        assert x.is_contiguous() and dy.is_contiguous()
        assert all(a is None or (a.is_contiguous() and a.shape == x.shape and a.dtype == x.dtype) for a in (add1, add2))
        dx = torch.empty_like(x)
        nbytes = self.lib.ctclip_layernorm_bwd_workspace(rows, cols)
        ws = self.workspace(x.device, nbytes)
        rc = self.lib.ctclip_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta), _p(add1),
                                           _p(add2), rows, cols, dcode(x.dtype), _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_layernorm_bwd")
        return dx

    def layernorm_bwd_partials(self, dy, x, gamma, mean, rstd, add1=None, add2=None):
        """First half of layernorm_bwd: -> (dx, partials); the dgamma / dbeta fold is layernorm_bwd_reduce (possibly on another stream)."""
        rows, cols = x.shape
        assert x.is_contiguous() and dy.is_contiguous()
        assert all(a is None or (a.is_contiguous() and a.shape == x.shape and a.dtype == x.dtype) for a in (add1, add2))
        dx = torch.empty_like(x)
        nbytes = self.lib.ctclip_layernorm_bwd_workspace(rows, cols)
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        rc = self.lib.ctclip_layernorm_bwd_partials(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(add1), _p(add2), rows, cols,
                                                    dcode(x.dtype), _p(part), nbytes, _stream())
        _lib.check(rc, "ctclip_layernorm_bwd_partials")
        return dx, part

    def layernorm_bwd_reduce(self, part, dgamma, dbeta, rows, cols):
        _lib.check(self.lib.ctclip_layernorm_bwd_reduce(_p(part), _p(dgamma), _p(dbeta), rows, cols, _stream()), "ctclip_layernorm_bwd_reduce")

    def patch_ln(self, video, pt, p1, p2, kpad, eps, dtype):
        B, C, Fr, H, W = video.shape
        assert C == 1 and video.dtype == torch.float32 and video.is_contiguous()
        ntok = B * (Fr // pt) * (H // p1) * (W // p2)
        out = torch.empty((ntok, kpad), dtype=dtype, device=video.device)
        rc = self.lib.ctclip_patch_ln_fwd(_p(video), _p(out), B, Fr, H, W, pt, p1, p2, kpad, float(eps), dcode(dtype), _stream())
        _lib.check(rc, "ctclip_patch_ln_fwd")
        return out

    def l2norm_rows(self, x, out_dtype, eps=1e-12):
        rows, cols = x.shape
        ldx = _rowmajor(x, "l2norm x")
        y = torch.empty((rows, cols), dtype=out_dtype, device=x.device)
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        rc = self.lib.ctclip_l2norm_rows(_p(x), _p(y), _p(inv), rows, cols, ldx, float(eps), dcode(x.dtype), dcode(out_dtype),
                                         _stream())
        _lib.check(rc, "ctclip_l2norm_rows")
        return y, inv

    def row_inv_norms(self, x, eps=1e-12):
        """1 / max(|row|, eps) in f32 (one read of x, nothing else written)"""
        rows, cols = x.shape
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        rc = self.lib.ctclip_l2norm_split3(_p(x), None, _p(inv), rows, cols, _rowmajor(x, "row_inv_norms x"), float(eps), dcode(x.dtype), 0, _stream())
        _lib.check(rc, "ctclip_l2norm_split3")
        return inv

    def l2norm_split3(self, x, order, eps=1e-12):
        """Unit rows as the bf16 expansion [hi | hi | lo] (order 0) / [hi | lo | hi] (order 1): (rows, 3 * cols) bf16; order 2: [hi | lo],
        (rows, 2 * cols): the codebook operand of gemm_argmax_hilo.  Also returns the inverse norms."""
        rows, cols = x.shape
        y = torch.empty((rows, (2 if order == 2 else 3) * cols), dtype=torch.bfloat16, device=x.device)
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        rc = self.lib.ctclip_l2norm_split3(_p(x), _p(y), _p(inv), rows, cols, _rowmajor(x, "l2norm_split3 x"), float(eps),
                                           dcode(x.dtype), int(order), _stream())
        _lib.check(rc, "ctclip_l2norm_split3")
        return y, inv

    def segment_sum(self, keys, x, out, nseg, rowscale=None, counts=None, accumulate=False, key_mod=0):
        """out[key[r]] (+)= rowscale[r] * x[r] in ascending row order (deterministic); keys None: key(r) = r % key_mod."""
        M, d = x.shape
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == nseg * d
        assert keys is None or (keys.dtype == torch.int64 and keys.is_contiguous() and keys.numel() == M)
        ws = self.workspace(x.device, self.lib.ctclip_segment_sum_workspace(M, nseg))
        rc = self.lib.ctclip_segment_sum(_p(keys), int(key_mod), _p(x), _rowmajor(x, "segment_sum x"), _p(rowscale), _p(out), _p(counts),
                                         M, d, nseg, int(accumulate), dcode(x.dtype), _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_segment_sum")
        return out

    # ------------------------------------------------------------------ PEG
    def peg_fwd(self, x, w, bias):
        B, D1, D2, D3, C = x.shape
        assert x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
        y = torch.empty_like(x)
        rc = self.lib.ctclip_peg_fwd(_p(x), _p(w), _p(bias), _p(y), B, D1, D2, D3, C, dcode(x.dtype), _stream())
        _lib.check(rc, "ctclip_peg_fwd")
        return y

    def peg_fwd_comp(self, x, w, bias, e_in=None):
        """The PEG residual add on the compensated residual stream: s = x + e_in + conv(x) + bias (f32) -> (y, e) = (bf16(s), bf16(s - y));
        None when the marching kernels do not serve the grid (the caller then runs peg_fwd and this add carries no residue)."""
        B, D1, D2, D3, C = x.shape
        assert x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
        if x.dtype != torch.bfloat16:
            return None
        assert e_in is None or (e_in.dtype == x.dtype and e_in.is_contiguous() and e_in.numel() == x.numel())
        y, e = torch.empty_like(x), torch.empty_like(x)
        rc = self.lib.ctclip_peg_fwd_comp(_p(x), _p(w), _p(bias), _p(e_in), _p(y), _p(e), B, D1, D2, D3, C, dcode(x.dtype), _stream())
        if rc == -2:      # CTCLIP_EUNSUPPORTED
            return None
        _lib.check(rc, "ctclip_peg_fwd_comp")
        return y, e

    def gemm_residual_comp(self, a, b, residual, comp):
        """-> (y, e) = the bf16 pair of s = a @ b^T + residual + comp (f32): y = bf16(s), e = bf16(s - y); None when the large-tile kernel
        does not serve the shape."""
        M, K = a.shape
        N = b.shape[0]
        if a.dtype != torch.bfloat16 or M % 256 or N % 128 or K % 64:
            return None
        for t in (residual, comp):
            assert t.dtype == torch.bfloat16 and tuple(t.shape) == (M, N) and t.stride(1) == 1 and t.stride(0) == residual.stride(0)
        y = torch.empty((M, N), dtype=a.dtype, device=a.device)
        e = torch.empty_like(y)
        rc = self.lib.ctclip_gemm_residual_comp(_p(a), _p(b), _p(y), _p(e), _p(residual), _p(comp), M, N, K, _rowmajor(a, "a"),
                                                _rowmajor(b, "b"), N, _rowmajor(residual, "residual"), dcode(a.dtype), _stream())
        if rc == -2:      # CTCLIP_EUNSUPPORTED
            return None
        _lib.check(rc, "ctclip_gemm_residual_comp")
        return y, e

    def peg_bwd(self, dy, x, w, dw=None, db=None, want_dx=True):
        """want_dx=False: weight gradient only (returns None); dw=None: grad-input only."""
        B, D1, D2, D3, C = x.shape
        assert dy.is_contiguous() and x.is_contiguous()
        dx = torch.empty_like(x) if want_dx else None
        ws = self.workspace(x.device, self.lib.ctclip_peg_bwd_workspace(B, D1, D2, C)) if dw is not None else None
        rc = self.lib.ctclip_peg_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), B, D1, D2, D3, C, dcode(x.dtype), _p(ws),
                                     ws.numel() if ws is not None else 0, _stream())
        _lib.check(rc, "ctclip_peg_bwd")
        return dx

    # ------------------------------------------------------------------ attention
    def head_transpose(self, x, nseq, H, L, D):
        Lp = (L + 7) // 8 * 8
        xt = torch.empty((nseq, H, D, Lp), dtype=x.dtype, device=x.device)
        rc = self.lib.ctclip_head_transpose(_p(x), _p(xt), nseq, H, L, Lp, D, _rowmajor(x, "head_transpose x"), dcode(x.dtype),
                                            _stream())
        _lib.check(rc, "ctclip_head_transpose")
        return xt

    def qk_norm_fwd(self, x, scale_vec, H, D):
        M = x.shape[0]
        y = torch.empty((M, H * D), dtype=x.dtype, device=x.device)
        inv = torch.empty((M, H), dtype=torch.float32, device=x.device)
        rc = self.lib.ctclip_qk_norm_fwd(_p(x), _p(scale_vec), _p(y), _p(inv), M, H, D, _rowmajor(x, "x"), H * D, dcode(x.dtype),
                                         _stream())
        _lib.check(rc, "ctclip_qk_norm_fwd")
        return y, inv

    def qk_norm_bwd(self, dy, x, inv, scale_vec, dx, dscale, H, D):
        M = x.shape[0]
        ws = self.workspace(x.device, self.lib.ctclip_qk_norm_bwd_workspace(M, H, D)) if dscale is not None else None
        rc = self.lib.ctclip_qk_norm_bwd(_p(dy), _p(x), _p(inv), _p(scale_vec), _p(dx), _p(dscale), M, H, D, _rowmajor(dy, "dy"),
                                         _rowmajor(x, "x"), _rowmajor(dx, "dx"), dcode(x.dtype), _p(ws), ws.numel() if ws is not None else 0,
                                         _stream())
        _lib.check(rc, "ctclip_qk_norm_bwd")
        return dx

    def attn_fwd(self, q, k, vt, bias, keymask, nseq, H, L, D, scale, want_lse=True, bias_grid=None, dropout=None):
        """bias: (H, L, L) f32, or with bias_grid = (gh, gw) the relative-position table (nclass, H) (ctclip_attn_fwd).
        dropout = (p, seed): attention-probability dropout."""
        gh, gw = bias_grid if bias_grid is not None else (0, 0)
        dp, dseed = dropout if dropout is not None else (0.0, 0)
        M = nseq * L
        Lp = vt.shape[-1]
        o = torch.empty((M, H * D), dtype=q.dtype, device=q.device)
        lse = torch.empty((nseq, H, L), dtype=torch.float32, device=q.device) if want_lse else None
        rc = self.lib.ctclip_attn_fwd(_p(q), _p(k), _p(vt), _p(bias), gh, gw, _p(keymask), _p(o), _p(lse), nseq, H, L, Lp, D,
                                      _rowmajor(q, "q"), _rowmajor(k, "k"), H * D, float(scale), float(dp), int(dseed), dcode(q.dtype),
                                      _stream())
        _lib.check(rc, "ctclip_attn_fwd")
        return o, lse

    def attn_bwd(self, q, k, v, qt, kt, o, dout, dot, lse, bias, keymask, dq, dk, dv, dbias, nseq, H, L, D, scale, bias_grid=None,
                 dropout=None):
        Lp = qt.shape[-1]
        gh, gw = bias_grid if bias_grid is not None else (0, 0)
        dp, dseed = dropout if dropout is not None else (0.0, 0)
        delta = torch.empty((nseq, H, L), dtype=torch.float32, device=q.device)
        ws = None
        if dbias is not None:
            ws = self.workspace(q.device, self.lib.ctclip_attn_bwd_workspace(nseq, H, L))
        rc = self.lib.ctclip_attn_bwd(_p(q), _p(k), _p(v), _p(qt), _p(kt), _p(o), _p(dout), _p(dot), _p(lse), _p(bias), gh, gw,
                                      _p(keymask), _p(delta), _p(dq), _p(dk), _p(dv), _p(dbias), nseq, H, L, Lp, D,
                                      _rowmajor(q, "q"), _rowmajor(k, "k"), _rowmajor(v, "v"), _rowmajor(o, "o"),
                                      _rowmajor(dout, "dout"), _rowmajor(dq, "dq"), _rowmajor(dk, "dk"), _rowmajor(dv, "dv"),
                                      float(scale), float(dp), int(dseed), dcode(q.dtype), _p(ws), ws.numel() if ws is not None else 0,
                                      _stream())
        _lib.check(rc, "ctclip_attn_bwd")

    # ------------------------------------------------------------------ input pipeline (csrc/preprocess.hip)
    def preprocess_volume(self, vox, slope, intercept, xy_spacing, z_spacing, target_xy=0.75, target_z=1.5, out_shape=(480, 480, 240),
                          hu_range=(-1000.0, 1000.0), hu_div=1000.0, pad_value=-1.0, out=None):
        """vox: (H, W, D) int16 / f32 / f64 voxel array on the device -> (1, out_d, out_h, out_w) f32 model input (into `out` when given:
        a contiguous f32 tensor of that shape, e.g. one volume of a batch buffer)."""
        code = {torch.int16: 0, torch.float32: 1, torch.float64: 2}[vox.dtype]
        assert vox.dim() == 3 and vox.is_contiguous()
        H, W, D = vox.shape
        oh, ow, od = out_shape
        if out is None:
            out = torch.empty((1, od, oh, ow), dtype=torch.float32, device=vox.device)
        else:
            assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == od * oh * ow and out.device == vox.device
        rc = self.lib.ctclip_preprocess_volume(_p(vox), code, H, W, D, float(slope), float(intercept), float(xy_spacing), float(z_spacing),
                                               float(target_xy), float(target_z), _p(out), oh, ow, od, float(hu_range[0]), float(hu_range[1]),
                                               float(hu_div), float(pad_value), _stream())
        _lib.check(rc, "ctclip_preprocess_volume")
        return out

    # ------------------------------------------------------------------ short-sequence attention (csrc/attn_short.hip)
    def attn_short_supported(self, dtype, L, D):
        return dtype == torch.bfloat16 and bool(self.lib.ctclip_attn_short_supported(int(L), int(D), dcode(dtype)))

    def attn_short_fwd(self, q, kv, q_scale, k_scale, nseq, L, H, scale):
        """q (M, H*32), kv (M, 2*H*32) = [k | v] row-major bf16 -> o (M, H*32)."""
        M = q.shape[0]
        o = torch.empty((M, H * 32), dtype=q.dtype, device=q.device)
        rc = self.lib.ctclip_attn_short_fwd(_p(q), _rowmajor(q, "q"), _p(kv), _rowmajor(kv, "kv"), _p(q_scale), _p(k_scale), _p(o), H * 32,
                                            nseq, H, L, float(scale), _stream())
        _lib.check(rc, "ctclip_attn_short_fwd")
        return o

    def attn_short_bwd(self, q, kv, q_scale, k_scale, do, nseq, L, H, scale, dq_scale=None, dk_scale=None):
        """-> dq (M, H*32), dkv (M, 2*H*32); dq_scale / dk_scale (32) f32 are accumulated into when given."""
        M = q.shape[0]
        dq = torch.empty((M, H * 32), dtype=q.dtype, device=q.device)
        dkv = torch.empty((M, 2 * H * 32), dtype=q.dtype, device=q.device)
        ws = self.workspace(q.device, self.lib.ctclip_attn_short_bwd_workspace(nseq, H))
        rc = self.lib.ctclip_attn_short_bwd(_p(q), _rowmajor(q, "q"), _p(kv), _rowmajor(kv, "kv"), _p(q_scale), _p(k_scale), _p(do),
                                            _rowmajor(do, "do"), _p(dq), H * 32, _p(dkv), 2 * H * 32, _p(dq_scale), _p(dk_scale), nseq, H, L,
                                            float(scale), _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_attn_short_bwd")
        return dq, dkv

    # ------------------------------------------------------------------ attention, second generation (csrc/attn2.hip)
    def attn2_supported(self, dtype, H, L, D, bias_grid, has_bias):
        gh, gw = bias_grid if bias_grid is not None else (0, 0)
        return dtype == torch.bfloat16 and bool(self.lib.ctclip_attn2_supported(H, L, D, gh, gw, int(has_bias)))

    def attn2_prep(self, q, k, v, q_scale, k_scale, scale, H):
        """row-major q, k, v (M, H*32) views -> head-planar q~, k^, v (H, M, 32) + inverse norms (M, H)."""
        M = q.shape[0]
        dev = q.device
        qh, kh, vh = (torch.empty((H, M, 32), dtype=q.dtype, device=dev) for _ in range(3))
        qinv, kinv = (torch.empty((M, H), dtype=torch.float32, device=dev) for _ in range(2))
        rc = self.lib.ctclip_attn2_prep(_p(q), _p(k), _p(v), _rowmajor(q, "q"), _rowmajor(k, "k"), _rowmajor(v, "v"), _p(q_scale),
                                        _p(k_scale), float(scale), _p(qh), _p(kh), _p(vh), _p(qinv), _p(kinv), M, H, _stream())
        _lib.check(rc, "ctclip_attn2_prep")
        return qh, kh, vh, qinv, kinv

    def gemm_headnorm(self, a, b, sections):
        """The attention-operand epilogue: a (M, K) @ b (nsec * 256, K)^T written per 256-column section as head-planar (8, M, 32) bf16.
        sections: [(scale (32,) f32 | None, mult)]: with a scale vector the section is l2-normalised per head and scaled (q~ / k^ of
        ctclip_attn2_prep) and its inverse norms (M, 8) f32 are returned, None = plain copy (v).  -> [(planar, inv | None)] or None when the
        large-tile kernel does not serve the shape."""
        M, K = a.shape
        nsec = len(sections)
        if a.dtype != torch.bfloat16 or M % 256 or K % 64 or b.shape[0] != nsec * 256 or not 1 <= nsec <= 3:
            return None
        outs, args = [], []
        for i in range(3):
            if i < nsec:
                sc, mult = sections[i]
                o = torch.empty((8, M, 32), dtype=a.dtype, device=a.device)
                inv = torch.empty((M, 8), dtype=torch.float32, device=a.device) if sc is not None else None
                if sc is not None:
                    assert sc.dtype == torch.float32 and sc.numel() == 32 and sc.is_contiguous()
                outs.append((o, inv))
                args += [_p(o), _p(inv), _p(sc), float(mult)]
            else:
                args += [None, None, None, 0.0]
        rc = self.lib.ctclip_gemm_headnorm(_p(a), _p(b), M, nsec, K, _rowmajor(a, "a"), _rowmajor(b, "b"), *args, dcode(a.dtype), _stream())
        if rc == -2:      # CTCLIP_EUNSUPPORTED
            return None
        _lib.check(rc, "ctclip_gemm_headnorm")
        return outs

    def attn2_fwd(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, nseq, L):
        H, M, _ = qh.shape
        gh, gw = bias_grid if tab is not None else (0, 0)
        o = torch.empty((M, H * 32), dtype=qh.dtype, device=qh.device)
        lse2 = torch.empty((H, M), dtype=torch.float32, device=qh.device)
        rc = self.lib.ctclip_attn2_fwd(_p(qh), _p(kh), _p(vh), _p(tab), gh, gw, _p(q_scale), _p(k_scale), float(scale), _p(o), H * 32,
                                       _p(lse2), nseq, H, L, _stream())
        _lib.check(rc, "ctclip_attn2_fwd")
        return o, lse2

    def attn2_bwd(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, nseq, L, want_dtab, defer_dtab=False):
        """defer_dtab: skip the table-gradient pass and return (dqh, dkh, dvh, ws) with ws = a workspace tensor of this call's own that
        attn2_bwd_dbias finishes the job from (on another stream, later)."""
        H, M, _ = qh.shape
        gh, gw = bias_grid if tab is not None else (0, 0)
        dqh, dkh, dvh = torch.empty_like(qh), torch.empty_like(kh), torch.empty_like(vh)
        if defer_dtab:
            ws = torch.empty(self.lib.ctclip_attn2_bwd_workspace(nseq, H, L, gh, gw), dtype=torch.uint8, device=qh.device)
            rc = self.lib.ctclip_attn2_bwd(_p(qh), _p(kh), _p(vh), _p(tab), gh, gw, _p(q_scale), _p(k_scale), float(scale), _p(o),
                                           _rowmajor(o, "o"), _p(dout), _rowmajor(dout, "dout"), _p(lse2), _p(dqh), _p(dkh), _p(dvh), None,
                                           nseq, H, L, _p(ws), ws.numel(), _stream())
            _lib.check(rc, "ctclip_attn2_bwd")
            return dqh, dkh, dvh, ws
        dtab = torch.empty_like(tab) if (want_dtab and tab is not None) else None
        ws = self.workspace(qh.device, self.lib.ctclip_attn2_bwd_workspace(nseq, H, L, gh if dtab is not None else 0, gw))
        rc = self.lib.ctclip_attn2_bwd(_p(qh), _p(kh), _p(vh), _p(tab), gh, gw, _p(q_scale), _p(k_scale), float(scale), _p(o),
                                       _rowmajor(o, "o"), _p(dout), _rowmajor(dout, "dout"), _p(lse2), _p(dqh), _p(dkh), _p(dvh), _p(dtab),
                                       nseq, H, L, _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_attn2_bwd")
        return dqh, dkh, dvh, dtab

    def attn2_bwd_tok(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, qinv, kinv, dq, dk, dv, dq_scale, dk_scale, nseq, L,
                      want_dtab, defer_dtab=False):
        """Backward of attn2_fwd straight to ROW-MAJOR dq (M, HD), dk / dv (column views of one (M, 2 HD) buffer) with the l2norm backward
        applied and dq_scale / dk_scale accumulated: the slab key pass writes dk / dv itself (no planar dk^ / dv round trip), the q half goes
        through the q-only un-prep.  -> (dtab | None, ws | None) -- ws when defer_dtab (attn2_bwd_dbias finishes the table gradient from it) --
        or None when the slab kernels do not serve the shape (caller: attn2_bwd + attn2_unprep)."""
        H, M, _ = qh.shape
        gh, gw = bias_grid if tab is not None else (0, 0)
        dqh = torch.empty_like(qh)
        nbytes = self.lib.ctclip_attn2_bwd_tok_workspace(nseq, H, L, gh, gw)
        dtab = torch.empty_like(tab) if (want_dtab and tab is not None and not defer_dtab) else None
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qh.device) if defer_dtab else self.workspace(qh.device, nbytes)
        rc = self.lib.ctclip_attn2_bwd_tok(_p(qh), _p(kh), _p(vh), _p(tab), gh, gw, _p(q_scale), _p(k_scale), float(scale), _p(o), _rowmajor(o, "o"),
                                           _p(dout), _rowmajor(dout, "dout"), _p(lse2), _p(kinv), _p(dqh), _p(dk), _rowmajor(dk, "dk"), _p(dv),
                                           _rowmajor(dv, "dv"), _p(dk_scale), _p(dtab), nseq, H, L, _p(ws), ws.numel(), _stream())
        if rc == -2:      # CTCLIP_EUNSUPPORTED
            return None
        _lib.check(rc, "ctclip_attn2_bwd_tok")
        ws2 = self.workspace(qh.device, self.lib.ctclip_attn2_unprep_workspace()) if defer_dtab else None
        wsu = ws2 if ws2 is not None else self.workspace(qh.device, max(nbytes, self.lib.ctclip_attn2_unprep_workspace()))
        # (not deferred: the shared per-stream workspace is reused -- stream order makes that safe: the reduce launches of bwd_tok are done)
        _lib.check(self.lib.ctclip_attn2_unprep_q(_p(dqh), _p(qh), _p(qinv), _p(q_scale), float(scale), _p(dq), _rowmajor(dq, "dq"), _p(dq_scale), M, H,
                                                  _p(wsu), wsu.numel(), _stream()), "ctclip_attn2_unprep_q")
        return dtab, (ws if defer_dtab else None)

    def attn2_bwd_fused(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, qinv, kinv, dq, dk, dv, dq_scale, dk_scale, nseq, L,
                        want_dtab):
        """Backward of attn2_fwd in ONE pass over the score tiles (csrc/attn2_bwd1.hip): row-major dq / dk / dv with the l2norm backward applied,
        dq_scale / dk_scale accumulated and the position-bias table gradient from the same sweep (fixed-point scatter, deterministic).
        -> (dtab | None,) or None when the kernel does not serve the shape (caller: attn2_bwd_tok)."""
        H, M, _ = qh.shape
        gh, gw = bias_grid if tab is not None else (0, 0)
        if not self.lib.ctclip_attn2_bwd_fused_supported(nseq, H, L, 32, gh, gw, int(tab is not None)):
            return None
        dtab = torch.empty_like(tab) if (want_dtab and tab is not None) else None
        ws = self.workspace(qh.device, self.lib.ctclip_attn2_bwd_fused_workspace(nseq, H, L, gh, gw))
        rc = self.lib.ctclip_attn2_bwd_fused(_p(qh), _p(kh), _p(vh), _p(tab), gh, gw, _p(q_scale), _p(k_scale), float(scale), _p(o), _rowmajor(o, "o"),
                                             _p(dout), _rowmajor(dout, "dout"), _p(lse2), _p(qinv), _p(kinv), _p(dq), _rowmajor(dq, "dq"), _p(dk),
                                             _rowmajor(dk, "dk"), _p(dv), _rowmajor(dv, "dv"), _p(dq_scale), _p(dk_scale), _p(dtab), nseq, H, L,
                                             _p(ws), ws.numel(), _stream())
        if rc == -2:      # CTCLIP_EUNSUPPORTED
            return None
        _lib.check(rc, "ctclip_attn2_bwd_fused")
        return (dtab,)

    def attn2_bwd_dbias(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, lse2, nseq, L, ws):
        """The table gradient (ncls, H) from the workspace a deferred attn2_bwd left behind."""
        H = qh.shape[0]
        gh, gw = bias_grid
        dtab = torch.empty_like(tab)
        rc = self.lib.ctclip_attn2_bwd_dbias(_p(qh), _p(kh), _p(vh), _p(tab), gh, gw, _p(q_scale), _p(k_scale), float(scale), _p(lse2), _p(dtab),
                                             nseq, H, L, _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_attn2_bwd_dbias")
        return dtab

    def attn2_unprep(self, dqh, dkh, dvh, qh, kh, qinv, kinv, q_scale, k_scale, scale, dq, dk, dv, dq_scale, dk_scale):
        H, M, _ = qh.shape
        ws = self.workspace(qh.device, self.lib.ctclip_attn2_unprep_workspace())
        rc = self.lib.ctclip_attn2_unprep(_p(dqh), _p(dkh), _p(dvh), _p(qh), _p(kh), _p(qinv), _p(kinv), _p(q_scale), _p(k_scale),
                                          float(scale), _p(dq), _p(dk), _p(dv), _rowmajor(dq, "dq"), _rowmajor(dk, "dk"), _rowmajor(dv, "dv"),
                                          _p(dq_scale), _p(dk_scale), M, H, _p(ws), ws.numel(), _stream())
        _lib.check(rc, "ctclip_attn2_unprep")

    def dropout(self, x, residual, p, seed, stream_id):
        """y = dropout(x) (+ residual); mask = philox(seed, element, stream_id).  The same call on dy is the backward."""
        assert x.is_contiguous() and (residual is None or (residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype))
        y = torch.empty_like(x)
        _lib.check(self.lib.ctclip_dropout(_p(x), _p(residual), _p(y), x.numel(), float(p), int(seed), int(stream_id), dcode(x.dtype),
                                           _stream()), "ctclip_dropout")
        return y

    def attn_dropout_mask(self, nseq, H, L, p, seed, device):
        m = torch.empty((nseq, H, L, L), dtype=torch.float32, device=device)
        _lib.check(self.lib.ctclip_attn_dropout_mask(_p(m), nseq, H, L, float(p), int(seed), _stream()), "ctclip_attn_dropout_mask")
        return m

    # ------------------------------------------------------------------ elementwise / streaming
    def accumulate(self, dst, src):
        """dst += src (f32, contiguous, same size)"""
        assert dst.dtype == src.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
        _lib.check(self.lib.ctclip_accumulate_f32(_p(dst), _p(src), dst.numel(), _stream()), "ctclip_accumulate_f32")
        return dst

    def geglu_weight_interleave(self, w, hp, dtype):
        """FeedForward[1].weight (2 * inner, K) f32 -> the (2 * hp, K) operand of gemm_geglu (x / gate rows interleaved in fours)."""
        two_inner, K = w.shape
        out = torch.empty((2 * hp, K), dtype=dtype, device=w.device)
        _lib.check(self.lib.ctclip_geglu_weight_interleave(_p(w), _p(out), two_inner // 2, hp, K, K, _stream()), "ctclip_geglu_weight_interleave")
        return out

    def gemm_geglu(self, x, w_il, hp, save_u=True):
        """g (M, hp) = x * gelu(gate) and, with save_u, u (M, 2 hp) = [x | gate] in one launch -> (u or None, g); None when the shape
        is not served (caller: gemm + geglu_fwd)."""
        M, K = x.shape
        if x.dtype != torch.bfloat16:
            return None
        u = torch.empty((M, 2 * hp), dtype=x.dtype, device=x.device) if save_u else None
        g = torch.empty((M, hp), dtype=x.dtype, device=x.device)
        timing = self._gemm_events
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = self.lib.ctclip_gemm_geglu(_p(x), _p(w_il), _p(u), _p(g), M, hp, K, _rowmajor(x, "x"), _rowmajor(w_il, "w"), 2 * hp, hp,
                                        dcode(x.dtype), _stream())
        if rc == -2:      # CTCLIP_EUNSUPPORTED
            return None
        if timing is not None:
            e1.record()
            timing.setdefault(("NT", "bf16", M, 2 * hp, K, torch.cuda.current_stream() != torch.cuda.default_stream(),
                               "+geglu" if save_u else "+geglu(g only)"), []).append((e0, e1))
        _lib.check(rc, "ctclip_gemm_geglu")
        return u, g

    def gemm_dgeglu(self, dy, wt, u):
        """du (M, 2 hp) = [dg gelu(gate) | dg x gelu'(gate)] with dg = dy @ wt^T formed in the accumulators only (wt = the out-projection
        weight transposed, (hp, K)) and u = [x | gate] (M, 2 hp) as stored by gemm_geglu; None when the shape is not served."""
        M, K = dy.shape
        hp = wt.shape[0]
        if dy.dtype != torch.bfloat16 or u.dtype != torch.bfloat16 or tuple(u.shape) != (M, 2 * hp):
            return None
        du = torch.empty_like(u)
        timing = self._gemm_events
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = self.lib.ctclip_gemm_dgeglu(_p(dy), _p(wt), _p(u), _p(du), M, hp, K, _rowmajor(dy, "dy"), _rowmajor(wt, "wt"), _rowmajor(u, "u"),
                                         _rowmajor(du, "du"), dcode(dy.dtype), _stream())
        if rc == -2:
            return None
        if timing is not None:
            e1.record()
            timing.setdefault(("NT", "bf16", M, hp, K, torch.cuda.current_stream() != torch.cuda.default_stream(), "+geglu-bwd"), []).append((e0, e1))
        _lib.check(rc, "ctclip_gemm_dgeglu")
        return du

    def shadow_refresh(self, jobs, version):
        """jobs: [dict(src_ptr / src_stride / src_shape of an f32 (rows, cols) parameter, dst bf16 2-D view, map, aux, transposed)] -> one
        launch (csrc/shadow.hip).  The device-side job table is rebuilt only when `version` changes (functional.refresh_shadows validates
        the raw pointers before every call)."""
        tab = self._shadow_tables.get(version)
        if tab is None:
            rows, tile0 = [], 0
            for j in jobs:
                dst = j["dst"]
                assert len(j["src_shape"]) == 2 and dst.dtype == torch.bfloat16 and dst.stride(1) == 1
                rows.append([j["src_ptr"], dst.data_ptr(), j["src_stride"], dst.stride(0), j["src_shape"][0], j["src_shape"][1], dst.shape[0], dst.shape[1],
                             int(j["map"]), int(j["aux"]), int(j["transposed"]), tile0])
                tile0 += ((dst.shape[0] + 63) // 64) * ((dst.shape[1] + 63) // 64)
            dev = jobs[0]["dst"].device
            table = torch.tensor(rows, dtype=torch.int64).to(dev)
            if len(self._shadow_tables) > 8:
                self._shadow_tables.clear()
            tab = self._shadow_tables[version] = (table, len(rows), tile0)
        table, n, ntiles = tab
        _lib.check(self.lib.ctclip_shadow_refresh(_p(table), n, ntiles, _stream()), "ctclip_shadow_refresh")

    def gemm_geglu_bwd(self, x, w_il, dg, hp):
        """du (M, 2 hp) = [dg * gelu(gate) | dg * x * gelu'(gate)] with (x, gate) = x @ w_il^T recomputed inside the launch (nothing of the
        forward is read); None when the shape is not served."""
        M, K = x.shape
        if x.dtype != torch.bfloat16 or dg.dtype != torch.bfloat16:
            return None
        assert dg.shape == (M, hp) and dg.stride(1) == 1
        du = torch.empty((M, 2 * hp), dtype=x.dtype, device=x.device)
        timing = self._gemm_events
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = self.lib.ctclip_gemm_geglu_bwd(_p(x), _p(w_il), _p(dg), _p(du), M, hp, K, _rowmajor(x, "x"), _rowmajor(w_il, "w"), dg.stride(0),
                                            2 * hp, dcode(x.dtype), _stream())
        if rc == -2:
            return None
        if timing is not None:
            e1.record()
            timing.setdefault(("NT", "bf16", M, 2 * hp, K, torch.cuda.current_stream() != torch.cuda.default_stream(),
                               "geglu-bwd recompute"), []).append((e0, e1))
        _lib.check(rc, "ctclip_gemm_geglu_bwd")
        return du

    def geglu_fwd(self, u):
        M, H2 = u.shape
        g = torch.empty((M, H2 // 2), dtype=u.dtype, device=u.device)
        _lib.check(self.lib.ctclip_geglu_fwd(_p(u), _p(g), M, H2 // 2, dcode(u.dtype), _stream()), "ctclip_geglu_fwd")
        return g

    def geglu_bwd(self, dg, u):
        M, H2 = u.shape
        du = torch.empty_like(u)
        _lib.check(self.lib.ctclip_geglu_bwd(_p(dg), _p(u), _p(du), M, H2 // 2, dcode(u.dtype), _stream()), "ctclip_geglu_bwd")
        return du

    def gelu_fwd(self, u):
        h = torch.empty_like(u)
        _lib.check(self.lib.ctclip_gelu_fwd(_p(u), _p(h), u.numel(), dcode(u.dtype), _stream()), "ctclip_gelu_fwd")
        return h

    def gelu_bwd(self, dh, u):
        du = torch.empty_like(u)
        _lib.check(self.lib.ctclip_gelu_bwd(_p(dh), _p(u), _p(du), u.numel(), dcode(u.dtype), _stream()), "ctclip_gelu_bwd")
        return du

    def leaky_relu_fwd(self, x, slope):
        y = torch.empty_like(x)
        _lib.check(self.lib.ctclip_leaky_relu_fwd(_p(x), _p(y), x.numel(), float(slope), _stream()), "ctclip_leaky_relu_fwd")
        return y

    def leaky_relu_bwd(self, dy, x, slope):
        dx = torch.empty_like(x)
        _lib.check(self.lib.ctclip_leaky_relu_bwd(_p(dy), _p(x), _p(dx), x.numel(), float(slope), _stream()),
                   "ctclip_leaky_relu_bwd")
        return dx

    def colsum(self, x, out, N=None):
        M = x.shape[0]
        N = x.shape[1] if N is None else N
        ws = self.workspace(x.device, self.lib.ctclip_colsum_workspace(M, N))
        _lib.check(self.lib.ctclip_colsum(_p(x), _p(out), M, N, _rowmajor(x, "colsum x"), dcode(x.dtype), _p(ws), ws.numel(), _stream()),
                   "ctclip_colsum")
        return out

    def patch_embed_param_bwd(self, G, W, g1, b1, dbp, dW, dg1, db1, accumulate):
        """Parameter-space epilogue of the patch-embedding backward: dW (+)= G * gamma1 + dbp (x) beta1, dgamma1 (+)= sum_n W G,
        dbeta1 (+)= W^T dbp (all f32, written / accumulated in place)."""
        N, K = W.shape
        for t in (G, W, dW):
            assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (N, K)
        for t in (g1, b1, dg1, db1):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == K
        assert dbp.dtype == torch.float32 and dbp.numel() == N
        _lib.check(self.lib.ctclip_patch_embed_param_bwd(_p(G), _p(W), _p(g1), _p(b1), _p(dbp), _p(dW), _p(dg1), _p(db1), N, K, int(accumulate),
                                                         _stream()), "ctclip_patch_embed_param_bwd")

    def permute0213(self, x):
        A, B, C, D = x.shape
        assert x.is_contiguous()
        y = torch.empty((A, C, B, D), dtype=x.dtype, device=x.device)
        _lib.check(self.lib.ctclip_permute0213(_p(x), _p(y), A, B, C, D, dcode(x.dtype), _stream()), "ctclip_permute0213")
        return y

    def transpose2d(self, x):
        R, C = x.shape
        y = torch.empty((C, R), dtype=x.dtype, device=x.device)
        _lib.check(self.lib.ctclip_transpose2d(_p(x), _p(y), R, C, _rowmajor(x, "transpose x"), R, dcode(x.dtype), _stream()),
                   "ctclip_transpose2d")
        return y

    def pool_fwd(self, x, out_dtype=None):
        B, t, R = x.shape
        assert x.is_contiguous()
        y = torch.empty((B, R), dtype=out_dtype or x.dtype, device=x.device)
        _lib.check(self.lib.ctclip_pool_fwd(_p(x), _p(y), B, t, R, dcode(x.dtype), dcode(y.dtype), _stream()), "ctclip_pool_fwd")
        return y

    def pool_bwd(self, dy, t, out_dtype=None):
        B, R = dy.shape
        assert dy.is_contiguous()
        dx = torch.empty((B, t, R), dtype=out_dtype or dy.dtype, device=dy.device)
        _lib.check(self.lib.ctclip_pool_bwd(_p(dy), _p(dx), B, t, R, dcode(dy.dtype), dcode(dx.dtype), _stream()), "ctclip_pool_bwd")
        return dx

    def convert_pad(self, src, rows_dst, cols_dst, dtype, colscale=None, out=None):
        rows, cols = src.shape
        if out is None:
            out = torch.empty((rows_dst, cols_dst), dtype=dtype, device=src.device)
        rc = self.lib.ctclip_convert_pad(_p(src), _p(out), _p(colscale), rows, cols, _rowmajor(src, "convert src"), rows_dst,
                                         cols_dst, _rowmajor(out, "convert dst"), dcode(src.dtype), dcode(out.dtype), _stream())
        _lib.check(rc, "ctclip_convert_pad")
        return out

    def cpb_expand(self, tab, gh, gw):
        ncls, H = tab.shape
        L = gh * gw
        bias = torch.empty((H, L, L), dtype=torch.float32, device=tab.device)
        _lib.check(self.lib.ctclip_cpb_expand(_p(tab), _p(bias), H, gh, gw, _stream()), "ctclip_cpb_expand")
        return bias

    def cpb_reduce(self, dbias, gh, gw):
        H = dbias.shape[0]
        dtab = torch.empty(((2 * gh - 1) * (2 * gw - 1), H), dtype=torch.float32, device=dbias.device)
        _lib.check(self.lib.ctclip_cpb_reduce(_p(dbias), _p(dtab), H, gh, gw, _stream()), "ctclip_cpb_reduce")
        return dtab

    def bert_embed_fwd(self, ids, word, pos, type0, dtype):
        B, T = ids.shape
        Hd = word.shape[1]
        x = torch.empty((B * T, Hd), dtype=dtype, device=word.device)
        rc = self.lib.ctclip_bert_embed_fwd(_p(ids), _p(word), _p(pos), _p(type0), _p(x), B * T, T, Hd, dcode(dtype), _stream())
        _lib.check(rc, "ctclip_bert_embed_fwd")
        return x

    def bert_embed_bwd(self, ids, dx, dword, dpos, dtype0):
        """Embedding-table gradients as deterministic segmented sums (no float atomics): rows of dx grouped by token id / position."""
        B, T = ids.shape
        if dword is not None:
            self.segment_sum(ids.reshape(-1), dx, dword, dword.shape[0], accumulate=True)
        if dpos is not None:
            self.segment_sum(None, dx, dpos, dpos.shape[0], accumulate=True, key_mod=T)
        if dtype0 is not None:            # every row has token type 0: a plain (deterministic) column sum
            self.colsum(dx, dtype0[0])

    # ------------------------------------------------------------------ VQ
    def vq_gather(self, embed, idx, dtype):
        M = idx.numel()
        d = embed.shape[1]
        out = torch.empty((M, d), dtype=dtype, device=embed.device)
        _lib.check(self.lib.ctclip_vq_gather(_p(embed), _p(idx), _p(out), M, d, dcode(dtype), _stream()), "ctclip_vq_gather")
        return out

    def vq_ema(self, idx, x, inv, cluster_size, embed, decay):
        """bins = histogram(idx), esum[c] = sum of the unit rows x[r] * inv[r] assigned to code c (row order: deterministic)."""
        C, d = embed.shape
        stats = torch.empty(C * (d + 1), dtype=torch.float32, device=embed.device)      # ONE buffer [bins | esum]: one all-reduce under data parallelism
        bins, esum = stats[:C], stats[C:].view(C, d)
        self.segment_sum(idx.reshape(-1), x, esum, C, rowscale=inv, counts=bins)
        return bins, esum

    def vq_ema_update(self, cluster_size, embed, bins, esum, decay):
        C, d = embed.shape
        _lib.check(self.lib.ctclip_vq_ema_update(_p(cluster_size), _p(embed), _p(bins), _p(esum), C, d, float(decay), _stream()),
                   "ctclip_vq_ema_update")

    # ------------------------------------------------------------------ CLIP head
    def visual_latent_fwd(self, x, w):
        Bm, K = x.shape
        N = w.shape[0]
        assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == K
        y = torch.empty((Bm, N), dtype=torch.float32, device=x.device)
        ws = self.workspace(x.device, self.lib.ctclip_visual_latent_fwd_workspace(min(8, Bm), N, K))
        for b0 in range(0, Bm, 8):
            nb = min(8, Bm - b0)
            rc = self.lib.ctclip_visual_latent_fwd(_p(x[b0:]), _p(w), _p(y[b0:]), nb, N, K, dcode(x.dtype), _p(ws), ws.numel(), _stream())
            _lib.check(rc, "ctclip_visual_latent_fwd")
        return y

    def visual_latent_bwd(self, dy, x, w, dw=None, accumulate=False, want_dx=True):
        Bm, K = x.shape
        N = w.shape[0]
        assert dy.dtype == torch.float32 and dy.is_contiguous()
        dx = torch.empty_like(x) if want_dx else None
        # rows per launch: 8 (the training step's batch), 24 in f32 when there are more -- the VocabFine step's 18 pooled vectors then
        # stream the 604-MB weight and its gradient once instead of three times
        step = 24 if (Bm > 8 and x.dtype == torch.float32) else 8
        for b0 in range(0, Bm, step):
            nb = min(step, Bm - b0)
            rc = self.lib.ctclip_visual_latent_bwd(_p(dy[b0:]), _p(x[b0:]), _p(w), _p(dx[b0:]) if want_dx else None, _p(dw), nb, N,
                                                   K, int(accumulate or b0 > 0), dcode(x.dtype), _stream())
            _lib.check(rc, "ctclip_visual_latent_bwd")
        return dx

    CLIP_LOSS_ONE_BLOCK = 128      # gathered batch the single-block kernel serves (its G x G logits live in LDS)

    def clip_loss(self, tl, il, temperature, want_grads=True, want_logits=False):
        G, Dl = tl.shape
        assert tl.dtype == torch.float32 and il.dtype == torch.float32 and tl.is_contiguous() and il.is_contiguous()
        dev = tl.device
        if G > self.CLIP_LOSS_ONE_BLOCK or Dl > 1024:
            return self._clip_loss_large(tl, il, temperature, want_grads, want_logits)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        logits = torch.empty((G, G), dtype=torch.float32, device=dev) if want_logits else None
        dtl = torch.empty_like(tl) if want_grads else None
        dil = torch.empty_like(il) if want_grads else None
        dtemp = torch.zeros(1, dtype=torch.float32, device=dev) if want_grads else None
        rc = self.lib.ctclip_clip_loss(_p(tl), _p(il), _p(temperature), _p(out), _p(logits), _p(dtl), _p(dil), _p(dtemp), G, Dl,
                                       _stream())
        _lib.check(rc, "ctclip_clip_loss")
        return out, logits, dtl, dil, dtemp

    def _clip_loss_large(self, tl, il, temperature, want_grads, want_logits):
        """Any gathered batch: logits and their gradient in global memory, f32 GEMMs around ctclip_clip_loss_logits (csrc/head.hip)."""
        G, Dl = tl.shape
        dev = tl.device
        ut, tinv = self.l2norm_rows(tl, torch.float32, eps=1e-12)
        uv, iinv = self.l2norm_rows(il, torch.float32, eps=1e-12)
        Gp = (G + 3) // 4 * 4                    # 16-byte aligned rows: S is a k-contiguous operand of the gradient GEMMs
        S = torch.zeros((G, Gp), dtype=torch.float32, device=dev)[:, :G]
        self.gemm(ut, uv, out=S)                 # (G, G) f32 cosines; exp(temperature) is applied on the device
        logits = S.clone() if want_logits else None
        out = torch.empty(2, dtype=torch.float32, device=dev)
        dtemp = torch.zeros(1, dtype=torch.float32, device=dev) if want_grads else None
        ws = torch.empty(4 * G, dtype=torch.float32, device=dev)
        _lib.check(self.lib.ctclip_clip_loss_logits(_p(S), S.stride(0), _p(temperature), _p(out), _p(dtemp), G, _p(ws), ws.numel() * 4, _stream()),
                   "ctclip_clip_loss_logits")
        if logits is not None:
            self.scale_by_scalar(logits, out[1:2])
        if not want_grads:
            return out, logits, None, None, None
        dut = self.gemm(S, uv, a_kc=True, b_kc=False)                              # (temp dS) Uv
        duv = self.gemm(S, ut, a_kc=False, b_kc=False, M=G, N=Dl, K=G)             # (temp dS)^T Ut
        dtl, dil = torch.empty_like(tl), torch.empty_like(il)
        for raw, inv, du, dst in ((tl, tinv, dut, dtl), (il, iinv, duv, dil)):
            _lib.check(self.lib.ctclip_l2norm_bwd_rows(_p(raw), _p(inv), _p(du), _p(dst), G, Dl, _stream()), "ctclip_l2norm_bwd_rows")
        return out, logits, dtl, dil, dtemp

    def scale_by_scalar(self, x, scalar):
        _lib.check(self.lib.ctclip_scale_by_scalar(_p(x), _p(scalar), x.numel(), _stream()), "ctclip_scale_by_scalar")
        return x

    # ------------------------------------------------------------------ fine-tuning heads (csrc/finetune.hip)
    def relu_dropout(self, x, dy, p, seed, stream_id):
        assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 4 == 0
        out = torch.empty_like(x)
        _lib.check(self.lib.ctclip_relu_dropout(_p(x), _p(dy), _p(out), x.numel(), float(p), int(seed), int(stream_id), _stream()),
                   "ctclip_relu_dropout")
        return out

    def bce_logits(self, logits, targets, pos_weight):
        Bn, C = logits.shape
        assert logits.dtype == torch.float32 and targets.dtype == torch.float32
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        _lib.check(self.lib.ctclip_bce_logits(_p(logits), _p(targets), _p(pos_weight), _p(loss), _p(dlogits), Bn, C, _stream()),
                   "ctclip_bce_logits")
        return loss, dlogits

    def pair_softmax_mse(self, sims):
        n = sims.shape[0]
        assert sims.dtype == torch.float32 and sims.shape[1] == 2
        loss = torch.empty(1, dtype=torch.float32, device=sims.device)
        dsims = torch.empty_like(sims)
        _lib.check(self.lib.ctclip_pair_softmax_mse(_p(sims), _p(loss), _p(dsims), n, _stream()), "ctclip_pair_softmax_mse")
        return loss, dsims

    def latent_similarity(self, text, image, temperature, dsims=None):
        """forward: sims (max(nt, ni)); backward (dsims given): (dtext, dimage, dtemp)."""
        nt, D = text.shape
        ni = image.shape[0]
        assert text.dtype == image.dtype == torch.float32 and text.is_contiguous() and image.is_contiguous() and image.shape[1] == D
        if dsims is None:
            sims = torch.empty(max(nt, ni), dtype=torch.float32, device=text.device)
            _lib.check(self.lib.ctclip_latent_similarity(_p(text), _p(image), _p(temperature), None, _p(sims), None, None, None, nt, ni, D,
                                                         _stream()), "ctclip_latent_similarity")
            return sims
        dt, di = torch.empty_like(text), torch.empty_like(image)
        dtemp = torch.empty(1, dtype=torch.float32, device=text.device)
        _lib.check(self.lib.ctclip_latent_similarity(_p(text), _p(image), _p(temperature), _p(dsims), None, _p(dt), _p(di), _p(dtemp), nt, ni, D,
                                                     _stream()), "ctclip_latent_similarity")
        return dt, di, dtemp

    # ------------------------------------------------------------------ optimiser
    def grad_norm_clip(self, g, max_norm, extra_sq=None):
        out = torch.empty(2, dtype=torch.float32, device=g.device)
        ws = self.workspace(g.device, self.lib.ctclip_grad_norm_workspace())
        rc = self.lib.ctclip_grad_norm_clip(_p(g), g.numel(), _p(extra_sq), float(max_norm or 0.0), _p(out), _p(ws), ws.numel(),
                                            _stream())
        _lib.check(rc, "ctclip_grad_norm_clip")
        return out

    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, step, weight_decay=0.0, clip=None, decay_mask4=None, zero_grad=False):
        """zero_grad: the gradient buffer is cleared in the same pass (ctclip_adam_step_zero_grad: no separate fill launch)."""
        assert decay_mask4 is None or (decay_mask4.dtype == torch.uint8 and decay_mask4.numel() >= (p.numel() + 3) // 4)
        fn = self.lib.ctclip_adam_step_zero_grad if zero_grad else self.lib.ctclip_adam_step
        rc = fn(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                int(step), float(weight_decay), _p(clip), _p(decay_mask4), _stream())
        _lib.check(rc, "ctclip_adam_step")


_active = None


def get():
    """The active backend; instantiates the HIP one on first use (ImportError if the .so is missing)."""
    global _active
    if _active is None:
        _active = HipBackend()
    return _active


def use(backend):
    """TEST HOOK: swap the primitive implementation (tests/ref_backend.py).  Returns the previous one."""
    global _active
    prev, _active = _active, backend
    return prev
