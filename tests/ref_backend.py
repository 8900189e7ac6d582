"""TEST INFRASTRUCTURE ONLY -- pure-torch checker with the interface of ``ct_clip_amd.backend.HipBackend``.

Two uses: (1) CPU tests swap it in (``backend.use(RefBackend())``) to validate the host-side composition and the
hand-derived backward formulas of ``ct_clip_amd.functional`` against the oracle without a GPU; (2) GPU tests call the
same method on both backends with the same inputs to check every HIP kernel.  Never used by the product.
"""
import torch
import torch.nn.functional as F


def _f(t):
    return t.float()


class RefBackend:
    name = "torch-reference (test only)"

    # ---- GEMM
    def gemm(self, a, b, *, a_kc=True, b_kc=True, bias=None, residual=None, out=None, out_dtype=None, accumulate=False,
             alpha=1.0, split_k=1, M=None, N=None, K=None):
        A = a if a_kc else a.t()
        Bm = b if b_kc else b.t()
        M = A.shape[0] if M is None else M
        N = Bm.shape[0] if N is None else N
        K = min(A.shape[1], Bm.shape[1]) if K is None else K
        y = alpha * (_f(A[:M, :K]) @ _f(Bm[:N, :K]).t())
        if bias is not None:
            y = y + bias[:N]
        if residual is not None:
            y = y + _f(residual[:M, :N])
        if out is None:
            return y.to(out_dtype or a.dtype)
        if accumulate:
            out[:M, :N] += y.to(out.dtype)
        else:
            out[:M, :N] = y.to(out.dtype)
        return out

    def gemm_nt2_select(self, mask):
        return 0

    def gemm_dw_db(self, dy, x, dw, db, accumulate=True):
        if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or dy.shape[0] % 64 or dy.shape[0] < 128:
            return False
        gw, gb = _f(dy).t() @ _f(x), _f(dy).sum(0)
        if accumulate:
            dw += gw; db += gb
        else:
            dw.copy_(gw); db.copy_(gb)
        return True

    def gemm_argmax(self, a, b):
        s = _f(a) @ _f(b).t()
        val, idx = s.max(dim=-1)
        return idx, val

    @staticmethod
    def gemm_argmax_hilo_ok(a, ncodes):
        M, d = a.shape
        return a.dtype == torch.bfloat16 and d % 64 == 0 and ((M + 255) // 256) * ((ncodes + 255) // 256) >= 160

    def gemm_argmax_hilo(self, a, b2):
        K = a.shape[1]
        e = _f(b2[:, :K]) + _f(b2[:, K:])
        val, idx = (_f(a) @ e.t()).max(dim=-1)
        return idx, val

    # ---- norms
    def layernorm_fwd(self, x, gamma, beta, eps, want_stats=True):
        xf = _f(x)
        mean = xf.mean(-1)
        var = xf.var(-1, unbiased=False)
        rstd = torch.rsqrt(var + eps)
        y = (xf - mean[:, None]) * rstd[:, None]
        if gamma is not None:
            y = y * gamma
        if beta is not None:
            y = y + beta
        return y.to(x.dtype), mean, rstd

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dgamma=None, dbeta=None, add1=None, add2=None):
        xh = (_f(x) - mean[:, None]) * rstd[:, None]
        g = _f(dy) * (gamma if gamma is not None else 1.0)
        dx = rstd[:, None] * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
        for a in (add1, add2):
            if a is not None:
                dx = dx + _f(a)
        if dgamma is not None:
            dgamma += (_f(dy) * xh).sum(0)
        if dbeta is not None:
            dbeta += _f(dy).sum(0)
        return dx.to(x.dtype)

    def layernorm_bwd_partials(self, dy, x, gamma, mean, rstd, add1=None, add2=None):
        xh = (_f(x) - mean[:, None]) * rstd[:, None]
        part = torch.stack([(_f(dy) * xh).sum(0), _f(dy).sum(0)])
        return self.layernorm_bwd(dy, x, gamma, mean, rstd, None, None, add1, add2), part

    def layernorm_bwd_reduce(self, part, dgamma, dbeta, rows, cols):
        if dgamma is not None:
            dgamma += part[0]
        if dbeta is not None:
            dbeta += part[1]

    def patch_ln(self, video, pt, p1, p2, kpad, eps, dtype):
        b, c, f, H, W = video.shape
        t, h, w = f // pt, H // p1, W // p2
        x = video.reshape(b, c, t, pt, h, p1, w, p2).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b * t * h * w, -1)
        x = F.layer_norm(x, x.shape[-1:], None, None, eps)
        out = torch.zeros((x.shape[0], kpad), dtype=dtype, device=video.device)
        out[:, :x.shape[1]] = x.to(dtype)
        return out

    def l2norm_rows(self, x, out_dtype, eps=1e-12):
        xf = _f(x)
        inv = 1.0 / xf.norm(dim=-1).clamp(min=eps)
        return (xf * inv[:, None]).to(out_dtype), inv

    def l2norm_split3(self, x, order, eps=1e-12):
        xf = _f(x)
        inv = 1.0 / xf.norm(dim=-1).clamp_min(eps)
        t = xf * inv[:, None]
        hi = t.to(torch.bfloat16)
        lo = (t - hi.float()).to(torch.bfloat16)
        if order == 2:
            return torch.cat([hi, lo], dim=1), inv
        return torch.cat([hi, hi, lo] if order == 0 else [hi, lo, hi], dim=1), inv

    def row_inv_norms(self, x, eps=1e-12):
        return 1.0 / _f(x).norm(dim=-1).clamp_min(eps)

    def segment_sum(self, keys, x, out, nseg, rowscale=None, counts=None, accumulate=False, key_mod=0):
        M, d = x.shape
        k = keys.reshape(-1) if keys is not None else torch.arange(M, device=x.device) % key_mod
        v = _f(x) if rowscale is None else _f(x) * rowscale[:, None]
        acc = torch.zeros((nseg, d), dtype=torch.float32, device=x.device).index_add_(0, k, v)
        o = out.view(nseg, d)
        if accumulate:
            o += acc
        else:
            o.copy_(acc)
        if counts is not None:
            counts.copy_(torch.bincount(k, minlength=nseg).float())
        return out

    # ---- PEG
    def peg_fwd(self, x, w, bias):
        C = x.shape[-1]
        xc = _f(x).permute(0, 4, 1, 2, 3)
        y = F.conv3d(F.pad(xc, (1, 1, 1, 1, 2, 0)), w.view(C, 1, 3, 3, 3), bias, groups=C)
        return (y.permute(0, 2, 3, 4, 1) + _f(x)).to(x.dtype).contiguous()

    def attn2_bwd_tok(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, qinv, kinv, dq, dk, dv, dq_scale, dk_scale, nseq, L,
                      want_dtab, defer_dtab=False):
        assert not defer_dtab, "the checker has no streams: the table gradient is never deferred on the CPU"
        dqh, dkh, dvh, dtab = self.attn2_bwd(qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, nseq, L, want_dtab)
        ws = None
        self.attn2_unprep(dqh, dkh, dvh, qh, kh, qinv, kinv, q_scale, k_scale, scale, dq, dk, dv, dq_scale, dk_scale)
        return dtab, ws

    def attn2_bwd_fused(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, qinv, kinv, dq, dk, dv, dq_scale, dk_scale, nseq, L,
                        want_dtab):
        dqh, dkh, dvh, dtab = self.attn2_bwd(qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, nseq, L, want_dtab)
        self.attn2_unprep(dqh, dkh, dvh, qh, kh, qinv, kinv, q_scale, k_scale, scale, dq, dk, dv, dq_scale, dk_scale)
        return (dtab,)

    def gemm_headnorm(self, a, b, sections):
        M, K = a.shape
        nsec = len(sections)
        if a.dtype != torch.bfloat16 or M % 256 or K % 64 or b.shape[0] != nsec * 256:
            return None
        y = (_f(a) @ _f(b).t()).to(a.dtype)                       # the bf16-rounded projection (what ctclip_attn2_prep reads)
        outs = []
        for i, (sc, mult) in enumerate(sections):
            t = _f(y[:, i * 256:(i + 1) * 256]).view(M, 8, 32)
            if sc is None:
                outs.append((t.permute(1, 0, 2).contiguous().to(a.dtype), None))
                continue
            inv = 1.0 / t.norm(dim=-1).clamp_min(1e-12)
            outs.append(((t * (inv[..., None] * sc * mult)).permute(1, 0, 2).contiguous().to(a.dtype), inv.contiguous()))
        return outs

    def peg_fwd_comp(self, x, w, bias, e_in=None):
        if x.dtype != torch.bfloat16:
            return None
        C = x.shape[-1]
        xc = _f(x).permute(0, 4, 1, 2, 3)
        s = F.conv3d(F.pad(xc, (1, 1, 1, 1, 2, 0)), w.view(C, 1, 3, 3, 3), bias, groups=C).permute(0, 2, 3, 4, 1) + _f(x)
        if e_in is not None:
            s = s + _f(e_in).view_as(s)
        y = s.to(x.dtype).contiguous()
        return y, (s - _f(y)).to(x.dtype).contiguous()

    def gemm_residual_comp(self, a, b, residual, comp):
        M, K = a.shape
        N = b.shape[0]
        if a.dtype != torch.bfloat16 or M % 256 or N % 128 or K % 64:
            return None
        s = _f(a) @ _f(b).t() + _f(residual) + _f(comp)
        y = s.to(a.dtype)
        return y, (s - _f(y)).to(a.dtype)

    def peg_bwd(self, dy, x, w, dw=None, db=None, want_dx=True):
        C = x.shape[-1]
        xx = _f(x).detach().requires_grad_(True)
        ww = w.detach().clone().requires_grad_(True)
        bb = torch.zeros(C, device=x.device, requires_grad=True)
        with torch.enable_grad():
            xc = xx.permute(0, 4, 1, 2, 3)
            y = F.conv3d(F.pad(xc, (1, 1, 1, 1, 2, 0)), ww.view(C, 1, 3, 3, 3), bb, groups=C).permute(0, 2, 3, 4, 1) + xx
            gx, gw, gb = torch.autograd.grad(y, (xx, ww, bb), _f(dy))
        if dw is not None:
            dw += gw
        if db is not None:
            db += gb
        return gx.to(x.dtype).contiguous() if want_dx else None

    # ---- attention
    def head_transpose(self, x, nseq, H, L, D):
        Lp = (L + 7) // 8 * 8
        xt = torch.zeros((nseq, H, D, Lp), dtype=x.dtype, device=x.device)
        xt[..., :L] = x[:, :H * D].reshape(nseq, L, H, D).permute(0, 2, 3, 1)
        return xt

    def qk_norm_fwd(self, x, scale_vec, H, D):
        M = x.shape[0]
        xf = _f(x[:, :H * D]).reshape(M, H, D)
        inv = 1.0 / xf.norm(dim=-1).clamp(min=1e-12)
        y = xf * inv[..., None] * scale_vec
        return y.reshape(M, H * D).to(x.dtype), inv

    def qk_norm_bwd(self, dy, x, inv, scale_vec, dx, dscale, H, D):
        M = x.shape[0]
        xf = _f(x[:, :H * D]).reshape(M, H, D)
        u = xf * inv[..., None]
        g = _f(dy[:, :H * D]).reshape(M, H, D) * scale_vec
        r = inv[..., None] * (g - u * (u * g).sum(-1, keepdim=True))
        dx[:, :H * D] = r.reshape(M, H * D).to(dx.dtype)
        if dscale is not None:
            dscale += (_f(dy[:, :H * D]).reshape(M, H, D) * u).sum((0, 1))
        return dx

    def _scores(self, q, k, bias, keymask, nseq, H, L, D, scale):
        qf = _f(q[:, :H * D]).reshape(nseq, L, H, D).permute(0, 2, 1, 3)
        kf = _f(k[:, :H * D]).reshape(nseq, L, H, D).permute(0, 2, 1, 3)
        s = torch.einsum("shid,shjd->shij", qf, kf) * scale
        if bias is not None:
            s = s + bias[None]
        if keymask is not None:
            s = s + keymask[:, None, None, :]
        return qf, kf, s

    # ---- Philox4x32-10 in torch (ct_clip_amd/csrc/common.h philox4x32): the dropout masks of the HIP kernels, bit for bit
    @staticmethod
    def philox(seed, index, stream):
        M32 = 0xFFFFFFFF
        c0, c1 = index & M32, (index >> 32) & M32
        c2, c3 = torch.full_like(index, stream), torch.zeros_like(index)
        k0, k1 = seed & M32, (seed >> 32) & M32
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2          # < 2^64: exact in int64 only up to 2^63, so split the product
            hi0, lo0 = (((0xD2511F53 >> 16) * c0 + ((0xD2511F53 & 0xFFFF) * c0 >> 16)) >> 16) & M32, p0 & M32
            hi1, lo1 = (((0xCD9E8D57 >> 16) * c2 + ((0xCD9E8D57 & 0xFFFF) * c2 >> 16)) >> 16) & M32, p1 & M32
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
        return c0, c1, c2, c3

    @staticmethod
    def _mult(word, p):
        return torch.where((word >> 8).to(torch.float32) * (1.0 / 16777216.0) >= torch.tensor(p, dtype=torch.float32), 1.0 / (1.0 - p), 0.0)

    # ---- attention, second generation (head-planar operands; spec of csrc/attn2.hip)
    LOG2E = 1.4426950408889634

    # short-sequence cosine attention (csrc/attn_short.hip): torch f32 on the same inputs
    def attn_short_supported(self, dtype, L, D):
        return dtype == torch.bfloat16 and D == 32 and 1 <= L <= 32

    @staticmethod
    def _short_graph(q, kv, q_scale, k_scale, nseq, L, H, scale):
        HD = H * 32
        qf, kf, vf = (t.float().view(nseq, L, H, 32).transpose(1, 2) for t in (q, kv[:, :HD], kv[:, HD:]))
        qn = F.normalize(qf, dim=-1) * q_scale
        kn = F.normalize(kf, dim=-1) * k_scale
        p = torch.softmax(qn @ kn.transpose(-1, -2) * scale, dim=-1)
        return (p @ vf).transpose(1, 2).reshape(nseq * L, HD)

    def attn_short_fwd(self, q, kv, q_scale, k_scale, nseq, L, H, scale):
        return self._short_graph(q, kv, q_scale.float(), k_scale.float(), nseq, L, H, scale).to(q.dtype)

    def attn_short_bwd(self, q, kv, q_scale, k_scale, do, nseq, L, H, scale, dq_scale=None, dk_scale=None):
        qd, kvd = q.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
        qs, ks = q_scale.detach().float().clone().requires_grad_(True), k_scale.detach().float().clone().requires_grad_(True)
        with torch.enable_grad():
            o = self._short_graph(qd, kvd, qs, ks, nseq, L, H, scale)
            dq, dkv, dqs, dks = torch.autograd.grad(o, (qd, kvd, qs, ks), do.float())
        if dq_scale is not None:
            dq_scale += dqs
        if dk_scale is not None:
            dk_scale += dks
        return dq.to(q.dtype), dkv.to(q.dtype)

    def attn2_supported(self, dtype, H, L, D, bias_grid, has_bias):
        if dtype != torch.bfloat16 or D != 32 or L % 32 or L < 64 or L > 1024:
            return False
        if has_bias:
            gh, gw = bias_grid
            return gh * gw == L and gw % 8 == 0 and (2 * gh - 1) * (2 * gw - 1) <= 4096
        return True

    def attn2_prep(self, q, k, v, q_scale, k_scale, scale, H):
        M = q.shape[0]
        c = scale * self.LOG2E

        def nrm(x, sv, mul):
            xf = _f(x).reshape(M, H, 32)
            inv = 1.0 / xf.norm(dim=-1).clamp_min(1e-12)
            return (xf * inv[..., None] * sv * mul).permute(1, 0, 2).contiguous().to(x.dtype), inv
        qh, qinv = nrm(q, q_scale, c)
        kh, kinv = nrm(k, k_scale, 1.0)
        vh = v.reshape(M, H, 32).permute(1, 0, 2).contiguous()
        return qh, kh, vh, qinv, kinv

    def _z(self, qh, kh, tab, bias_grid, nseq, L):
        H = qh.shape[0]
        z = torch.einsum("hsid,hsjd->hsij", _f(qh).reshape(H, nseq, L, 32), _f(kh).reshape(H, nseq, L, 32))     # log2-domain logits
        if tab is not None:
            gh, gw = bias_grid
            z = z + (tab.t()[:, self._cls(gh, gw, tab.device)] * self.LOG2E)[:, None]
        return z

    def attn2_fwd(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, nseq, L):
        H, M, _ = qh.shape
        z = self._z(qh, kh, tab, bias_grid, nseq, L)
        lse2 = torch.logsumexp(z * (1.0 / self.LOG2E), dim=-1) * self.LOG2E
        p = torch.exp2(z - lse2[..., None])
        o = torch.einsum("hsij,hsjd->hsid", p, _f(vh).reshape(H, nseq, L, 32))
        return o.permute(1, 2, 0, 3).reshape(M, H * 32).to(qh.dtype), lse2.reshape(H, M)

    def attn2_bwd(self, qh, kh, vh, tab, bias_grid, q_scale, k_scale, scale, o, dout, lse2, nseq, L, want_dtab):
        H, M, _ = qh.shape
        c = scale * self.LOG2E
        z = self._z(qh, kh, tab, bias_grid, nseq, L)
        p = torch.exp2(z - lse2.reshape(H, nseq, L)[..., None])
        do = _f(dout).reshape(nseq, L, H, 32).permute(2, 0, 1, 3)
        of = _f(o).reshape(nseq, L, H, 32).permute(2, 0, 1, 3)
        v4, k4, q4 = (_f(t).reshape(H, nseq, L, 32) for t in (vh, kh, qh))
        dp = torch.einsum("hsid,hsjd->hsij", do, v4)
        delta = (do * of).sum(-1)
        ds = p * (dp - delta[..., None])
        dqh = scale * torch.einsum("hsij,hsjd->hsid", ds, k4)
        dkh = (scale / c) * torch.einsum("hsij,hsid->hsjd", ds, q4)
        dvh = torch.einsum("hsij,hsid->hsjd", p, do)
        dtab = None
        if want_dtab and tab is not None:
            gh, gw = bias_grid
            ncls = (2 * gh - 1) * (2 * gw - 1)
            dtab = torch.zeros((ncls, H), dtype=torch.float32, device=qh.device)
            dtab.index_add_(0, self._cls(gh, gw, qh.device).reshape(-1), ds.sum(1).reshape(H, L * L).t().contiguous())
        return tuple(t.reshape(H, M, 32).to(qh.dtype) for t in (dqh, dkh, dvh)) + (dtab,)

    def attn2_unprep(self, dqh, dkh, dvh, qh, kh, qinv, kinv, q_scale, k_scale, scale, dq, dk, dv, dq_scale, dk_scale):
        H, M, _ = qh.shape
        c = scale * self.LOG2E

        def back(dxh, xh, inv, sv, mul, out, dsv):
            g_hat = _f(dxh).permute(1, 0, 2)                       # (M, H, 32) gradient w.r.t. x^ = u * sv
            u = _f(xh).permute(1, 0, 2) / (sv * mul)
            if dsv is not None:
                dsv += (g_hat * u).sum((0, 1))
            g = g_hat * sv
            out.copy_((inv[..., None] * (g - u * (u * g).sum(-1, keepdim=True))).reshape(M, H * 32).to(out.dtype))
        back(dqh, qh, qinv, q_scale, c, dq, dq_scale)
        back(dkh, kh, kinv, k_scale, 1.0, dk, dk_scale)
        dv.copy_(dvh.permute(1, 0, 2).reshape(M, H * 32))

    def dropout(self, x, residual, p, seed, stream_id):
        n = x.numel()
        idx = torch.arange(n // 4, dtype=torch.int64)            # integer Philox on the CPU (no device JIT of int64 kernels)
        words = torch.stack(self.philox(seed, idx, stream_id), dim=1).reshape(-1)
        y = _f(x).reshape(-1) * self._mult(words, p).to(x.device)
        if residual is not None:
            y = y + _f(residual).reshape(-1)
        return y.reshape(x.shape).to(x.dtype)

    def attn_dropout_mask(self, nseq, H, L, p, seed, device):
        """csrc/common.h attn_drop_block / attn_drop_pick (round 6): one Philox call (stream 1) per block of 2 queries x 4 keys, 16 bits per element."""
        import math
        import numpy as np
        nq2, nk4 = (L + 1) // 2, (L + 3) // 4
        nblk = nseq * H * nq2 * nk4
        words = torch.stack(self.philox(seed, torch.arange(nblk, dtype=torch.int64), 1), dim=0)          # (4, nblk)
        sh = torch.arange(nseq * H, dtype=torch.int64)[:, None, None]
        qi = torch.arange(L, dtype=torch.int64)[None, :, None]
        kj = torch.arange(L, dtype=torch.int64)[None, None, :]
        blk = (sh * nq2 + (qi >> 1)) * nk4 + (kj >> 2)
        wi = (qi & 1) * 2 + ((kj & 3) >> 1) + torch.zeros_like(blk)
        word = words.reshape(-1)[wi * nblk + blk]
        u = (word >> (16 * (kj & 1))) & 0xFFFF
        thr = int(math.floor(float(np.float32(np.float32(p) * np.float32(65536.0)) + np.float32(0.5))))
        keep = torch.where(u >= thr, 1.0 / (1.0 - p), 0.0).to(torch.float32)
        return keep.reshape(nseq, H, L, L).to(device)

    def attn_fwd(self, q, k, vt, bias, keymask, nseq, H, L, D, scale, want_lse=True, bias_grid=None, dropout=None):
        if bias_grid is not None:
            bias = self.cpb_expand(bias, *bias_grid)
        qf, kf, s = self._scores(q, k, bias, keymask, nseq, H, L, D, scale)
        lse = torch.logsumexp(s, dim=-1)
        p = torch.exp(s - lse[..., None])
        if dropout is not None:
            p = p * self.attn_dropout_mask(nseq, H, L, dropout[0], dropout[1], p.device)
        v = _f(vt[..., :L]).permute(0, 1, 3, 2)  # (nseq, H, L, D)
        o = torch.einsum("shij,shjd->shid", p, v).permute(0, 2, 1, 3).reshape(nseq * L, H * D)
        return o.to(q.dtype).contiguous(), lse

    def attn_bwd(self, q, k, v, qt, kt, o, dout, dot, lse, bias, keymask, dq, dk, dv, dbias, nseq, H, L, D, scale, bias_grid=None,
                 dropout=None):
        if bias_grid is not None:
            bias = self.cpb_expand(bias, *bias_grid)
        qf, kf, s = self._scores(q, k, bias, keymask, nseq, H, L, D, scale)
        p = torch.exp(s - lse[..., None])
        dm = self.attn_dropout_mask(nseq, H, L, dropout[0], dropout[1], p.device) if dropout is not None else None
        vf = _f(v[:, :H * D]).reshape(nseq, L, H, D).permute(0, 2, 1, 3)
        dof = _f(dout[:, :H * D]).reshape(nseq, L, H, D).permute(0, 2, 1, 3)
        of = _f(o[:, :H * D]).reshape(nseq, L, H, D).permute(0, 2, 1, 3)
        delta = (dof * of).sum(-1)
        dvf = torch.einsum("shij,shid->shjd", p if dm is None else p * dm, dof)
        dp = torch.einsum("shid,shjd->shij", dof, vf)
        if dm is not None:
            dp = dp * dm
        ds = p * (dp - delta[..., None])
        dqf = torch.einsum("shij,shjd->shid", ds, kf) * scale
        dkf = torch.einsum("shij,shid->shjd", ds, qf) * scale
        back = lambda t: t.permute(0, 2, 1, 3).reshape(nseq * L, H * D)
        dq[:, :H * D] = back(dqf).to(dq.dtype)
        dk[:, :H * D] = back(dkf).to(dk.dtype)
        dv[:, :H * D] = back(dvf).to(dv.dtype)
        if dbias is not None:
            if bias_grid is not None:
                dbias.copy_(self.cpb_reduce(ds.sum(0), *bias_grid))   # table mode: overwritten (ctclip_attn_bwd)
            else:
                dbias += ds.sum(0)

    # ---- elementwise
    def accumulate(self, dst, src):
        dst += src
        return dst

    def geglu_weight_interleave(self, w, hp, dtype):
        two_inner, K = w.shape
        inner = two_inner // 2
        out = torch.zeros((2 * hp, K), dtype=torch.float32, device=w.device)
        j = torch.arange(inner, device=w.device)
        out[8 * (j // 4) + j % 4] = w[:inner].float()
        out[8 * (j // 4) + 4 + j % 4] = w[inner:].float()
        return out.to(dtype)

    def shadow_refresh(self, jobs, version):
        """csrc/shadow.hip restated: dst = gather (plain: dst[r][c] = src[map(r)][c]; transposed: dst[r][c] = src[map(c)][r]), 0 outside."""
        for j in jobs:
            src, dst, mp, aux, tr = j["src_ref"]().detach().float(), j["dst"], j["map"], j["aux"], j["transposed"]
            ext = dst.shape[1] if tr else dst.shape[0]          # the mapped extent
            other = dst.shape[0] if tr else dst.shape[1]        # extent along the source columns
            n = torch.arange(ext)
            if mp == 0:
                idx = torch.where(n < src.shape[0], n, torch.full_like(n, -1))
            elif mp == 1:
                hp = ext // 2
                idx = torch.where(n < hp, torch.where(n < aux, n, torch.full_like(n, -1)),
                                  torch.where(n - hp < aux, aux + n - hp, torch.full_like(n, -1)))
            else:
                jj, part = 4 * (n >> 3) + (n & 3), (n >> 2) & 1
                idx = torch.where(jj < aux, part * aux + jj, torch.full_like(n, -1))
            m = torch.zeros(ext, other, dtype=torch.float32, device=src.device)
            ok = idx >= 0
            kc = min(other, src.shape[1])
            m[ok, :kc] = src[idx[ok].to(src.device)][:, :kc]
            dst.copy_((m.t() if tr else m).to(dst.dtype))

    def gemm_dgeglu(self, dy, wt, u):
        """out-projection grad-input + GEGLU backward: dg = dy wt^T stays f32 (never rounded to the storage dtype)."""
        if dy.dtype != torch.bfloat16 or dy.shape[0] % 256 or wt.shape[0] % 128:
            return None
        dg = _f(dy) @ _f(wt).t()
        uu = _f(u).detach().requires_grad_(True)
        with torch.enable_grad():
            a, gt = uu.chunk(2, dim=-1)
            (gu,) = torch.autograd.grad(a * F.gelu(gt), uu, dg)
        return gu.to(u.dtype)

    def _geglu_parts(self, x, w_il, hp):
        y = _f(x) @ _f(w_il).t()                                  # interleaved columns
        y = y.view(x.shape[0], hp // 4, 2, 4)
        return y[:, :, 0, :].reshape(x.shape[0], hp), y[:, :, 1, :].reshape(x.shape[0], hp)

    def gemm_geglu(self, x, w_il, hp, save_u=True):
        if x.dtype != torch.bfloat16 or x.shape[0] % 256 or (2 * hp) % 256:
            return None
        xs, gate = self._geglu_parts(x, w_il, hp)
        g = xs * F.gelu(gate)
        return (torch.cat([xs, gate], dim=1).to(x.dtype) if save_u else None), g.to(x.dtype)

    def gemm_geglu_bwd(self, x, w_il, dg, hp):
        """du = [dg gelu(gate) | dg x gelu'(gate)] with (x, gate) recomputed in f32 from the layer input (no rounding of u to bf16)."""
        if x.dtype != torch.bfloat16 or x.shape[0] % 256 or (2 * hp) % 256:
            return None
        xs, gate = self._geglu_parts(x, w_il, hp)
        uu = torch.cat([xs, gate], dim=1).detach().requires_grad_(True)
        with torch.enable_grad():
            a, gt = uu.chunk(2, dim=-1)
            (gu,) = torch.autograd.grad(a * F.gelu(gt), uu, _f(dg))
        return gu.to(x.dtype)

    def geglu_fwd(self, u):
        a, g = _f(u).chunk(2, dim=-1)
        return (a * F.gelu(g)).to(u.dtype)

    def geglu_bwd(self, dg, u):
        uu = _f(u).detach().requires_grad_(True)
        with torch.enable_grad():
            a, g = uu.chunk(2, dim=-1)
            y = a * F.gelu(g)
            (gu,) = torch.autograd.grad(y, uu, _f(dg))
        return gu.to(u.dtype)

    def gelu_fwd(self, u):
        return F.gelu(_f(u)).to(u.dtype)

    def gelu_bwd(self, dh, u):
        uu = _f(u).detach().requires_grad_(True)
        with torch.enable_grad():
            (gu,) = torch.autograd.grad(F.gelu(uu), uu, _f(dh))
        return gu.to(u.dtype)

    def leaky_relu_fwd(self, x, slope):
        return F.leaky_relu(x, slope)

    def leaky_relu_bwd(self, dy, x, slope):
        return torch.where(x > 0, dy, dy * slope)

    def colsum(self, x, out, N=None):
        N = x.shape[1] if N is None else N
        out += _f(x[:, :N]).sum(0)
        return out

    def patch_embed_param_bwd(self, G, W, g1, b1, dbp, dW, dg1, db1, accumulate):
        vW, vg, vb = G * g1[None, :] + dbp[:, None] * b1[None, :], (W * G).sum(0), W.t() @ dbp
        if accumulate:
            dW.add_(vW); dg1.add_(vg); db1.add_(vb)
        else:
            dW.copy_(vW); dg1.copy_(vg); db1.copy_(vb)

    def permute0213(self, x):
        return x.permute(0, 2, 1, 3).contiguous()

    def transpose2d(self, x):
        return x.t().contiguous()

    def pool_fwd(self, x, out_dtype=None):
        return _f(x).mean(1).to(out_dtype or x.dtype)

    def pool_bwd(self, dy, t, out_dtype=None):
        return (_f(dy)[:, None, :] / t).expand(-1, t, -1).to(out_dtype or dy.dtype).contiguous()

    def convert_pad(self, src, rows_dst, cols_dst, dtype, colscale=None, out=None):
        rows, cols = src.shape
        if out is None:
            out = torch.empty((rows_dst, cols_dst), dtype=dtype, device=src.device)
        out.zero_()
        v = _f(src)
        if colscale is not None:
            v = v * colscale
        out[:rows, :cols] = v.to(out.dtype)
        return out

    def _cls(self, gh, gw, device):
        L = gh * gw
        i = torch.arange(L, device=device)
        iy, ix = i // gw, i % gw
        return (iy[:, None] - iy[None, :] + gh - 1) * (2 * gw - 1) + (ix[:, None] - ix[None, :] + gw - 1)

    def cpb_expand(self, tab, gh, gw):
        cls = self._cls(gh, gw, tab.device)
        return tab[cls].permute(2, 0, 1).contiguous()

    def cpb_reduce(self, dbias, gh, gw):
        H = dbias.shape[0]
        cls = self._cls(gh, gw, dbias.device).reshape(-1)
        dtab = torch.zeros(((2 * gh - 1) * (2 * gw - 1), H), device=dbias.device)
        dtab.index_add_(0, cls, dbias.reshape(H, -1).t())
        return dtab

    def bert_embed_fwd(self, ids, word, pos, type0, dtype):
        B, T = ids.shape
        x = word[ids] + pos[:T][None] + type0[None, None]
        return x.reshape(B * T, -1).to(dtype)

    def bert_embed_bwd(self, ids, dx, dword, dpos, dtype0):
        B, T = ids.shape
        g = _f(dx)
        if dword is not None:
            dword.index_add_(0, ids.reshape(-1), g)
        if dpos is not None:
            dpos[:T] += g.reshape(B, T, -1).sum(0)
        if dtype0 is not None:
            dtype0[0] += g.sum(0)

    # ---- VQ
    def vq_gather(self, embed, idx, dtype):
        return embed[idx.reshape(-1)].to(dtype)

    def vq_ema(self, idx, x, inv, cluster_size, embed, decay):
        C, d = embed.shape
        xn = _f(x) * inv[:, None]
        stats = torch.zeros(C * (d + 1), dtype=torch.float32, device=embed.device)      # one buffer [bins | esum], as the HIP backend
        bins, esum = stats[:C], stats[C:].view(C, d)
        bins.index_add_(0, idx.reshape(-1), torch.ones(idx.numel(), device=embed.device))
        esum.index_add_(0, idx.reshape(-1), xn)
        return bins, esum

    def vq_ema_update(self, cluster_size, embed, bins, esum, decay):
        cluster_size.mul_(decay).add_(bins * (1 - decay))
        enorm = F.normalize(esum / bins.clamp(min=1.0)[:, None], dim=-1)
        enorm = torch.where((bins == 0)[:, None], F.normalize(embed, dim=-1), enorm)
        embed.mul_(decay).add_(enorm * (1 - decay))

    # ---- head
    def visual_latent_fwd(self, x, w):
        return _f(x) @ _f(w).t()

    def visual_latent_bwd(self, dy, x, w, dw=None, accumulate=False, want_dx=True):
        if dw is not None:
            g = dy.t() @ _f(x)
            if accumulate:
                dw += g
            else:
                dw.copy_(g)
        return (dy @ _f(w)).to(x.dtype) if want_dx else None

    def clip_loss(self, tl, il, temperature, want_grads=True, want_logits=False):
        a = tl.detach().clone().requires_grad_(True)
        b = il.detach().clone().requires_grad_(True)
        th = temperature.detach().clone().reshape(1).requires_grad_(True)
        with torch.enable_grad():
            s = F.normalize(a, dim=-1) @ F.normalize(b, dim=-1).t() * th.exp()
            e1, e2 = s.exp(), s.t().exp()
            l1 = (-torch.log(e1.diagonal() + 1e-20) + torch.log(e1.sum(-1) + 1e-20)).mean()
            l2 = (-torch.log(e2.diagonal() + 1e-20) + torch.log(e2.sum(-1) + 1e-20)).mean()
            loss = (l1 + l2) / 2
            if want_grads:
                ga, gb, gt = torch.autograd.grad(loss, (a, b, th))
            else:
                ga = gb = gt = None
        out = torch.stack([loss.detach(), th.detach().exp()[0]])
        return out, (s.detach() if want_logits else None), ga, gb, gt

    def scale_by_scalar(self, x, scalar):
        x.mul_(scalar)
        return x

    # ---- optimiser
    def relu_dropout(self, x, dy, p, seed, stream_id):
        n4 = x.numel() // 4
        if p > 0:
            idx = torch.arange(n4, dtype=torch.int64, device=x.device)
            w = torch.stack(self.philox(torch.full_like(idx, seed), idx, stream_id), dim=1).reshape(-1)
            keep = self._mult(w, p).reshape(x.shape)
        else:
            keep = torch.ones_like(x)
        return (torch.relu(x) * keep) if dy is None else (dy * keep * (x > 0))

    def bce_logits(self, logits, targets, pos_weight):
        lg = logits.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            loss = F.binary_cross_entropy_with_logits(lg, targets, pos_weight=pos_weight)
            (g,) = torch.autograd.grad(loss, lg)
        return loss.detach().reshape(1), g

    def pair_softmax_mse(self, sims):
        s = sims.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            p = torch.softmax(s, dim=1)
            target = torch.tensor([1.0, 0.0], device=s.device).expand_as(p)
            loss = F.mse_loss(p.reshape(-1), target.reshape(-1))
            (g,) = torch.autograd.grad(loss, s)
        return loss.detach().reshape(1), g

    def latent_similarity(self, text, image, temperature, dsims=None):
        t = text.detach().clone().requires_grad_(True)
        v = image.detach().clone().requires_grad_(True)
        tp = temperature.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            sims = (F.normalize(t, dim=-1) * F.normalize(v, dim=-1)).sum(-1) * tp.exp()
            if dsims is None:
                return sims.detach()
            dt, dv, dtp = torch.autograd.grad(sims, (t, v, tp), dsims)
        return dt, dv, dtp.reshape(1)

    def grad_norm_clip(self, g, max_norm, extra_sq=None):
        sq = (g.double() ** 2).sum()
        if extra_sq is not None:
            sq = sq + extra_sq.double()[0]
        norm = sq.sqrt().float()
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0) if max_norm else torch.ones_like(norm)
        return torch.stack([norm, coef])

    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, step, weight_decay=0.0, clip=None, decay_mask4=None, zero_grad=False):
        gg = g * (clip[1] if clip is not None else 1.0)
        if weight_decay:
            if decay_mask4 is None:
                p.mul_(1 - lr * weight_decay)
            else:
                keep = decay_mask4.to(torch.bool).repeat_interleave(4)[:p.numel()]
                p.mul_(torch.where(keep, torch.full_like(p, 1 - lr * weight_decay), torch.ones_like(p)))
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        p.addcdiv_(m, v.sqrt() / (bc2 ** 0.5) + eps, value=-lr / bc1)
        if zero_grad:
            g.zero_()
