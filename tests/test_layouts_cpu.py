"""The index arithmetic the HIP kernels rest on, restated in Python and checked exhaustively (no GPU): operand permutations are
bijections, LDS images are bank-conflict free for the service groups MI355X_MICROARCH.md lists, the transposing-read image of
gemm_tn.hip delivers the fragment the MFMA expects under the lane mapping measured by tools/tr_probe.hip, and the relative-position
bias classes of a fragment run are consecutive.  Each function names the kernel lines it mirrors."""
import itertools

B128_GROUPS = [[*range(0, 4), *range(12, 16), *range(20, 28)], [*range(4, 12), *range(16, 20), *range(28, 32)]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]          # ds_read_b128: four groups of 16 lanes, one LDS cycle each


def slots16(addrs):
    """16-byte slot of the 256-byte bank row touched by each address"""
    return [(a // 16) % 16 for a in addrs]


def test_gemm_nt_b_row_permutation_and_epilogue_columns():
    # gemm_nt.hip enter_b: LDS row rho of the B panel holds tile column (rho & 0x80) + (rho & 15) * 8 + ((rho >> 4) & 7)
    col = lambda rho: (rho & 0x80) + ((rho & 15) << 3) + ((rho >> 4) & 7)
    assert sorted(col(r) for r in range(256)) == list(range(256))
    # fragment b of wave column wn reads LDS rows wn*128 + b*16 + li: lane li then owns columns wn*128 + li*8 + b, b = 0..7
    for wn, li in itertools.product(range(2), range(16)):
        assert [col(wn * 128 + b * 16 + li) for b in range(8)] == [wn * 128 + li * 8 + b for b in range(8)]


def test_gemm_nt_fragment_reads_are_bank_conflict_free():
    # gemm_nt.hip swz(): row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); lane (li, lg) reads row base + li, chunk ks*4 + lg
    swz = lambda row, chunk: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
    for base, ks in itertools.product(range(0, 256, 16), range(2)):
        for grp in B128_GROUPS:
            addrs = [swz(base + (l & 15), ks * 4 + (l >> 4)) for l in grp]
            assert len(set(slots16(addrs))) == 16


def tr_read(lds, lane_addr):
    """ds_read_b64_tr_b16 as measured (tools/tr_probe.hip): in each 16-lane group, output lane i element j = element i & 3 of the
    four bf16 addressed by lane 4 j + (i >> 2).  lds: dict byte address -> value of the 2-byte element at that address."""
    out = {}
    for g in range(4):
        for i in range(16):
            out[g * 16 + i] = [lds[lane_addr[g * 16 + 4 * j + (i >> 2)] + 2 * (i & 3)] for j in range(4)]
    return out


def test_gemm_tn_transposed_fragments():
    # gemm_tn.hip loader: piece = wave*4 + j, quarter qd = piece >> 3, k row r = (piece & 7) * 8 + (lane >> 3), LDS position
    # pos = lane & 7 holds source sub-chunk c = (pos >> 1) ^ ((r >> 1) & 3), half pos & 1  ->  tile column qd*64 + c*16 + (pos&1)*8 + e
    lds = {}
    for piece, lane in itertools.product(range(32), range(64)):
        qd, r, pos = piece >> 3, (piece & 7) * 8 + (lane >> 3), lane & 7
        c = (pos >> 1) ^ ((r >> 1) & 3)
        col0 = qd * 64 + c * 16 + (pos & 1) * 8
        for e in range(8):
            lds[piece * 1024 + lane * 16 + 2 * e] = (r, col0 + e)          # (k, column) stored at this LDS address
    assert len(lds) == 64 * 256                                            # every (k, column) of the 64 x 256 panel exactly once
    assert sorted(lds.values()) == [(k, c) for k in range(64) for c in range(256)]
    # fragment read (read_frag / fadr): lane (li, lg) addresses row ks*32 + 16*second + lg*4 + (li >> 2), sub-chunk c, piece li & 3
    for q, c, ks in itertools.product(range(4), range(4), range(2)):
        frag = {l: [] for l in range(64)}
        for second in range(2):
            addr = {}
            for l in range(64):
                li, lg = l & 15, l >> 4
                frow = lg * 4 + (li >> 2)
                r = ks * 32 + 16 * second + frow
                addr[l] = q * 8192 + r * 128 + ((c ^ ((frow >> 1) & 3)) << 5) + (li & 3) * 8
                # bank check: a half-wave (32 lanes) must cover 32 distinct 8-byte slots of the 256-byte bank row
            for half in range(2):
                assert len({(addr[l] // 8) % 32 for l in range(half * 32, half * 32 + 32)}) == 32
            got = tr_read(lds, addr)
            for l in range(64):
                frag[l] += got[l]
        for l in range(64):
            li, lg = l & 15, l >> 4
            # lane (li, lg) holds column q*64 + c*16 + li at the eight k-slots 4 lg + j and 16 + 4 lg + j of this sub-step
            assert frag[l] == [(ks * 32 + s * 16 + lg * 4 + j, q * 64 + c * 16 + li) for s in range(2) for j in range(4)]


def test_attention_lds_tile_rows_are_conflict_free():
    # attn.hip lds_frag: tile + pi32(c) * 80 + half * 16 (+ 32): rows are 80 bytes apart
    pi32 = lambda c: (c & 3) | ((c & 4) << 1) | ((c & 8) >> 1) | (c & 16)
    for off in (0, 32):
        for grp in B128_GROUPS:
            addrs = [pi32(l & 31) * 80 + (l >> 5) * 16 + off for l in grp]
            assert len(set(slots16(addrs))) == 16


def test_relative_bias_classes_of_a_fragment_run_are_consecutive():
    # attn.hip tile_logits_fast: u(t) = (t // gw) * (2 gw - 1) + t % gw, class(i, j) = u(i) - u(j) + c0; the eight keys of a run
    # (8-aligned) lie in one image row when gw % 8 == 0, so class(i, j0 + e) = class(i, j0) - e
    for gh, gw in ((24, 24), (3, 8), (5, 16)):
        u = lambda t: (t // gw) * (2 * gw - 1) + t % gw
        c0 = (gh - 1) * (2 * gw - 1) + (gw - 1)
        cls = lambda i, j: (i // gw - j // gw + gh - 1) * (2 * gw - 1) + (i % gw - j % gw + gw - 1)
        L = gh * gw
        for i in range(L):
            for j0 in range(0, L, 8):
                base = u(i) - u(j0) + c0
                assert [cls(i, j0 + e) for e in range(8)] == [base - e for e in range(8)]
                assert 0 <= base - 7 and base < (2 * gh - 1) * (2 * gw - 1)


# ---------------------------------------------------------------------------------------------------- attn2.hip
A2_SWZ = lambda row, chunk: row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)
A2_PI32 = lambda c: (c & 3) | ((c & 4) << 1) | ((c & 8) >> 1) | (c & 16)


def a2_image():
    """loader of attn2.hip: thread (row, chunk) stores elements (row, 8 chunk + e) of a 32 x 32 bf16 tile at swz(row, chunk) + 2 e"""
    lds = {}
    for row, chunk, e in itertools.product(range(32), range(4), range(8)):
        lds[A2_SWZ(row, chunk) + 2 * e] = (row, 8 * chunk + e)
    assert len(lds) == 32 * 32
    return lds


def test_attn2_row_fragments_are_bank_conflict_free():
    # attn2.hip lds_rows: two b128 reads at swz(row, half) and swz(row, 2 + half); row = pi32(lane & 31) (A operands) or lane & 31 (B)
    for perm in (A2_PI32, lambda c: c):
        for off in (0, 2):
            for grp in B128_GROUPS:
                addrs = [A2_SWZ(perm(l & 31), off + (l >> 5)) for l in grp]
                assert len(set(slots16(addrs))) == 16


def test_attn2_transposed_fragments():
    # attn2.hip tr_offsets / lds_cols: lane (i = lane & 31, half) must receive tokens 8 half + e and 16 + 8 half + e of head dim pi32(i)
    lds = a2_image()

    def tr_off(lane):
        t, grp, half = lane & 15, (lane >> 4) & 1, lane >> 5
        sg = ((t & 1) << 1) | ((t >> 1) & 1)
        return [A2_SWZ(8 * half + 4 * j0 + (t >> 2), 2 * grp + (sg >> 1)) + 8 * (sg & 1) for j0 in range(2)]
    frag = {l: [] for l in range(64)}
    for base in (0, 1024):
        for j0 in range(2):
            addr = {l: base + tr_off(l)[j0] for l in range(64)}
            for hw in range(2):    # a half-wave must cover 32 distinct 8-byte slots of the 256-byte bank row
                assert len({(addr[l] // 8) % 32 for l in range(hw * 32, hw * 32 + 32)}) == 32
            got = tr_read(lds, addr)
            for l in range(64):
                frag[l] += got[l]
    for l in range(64):
        i, half = l & 31, l >> 5
        assert frag[l] == [(8 * half + e, A2_PI32(i)) for e in range(8)] + [(16 + 8 * half + e, A2_PI32(i)) for e in range(8)]


def test_attn2_bias_tile_addresses():
    # attn2.hip bias_tile: rows = keys -> reversed table entry ncls - 1 - (u(q) - u(k0) + c0) + e ; rows = queries -> u(q0) - u(k) + c0 + e
    for gh, gw in ((24, 24), (3, 8), (5, 16)):
        u = lambda t: (t // gw) * (2 * gw - 1) + t % gw
        c0 = (gh - 1) * (2 * gw - 1) + (gw - 1)
        ncls = (2 * gh - 1) * (2 * gw - 1)
        cls = lambda i, j: (i // gw - j // gw + gh - 1) * (2 * gw - 1) + (i % gw - j % gw + gw - 1)
        L = gh * gw
        for i in range(L):
            for j0 in range(0, L, 8):
                rbase = ncls - 1 - (u(i) - u(j0) + c0)
                assert [ncls - 1 - cls(i, j0 + e) for e in range(8)] == [rbase + e for e in range(8)] and 0 <= rbase and rbase + 7 < ncls
                fbase = u(j0) - u(i) + c0
                assert [cls(j0 + e, i) for e in range(8)] == [fbase + e for e in range(8)] and 0 <= fbase and fbase + 7 < ncls


def test_attn2_work_item_decode_covers_every_item_once():
    # attn2.hip decode_item: workgroup b -> item (b & 7) * per + (b >> 3), per = ceil(nitems / 8); grid = 8 * per
    for nitems in (1, 7, 8, 9, 4608, 4611):
        per = (nitems + 7) >> 3
        seen = [(b & 7) * per + (b >> 3) for b in range(8 * per)]
        assert sorted(w for w in seen if w < nitems) == list(range(nitems))


# ---------------------------------------------------------------------------------------------------- PEG marching kernels
def _peg_march_model(x, w, bias, direction, TB, PSEG=4, NS=6):
    """csrc/peg_lds.hip peg_march_kernel step for step: ring of NS LDS slots (halo rows / zero columns), plane m + NS - 3 fetched in
    step m, tap d1 <-> slot (m + d1 - 2) % NS, row r + d2, input column jj scattered into three rotating accumulators."""
    DEPTH = NS - 3
    import numpy as np
    B, D1, D2, D3, C = x.shape
    L = D3 // PSEG
    y = np.zeros_like(x)
    wk = w.reshape(C, 3, 3, 3).copy()
    if direction < 0:
        wk = wk[:, :, ::-1, ::-1].copy()
    wk[:, 2, 1, 1] += 1.0                                       # the residual in the centre tap
    bv = bias if direction > 0 else np.zeros(C, x.dtype)
    for b in range(B):
        for beta0 in range(0, D2, TB):
            slots = np.zeros((NS, TB + 2, D3 + 2, C), x.dtype)
            plane_of = (lambda m: m) if direction > 0 else (lambda m: D1 - 1 - m)

            def dma(m):
                for i in range(TB + 2):
                    beta = beta0 - 1 + i
                    if 0 <= beta < D2:
                        slots[m % NS, i, 1:D3 + 1] = x[b, plane_of(m), beta]
            for m in range(min(DEPTH, D1)):
                dma(m)
            for m in range(D1):
                if m + DEPTH < D1:
                    dma(m + DEPTH)                               # overwrites the slot of march plane m - 3
                for r in range(TB):
                    if beta0 + r >= D2:
                        continue
                    for seg in range(PSEG):
                        g0 = seg * L
                        accm, acc0, accp = np.zeros(C, x.dtype), bv.copy(), bv.copy()
                        for jj in range(L + 2):
                            for d1 in range(3):
                                for d2 in range(3):
                                    xs = slots[(m + d1 - 2) % NS, r + d2, g0 + jj]
                                    if jj <= L - 1:
                                        accp = accp + wk[:, d1, d2, 0] * xs
                                    if 1 <= jj <= L:
                                        acc0 = acc0 + wk[:, d1, d2, 1] * xs
                                    if jj >= 2:
                                        accm = accm + wk[:, d1, d2, 2] * xs
                            if jj >= 2:
                                y[b, plane_of(m), beta0 + r, g0 + jj - 2] = accm
                            accm, acc0, accp = acc0, accp, bv.copy()
    return y


def _peg_wgrad_model(dy, x, TB, PSEG=4, NS=5):
    """peg_wgrad_march_kernel: own dy window (zero outside the thread's quarter), x column jj meets dy[col + 1], dy[col], dy[col - 1]."""
    import numpy as np
    B, D1, D2, D3, C = x.shape
    L = D3 // PSEG
    dw, db = np.zeros((C, 3, 3, 3), x.dtype), np.zeros(C, x.dtype)
    for b in range(B):
        for beta0 in range(0, D2, TB):
            slots = np.zeros((NS, TB + 2, D3 + 2, C), x.dtype)
            gsl = np.zeros((2, TB, D3, C), x.dtype)

            def dma_x(m):
                for i in range(TB + 2):
                    beta = beta0 - 1 + i
                    if 0 <= beta < D2:
                        slots[m % NS, i, 1:D3 + 1] = x[b, m, beta]

            def dma_g(m):
                for i in range(TB):                              # (rows past D2 are never written: they stay zero)
                    if beta0 + i < D2:
                        gsl[m & 1, i] = dy[b, m, beta0 + i]
            dma_x(0), dma_g(0)
            if D1 > 1:
                dma_x(1)
            for m in range(D1):
                if m + 1 < D1:
                    dma_g(m + 1)
                if m + 2 < D1:
                    dma_x(m + 2)
                for r in range(TB):
                    for seg in range(PSEG):
                        g0 = seg * L
                        gm, gc = np.zeros(C, x.dtype), np.zeros(C, x.dtype)
                        for jj in range(L + 2):
                            gq = gsl[m & 1, r, g0 + jj] if jj <= L - 1 else np.zeros(C, x.dtype)
                            if jj <= L - 1:
                                db += gq
                            for d1 in range(3):
                                for d2 in range(3):
                                    xs = slots[(m + d1 - 2) % NS, r + d2, g0 + jj]
                                    if jj <= L - 1:
                                        dw[:, d1, d2, 0] += gq * xs
                                    if 1 <= jj <= L:
                                        dw[:, d1, d2, 1] += gc * xs
                                    if jj >= 2:
                                        dw[:, d1, d2, 2] += gm * xs
                            gm, gc = gc, gq
    return dw.reshape(C, 27), db


def test_peg_marching_kernels_index_arithmetic():
    """The march (ring slots, mirrored taps of the grad-in pass, quarter-row scatter, own-dy window of the weight gradient) against
    F.conv3d with the reference's causal padding (attention.py:63-84) and its autograd."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    for (B, D1, D2, D3, C, TB) in [(1, 4, 5, 8, 2, 4), (2, 3, 12, 4, 1, 12), (1, 5, 13, 8, 2, 12)]:
        x = torch.randn(B, D1, D2, D3, C, generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn(C, 27, generator=g, dtype=torch.float64, requires_grad=True)
        bias = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(B, D1, D2, D3, C, generator=g, dtype=torch.float64)
        xc = x.permute(0, 4, 1, 2, 3)
        yr = F.conv3d(F.pad(xc, (1, 1, 1, 1, 2, 0)), w.view(C, 1, 3, 3, 3), bias, groups=C).permute(0, 2, 3, 4, 1) + x
        dxr, dwr, dbr = torch.autograd.grad(yr, (x, w, bias), dy)
        y = _peg_march_model(x.detach().numpy(), w.detach().numpy(), bias.detach().numpy(), +1, TB)
        np.testing.assert_allclose(y, yr.detach().numpy(), rtol=1e-10, atol=1e-10)
        dx = _peg_march_model(dy.numpy(), w.detach().numpy(), None, -1, TB)
        np.testing.assert_allclose(dx, dxr.numpy(), rtol=1e-10, atol=1e-10)
        dw, db = _peg_wgrad_model(dy.numpy(), x.detach().numpy(), TB)
        np.testing.assert_allclose(dw, dwr.numpy(), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(db, dbr.numpy(), rtol=1e-10, atol=1e-10)


def test_peg_marching_lds_reads_are_bank_conflict_free():
    """ds_read_b32 is served in two groups of 32 lanes; a group = two rows of the tile x 16 channel pairs.  With an odd number of
    64-byte positions per LDS row the two rows fall into different halves of the 128-byte bank window."""
    for D3 in (8, 16, 24, 32):
        for rsp, what in (((D3 + 2) | 1, "x ring"), (D3 | 1, "dy plane")):
            for g0 in range(0, D3, D3 // 4):
                for jj in range(D3 // 4 + 2):
                    for half in (0, 1):
                        banks = set()
                        for lane in range(32 * half, 32 * half + 32):
                            r, pr = lane >> 4, lane & 15
                            a = (r * rsp + g0 + jj) * 64 + pr * 4
                            banks.add((a // 4) % 32)
                        assert len(banks) == 32, (D3, what, g0, jj)


# ---------------------------------------------------------------------------------------------------- short-sequence attention
class _ShortAttnModel:
    """csrc/attn_short.hip lane by lane: layout R, the 32x32x16 MFMA operand / accumulator maps, ds_read_b64_tr_b16."""

    @staticmethod
    def dims(half):                                  # the 16 head dims (or tokens) of a lane, register order 4 j + r
        return [8 * j + 4 * half + r for j in range(4) for r in range(4)]

    @classmethod
    def rows(cls, X):                                # (32, 32) row-major -> per-lane registers [64][16]
        import numpy as np
        return np.array([[X[lane & 31, d] for d in cls.dims(lane >> 5)] for lane in range(64)])

    @classmethod
    def mma(cls, A, B):
        """v_mfma_f32_32x32x16 x 2: lane (m | n = lane & 31, half) supplies contraction slots (t, half, i) = its registers 8 t + i;
        result lane (n, half) register 4 j + r = D[8 j + 4 half + r][n]."""
        import numpy as np
        Dm = np.zeros((32, 32))
        for m in range(32):
            for n in range(32):
                Dm[m, n] = sum(A[m + 32 * half, k] * B[n + 32 * half, k] for half in range(2) for k in range(16))
        return np.array([[Dm[d, lane & 31] for d in cls.dims(lane >> 5)] for lane in range(64)])

    @staticmethod
    def tile_off(row, byte):                         # 16-byte chunk c of row r lives at chunk position c ^ ((r >> 2) & 3)
        return row * 64 + ((((byte >> 4) ^ (row >> 2)) & 3) << 4) + (byte & 15)

    @classmethod
    def tile_image(cls, X, how):
        """the 2-KB tile as {byte address: element}: written by tile_write from layout R, or by the two LDS-DMA instructions of tile_fetch"""
        img = {}
        if how == "write":
            R = cls.rows(X)
            for lane in range(64):
                row, half = lane & 31, lane >> 5
                for j in range(4):
                    for r in range(4):                                           # ds_write_b64 at tile_off(row, 16 j + 8 half)
                        img[cls.tile_off(row, 16 * j + 8 * half) + 2 * r] = R[lane, 4 * j + r]
        else:
            for e in range(2):
                for lane in range(64):                                           # lane's 16 bytes land at e * 1024 + 16 lane
                    row = 16 * e + (lane >> 2)
                    c = (lane ^ (row >> 2)) & 3                                  # the global chunk the lane fetches
                    for k in range(8):
                        img[e * 1024 + 16 * lane + 2 * k] = X[row, 8 * c + k]
        return img

    @classmethod
    def tile_cols(cls, X, how="write"):
        """tile_cols: lane (m = lane & 31, half) gets X[token][m] for its 16 tokens, through the measured ds_read_b64_tr_b16 rule."""
        import numpy as np
        img = cls.tile_image(X, how)
        out = np.zeros((64, 16))
        for lane in range(64):
            g16, i = lane >> 4, lane & 15
            for blk in range(4):                     # reads: (a0, +0), (a1, +0), (a0, +1024), (a1, +1024)
                for j in range(4):                   # element j of the result comes from the address of lane 4 j + (i >> 2) of the group
                    src = 16 * g16 + 4 * j + (i >> 2)
                    t16, grp, half = src & 15, (src >> 4) & 1, src >> 5
                    tok, byte = 4 * half + (t16 >> 2) + (8 if blk & 1 else 0), 32 * grp + 8 * (t16 & 3)
                    addr = cls.tile_off(tok, byte) + (1024 if blk >= 2 else 0)
                    out[lane, 4 * blk + j] = img[addr + 2 * (i & 3)]
        return out

    @classmethod
    def tile_rows(cls, X, how):
        """tile_read: layout R of the lane's row from the tile image"""
        import numpy as np
        img = cls.tile_image(X, how)
        return np.array([[img[cls.tile_off(lane & 31, 16 * j + 8 * (lane >> 5)) + 2 * r] for j in range(4) for r in range(4)] for lane in range(64)])

    @classmethod
    def tile_store_rows(cls, X):
        """tile_write of layout R then tile_store: (row, chunk, eight elements) of every 16-byte store"""
        img = cls.tile_image(X, "write")
        out = {}
        for e in range(2):
            for lane in range(64):
                row = 16 * e + (lane >> 2)
                c = (lane ^ (row >> 2)) & 3
                out[(row, c)] = [img[e * 1024 + 16 * lane + 2 * k] for k in range(8)]
        return out


def test_short_attention_lane_maps():
    """Forward and backward of attn_short.hip in the kernel's own register layouts against plain matrix algebra."""
    import numpy as np
    M = _ShortAttnModel
    rng = np.random.default_rng(0)
    Q, K, V, dO = (rng.standard_normal((32, 32)) for _ in range(4))
    # transposed fragments: lane (m, half) holds X[token][m] for tokens dims(half) -- from a tile written from registers or by the DMA
    for how in ("write", "dma"):
        T = M.tile_cols(V, how)
        for lane in range(64):
            assert np.allclose(T[lane], [V[tok, lane & 31] for tok in M.dims(lane >> 5)])
        assert np.allclose(M.tile_rows(V, how), M.rows(V))                                       # tile_read gives layout R
    for (row, c), vals in M.tile_store_rows(V).items():                                          # staged stores: 16 B = dims 8 c .. 8 c + 7
        assert np.allclose(vals, V[row, 8 * c:8 * c + 8])
    # swizzled row reads: at most two lanes of a 32-lane ds_read_b64 group share a bank pair
    for half in range(2):
        for j in range(4):
            banks = [M.tile_off(row, 16 * j + 8 * half) // 8 % 32 for row in range(32)]
            assert max(banks.count(b) for b in set(banks)) <= 2
    # S^T = K Q^T with queries as columns; O^T = V^T P
    St = M.mma(M.rows(K), M.rows(Q))
    S = Q @ K.T
    for lane in range(64):
        assert np.allclose(St[lane], [S[lane & 31, key] for key in M.dims(lane >> 5)])
    P = np.exp(S - S.max(1, keepdims=True)); P /= P.sum(1, keepdims=True)
    Pt = np.array([[P[lane & 31, key] for key in M.dims(lane >> 5)] for lane in range(64)])      # what the lanes hold after the softmax
    Ot = M.mma(M.tile_cols(V), Pt)
    O = P @ V
    for lane in range(64):
        assert np.allclose(Ot[lane], [O[lane & 31, d] for d in M.dims(lane >> 5)])               # layout R of the query row: store_row
    # backward, queries as columns: dP^T = V dO^T, dQ = dS K
    dPt = M.mma(M.rows(V), M.rows(dO))
    dP = dO @ V.T
    for lane in range(64):
        assert np.allclose(dPt[lane], [dP[lane & 31, key] for key in M.dims(lane >> 5)])
    dS = P * (dP - (P * dP).sum(1, keepdims=True))
    dSt = np.array([[dS[lane & 31, key] for key in M.dims(lane >> 5)] for lane in range(64)])
    dQt = M.mma(M.tile_cols(K), dSt)
    for lane in range(64):
        assert np.allclose(dQt[lane], [(dS @ K)[lane & 31, d] for d in M.dims(lane >> 5)])
    # keys as columns: S = Q K^T (rows = queries), dK = dS^T Q, dV = P^T dO
    S2 = M.mma(M.rows(Q), M.rows(K))
    for lane in range(64):
        assert np.allclose(S2[lane], [S[q, lane & 31] for q in M.dims(lane >> 5)])
    dS2 = np.array([[dS[q, lane & 31] for q in M.dims(lane >> 5)] for lane in range(64)])
    P2 = np.array([[P[q, lane & 31] for q in M.dims(lane >> 5)] for lane in range(64)])
    dKt, dVt = M.mma(M.tile_cols(Q), dS2), M.mma(M.tile_cols(dO), P2)
    for lane in range(64):
        assert np.allclose(dKt[lane], [(dS.T @ Q)[lane & 31, d] for d in M.dims(lane >> 5)])
        assert np.allclose(dVt[lane], [(P.T @ dO)[lane & 31, d] for d in M.dims(lane >> 5)])


# ---------------------------------------------------------------------------------------------------------------- attn2_bwd1.hip (round 4)
def _bwd1_plan(nkb, nw=8):
    """plan1() of csrc/attn2_bwd1.hip: positions per wave = the smallest P >= max(ceil(nkb^2 / 8), nkb) with P d != 0 (mod nkb) for d = 1..7"""
    nt = nkb * nkb
    p = max((nt + nw - 1) // nw, nkb)
    while any((p * d) % nkb == 0 for d in range(1, nw)):
        p += 1
    return nt, p


def test_bwd1_tile_schedule_properties():
    """The one-pass attention backward walks the nkb x nkb score tiles row-major (key block, query tile); wave w owns positions [P w, P w + P).
    For every supported nkb (L = 256 .. 576): every tile exactly once; in every step the eight waves are on eight DIFFERENT query tiles (so the
    read-modify-write of dQ^T[t] never collides); a key block is shared by at most two waves; and the tabulated number of earlier updates of a
    tile (etab, bwd1_body) equals a brute-force count -- the tile counters then admit the updates of a tile in step order."""
    for nkb in range(8, 19):
        nt, p = _bwd1_plan(nkb)
        pos = lambda w, s: p * w + s
        seen = {}
        for w in range(8):
            for s in range(p):
                g = pos(w, s)
                if g < nt:
                    assert g not in seen
                    seen[g] = (w, s)
        assert sorted(seen) == list(range(nt)), nkb
        for s in range(p):
            tiles = [pos(w, s) % nkb for w in range(8) if pos(w, s) < nt]
            assert len(set(tiles)) == len(tiles), (nkb, s, tiles)
        for kb in range(nkb):
            owners = {seen[kb * nkb + t][0] for t in range(nkb)}
            assert len(owners) <= 2 and (len(owners) == 1 or max(owners) - min(owners) == 1), (nkb, kb, owners)
        # etab[w][s]: closed form of the kernel vs brute force
        for w in range(8):
            for s in range(p):
                if pos(w, s) >= nt:
                    continue
                tt = pos(w, s) % nkb
                brute = sum(1 for w8 in range(8) for s8 in range(s) if pos(w8, s8) < nt and pos(w8, s8) % nkb == tt)
                n = 0
                for w8 in range(8):
                    ln = min(max(nt - p * w8, -10 ** 9), p)
                    lim = min(s, ln)
                    f = (tt - p * w8) % nkb
                    if f < lim:
                        n += (lim - 1 - f) // nkb + 1
                assert n == brute, (nkb, w, s, n, brute)


def test_bwd1_fixed_point_rounding_and_class_counts():
    """The table gradient is scattered as integers: round(x) = the low mantissa bits of float32(x + 1.5 * 2^23) for |x| < 2^22 (round to nearest
    even, like the FMA that produces it), 576 addends of magnitude < 2^21 stay below 2^31, and the number of (query, key) pairs of an offset class
    -- subtracted as count x bits(1.5 * 2^23) when the table is flushed -- is (gh - |dy|)(gw - |dx|)."""
    import numpy as np
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-2 ** 21, 2 ** 21, 200000), rng.uniform(-4, 4, 50000), np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.0, 2 ** 21 - 1.0])])
    x = x.astype(np.float32)
    bits = (x + np.float32(12582912.0)).astype(np.float32).view(np.uint32)
    got = (bits - np.uint32(0x4B400000)).view(np.int32)
    want = np.rint(x.astype(np.float64)).astype(np.int64)          # numpy rounds half to even
    assert np.array_equal(got.astype(np.int64), want)
    assert 576 * 2 ** 21 * 1.05 < 2 ** 31
    gh, gw = 24, 24
    counts = {}
    for q in range(gh * gw):
        for k in range(gh * gw):
            c = ((q // gw - k // gw + gh - 1), (q % gw - k % gw + gw - 1))
            counts[c] = counts.get(c, 0) + 1
    for (dyi, dxi), n in counts.items():
        assert n == (gh - abs(dyi - (gh - 1))) * (gw - abs(dxi - (gw - 1)))
    assert len(counts) == (2 * gh - 1) * (2 * gw - 1)
    # modular arithmetic of the flush: u32 sum of (bits) minus count * MAGIC_BITS = the signed integer sum, whatever the wrap-arounds
    vals = rng.integers(-2 ** 21, 2 ** 21, 576)
    acc = np.uint32(0)
    with np.errstate(over="ignore"):
        for v in vals:
            acc = np.uint32(acc + (np.float32(v) + np.float32(12582912.0)).view(np.uint32))
        back = np.uint32(acc - np.uint32(576) * np.uint32(0x4B400000)).view(np.int32)
    assert int(back) == int(vals.sum())


def test_bwd1_step_table_fits_its_register_and_its_lds_alias():
    """Round 4, second form: a wave keeps its row of the step table in ONE register (lane = step: P <= 64) and the table is built where the L
    per-query floats log2 K - lse2 live later (8 P <= L); plan1() declines a shape that breaks either."""
    for nkb in range(8, 21):                                     # L = 256 .. 640 (4 L <= NPIECE * 512)
        _, p = _bwd1_plan(nkb)
        assert p <= 64 and 8 * p <= 32 * nkb, (nkb, p)


def test_bwd1_three_term_bf16_split_carries_an_f32():
    """The per-query terms (-delta_q, log2 K - lse2_q) enter the score / dP tiles as a matrix product: A row = the f32 value split into three bf16
    terms (hi = bf16(v), lo = bf16(v - hi), lolo = bf16(v - hi - lo)), B = ones.  The f32 sum of the three products must return the value to
    f32 rounding (24 mantissa bits in three 8-bit pieces), for magnitudes from 1e-28 to 1e+30 (below ~1e-33 the third term is subnormal and flushed:
    2^-20 relative there, on values that are zero for every purpose of the kernel) and for exact integers (log2 K)."""
    import numpy as np

    def bf16(x):                                                  # round-to-nearest-even on the upper 16 bits, as v_cvt_pk_bf16_f32
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(1)
    v = np.concatenate([(1.0 + np.abs(rng.standard_normal(100000))) * rng.choice([-1.0, 1.0], 100000) * 10.0 ** rng.uniform(-28, 30, 100000), np.arange(-100, 101, dtype=np.float64),
                        -rng.uniform(0, 40, 50000)]).astype(np.float32)
    hi = bf16(v)
    r1 = (v - hi).astype(np.float32)
    lo = bf16(r1)
    r2 = (r1 - lo).astype(np.float32)
    ll = bf16(r2)
    back = ((hi.astype(np.float32) + lo) + ll).astype(np.float32)     # the MFMA adds the three products in f32
    err = np.abs(back.astype(np.float64) - v.astype(np.float64))
    assert np.all(err <= np.abs(v.astype(np.float64)) * 2.0 ** -23), float((err / np.maximum(np.abs(v), 1e-38)).max())
    ints = np.arange(-100, 101, dtype=np.float32)
    assert np.array_equal(bf16(ints), ints)                       # log2 K is an integer in [-100, 100]: ONE term already exact


def test_bwd1_touch_chunks_cover_every_line_of_the_next_item():
    """bwd1_unprep_q touches the next item's operands one dword per 128-byte line: wave-uniform chunks of 64 lines, chunk j * 8 + wave (j < NTOUCH = 5)
    walking Q~ | V | K^ (L / 2 lines each) | dout | o (L rows, one line per row) | lse2 (L / 32 lines), every segment padded to whole chunks and
    lines past a segment clamped to its last one.  Every line of every segment must be touched, for every supported L."""
    NTOUCH, NW = 5, 8
    for L in range(256, 641, 32):
        hl, cq, cr = L // 2, (L // 2 + 63) >> 6, (L + 63) >> 6
        assert 3 * cq + 2 * cr + 1 <= NTOUCH * NW, L
        seen = {sg: set() for sg in range(6)}
        for j in range(NTOUCH):
            for wave in range(NW):
                r, sg = j * NW + wave, 0
                if r >= cq:
                    r -= cq; sg = 1
                if sg == 1 and r >= cq:
                    r -= cq; sg = 2
                if sg == 2 and r >= cq:
                    r -= cq; sg = 3
                if sg == 3 and r >= cr:
                    r -= cr; sg = 4
                if sg == 4 and r >= cr:
                    r -= cr; sg = 5
                nl = hl if sg < 3 else (L if sg < 5 else L // 32)
                for lane in range(64):
                    seen[sg].add(min(r * 64 + lane, nl - 1))
        for sg in range(6):
            nl = hl if sg < 3 else (L if sg < 5 else L // 32)
            assert seen[sg] == set(range(nl)), (L, sg)
