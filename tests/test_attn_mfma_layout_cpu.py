"""CPU tier: the lane algebra of the matrix-core decode-attention walk (tiny-llm_amd/csrc/attn_mfma.h), restated in numpy.

The kernel's claim is that NOTHING crosses lanes between its two matrix products: the transposed score tile that
`v_mfma_f32_16x16x32_bf16` leaves in a lane is already the B operand of the second product, provided V's rows are requested in the
order "tokens 4 g .. 4 g + 3 of the first 16-token tile, then of the second".  This file simulates a 64-lane wave with the operand /
result layouts of that instruction (A[m][k]: lane m + 16 (k / 8), element k % 8; B[k][n]: lane n + 16 (k / 8), element k % 8;
D[m][n]: lane n + 16 (m / 4), element m % 4 -- the layouts csrc/qmv3.h and csrc/qmm6.h are built on and the GPU tests pin) and walks
32 tokens exactly as the kernel's index expressions do: coalesced row chunks, the K image in LDS, the Q operand, the packed weights,
the `v_perm_b32` pick of V's A operand, the dims a lane's accumulators stand for.  The result must be plain attention.  Executable
documentation of the layout, not a substitute for the device parity tests (tests/test_decode_kernels_gpu.py, long contexts)."""

import numpy as np

D, WT, RQ = 128, 32, 4  # head size, tokens of a wave per stage, query heads of the GQA group


def mfma_16x16x32(a_regs, b_regs, c_regs):
    """a_regs, b_regs: [64 lanes][8 elements]; c_regs: [64][4].  D = A (16 x 32) . B (32 x 16) + C in the instruction's layouts."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for lane in range(64):
        r, c = lane % 16, lane // 16
        A[r, 8 * c:8 * c + 8] = a_regs[lane]
        B[8 * c:8 * c + 8, r] = b_regs[lane]
    Dm = A @ B
    out = c_regs.copy()
    for lane in range(64):
        n, g = lane % 16, lane // 16
        for i in range(4):
            out[lane, i] += Dm[4 * g + i, n]
    return out


def row_of(g, e):
    """Row (of the wave's 32) that lane group g requests with its e-th 16-byte load: the order of the second product's reduction index."""
    return 4 * g + e if e < 4 else 16 + 4 * g + (e - 4)


def test_rows_are_requested_once_each_and_coalesced():
    seen = sorted(row_of(g, e) for g in range(4) for e in range(8))
    assert seen == list(range(WT))
    # one load instruction (fixed e): the 16 lanes of a group read ONE row's 256 bytes (c = 0 .. 15 -> dims 8 c .. 8 c + 7)
    for e in range(8):
        for g in range(4):
            cols = sorted(d for c in range(16) for d in range(8 * c, 8 * c + 8))
            assert cols == list(range(D)) and 0 <= row_of(g, e) < WT


def walk_32_tokens(K, V, Q, valid):
    """One stage of one wave as the kernel's index expressions lay it out.  K, V [32][128], Q [4][128], valid [32] -> (m, l, O) per head,
    O [4][128], with the softmax weights exp2(score - max) unnormalised, as the kernel keeps them."""
    lanes = range(64)
    # coalesced requests: lane (c, g) holds dims 8 c .. 8 c + 7 of rows row_of(g, 0 .. 7)
    kreg = np.zeros((64, 8, 8))
    vreg = np.zeros((64, 8, 8))
    for lane in lanes:
        c, g = lane % 16, lane // 16
        for e in range(8):
            kreg[lane, e] = K[row_of(g, e), 8 * c:8 * c + 8]
            vreg[lane, e] = V[row_of(g, e), 8 * c:8 * c + 8]
    # K image in LDS: row-major [32][128] (the 16 bytes of padding per row only move banks)
    image = np.zeros((WT, D))
    for lane in lanes:
        c, g = lane % 16, lane // 16
        for e in range(8):
            image[row_of(g, e), 8 * c:8 * c + 8] = kreg[lane, e]
    np.testing.assert_array_equal(image, K)
    # first product: S^T tile ab = K rows (A: lane = token, dims 32 g + 8 j ..) . Q^T (B: lane = head, the same dims; heads >= 4 zero)
    s = np.zeros((2, 64, 4))
    for ab in range(2):
        for j in range(4):
            a = np.zeros((64, 8))
            b = np.zeros((64, 8))
            for lane in lanes:
                c, g = lane % 16, lane // 16
                a[lane] = image[16 * ab + c, 32 * g + 8 * j:32 * g + 8 * j + 8]
                if c < RQ:
                    b[lane] = Q[c, 32 * g + 8 * j:32 * g + 8 * j + 8]
            s[ab] = mfma_16x16x32(a, b, s[ab])
    # a lane (head c, g) now holds tokens 16 ab + 4 g + i: per-lane softmax statistics, the maximum closed over the four lane groups
    sv = np.full((64, 8), -3e38)
    for lane in lanes:
        c, g = lane % 16, lane // 16
        for ab in range(2):
            for i in range(4):
                tok = 16 * ab + 4 * g + i
                if c < RQ:
                    np.testing.assert_allclose(s[ab, lane, i], K[tok] @ Q[c], rtol=1e-12, atol=1e-9)
                if valid[tok]:
                    sv[lane, 4 * ab + i] = s[ab, lane, i]
    tm = sv.max(axis=1)
    for bit in (16, 32):  # v_permlane16_swap / v_permlane32_swap: the lane whose index differs in that bit
        tm = np.maximum(tm, tm[[lane ^ bit for lane in lanes]])
    m = np.maximum(-1e30, tm)
    pw = np.exp2(sv - m[:, None])  # masked scores sit far below the initial maximum: weight 0 without a second select
    l_lane = pw.sum(axis=1)
    # second product: O^T tile j = V^T (A: lane (c, g) picks element j of its eight chunks) . P^T (B: the lane's eight weights)
    o = np.zeros((8, 64, 4))
    for j in range(8):
        a = np.zeros((64, 8))
        for lane in lanes:
            for e in range(8):
                a[lane, e] = vreg[lane, e, j]  # v_perm_b32 packs elements j of chunks (2 q, 2 q + 1) into word q
        o[j] = mfma_16x16x32(a, pw, o[j])
    # a lane (head c, g') holds, of tile j, rows m = 4 g' + i <-> dims 8 m + j = 32 g' + 8 i + j
    O = np.zeros((RQ, D))
    l = np.zeros(RQ)
    for lane in lanes:
        c, g = lane % 16, lane // 16
        if c < RQ:
            l[c] += l_lane[lane]  # closed over the lane groups once, after the walk
            for j in range(8):
                for i in range(4):
                    O[c, 32 * g + 8 * i + j] = o[j, lane, i]
    return m[:RQ], l, O


def test_a_stage_of_the_walk_is_plain_attention():
    rng = np.random.default_rng(7)
    for n_valid in (32, 19, 1, 0):
        K = rng.standard_normal((WT, D))
        V = rng.standard_normal((WT, D))
        Q = rng.standard_normal((RQ, D)) / np.sqrt(D)
        valid = np.arange(WT) < n_valid
        m, l, O = walk_32_tokens(K, V, Q, valid)
        for h in range(RQ):
            scores = K @ Q[h]
            if n_valid == 0:
                assert m[h] == -1e30 and l[h] == 0.0 and not O[h].any()
                continue
            want_m = scores[valid].max()
            w = np.where(valid, np.exp2(scores - want_m), 0.0)
            np.testing.assert_allclose(m[h], want_m, rtol=1e-12)
            np.testing.assert_allclose(l[h], w.sum(), rtol=1e-12)
            np.testing.assert_allclose(O[h], w @ V, rtol=1e-10, atol=1e-12)


def test_the_k_image_reads_are_free_of_bank_conflicts():
    """16-byte reads of 16 rows at one column: with 272-byte rows the 16 lanes of a group cover all 64 banks (bank = byte / 4 mod 64);
    with plain 256-byte rows they would all fall on the same four."""
    for stride, distinct in ((272, 64), (256, 4)):
        for g in range(4):
            for j in range(4):
                banks = set()
                for c in range(16):
                    byte = c * stride + 64 * g + 16 * j
                    banks.update(((byte // 4) + k) % 64 for k in range(4))
                assert len(banks) == distinct


def _bf16_round(x):
    """Round-to-nearest-even to bf16, returned as float32 (the v_cvt_pk_bf16_f32 the kernel packs its weights with)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def test_weights_as_bf16_hi_plus_lo_keep_sixteen_mantissa_bits():
    """P goes through the second product as two bf16 operands: hi = bf16(p), lo = bf16(p - hi).  hi + lo is within 2^-16 of p
    (relative), where one bf16 operand alone is within 2^-8: the weighted sum of V then differs from the fp32 walk by less than the
    output's own bf16 rounding, which is why the 1-ulp bar of the long-context parity cases held."""
    rng = np.random.default_rng(11)
    p = np.exp2(-rng.uniform(0.0, 24.0, size=100_000)).astype(np.float32)
    hi = _bf16_round(p)
    lo = _bf16_round(p - hi)
    rel_hi = np.abs(hi.astype(np.float64) - p) / p
    rel_both = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - p) / p
    assert rel_hi.max() <= 2.0 ** -8 and rel_hi.max() > 2.0 ** -10
    assert rel_both.max() <= 2.0 ** -16
