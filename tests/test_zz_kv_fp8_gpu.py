"""GPU tier: FP8 (OCP E4M3) KV pages -- SURVEY section 8 row f4, second half ("quantized-KV").

The reference has no quantised cache (README.md:134-135), so there is no reference output to match; what is held here:
  * the device codec (csrc/kv8.h: v_cvt_pk_fp8_f32, power-of-two row scales) equals `oracle/kv_fp8.py` BIT FOR BIT -- codes and
    scales -- and that oracle is pinned to torch.float8_e4m3fn (tests/test_kv_fp8_oracle_cpu.py);
  * every attention kernel over FP8 pages equals THE SAME kernel over the dequantised bf16 pages bit for bit (the scales are powers of
    two and commute with every rounding): the operator's decode and FlashAttention kernels, the engine's VALU walk in its three page
    modes and its matrix-core walk, up to 32,768 tokens -- so everything the bf16 kernels are held to (tests/test_ops_gpu.py,
    tests/test_decode_kernels_gpu.py) carries over;
  * the fused engine with kv_format="fp8" against `OracleQwen3(kv_format="fp8")` and the float64 truth, both replay routes bit-identical,
    forks (copy-on-write of a quantised tail page), batched steps, chunked prefill.
Reference ops whose quantised twins these are: paged_cache_update / paged_attention (paged_attention.cpp:14-70, 77-225; kernels
paged_attention.metal:82-506), op order qwen3_week3.py:63-86."""
import os

import numpy as np
import pytest
import torch

from helpers import TINY_CFG, assert_bf16_close, check_against_truth, to_mlx_shaped
from oracle import kv_fp8
from oracle import tiny_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
HQ, HKV, D, PAGE = 32, 8, 128, 128
EPS, THETA = 1e-6, 1e6


@pytest.fixture(scope="module")
def ext():
    import tiny_llm_ext_hip

    tiny_llm_ext_hip.load_library(".")
    return tiny_llm_ext_hip


def _t(a, dtype=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def _host(t):
    return t.float().cpu().numpy()


def _rows(rng, n, sigma):
    return O.bf16(rng.standard_normal((n, D), dtype=np.float32) * sigma)


# ---- codec ---------------------------------------------------------------------------------------------------------------------------
def test_device_codec_equals_the_oracle_bit_for_bit(ext):
    rng = np.random.default_rng(0)
    finite = kv_fp8.decode_e4m3(np.arange(256, dtype=np.uint8))
    finite = finite[~np.isnan(finite)]
    blocks = [_rows(rng, 300, s) for s in (1e-4, 0.02, 1.0, 40.0, 3e4)]
    # rows on the E4M3 grid and on its ties (x a power of two), an all-zero row, a row with one huge value, bf16 extremes
    grid = np.resize(finite, (40, D)).astype(np.float32)
    ties = O.bf16(np.resize((np.sort(finite)[:-1] + np.sort(finite)[1:]) / 2, (40, D)).astype(np.float32))
    special = np.zeros((4, D), np.float32)
    special[1, 5] = 3.0e38
    special[2, :] = 2.0 ** -120
    special[3, ::2] = -447.0
    x = np.concatenate(blocks + [grid, grid * 2.0 ** -7, ties, special])
    codes, scales = ext.kv_fp8_quantize_rows(_t(x))
    want_c, want_s = kv_fp8.quantize_rows(x)
    np.testing.assert_array_equal(scales.cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    np.testing.assert_array_equal(codes.cpu().numpy(), want_c)
    back = _host(ext.kv_fp8_dequantize_rows(codes, scales))
    np.testing.assert_array_equal(back, kv_fp8.dequantize_rows(want_c, want_s))


def test_paged_cache_update_fp8_equals_the_oracle(ext):
    rng = np.random.default_rng(1)
    P, H, page = 7, 8, 32
    pages = torch.zeros((P, H, page, D), dtype=torch.uint8, device=DEV)
    scales = torch.zeros((P, H, page), dtype=torch.float32, device=DEV)
    want_p, want_s = np.zeros((P, H, page, D), np.uint8), np.zeros((P, H, page), np.float32)
    for page_id, start, n in [(3, 0, 32), (5, 7, 1), (0, 30, 2), (5, 8, 17)]:
        vals = O.bf16(rng.standard_normal((1, H, n, D), dtype=np.float32) * rng.choice([0.01, 1.0, 9.0]))
        out = ext.paged_cache_update_fp8(pages, scales, _t(vals), page_id, start)
        assert out[0] is pages and out[1] is scales  # in place, like the reference's paged_cache_update
        kv_fp8.paged_cache_update(want_p, want_s, vals, page_id, start)
    np.testing.assert_array_equal(pages.cpu().numpy(), want_p)
    np.testing.assert_array_equal(scales.cpu().numpy(), want_s)
    with pytest.raises(RuntimeError, match="outside page storage"):
        ext.paged_cache_update_fp8(pages, scales, _t(np.zeros((1, H, 5, D), np.float32)), 2, 30)


# ---- operator: paged_attention over FP8 pages == over the dequantised bf16 pages ------------------------------------------------------
def _paged_case(rng, ctxs, page, hq=HQ, hkv=HKV, sigma=1.0):
    B = len(ctxs)
    need = [(c + page - 1) // page for c in ctxs]
    P = sum(need) + 3
    ids = list(rng.permutation(P))
    table = -np.ones((B, max(max(need), 1) + 1), dtype=np.int32)
    for b in range(B):
        for j in range(need[b]):
            table[b, j] = ids.pop()
    k = O.bf16(rng.standard_normal((P, hkv, page, D), dtype=np.float32) * sigma)
    v = O.bf16(rng.standard_normal((P, hkv, page, D), dtype=np.float32) * sigma)
    kc, ks = kv_fp8.quantize_rows(k)
    vc, vs = kv_fp8.quantize_rows(v)
    return table, (kc, ks, vc, vs)


@pytest.mark.parametrize("L,ctxs,page", [(1, [1], 16), (1, [700, 33], 128), (3, [129, 5], 64), (8, [2000], 128), (9, [9], 16), (65, [200, 65], 32),
                                         (200, [1500], 128), (512, [8192], 128), (40, [300, 40, 3000], 16)])
def test_paged_attention_fp8_equals_the_bf16_kernels_on_dequantised_pages(ext, L, ctxs, page):
    rng = np.random.default_rng(L * 1000 + sum(ctxs))
    table, (kc, ks, vc, vs) = _paged_case(rng, ctxs, page)
    B = len(ctxs)
    q = O.bf16(rng.standard_normal((B * HQ, L, D), dtype=np.float32))
    args = (_t(table, torch.int32), _t(np.asarray(ctxs), torch.int32), D ** -0.5, True)
    kw = dict(num_kv_heads=HKV, num_heads=HQ, max_context_hint=max(ctxs))
    got = ext.paged_attention_fp8(_t(q), _t(kc, torch.uint8), _t(ks, torch.float32), _t(vc, torch.uint8), _t(vs, torch.float32), *args, **kw)
    kd, vd = kv_fp8.dequantize_rows(kc, ks), kv_fp8.dequantize_rows(vc, vs)
    twin = ext.paged_attention(_t(q), _t(kd), _t(vd), *args, **kw)
    np.testing.assert_array_equal(_host(got), _host(twin))
    # ... and a few rows against the oracle itself (the bf16 kernels' own bar)
    for b, r in {(0, 0), (B - 1, L - 1), (B // 2, L // 2)}:
        vis = max(ctxs[b] - L + r + 1, 0)
        want = O.paged_attention(q[b * HQ:(b + 1) * HQ, r:r + 1], kd, vd, table[b:b + 1], np.asarray([vis], np.int32), D ** -0.5, True, HKV, HQ,
                                 "bf16", round_p=L > 8)
        # (FlashAttention rounds P to bf16 before P.V like the reference, paged_attention.metal:439-444: 2^-9 of the largest |v| on top)
        floor = 1.5e-3 + (2.0 ** -9 * float(np.max(np.abs(vd))) if L > 8 else 0.0)
        assert_bf16_close(_host(got)[b * HQ:(b + 1) * HQ, r], want[:, 0], ulps=1.0, abs_floor=floor, what=f"L={L} ctx={ctxs[b]} row {r}")


def test_paged_attention_fp8_refuses_what_it_cannot_take(ext):
    z8 = torch.zeros((2, 2, 16, 64), dtype=torch.uint8, device=DEV)
    zs = torch.zeros((2, 2, 16), dtype=torch.float32, device=DEV)
    q = torch.zeros((4, 1, 64), dtype=torch.bfloat16, device=DEV)
    bt, cl = torch.zeros((1, 2), dtype=torch.int32, device=DEV), torch.ones((1,), dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="head dimension 128"):
        ext.paged_attention_fp8(q, z8, zs, z8, zs, bt, cl, 1.0, True, num_kv_heads=2, num_heads=4)
    with pytest.raises(RuntimeError, match="uint8 pages"):
        ext.paged_attention_fp8(q, z8.to(torch.bfloat16), zs, z8, zs, bt, cl, 1.0, True, num_kv_heads=2, num_heads=4)


# ---- engine kernels: the decode attention launch over FP8 pages ------------------------------------------------------------------------
def _decode_case(rng, ctxs, k_norm_zero):
    """ctx = tokens already cached.  With k_norm_zero the appended K row is all zeros and the new V row lies on the E4M3 grid: the
    quantisation of the token being decoded is then the identity and the FP8 launch must equal the bf16 launch bit for bit."""
    B = len(ctxs)
    table, (kc, ks, vc, vs) = _paged_case(rng, [c + 1 for c in ctxs], PAGE)
    rows = O.bf16(rng.standard_normal((B, HQ + 2 * HKV, D), dtype=np.float32))
    kn = O.bf16(1.0 + 0.1 * rng.standard_normal((D,), dtype=np.float32))
    if k_norm_zero:
        kn = np.zeros((D,), np.float32)
        rows[:, HQ + HKV:] = kv_fp8.round_trip(rows[:, HQ + HKV:])
    qn = O.bf16(1.0 + 0.1 * rng.standard_normal((D,), dtype=np.float32))
    return table, np.asarray(ctxs, np.int32), rows.reshape(B, -1), qn, kn, (kc, ks, vc, vs)


def _run_decode_fp8(ext, case, max_context):
    table, ctx, qkv, qn, kn, (kc, ks, vc, vs) = case
    pools = [_t(kc, torch.uint8), _t(ks, torch.float32), _t(vc, torch.uint8), _t(vs, torch.float32)]
    out, info = ext.decode_attention_fused_fp8(_t(qkv), _t(qn), _t(kn), *pools, _t(table, torch.int32), _t(ctx, torch.int32), num_heads=HQ,
                                               num_kv_heads=HKV, rope_theta=THETA, eps=EPS, max_context=max_context)
    torch.cuda.synchronize()
    return _host(out), [p.cpu().numpy() for p in pools], info


def _run_decode_bf16(ext, case, max_context):
    table, ctx, qkv, qn, kn, (kc, ks, vc, vs) = case
    kd, vd = _t(kv_fp8.dequantize_rows(kc, ks)), _t(kv_fp8.dequantize_rows(vc, vs))
    out, info = ext.decode_attention_fused(_t(qkv), _t(qn), _t(kn), kd, vd, _t(table, torch.int32), _t(ctx, torch.int32), num_heads=HQ,
                                           num_kv_heads=HKV, rope_theta=THETA, eps=EPS, max_context=max_context)
    torch.cuda.synchronize()
    return _host(out), _host(kd), _host(vd), info


# (head counts / windows: tl_decode_attention_plan; one sequence up to 4,096 tokens walks on the VALU, longer ones and groups of sequences
# beyond 128-token windows on the matrix cores)
PLANS = [([0], {}), ([63], {}), ([64], {}), ([200], {}), ([300], {}), ([1000], {}), ([4095], {}), ([8191], {}), ([32767], {}),
         ([8192, 5000, 129], {}), ([300, 17, 2000], {"TL_ATTN_MFMA": "0"}), ([8191], {"TL_ATTN_MFMA": "0"}), ([8191], {"TL_ATTN_RQ": "1"}),
         ([700] * 12, {}), ([150] * 40, {}), ([32768, 1, 700, 20000], {"TL_ATTN_MAX_SPLITS": "256"})]


@pytest.mark.parametrize("ctxs,env", PLANS)
def test_decode_walks_over_fp8_pages_equal_the_bf16_walks_bit_for_bit(ext, ctxs, env, monkeypatch):
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MFMA", "TL_ATTN_MIN_TOKENS"):
        monkeypatch.delenv(name, raising=False)
    for name, value in env.items():
        monkeypatch.setenv(name, value)
    rng = np.random.default_rng(sum(ctxs) + len(ctxs))
    case = _decode_case(rng, ctxs, k_norm_zero=True)
    got, pools, info = _run_decode_fp8(ext, case, max(ctxs))
    twin, kd, vd, info2 = _run_decode_bf16(ext, case, max(ctxs))
    assert info == info2
    on_matrix_cores = env.get("TL_ATTN_MFMA") != "0" and info["heads_per_workgroup"] == 4 and info["tokens_per_split"] >= 128
    if on_matrix_cores:  # csrc/attn_mfma.h: the same instruction stream over the same values
        np.testing.assert_array_equal(got, twin, err_msg=f"ctxs={ctxs} {info}")
    else:
        # the VALU walk (attn_decode_fused_kernel): hipcc contracts `acc * f + p * v` into one of its two fused forms per INSTANTIATION,
        # so the two kernels may differ in the last fp32 bit of a value sum: equal but for a rare one-step bf16 flip
        diff = got != twin
        assert diff.mean() <= 2e-3, f"ctxs={ctxs} {info}: {int(diff.sum())} of {diff.size} elements differ"
        assert_bf16_close(got, twin, ulps=1.0, abs_floor=0.0, what=f"ctxs={ctxs} {info}")
    # the appended rows: what the bf16 launch appended, quantised
    np.testing.assert_array_equal(kv_fp8.dequantize_rows(pools[0], pools[1]), kd)
    np.testing.assert_array_equal(kv_fp8.dequantize_rows(pools[2], pools[3]), vd)


@pytest.mark.parametrize("ctxs,env", [([0], {}), ([100], {}), ([700], {}), ([5000, 129, 8192], {}), ([2000, 30], {"TL_ATTN_MFMA": "0"}), ([150] * 20, {})])
def test_decode_attention_fp8_quantises_the_token_being_decoded(ext, ctxs, env, monkeypatch):
    """General case: the new K row (q/k-norm + RoPE) and V row are quantised in registers -- the page receives the oracle's codes and
    scale for the row the bf16 launch would have appended, and the step attends to the dequantised row (oracle: paged attention over the
    dequantised pages with the round trip of the new rows in place)."""
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MFMA", "TL_ATTN_MIN_TOKENS"):
        monkeypatch.delenv(name, raising=False)
    for name, value in env.items():
        monkeypatch.setenv(name, value)
    rng = np.random.default_rng(7 + sum(ctxs))
    case = _decode_case(rng, ctxs, k_norm_zero=False)
    table, ctx, qkv, qn, kn, (kc, ks, vc, vs) = case
    got, pools, info = _run_decode_fp8(ext, case, max(ctxs))
    _, kd_twin, vd_twin, _ = _run_decode_bf16(ext, case, max(ctxs))  # its pages now hold the UNQUANTISED appended rows
    B = len(ctxs)
    kd, vd = kv_fp8.dequantize_rows(kc, ks), kv_fp8.dequantize_rows(vc, vs)
    for b in range(B):
        pid, slot = int(table[b, ctx[b] // PAGE]), int(ctx[b] % PAGE)
        for codes, scales, twin, name in ((pools[0], pools[1], kd_twin, "K"), (pools[2], pools[3], vd_twin, "V")):
            want_c, want_s = kv_fp8.quantize_rows(twin[pid, :, slot])
            np.testing.assert_array_equal(scales[pid, :, slot].view(np.uint32), want_s.view(np.uint32), err_msg=f"{name} scale, sequence {b}")
            np.testing.assert_array_equal(codes[pid, :, slot], want_c, err_msg=f"{name} codes, sequence {b}")
        kd[pid, :, slot] = kv_fp8.round_trip(kd_twin[pid, :, slot])
        vd[pid, :, slot] = kv_fp8.round_trip(vd_twin[pid, :, slot])
    # nothing else in the pools moved
    untouched = np.ones(kc.shape[:3], bool)
    for b in range(B):
        untouched[int(table[b, ctx[b] // PAGE]), :, int(ctx[b] % PAGE)] = False
    np.testing.assert_array_equal(pools[0][untouched], kc[untouched])
    np.testing.assert_array_equal(pools[3][untouched], vs[untouched])
    rows = qkv.reshape(B, HQ + 2 * HKV, D)
    q = O.rope(O.rms_norm_fast(rows[:, :HQ], qn, EPS)[:, None], ctx, D, THETA, False, "bf16")
    want = O.paged_attention(q.transpose(0, 2, 1, 3).reshape(B * HQ, 1, D), kd, vd, table, ctx + 1, D ** -0.5, True, HKV, HQ)
    assert_bf16_close(got, want.reshape(B, HQ * D), ulps=1.0, abs_floor=1.5e-3, what=f"ctxs={ctxs} {info}")


# ---- the fused engine with FP8 pages -----------------------------------------------------------------------------------------------------
def _engine_run(model, kv_format, prompts, steps, route="aql", page=16, chunk=2048, max_prefill_rows=256):
    from tiny_llm_hip.engine import DecodeEngine

    old = os.environ.pop("TL_AQL", None)
    if route == "hipgraph":
        os.environ["TL_AQL"] = "0"
    try:
        n = len(prompts)
        pages = sum((len(p) + steps + 2 * page) // page + 1 for p in prompts) + 2
        eng = DecodeEngine(model, page_size=page, num_pages=pages, max_batch=n, max_prefill_rows=max_prefill_rows, kv_format=kv_format)
        try:
            rows = []
            for i, p in enumerate(prompts):
                eng.begin(i)
                eng.prefill(i, p, chunk=chunk)
            if n == 1:
                rows.append(eng.logits(1).float().cpu().numpy()[0])
            for _ in range(steps):
                eng.decode(1, batch=n)
                rows.append(eng.logits(n).float().cpu().numpy().copy())
            ids = [eng.read_tokens(i, steps + 1) for i in range(n)]  # the prefill's first token, then one per step
            route_used = eng.replay_route()
            kv_bytes = eng.stats()["kv_bytes"]
            for i in range(n):
                eng.release(i)
            assert eng.stats()["pages_in_use"] == 0
            return ids, rows, route_used, kv_bytes
        finally:
            eng.close()
    finally:
        os.environ.pop("TL_AQL", None)
        if old is not None:
            os.environ["TL_AQL"] = old


@pytest.fixture(scope="module")
def tiny():
    w = O.make_qwen3_weights(TINY_CFG, seed=11, sigma=0.05)
    return w, to_mlx_shaped(TINY_CFG, w)


@pytest.mark.parametrize("prompt_len,steps,chunk,page", [(5, 6, 2048, 16), (40, 8, 16, 16), (150, 5, 64, 64), (300, 4, 128, 128)])
def test_engine_with_fp8_pages_against_the_quantised_oracle_and_the_truth(tiny, prompt_len, steps, chunk, page):
    w, model = tiny
    prompt = [int(t) for t in np.random.default_rng(prompt_len).integers(1, TINY_CFG["vocab_size"], size=prompt_len)]
    ids, rows, route, kv8_bytes = _engine_run(model, "fp8", [prompt], steps, page=page, chunk=chunk)
    assert route.startswith("aql"), route
    _, _, _, bf16_bytes = _engine_run(model, "bf16", [prompt], steps, page=page, chunk=chunk)
    assert kv8_bytes * 256 == bf16_bytes * 132  # 128 codes + a 4-byte scale per row of 256 bytes
    oracle, truth = O.OracleQwen3(TINY_CFG, w, kv_format="fp8"), O.TruthQwen3(TINY_CFG, w)
    for lo in range(0, prompt_len, chunk):  # the oracle sees the same chunks (rows > 8 take the tile-GEMM arithmetic)
        o = oracle.forward(prompt[lo:lo + chunk])[0, -1]
        t = truth.forward(prompt[lo:lo + chunk])[0, -1]
    want_o, want_t = [o], [t]
    for tok in ids[0][:steps]:  # teacher-forced on the ids the engine produced
        want_o.append(oracle.forward([tok])[0, -1])
        want_t.append(truth.forward([tok])[0, -1])
    got = np.stack([rows[0]] + [r[0] for r in rows[1:]])
    check_against_truth(got, np.stack(want_o), np.stack(want_t), what=f"engine, FP8 pages, {prompt_len}-token prompt in chunks of {chunk}, {steps} steps")


def test_engine_fp8_routes_are_bit_identical_and_batched_steps_agree(tiny):
    w, model = tiny
    rng = np.random.default_rng(5)
    prompts = [[int(t) for t in rng.integers(1, TINY_CFG["vocab_size"], size=n)] for n in (30, 7, 100, 55, 18, 64)]
    a = _engine_run(model, "fp8", prompts, 12, route="aql")
    b = _engine_run(model, "fp8", prompts, 12, route="hipgraph")
    assert a[2].startswith("aql") and b[2].startswith("hipgraph")
    assert a[0] == b[0]
    for x, y in zip(a[1], b[1]):
        np.testing.assert_array_equal(x, y)
    # the quantised cache is really in the loop: a bf16 engine answers with other logits
    c = _engine_run(model, "bf16", prompts, 12, route="aql")
    assert any(not np.array_equal(x, y) for x, y in zip(a[1], c[1]))
    # rows of the batched step (6 rows: the batched matmuls, attention reading fp32 slice partials) against the quantised oracle
    for i in (0, 2):
        oracle, truth = O.OracleQwen3(TINY_CFG, w, kv_format="fp8"), O.TruthQwen3(TINY_CFG, w)
        oracle.forward(prompts[i]), truth.forward(prompts[i])
        want_o = [oracle.forward([tok])[0, -1] for tok in a[0][i][:12]]
        want_t = [truth.forward([tok])[0, -1] for tok in a[0][i][:12]]
        check_against_truth(np.stack([r[i] for r in a[1]]), np.stack(want_o), np.stack(want_t), what=f"batched step, FP8 pages, sequence {i}")


def test_engine_fp8_fork_copies_a_quantised_tail_page(tiny):
    from tiny_llm_hip.engine import DecodeEngine

    w, model = tiny
    prompt = [int(t) for t in np.random.default_rng(9).integers(1, TINY_CFG["vocab_size"], size=37)]  # 37 = 2 pages of 16 + 5: a shared partial page
    eng = DecodeEngine(model, page_size=16, num_pages=40, max_batch=2, max_prefill_rows=64, kv_format="fp8")
    try:
        eng.begin(0)
        eng.prefill(0, prompt, chunk=64)
        eng.fork(0, 1)
        eng.decode(9, batch=2)
        a, b = eng.read_tokens(0, 9), eng.read_tokens(1, 9)  # the last nine (the fork has no prefill token of its own)
        assert a == b  # same state, same greedy continuation -- through a copied quantised page
        eng.release(0)
        eng.release(1)
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()
    solo = _engine_run(model, "fp8", [prompt], 9, chunk=64, max_prefill_rows=64)
    assert solo[0][0][1:] == a
