"""GPU tier: the bf16 FlashAttention prefill kernel on 8 waves per workgroup (round 6: one workgroup per CU, the K/V tile double-buffered in
LDS and shared by 4 heads x 2 query blocks, one barrier per stage, page ids by scalar loads) against its 4-wave twin (two workgroups per CU,
single-buffered tile, two barriers per stage).  A (head, 32-row query block) runs the same instruction sequence over the same values in both:
the outputs must be BIT-IDENTICAL -- which carries every oracle-held property of the 4-wave kernel (tests/test_ops_gpu.py,
tests/test_decode_kernels_gpu.py) over to the default.  Reference kernel: paged_attention_mma_bf16_d128, paged_attention.metal:250-506."""
import numpy as np
import pytest
import torch

from helpers import assert_bf16_close
from oracle import kv_fp8
from oracle import tiny_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
HQ, HKV, D = 32, 8, 128


@pytest.fixture(scope="module")
def ext():
    import tiny_llm_ext_hip

    tiny_llm_ext_hip.load_library(".")
    return tiny_llm_ext_hip


def _case(seed, L, ctxs, page, hq=HQ, hkv=HKV):
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = len(ctxs)
    need = [(c + page - 1) // page for c in ctxs]
    P = sum(need) + 3
    ids = torch.randperm(P, generator=g).tolist()
    table = -torch.ones((B, max(need) + 1), dtype=torch.int32)
    for b in range(B):
        for j in range(need[b]):
            table[b, j] = ids.pop()
    k = torch.randn((P, hkv, page, D), generator=g).to(torch.bfloat16)
    v = torch.randn((P, hkv, page, D), generator=g).to(torch.bfloat16)
    q = torch.randn((B * hq, L, D), generator=g).to(torch.bfloat16)
    return q.to(DEV), k.to(DEV), v.to(DEV), table.to(DEV), torch.tensor(ctxs, dtype=torch.int32, device=DEV)


CASES = [(64, [64], 64), (65, [200, 65], 64), (100, [300, 100], 128), (512, [8192], 128), (2048, [2048], 128), (1000, [5000], 64),
         (4096, [8192], 128), (300, [300, 2000, 777], 128), (128, [16384], 256)]


@pytest.mark.parametrize("L,ctxs,page", CASES)
@pytest.mark.parametrize("causal", [True, False])
def test_eight_wave_kernel_equals_the_four_wave_twin_bit_for_bit(ext, L, ctxs, page, causal):
    q, k, v, table, cl = _case(L + sum(ctxs), L, ctxs, page)
    kw = dict(num_kv_heads=HKV, num_heads=HQ, max_context_hint=max(ctxs))
    before = ext.paged_attention_waves(0)
    try:
        ext.paged_attention_waves(4)
        four = ext.paged_attention(q, k, v, table, cl, D ** -0.5, causal, **kw)
        ext.paged_attention_waves(8)
        eight = ext.paged_attention(q, k, v, table, cl, D ** -0.5, causal, **kw)
        torch.cuda.synchronize()
    finally:
        ext.paged_attention_waves(before)
    assert not torch.isnan(eight.float()).any()
    assert torch.equal(four, eight)


def test_eight_wave_kernel_with_other_head_groupings_and_fp8_pages(ext):
    """GQA 2:1 and 1:1 (8 waves = 2 or 1 heads x 4 or 8 query blocks) and FP8 pages (codes converted when a chunk is stored into the tile)."""
    for hq, hkv in ((16, 8), (8, 8), (32, 4)):
        q, k, v, table, cl = _case(hq, 500, [1500, 500], 128, hq, hkv)
        kw = dict(num_kv_heads=hkv, num_heads=hq, max_context_hint=1500)
        outs = []
        for nw in (4, 8):
            ext.paged_attention_waves(nw)
            outs.append(ext.paged_attention(q, k, v, table, cl, D ** -0.5, True, **kw))
        ext.paged_attention_waves(8)
        assert torch.equal(outs[0], outs[1]), (hq, hkv)
    q, k, v, table, cl = _case(3, 700, [2100], 128)
    kc, ks = ext.kv_fp8_quantize_rows(k)
    vc, vs = ext.kv_fp8_quantize_rows(v)
    kw = dict(num_kv_heads=HKV, num_heads=HQ, max_context_hint=2100)
    outs = []
    for nw in (4, 8):
        ext.paged_attention_waves(nw)
        outs.append(ext.paged_attention_fp8(q, kc, ks, vc, vs, table, cl, D ** -0.5, True, **kw))
    ext.paged_attention_waves(8)
    assert torch.equal(outs[0], outs[1])
    twin = ext.paged_attention(q, ext.kv_fp8_dequantize_rows(kc, ks), ext.kv_fp8_dequantize_rows(vc, vs), table, cl, D ** -0.5, True, **kw)
    assert torch.equal(outs[1], twin)


def test_eight_wave_kernel_against_the_oracle_on_sampled_rows(ext):
    """The default kernel itself against the oracle (P rounded to bf16 before P.V, paged_attention.metal:439-444), at a chunk shape of the
    engine's prefill: the last 4,096 rows of an 8,192-token context, 64-token softmax steps."""
    rng = np.random.default_rng(7)
    L, ctx, page = 4096, 8192, 128
    need = ctx // page
    P = need + 2
    table = np.asarray([rng.permutation(P)[:need]], dtype=np.int32)
    kp = O.bf16(rng.standard_normal((P, HKV, page, D), dtype=np.float32))
    vp = O.bf16(rng.standard_normal((P, HKV, page, D), dtype=np.float32))
    q = O.bf16(rng.standard_normal((HQ, L, D), dtype=np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.bfloat16)
    assert ext.paged_attention_waves(0) == 8
    got = ext.paged_attention(t(q), t(kp), t(vp), torch.from_numpy(table).to(DEV), torch.tensor([ctx], dtype=torch.int32, device=DEV),
                              D ** -0.5, True, num_kv_heads=HKV, num_heads=HQ, max_context_hint=ctx).float().cpu().numpy()
    for r in sorted({0, 31, 32, 63, 64, 2047, 2048, 4095, *rng.integers(0, L, size=6).tolist()}):
        vis = ctx - L + r + 1
        want = O.paged_attention(q[:, r:r + 1], kp, vp, table, np.asarray([vis], dtype=np.int32), D ** -0.5, True, HKV, HQ, "bf16", round_p=True)
        assert_bf16_close(got[:, r], want[:, 0], ulps=1.0, abs_floor=1.5e-3 + 2.0 ** -9 * 4.0, what=f"row {r} (sees {vis} tokens)")
    assert kv_fp8 is not None
