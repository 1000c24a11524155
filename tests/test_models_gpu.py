"""GPU tier: the host mirror of the reference models on the HIP operators — the reference's model-level ladder
(tests_refsol/test_week_2_day_1.py, test_week_2_day_6.py:92-153, test_week_3_day_3.py:386-402,
test_week_3_day_4.py:325-345): every Week-2 checkpoint of the kernel ladder, the Week-3 paged model and the Week-1
dense model must agree on log-probs for the same seeded W4 checkpoint, and all of them with the numpy oracle.

Tolerance (derived, see test_engine_gpu.py's docstring): every path is teacher-forced on the bf16 oracle's greedy ids and
its RAW logits are compared with the float64 no-rounding truth (oracle.TruthQwen3):  max|path - truth| must stay within
`factor` x the bf16 oracle's own max distance from the truth + 1 bf16 ulp.  factor = 1.5 for paths with the oracle's rounding
points (Week 3, Week-2 checkpoints from "rmsnorm" on), 2.5 for the early Week-2 checkpoints and Week 1, whose reference
semantics round MORE often (readable RMSNorm rounds twice, layer_norm.py:10-15; "kv-cache" and Week 1 also round every
dequantised weight to bf16, quantize.py:93-100).  Reference bars for comparison: rtol 0.1 / atol 2.0 on full models
(test_week_2_day_6.py:92-109), 1e-3 between two paths that share every kernel on a fake model (test_week_3_day_3.py:386-402).
"""

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O
from helpers import TINY_CFG, check_against_truth, to_mlx_shaped

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ckpt():
    w = O.make_qwen3_weights(TINY_CFG, seed=11, sigma=0.05)
    return w, to_mlx_shaped(TINY_CFG, w)


def run_cached(model, prompt, steps, forced=None):
    """prefill + decode through a KV-cached model; returns per-step last-row logits.  `forced`: the ids to feed (teacher
    forcing); otherwise the model's own greedy ids."""
    cache = model.create_kv_cache()
    try:
        toks = torch.tensor([prompt], dtype=torch.int32, device="cuda")
        out = [model(toks, 0, cache, logits_to_keep=1)[0, -1].float().cpu().numpy()]
        offset = len(prompt)
        for i in range(steps):
            tok = forced[i] if forced is not None else int(np.argmax(out[-1]))
            t = torch.tensor([[tok]], dtype=torch.int32, device="cuda")
            out.append(model(t, offset, cache, logits_to_keep=1)[0, -1].float().cpu().numpy())
            offset += 1
        return np.stack(out)
    finally:
        for c in cache:
            c.release()


def test_week2_ladder_week3_and_oracle_agree(ckpt):
    from tiny_llm_hip import Qwen3ModelWeek2, Qwen3ModelWeek3
    from tiny_llm_hip.qwen3_week2 import WEEK2_CHECKPOINTS

    w, mlx_model = ckpt
    prompt = [7, 300, 12, 901, 44, 5, 610, 73, 250, 18, 999]  # 11 tokens: the prefill takes the GEMM path
    ref, exact = O.OracleQwen3(TINY_CFG, w), O.TruthQwen3(TINY_CFG, w)
    want, truth, ids = [ref.forward(prompt)[0, -1]], [exact.forward(prompt)[0, -1]], []
    for _ in range(4):
        ids.append(int(np.argmax(want[-1])))
        want.append(ref.forward([ids[-1]])[0, -1])
        truth.append(exact.forward([ids[-1]])[0, -1])
    want, truth = np.stack(want), np.stack(truth)
    week3 = run_cached(Qwen3ModelWeek3(mlx_model, page_size=4), prompt, 4, forced=ids)
    check_against_truth(week3, want, truth, what="Qwen3ModelWeek3 op by op, TINY")
    for name in WEEK2_CHECKPOINTS:
        got = run_cached(Qwen3ModelWeek2(mlx_model, checkpoint=name), prompt, 4, forced=ids)
        more_roundings = WEEK2_CHECKPOINTS.index(name) < WEEK2_CHECKPOINTS.index("rmsnorm")
        check_against_truth(got, want, truth, what=f"Qwen3ModelWeek2 checkpoint {name}, TINY", factor=2.5 if more_roundings else 1.5)


def test_week1_dense_model_matches_cached_models(ckpt):
    """Week 1 re-runs the whole context without a cache, on dequantised bf16 weights (qwen3_week1.py:206-217)."""
    from tiny_llm_hip import Qwen3ModelWeek1, Qwen3ModelWeek3

    w, mlx_model = ckpt
    prompt = [9, 8, 700, 6, 55, 4]
    toks = torch.tensor([prompt], dtype=torch.int32, device="cuda")
    dense = Qwen3ModelWeek1(mlx_model)(toks)[0, -1].float().cpu().numpy()
    paged = run_cached(Qwen3ModelWeek3(mlx_model, page_size=4), prompt, 0)[0]
    want = O.OracleQwen3(TINY_CFG, w).forward(prompt)[0, -1]
    truth = O.TruthQwen3(TINY_CFG, w).forward(prompt)[0, -1]
    check_against_truth(paged[None], want[None], truth[None], what="Qwen3ModelWeek3 prefill, TINY")
    check_against_truth(dense[None], want[None], truth[None], what="Qwen3ModelWeek1 (dense bf16 weights), TINY", factor=2.5)


def test_week2_offset_mismatch_is_rejected(ckpt):
    from tiny_llm_hip import Qwen3ModelWeek2

    model = Qwen3ModelWeek2(ckpt[1])
    cache = model.create_kv_cache()
    toks = torch.tensor([[1, 2, 3]], dtype=torch.int32, device="cuda")
    model(toks, 0, cache, logits_to_keep=1)
    with pytest.raises(ValueError, match="does not match model offset"):
        model(toks[:, :1], 7, cache, logits_to_keep=1)
    for c in cache:
        c.release()


def test_paged_pool_counters_on_device():
    """The reference's pool literals with the real extension op (tests_refsol/test_week_3_day_3.py:238-270)."""
    from tiny_llm_hip import TinyKvPagedCache, TinyKvPagedPool

    pool = TinyKvPagedPool(page_size=4)
    cache = TinyKvPagedCache(pool=pool)
    g = torch.Generator(device="cuda").manual_seed(0)
    k = torch.randn((1, 2, 17, 4), device="cuda", generator=g)
    v = torch.randn((1, 2, 17, 4), device="cuda", generator=g)
    cache.update_and_fetch_paged(k, v)
    assert (pool.num_pages, pool.capacity, pool.storage_growths, pool.copied_pages_on_growth, pool.copied_bytes_on_growth) \
        == (5, 8, 2, 4, 1024)
    dk, dv = cache.gather_dense()
    assert torch.equal(dk, k) and torch.equal(dv, v)
    cache.release()
    assert pool.used_page_ids == set() and pool.num_free_pages == 5


def test_batch_generate_with_real_model(ckpt):
    """Continuous batching of the op-by-op Week-3 model (reference batch.py:136-285) over real kernels: every request
    finishes, pages come back, and a request served alone produces the same text."""
    from tiny_llm_hip import Qwen3ModelWeek3, batch_generate

    class Detok:
        def __init__(self, _):
            self.text = ""

        def add_token(self, token):
            self.text += f"{token},"

    class Tok:
        eos_token_id = -1
        _tokenizer = object()
        detokenizer = Detok(_tokenizer)

        def encode(self, prompt, add_special_tokens=False):
            return [int(t) for t in prompt.split()]

    model = Qwen3ModelWeek3(ckpt[1], page_size=4)
    prompts = ["5 6 7 8 9 10 11", "100 200 300", "42 43 44 45 46 47 48 49 50 51 52 53", "9"]
    batched = dict(batch_generate(model, Tok(), prompts, max_seq_len=20, batch_size=2, prefill_step=4))
    assert sorted(batched) == [0, 1, 2, 3] and all(batched.values())
    solo = dict(batch_generate(model, Tok(), [prompts[1]], max_seq_len=20, batch_size=1, prefill_step=4))
    assert solo[0].split(",")[0] == batched[1].split(",")[0]
    for pool in model.page_pools:
        assert pool.used_page_ids == set()


def test_week3_model_with_moe_layers():
    """Qwen3-MoE wiring of Qwen3ModelWeek3 (reference qwen3_week3.py:210-272): with ONE expert that is always selected the
    sparse layer must reproduce the dense MLP built from the same weights; with 4 experts / top-2 the model runs end to
    end through the grouped-expert GEMV (the block arithmetic itself is checked against the oracle in test_ops_gpu.py)."""
    import copy
    from types import SimpleNamespace

    from tiny_llm_hip import Qwen3ModelWeek3

    w = O.make_qwen3_weights(TINY_CFG, seed=21, sigma=0.05)
    dense = to_mlx_shaped(TINY_CFG, w)

    def stack(layer, n):
        return SimpleNamespace(weight=torch.stack([layer.weight] * n), scales=torch.stack([layer.scales] * n),
                               biases=torch.stack([layer.biases] * n), group_size=128, bits=4)

    def router(n):
        packed, scales, biases = O.quantize_affine(np.random.default_rng(5).standard_normal((n, TINY_CFG["hidden_size"])).astype(np.float32) * 0.2)
        return SimpleNamespace(weight=torch.from_numpy(packed.view(np.int32)).cuda(), scales=torch.from_numpy(scales).cuda().to(torch.bfloat16),
                               biases=torch.from_numpy(biases).cuda().to(torch.bfloat16), group_size=128, bits=4)

    def moe_variant(n_experts, top_k):
        m = copy.copy(dense)
        m.args = SimpleNamespace(**vars(dense.args), num_experts=n_experts, num_experts_per_tok=top_k, norm_topk_prob=True,
                                 decoder_sparse_step=1, mlp_only_layers=[])
        inner = copy.copy(dense.model)
        inner.layers = []
        for layer in dense.model.layers:
            lay = copy.copy(layer)
            lay.mlp = SimpleNamespace(gate=router(n_experts), switch_mlp=SimpleNamespace(
                gate_proj=stack(layer.mlp.gate_proj, n_experts), up_proj=stack(layer.mlp.up_proj, n_experts),
                down_proj=stack(layer.mlp.down_proj, n_experts)))
            inner.layers.append(lay)
        m.model = inner
        return m

    tokens = torch.tensor([[5, 17, 400, 3, 99, 250, 7]], dtype=torch.int32, device="cuda")

    def run(model_obj):
        net = Qwen3ModelWeek3(model_obj, page_size=16)
        cache = net.create_kv_cache()
        try:
            return net(tokens, 0, cache).float().cpu().numpy()
        finally:
            for c in cache:
                c.release()

    # every variant computes the dense model's function: all are held against ITS float64 truth (all 7 rows of the prompt)
    ids = tokens[0].tolist()
    oracle = O.OracleQwen3(TINY_CFG, w).forward(ids, logits_to_keep=None)[0]
    truth = O.TruthQwen3(TINY_CFG, w).forward(ids, logits_to_keep=None)[0]
    check_against_truth(run(dense)[0], oracle, truth, what="dense twin of the MoE model, TINY")
    got = run(moe_variant(1, 1))       # softmax over one expert = 1.0, renormalised top-1 = 1.0
    check_against_truth(got[0], oracle, truth, what="MoE 1 expert / top-1 == dense, TINY", factor=2.0)
    many = run(moe_variant(4, 2))      # identical experts, probabilities renormalised to sum to 1 -> the dense result again
    assert np.isfinite(many).all()
    # two experts' outputs are rounded to bf16, weighted by bf16 probabilities and summed in bf16 (moe.py:60-89): more
    # rounding points than the dense MLP the oracle models
    check_against_truth(many[0], oracle, truth, what="MoE 4 identical experts / top-2 == dense, TINY", factor=2.5)


@pytest.mark.gpu
def test_profile_week2_kernel_group_replay_runs_on_a_tiny_model():
    """benches/profile_week2_kernels.py: every default checkpoint / phase builds its model and replays its kernel groups
    (projections, attention, pointwise, KV growth on decode) on the HIP operators; shares sum to one."""
    from benches import profile_week2_kernels as P
    from tiny_llm_hip.synthetic import synthetic_qwen3

    cfg = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
               intermediate_size=512, vocab_size=1024, rope_theta=1000000, rms_norm_eps=1e-6, max_position_embeddings=4096,
               tie_word_embeddings=True)
    model = synthetic_qwen3(cfg, seed=1, sigma=0.05, device="cuda")
    for spec in P.DEFAULT_CASES:
        case = P.parse_case(spec)
        out = P.profile_case(model, case, warmup=1, iterations=2, seed=0)
        names = [c["name"] for c in out["categories"]]
        assert names == list(P.GROUPS[:4 if case.phase == "decode" else 3])
        assert abs(sum(c["share"] for c in out["categories"]) - 1.0) < 1e-9 and out["attributed_us"] > 0
