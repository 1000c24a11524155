"""CPU tier: the plans the decode path picks, read through the host-only entry points (include/tinyllm_engine.h:
tl_decode_gemv_plan, tl_decode_attention_plan) -- no device, no launch.  The numbers behind each rule are in
profiles/r03_labs/README.md (plan sweeps, window sizes, switch points); the same GEMV plans are asserted on the device against the
kernels that actually ran (tests/test_decode_kernels_gpu.py)."""

import ctypes

import pytest

QWEN3_4B = {"qkv": (6144, 2560), "wo": (2560, 4096), "gate_up": (19456, 2560), "down": (2560, 9728), "lm_head": (151936, 2560)}


@pytest.fixture(scope="module")
def lib(built_libs):
    import tiny_llm_ext_hip as ext

    return ext.lib()


def gemv_plan(lib, M, name):
    rows, cols = QWEN3_4B[name]
    out = (ctypes.c_int * 5)()
    ok = lib.tl_decode_gemv_plan(M, rows, cols, out)
    return ok, tuple(out)


def attention_plan(lib, batch, ctx, heads=32, kv_heads=8):
    out = (ctypes.c_int * 3)()
    assert lib.tl_decode_attention_plan(batch, ctx, heads, kv_heads, out) == 1
    return tuple(out)


def test_gemv_plans_at_the_qwen3_4b_projections(lib):
    """(activation rows per workgroup, reduction split, waves, groups per wave, workgroups): one row -- the kernels bench.py's roofline
    names; 2 and 4 rows -- gate|up keeps the finer cut, w_down takes 16 waves of 5 groups at 3-4 rows (round 3, tools/lab/plan_lab)."""
    want = {
        (1, "qkv"): (1, 2, 4, 10, 192), (1, "wo"): (1, 4, 4, 8, 160), (1, "gate_up"): (1, 4, 4, 5, 1216), (1, "down"): (1, 8, 8, 10, 160),
        (1, "lm_head"): (1, 2, 4, 10, 4748),
        (2, "qkv"): (2, 2, 4, 10, 192), (2, "gate_up"): (2, 4, 4, 5, 1216), (2, "down"): (2, 8, 8, 10, 160),
        (3, "down"): (4, 16, 16, 5, 160), (4, "gate_up"): (4, 4, 4, 5, 1216), (4, "down"): (4, 16, 16, 5, 160), (4, "wo"): (4, 4, 4, 8, 160),
    }
    for (M, name), plan in want.items():
        ok, got = gemv_plan(lib, M, name)
        assert ok == 1 and got == plan, f"{name} at {M} rows: {got}"
    assert lib.tl_decode_gemv_plan(1, 100, 2560, (ctypes.c_int * 5)()) == 0, "rows that are not a multiple of 16 go to the packed-dot GEMV"
    assert lib.tl_decode_gemv_plan(1, 2560, 9728 * 4, (ctypes.c_int * 5)()) == 0, "a reduction of 304 groups does not fit 8 x 10 groups per wave"
    assert lib.tl_decode_gemv_plan(0, 2560, 2560, (ctypes.c_int * 5)()) == 0


# (hidden, Hq * D, (Hq + 2 Hkv) * D, intermediate, vocab) of the dense Qwen3 family and of Qwen2-7B (public configs; head_dim 128)
MODEL_SHAPES = {
    "qwen3-0.6b": (1024, 2048, 4096, 3072, 151936), "qwen3-1.7b": (2048, 2048, 4096, 6144, 151936),
    "qwen3-4b": (2560, 4096, 6144, 9728, 151936), "qwen3-8b": (4096, 4096, 6144, 12288, 151936),
    "qwen3-14b": (5120, 5120, 7168, 17408, 151936), "qwen3-32b": (5120, 8192, 10240, 25600, 151936),
    "qwen2-7b": (3584, 3584, 4608, 18944, 152064),
}


@pytest.mark.parametrize("model", list(MODEL_SHAPES))
def test_every_taken_gemv_plan_has_a_compiled_kernel(lib, model):
    """Round-3 advisor finding: a planner rule promoted w_down of Qwen3-8B / 14B / Qwen2-7B at 3-4 rows to 16 waves x 8 / 10 groups,
    which no kernel is compiled for (the engine then failed instead of decoding).  A plan is "taken" only for a combination in the
    instantiation table (csrc/qmv3.hip Q3_TABLE); every other shape decodes through the packed-dot GEMV."""
    hidden, q_dim, qkv_dim, inter, vocab = MODEL_SHAPES[model]
    projections = {"qkv": (qkv_dim, hidden), "wo": (hidden, q_dim), "gate_up": (2 * inter, hidden), "down": (hidden, inter),
                   "lm_head": (vocab, hidden)}
    for name, (rows, cols) in projections.items():
        for M in range(1, 9):
            out = (ctypes.c_int * 5)()
            ok = lib.tl_decode_gemv_plan(M, rows, cols, out)
            MR, KS, CW, LM, blocks = tuple(out)
            compiled = lib.tl_decode_gemv_variant_compiled(MR, KS, CW, LM)
            assert ok == 0 or compiled == 1, f"{model} {name} at {M} rows: plan {tuple(out)} is taken but not compiled"
            if ok:
                assert MR >= M or MR == 8, f"{model} {name} at {M} rows: {tuple(out)}"
                assert KS * LM * 128 >= cols and blocks >= 1, f"{model} {name} at {M} rows: the plan does not cover the reduction: {tuple(out)}"
    # the cases the finding named: four rows against reductions of 96 / 136 groups (w_down of Qwen3-8B / 14B) are beyond 8 waves x 10
    # groups and no 16-wave kernel with 8 / 10 groups per wave exists: not taken (packed-dot GEMV), as before round 3
    out = (ctypes.c_int * 5)()
    assert lib.tl_decode_gemv_plan(4, 4096, 12288, out) == 0 and lib.tl_decode_gemv_plan(4, 5120, 17408, out) == 0
    assert lib.tl_decode_gemv_plan(4, 2560, 9728, out) == 1 and tuple(out)[:4] == (4, 16, 16, 5)  # the shape the rule was measured on


def test_the_planners_variant_predicate_is_the_instantiation_table(lib):
    """qmv3_has_variant (header, used by qmv3_plan) against the macro table the launcher is generated from."""
    import itertools

    table = {(mr, ks, cw, lm) for mr, ks, cw, lm in itertools.product(range(1, 17), range(1, 17), (4, 8, 16), range(1, 12))
             if lib.tl_decode_gemv_variant_compiled(mr, ks, cw, lm)}
    assert len(table) == 53  # round 5: exactly the combinations the planner reaches (test_every_compiled_gemv_variant_is_reached_by_some_shape)
    assert (4, 16, 16, 5) in table and (4, 16, 16, 8) not in table and (1, 8, 8, 10) in table and (1, 8, 4, 10) not in table and (1, 8, 8, 4) not in table and (1, 2, 8, 10) not in table


@pytest.mark.parametrize("batch,ctx,windows,heads_per_wg,max_window", [
    (1, 40, 1, 1, 64), (1, 100, 2, 1, 64), (1, 150, 4, 1, 64), (1, 255, 4, 1, 64),   # up to 256 tokens: 64-token windows
    (1, 300, 4, 1, 128), (1, 511, 4, 1, 128),                                        # 257..512: 128 (4 partials for the wo GEMV, not 8)
    (1, 700, 4, 1, 256), (1, 1500, 8, 1, 256), (1, 3000, 16, 1, 256),                # beyond: 256-token windows
    (1, 5000, 32, 4, 256), (1, 20000, 32, 4, 1024), (1, 32768, 32, 4, 2048),         # above 4,096 tokens a GQA group per workgroup, 32 windows
    (2, 300, 8, 1, 64), (2, 700, 4, 1, 256), (2, 1500, 8, 4, 256), (2, 3000, 16, 4, 256),  # two sequences switch at 1,024 tokens
    (4, 256, 8, 4, 64), (4, 1000, 4, 4, 256),                                        # 3+ sequences: always the GQA-group walk
    (4, 4500, 8, 4, 1024), (3, 8000, 8, 4, 1024), (2, 8000, 16, 4, 512), (4, 32000, 8, 4, 4096),  # ... on at most 256 workgroups (end of round 6)
    (8, 300, 4, 4, 128), (8, 1000, 4, 4, 256), (64, 300, 1, 4, 512),                 # 5+ sequences up to 1,024 tokens (end of round 6): windows of 128+ tokens,
    (5, 600, 4, 4, 256), (9, 300, 2, 4, 256), (16, 600, 2, 4, 512), (23, 600, 1, 4, 1024),   # ... at most one workgroup per CU (256),
    (5, 150, 1, 4, 256), (11, 200, 1, 4, 256), (12, 200, 1, 4, 256), (16, 200, 1, 4, 256),   # ... no split (and no merge launch) up to 256 tokens,
    (24, 200, 1, 4, 256), (32, 700, 1, 4, 768), (32, 1500, 1, 4, 2048), (20, 1500, 1, 4, 2048), (24, 8000, 1, 4, 8192), (24, 12000, 2, 4, 8192),  # from 17 sequences none up to 8,192              # ... from 24 sequences none up to 1,024; beyond 1,024 tokens the round-4 plan
    (8, 2000, 4, 4, 512), (16, 2000, 2, 4, 1024), (10, 3000, 2, 4, 2048), (12, 6000, 2, 4, 4096), (18, 3000, 1, 4, 4096), (17, 1500, 1, 4, 2048), (5, 2000, 4, 4, 512), (7, 4000, 4, 4, 1024), (6, 8000, 4, 4, 2048),  # (5-7 sequences: 4 windows there too)
])
def test_attention_plans_by_context_and_sequences(lib, monkeypatch, batch, ctx, windows, heads_per_wg, max_window):
    for name in ("TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MIN_TOKENS", "TL_ATTN_MFMA"):
        monkeypatch.delenv(name, raising=False)
    n_splits, per_split, rq = attention_plan(lib, batch, ctx)
    assert (n_splits, rq) == (windows, heads_per_wg), f"{batch} sequences, {ctx} tokens: {n_splits} windows of {per_split}, {rq} heads per workgroup"
    assert per_split % 64 == 0 and 64 <= per_split <= max_window and n_splits * per_split >= ctx + 1


def test_attention_plan_knobs_are_read(lib, monkeypatch):
    monkeypatch.setenv("TL_ATTN_MIN_TOKENS", "64")
    assert attention_plan(lib, 1, 3000)[0] == 64  # the round-2 plan: 64-token windows whatever the context
    monkeypatch.setenv("TL_ATTN_MAX_SPLITS", "8")
    assert attention_plan(lib, 1, 3000)[0] == 8
    monkeypatch.delenv("TL_ATTN_MIN_TOKENS")
    monkeypatch.delenv("TL_ATTN_MAX_SPLITS")
    monkeypatch.setenv("TL_ATTN_RQ", "4")
    assert attention_plan(lib, 1, 300)[2] == 4


def batched_plan(lib, M, rows, cols):
    out = (ctypes.c_int * 6)()
    ok = lib.tl_decode_batched_plan(M, rows, cols, out)
    return ok, tuple(out)


def test_register_resident_matmul_plans_at_qwen3_4b_shapes(lib):
    """csrc/qmm6.h: (16-row blocks per workgroup, groups per wave) such that MB x GPW x 16 fragment registers fit one wave per SIMD
    (<= 320); rows beyond a workgroup's block go to further workgroups over the same tiles; one workgroup per CU and row block.  The block count is
    the largest that fits -- except (round 6) where the rows are long and the tiles few (wo): there the cheapest by the planner's arithmetic
    (what a CU pulls + its walk), 16-row blocks over more workgroups."""
    for M, blocks in ((5, 1), (16, 1), (17, 2), (32, 2), (33, 3), (64, 4)):
        for name, (rows, cols) in QWEN3_4B.items():
            ok, (MB, GPW, sets, row_blocks, wgs, tpw) = batched_plan(lib, M, rows, cols)
            assert ok == 1, f"{name} at {M} rows"
            assert GPW * 4 >= cols // 128 and MB * GPW * 16 <= 320 and MB * row_blocks >= blocks, f"{name} at {M} rows: {(MB, GPW, row_blocks)}"
            # several row blocks: the workgroups of one tile range sit a multiple of 8 apart (one XCD); the surplus ones find no tile and leave
            spare = 7 if row_blocks > 1 else 0
            assert wgs * tpw >= rows // 16 and (wgs - 1 - spare) * tpw < rows // 16, f"{name} at {M} rows: tiles {rows // 16} over {wgs} x {tpw}"
            assert row_blocks == 1 or wgs % 8 == 0, f"{name} at {M} rows: {wgs} workgroups per row block"
            assert wgs * row_blocks <= 256 + 8 * row_blocks or tpw == 1, f"{name} at {M} rows: more than one workgroup per CU while a workgroup walks several tiles"
            assert sets == 1 if tpw == 1 else sets >= 2 or GPW > 8, f"{name} at {M} rows: a workgroup that walks tiles keeps a tile in flight"
            assert 4 * 7 + (sets - 1) * 2 * GPW <= 63, f"{name} at {M} rows: the counted waits of the transposer hold 6 bits"
    assert batched_plan(lib, 64, 19456, 2560)[1][:4] == (4, 5, 2, 1)
    assert batched_plan(lib, 64, 2560, 4096)[1] == (1, 8, 3, 4, 56, 3)  # wo: 4 x 56 workgroups of 16 rows x 3 tiles (until round 6: 2 x 80 of 32 rows x 2)
    assert batched_plan(lib, 33, 2560, 4096)[1][:4] == (1, 8, 2, 3) and batched_plan(lib, 32, 2560, 4096)[1][:4] == (1, 8, 2, 2)
    assert batched_plan(lib, 64, 151936, 2560)[1][:4] == (4, 5, 2, 1) and batched_plan(lib, 64, 6144, 2560)[1][:4] == (4, 5, 2, 1)  # lm_head, qkv keep the largest block
    assert batched_plan(lib, 64, 2560, 9728)[1][:2] == (1, 19) and batched_plan(lib, 64, 2560, 9728)[1][3] == 4
    assert batched_plan(lib, 65, 2560, 2560)[0] == 0 and batched_plan(lib, 8, 100, 2560)[0] == 0 and batched_plan(lib, 8, 2560, 100)[0] == 0


@pytest.mark.parametrize("model", list(MODEL_SHAPES))
def test_every_taken_batched_plan_has_a_compiled_kernel(lib, model):
    hidden, q_dim, qkv_dim, inter, vocab = MODEL_SHAPES[model]
    projections = {"qkv": (qkv_dim, hidden), "wo": (hidden, q_dim), "gate_up": (2 * inter, hidden), "down": (hidden, inter),
                   "lm_head": (vocab, hidden)}
    for name, (rows, cols) in projections.items():
        for M in (5, 8, 16, 17, 32, 33, 64):
            ok, (MB, GPW, sets, row_blocks, wgs, tpw) = batched_plan(lib, M, rows, cols)
            if ok:
                assert lib.tl_decode_batched_variant_compiled(MB, GPW) == 1, f"{model} {name} at {M} rows: {(MB, GPW)} is taken but not compiled"
                assert GPW * 4 >= (cols + 127) // 128, f"{model} {name} at {M} rows: the waves do not cover the reduction"


def test_the_batched_planners_variant_predicate_is_the_instantiation_table(lib):
    table = {(mb, gpw) for mb in range(1, 9) for gpw in range(1, 33) if lib.tl_decode_batched_variant_compiled(mb, gpw)}
    assert table == {(1, 2), (1, 4), (1, 5), (1, 8), (1, 19), (2, 2), (2, 4), (2, 5), (2, 8), (4, 2), (4, 4), (4, 5)}


def streaming_plan(lib, M, rows, cols):
    out = (ctypes.c_int * 4)()
    ok = lib.tl_decode_streaming_plan(M, rows, cols, out)
    return ok, tuple(out)


def test_row_streaming_matmul_plans_at_qwen3_4b_shapes(lib):
    """csrc/qmm7.h (round 6): gate|up and qkv of a 2,560-wide model -- a workgroup owns T tiles, a wave 5 of the 20 groups; row blocks are
    EXACTLY ceil(M / 16), 3 included (a 33-row step does not pay for 64); one workgroup per CU at most.  wo (4,096 columns: 8 groups per
    wave), w_down (19) and lm_head (37 tiles per CU) stay with the register-resident / K-sliced kernels."""
    for M in (5, 8, 9, 16, 17, 32, 33, 48, 49, 64):
        for name, T in (("qkv", 2), ("gate_up", 5)):
            rows, cols = QWEN3_4B[name]
            ok, (MB, Tp, GPW, wgs) = streaming_plan(lib, M, rows, cols)
            assert ok == 1 and (MB, Tp, GPW) == ((M + 15) // 16, T, 5), f"{name} at {M} rows: {(ok, MB, Tp, GPW)}"
            assert wgs <= 256 and wgs * Tp >= rows // 16 > (wgs - 1) * Tp, f"{name} at {M} rows: {wgs} workgroups x {Tp} tiles for {rows // 16}"
            assert lib.tl_decode_streaming_variant_compiled(Tp, GPW) == 1
        for name in ("wo", "down", "lm_head"):
            rows, cols = QWEN3_4B[name]
            assert streaming_plan(lib, M, rows, cols)[0] == 0, f"{name} at {M} rows"
    assert streaming_plan(lib, 65, 6144, 2560)[0] == 0 and streaming_plan(lib, 8, 100, 2560)[0] == 0 and streaming_plan(lib, 8, 2560, 100)[0] == 0


@pytest.mark.parametrize("model", list(MODEL_SHAPES))
def test_every_taken_streaming_plan_has_a_compiled_kernel(lib, model):
    hidden, q_dim, qkv_dim, inter, vocab = MODEL_SHAPES[model]
    projections = {"qkv": (qkv_dim, hidden), "wo": (hidden, q_dim), "gate_up": (2 * inter, hidden), "down": (hidden, inter), "lm_head": (vocab, hidden)}
    for name, (rows, cols) in projections.items():
        for M in (5, 8, 16, 17, 32, 33, 48, 64):
            ok, (MB, T, GPW, wgs) = streaming_plan(lib, M, rows, cols)
            if ok:
                assert lib.tl_decode_streaming_variant_compiled(T, GPW) == 1, f"{model} {name} at {M} rows: {(T, GPW)} is taken but not compiled"
                assert GPW * 4 >= (cols + 127) // 128 and 1 <= MB <= 4 and wgs * T >= rows // 16, f"{model} {name} at {M} rows"


def test_the_streaming_planners_variant_predicate_is_the_instantiation_table(lib):
    table = {(t, gpw) for t in range(1, 17) for gpw in range(1, 33) if lib.tl_decode_streaming_variant_compiled(t, gpw)}
    assert table == {(2, 5), (5, 5)}
    reached = set()
    out = (ctypes.c_int * 4)()
    for M in (5, 16, 17, 33, 64):
        for G in range(1, 80):
            for tiles in list(range(1, 64)) + [96, 128, 160, 192, 256, 320, 384, 512, 608, 640, 768, 1024, 1216, 1280, 2048, 4748, 9496]:
                if lib.tl_decode_streaming_plan(M, tiles * 16, G * 128, out) == 1:
                    reached.add(tuple(out)[1:3])
    assert reached == table, f"planner reaches {sorted(reached)}, compiled {sorted(table)}"


def test_every_compiled_gemv_variant_is_reached_by_some_shape(lib):
    """The other direction of the test above (round-4 review, item 8): an instantiation no planner rule selects is dead weight in the
    library (round 4 carried 98 GEMV combinations x 5 fused variants; 45 of the 98 could only be reached through lab-only overrides).
    Sweep the planner over 1..8 rows, reductions of 1..255 groups and a spread of tile counts, collect what it returns, and hold the
    instantiation table (csrc/qmv3.hip Q3_TABLE, through tl_decode_gemv_variant_compiled) to exactly that set."""
    reached = set()
    tile_counts = list(range(1, 64)) + [96, 128, 160, 192, 256, 320, 384, 512, 608, 640, 768, 1024, 1216, 2048, 4748, 9496]
    out = (ctypes.c_int * 5)()
    for M in range(1, 9):
        for G in range(1, 256):
            for tiles in tile_counts:
                if lib.tl_decode_gemv_plan(M, tiles * 16, G * 128, out) == 1:
                    reached.add(tuple(out)[:4])
    compiled = {(MR, KS, CW, LM) for MR in (1, 2, 4, 8) for KS in (1, 2, 4, 8, 16) for CW in (4, 8, 16) for LM in (4, 5, 8, 10)
                if lib.tl_decode_gemv_variant_compiled(MR, KS, CW, LM) == 1}
    assert reached <= compiled, f"taken but not compiled: {sorted(reached - compiled)}"
    assert compiled <= reached, f"compiled but unreachable (prune csrc/qmv3.hip Q3_TABLE and qmv3_has_variant): {sorted(compiled - reached)}"
    assert len(compiled) == 53
