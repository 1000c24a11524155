import sys, time
sys.path.insert(0, "tiny-llm_amd/extensions_hip"); sys.path.insert(0, "tiny-llm_amd")
import torch, numpy as np
import tiny_llm_ext_hip as ext
ext.load_library(".")
torch.manual_seed(0)
HQ, HKV, D, PAGE = 32, 8, 128, 128
def case(L, ctx, B=1):
    need = (ctx + PAGE - 1) // PAGE
    P = need * B + 3
    perm = torch.randperm(P)[: need * B].to(torch.int32).reshape(B, need).cuda()
    kp = torch.randn(P, HKV, PAGE, D, device="cuda").to(torch.bfloat16)
    vp = torch.randn(P, HKV, PAGE, D, device="cuda").to(torch.bfloat16)
    q = torch.randn(B * HQ, L, D, device="cuda").to(torch.bfloat16)
    cl = torch.full((B,), ctx, dtype=torch.int32, device="cuda")
    return q, kp, vp, perm, cl
def run(c, ctx):
    q, kp, vp, bt, cl = c
    return ext.paged_attention(q, kp, vp, bt, cl, D ** -0.5, True, num_kv_heads=HKV, num_heads=HQ, max_context_hint=ctx)
for L, ctx, B in [(64, 64, 1), (100, 300, 2), (512, 8192, 1), (2048, 8192, 1), (4096, 8192, 1), (4096, 32768, 1), (33, 2000, 3), (2048, 2048, 1)]:
    c = case(L, ctx, B)
    outs = {}
    for nw in (4, 8):
        ext.paged_attention_waves(nw)
        o = run(c, ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): o = run(c, ctx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        outs[nw] = (o, dt)
    same = torch.equal(outs[4][0], outs[8][0])
    flop = 4.0 * HQ * D * B * (L * (ctx - L) + L * L / 2)
    print(f"L={L} ctx={ctx} B={B}: 4 waves {outs[4][1]*1e6:8.1f} us ({flop/outs[4][1]/1e12:6.1f} TF)  8 waves {outs[8][1]*1e6:8.1f} us ({flop/outs[8][1]/1e12:6.1f} TF)  identical={same} nan={bool(torch.isnan(outs[8][0].float()).any())}")
