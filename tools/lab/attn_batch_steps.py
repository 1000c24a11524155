"""Lab: the decode attention launch (tl_decode_attention_fused: rope rows + attention [+ merge]) across sequence counts at ~140 tokens of context --
where does its time step?  usage: python tools/lab/attn_batch_steps.py [ctx]"""
import sys, time
sys.path.insert(0, "tiny-llm_amd/extensions_hip"); sys.path.insert(0, "tiny-llm_amd")
import torch
import tiny_llm_ext_hip as ext
ext.load_library(".")
HQ, HKV, D, PAGE = 32, 8, 128, 128
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 140
torch.manual_seed(0)
for B in (8, 16, 24, 30, 31, 32, 33, 34, 40, 48, 56, 64):
    need = (ctx + 1 + PAGE - 1) // PAGE
    P = need * B + 2
    table = torch.arange(need * B, dtype=torch.int32, device="cuda").reshape(B, need)
    kp = torch.randn(P, HKV, PAGE, D, device="cuda").to(torch.bfloat16)
    vp = torch.randn(P, HKV, PAGE, D, device="cuda").to(torch.bfloat16)
    qkv = torch.randn(B, (HQ + 2 * HKV) * D, device="cuda").to(torch.bfloat16)
    qn = torch.ones(D, device="cuda", dtype=torch.bfloat16); kn = torch.ones(D, device="cuda", dtype=torch.bfloat16)
    cl = torch.full((B,), ctx, dtype=torch.int32, device="cuda")
    def run():
        return ext.decode_attention_fused(qkv, qn, kn, kp, vp, table, cl, num_heads=HQ, num_kv_heads=HKV, rope_theta=1e6, eps=1e-6, max_context=ctx)
    out, info = run(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    ev0.record()
    for _ in range(n): run()
    ev1.record(); torch.cuda.synchronize()
    print(f"B={B:3d} ctx={ctx}: {ev0.elapsed_time(ev1) * 1e3 / n:7.2f} us per call (rope rows + attention{' + merge' if info['launches'] > 1 else ''})  plan {info}")
