import sys, os, time, json, random
ROOT="/root/repo"
for p in ("tiny-llm_amd", "tiny-llm_amd/extensions_hip"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
from tiny_llm_hip.engine import DecodeEngine
from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3
cfg = dict(QWEN3_CONFIGS["qwen3-4b"])
model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
rng = random.Random(1)
def leg(tag, long_first):
    if long_first:
        eng = DecodeEngine(model, page_size=128, num_pages=300, max_batch=1, max_prefill_rows=2048)
        eng.begin(0); eng.prefill(0, [rng.randrange(256, 100000) for _ in range(32768)], chunk=2048); eng.decode(8, batch=1); eng.synchronize(); eng.release(0); eng.close()
    B, plen, steps, page = 64, 128, 16, 128
    per_seq = (plen + 4 * steps + 8 + 2 * page) // page + 1
    eng = DecodeEngine(model, page_size=page, num_pages=per_seq * B + 2, max_batch=B, max_prefill_rows=128)
    for slot in range(B):
        eng.begin(slot); eng.prefill(slot, [rng.randrange(256, 100000) for _ in range(plen)], chunk=128)
    eng.decode(4, batch=B); eng.synchronize(); torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); eng.decode(steps, batch=B); eng.synchronize(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) / steps * 1e3, 4))
    print(tag, ts, eng.replay_route(), {k: v for k, v in eng.stats().items() if k in ("graph_captures", "graph_replays", "aql_steps", "decode_steps")}, flush=True)
    eng.close()
leg("plain", False); leg("after_32k", True); leg("plain_again", False)
os.environ["TL_AQL"] = "0"; leg("hipgraph", False)
