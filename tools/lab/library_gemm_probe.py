import torch, time
torch.manual_seed(0)
dev = "cuda"
shapes = [(2048, 2560, 6144, "qkv"), (2048, 4096, 2560, "wo"), (2048, 2560, 19456, "gate_up"), (2048, 9728, 2560, "down"),
          (8192, 2560, 19456, "gate_up_8k"), (8192, 9728, 2560, "down_8k"), (512, 2560, 19456, "gate_up_512")]
for M, N, K, name in shapes:
    a = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    w = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        o = a @ w.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        o = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: M {M} N {N} K {K}: {dt*1e6:.1f} us, {2*M*N*K/dt/1e12:.0f} TFLOP/s", flush=True)
