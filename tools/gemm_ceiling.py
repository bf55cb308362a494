"""Practical bf16 MFMA ceiling on this box: hipBLASLt GEMMs through torch.matmul (diagnostic; puts the conv kernels' TFLOP/s
into perspective against what the vendor library sustains, clocks and power limits included)."""
import torch
dev = torch.device('cuda')
for (M, N, K) in ((8192, 8192, 8192), (16384, 8192, 4608), (147456, 512, 4608), (589824, 256, 2304)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'GEMM {M}x{N}x{K}: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s')
