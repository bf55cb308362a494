"""VERDICT r4 item 8, measured then decided: is Winograd F(2x2, 3x3) worth a kernel family on the 512-channel 8x8 layers (encoder.conv.3.2 / 3.3,
decoder.conv.0.1: 3 x 302 MFLOP per frame forward, reference module/conv.py:210-223,335-338)?

F(2x2, 3x3) turns the layer into 16 independent GEMMs [tiles x Cin] x [Cin x Cout] over the transformed 4x4 input tiles (2.25x fewer MACs) plus an
input transform (4x expansion of the activation: 16 transformed values per 2x2 output tile) and an output transform.  Measured here, on the layer
at the headline size (2304 frames):
  (1) the product's direct kernel on this layer (srvp_conv_mfma, halo kernel), forward;
  (2) an UPPER BOUND for a Winograd kernel: the 16 batched GEMMs alone at the vendor library's speed (torch.bmm = hipBLASLt; a measuring stick,
      nothing in the product calls it) -- no transforms, no staging of the transformed operands;
  (3) the HBM time of materialising the transformed input and output once (what an unfused transform -> GEMM -> transform pipeline adds), from a
      device copy of the same number of bytes;
  (4) the arithmetic price: relative error of F(2x2,3x3) with the transformed operands rounded to bf16 (as an MFMA kernel would hold them)
      against the float64 convolution, next to the direct bf16 convolution's error, on a sample of frames.
One JSON line.    usage: python tools/winograd_bound.py"""
import ctypes as C
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srvp_amd import _lib as L
from srvp_amd.convnet import Block, Feat


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def winograd_f2x2_3x3(x, w, round_bf16):
    """x (N, C, H, W) float64, w (K, C, 3, 3) float64 -> (N, K, H, W), pad 1.  Transformed operands optionally rounded to bf16, fp32 accumulation."""
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    N, Cc, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)                       # (K, C, 4, 4)
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                           # (N, C, H/2, W/2, 4, 4)
    V = torch.einsum('ij,nchwjl,ml->nchwim', Bt, tiles, Bt)
    if round_bf16:
        U, V = U.to(torch.bfloat16).to(torch.float32), V.to(torch.bfloat16).to(torch.float32)
    else:
        U, V = U.float(), V.float()
    M = torch.einsum('kcim,nchwim->nkhwim', U, V)                        # fp32 accumulation over c
    Y = torch.einsum('ij,nkhwjl,ml->nkhwim', At, M.double(), At)         # (N, K, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def main():
    dev = torch.device('cuda')
    N, Cc, K, H = 2304, 512, 512, 8
    g = torch.Generator().manual_seed(3)
    # ---- (1) the product's kernel
    f0 = Feat(N, H, H, Cc, dev)
    f0.put_nhwc((torch.randn(64, H, H, Cc, generator=g) * 0.5).repeat(N // 64, 1, 1, 1).to(dev))
    spec = dict(kind='conv', key='w', bnkey='bn', cin=Cc, cout=K, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0], False, N, dev, True)
    blk._fwd = blk.fwd_descs()
    w = (torch.randn(K, Cc, 3, 3, generator=g) * 0.03)
    st = L.stream()
    blk.pack(w.to(dev), st)
    t_direct = timeit(lambda: L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st))
    flops = 2.0 * N * H * H * K * 9 * Cc
    # ---- (2) 16 batched GEMMs at vendor speed
    ntile = N * (H // 2) * (H // 2)
    Vt = torch.randn(16, ntile, Cc, device=dev, dtype=torch.bfloat16)
    Ut = torch.randn(16, Cc, K, device=dev, dtype=torch.bfloat16)
    t_bmm = timeit(lambda: torch.bmm(Vt, Ut))
    # ---- (3) transformed input written + read once, transformed output written + read once (bf16 in, fp32 out of the GEMMs)
    a = torch.empty(16 * ntile * Cc, device=dev, dtype=torch.bfloat16)
    b = torch.empty_like(a)
    o = torch.empty(16 * ntile * K, device=dev, dtype=torch.float32)
    o2 = torch.empty_like(o)
    t_copy = timeit(lambda: (b.copy_(a), o2.copy_(o)))
    # ---- (4) arithmetic, on 8 frames, float64 reference
    xs = (torch.randn(8, Cc, H, H, generator=g) * 0.5).to(torch.bfloat16).double()
    ws = w.to(torch.bfloat16).double()
    ref = F.conv2d(xs, ws, None, 1, 1)
    scale = ref.abs().max()
    e_win = ((winograd_f2x2_3x3(xs, ws, True) - ref).abs().max() / scale).item()
    e_win_exact = ((winograd_f2x2_3x3(xs, ws, False) - ref).abs().max() / scale).item()
    e_dir = ((F.conv2d(xs.float(), ws.float(), None, 1, 1).double() - ref).abs().max() / scale).item()
    print(json.dumps(dict(layer='512 -> 512 @ 8x8, 2304 frames (forward)', direct_ms=t_direct, direct_tflops=flops / t_direct / 1e9,
                          winograd_gemms_only_ms=t_bmm, winograd_gemms_tflops=16 * 2.0 * ntile * Cc * K / t_bmm / 1e9,
                          unfused_transform_traffic_ms=t_copy, upper_bound_gain_ms_per_layer=t_direct - t_bmm,
                          rel_err_winograd_bf16_transformed_operands=e_win, rel_err_winograd_fp32_operands=e_win_exact,
                          rel_err_direct_fp32_accumulation=e_dir, gate_2_pow_minus_7=2 ** -7)))


if __name__ == '__main__':
    main()
