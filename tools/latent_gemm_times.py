"""Every launch-form latent GEMM of one training step (srvp_gemm_f32, srvp_linear_wgrad_f32, srvp_colsum_f32, srvp_act_bwd_f32), replayed ALONE
with the arguments the step used: shape, microseconds, GFLOP/s -- are these kernels slow on their own, or only beside the convolutions?
    usage: [CFG=bair|kth|human|smmnist] [B=..] python tools/latent_gemm_times.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.train import train
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'bair')]
B = int(os.environ.get('B', cfg['batch'])); T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
for _ in range(3):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
NAMES = {'srvp_gemm_f32', 'srvp_linear_wgrad_f32', 'srvp_colsum_f32', 'srvp_act_bwd_f32', 'srvp_axpby_f32'}
calls = []
orig = L.call
def rec(name, *a):
    if name in NAMES:
        calls.append((name, a))
    return orig(name, *a)
import srvp_amd.latent as LT, srvp_amd.model as M
L.call = rec; LT.L.call = rec; M.L.call = rec
train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
L.call = orig
st = L.stream()
tot = 0.0
print(f'{len(calls)} launch-form latent launches per step at B={B} ({cfg["label"]})')
for name, a in calls:
    a = list(a); a[-1] = st
    for _ in range(3):
        orig(name, *a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        orig(name, *a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    if name == 'srvp_gemm_f32':
        Mm, N, K = a[9], a[10], a[11]
        desc = f'M={Mm} N={N} K={K} acc={a[13]}  {2.0 * Mm * N * K / us / 1e3:8.1f} GFLOP/s'
    elif name == 'srvp_linear_wgrad_f32':
        N, K, Mm = a[7], a[8], a[9]
        desc = f'dW[{N}][{K}] over M={Mm} rows  {2.0 * Mm * N * K / us / 1e3:8.1f} GFLOP/s'
    else:
        desc = ''
    print(f'{name:24s} {us:8.1f} us  {desc}')
print(f'sum alone: {tot / 1e3:.3f} ms per step')
