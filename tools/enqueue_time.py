"""CPU enqueue time vs GPU time of one training step (scratch diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd, bench
from srvp_amd import _lib as L
from srvp_amd.train import train, fused_step

cfg = bench.CONFIGS[os.environ.get('CFG', 'bair')]; B = int(os.environ.get('B', cfg['batch'])); T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
for _ in range(3):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
n = 0
orig = L.call
def counting(*a):
    global n
    n += 1
    return orig(*a)
L.call = counting
import srvp_amd.convnet, srvp_amd.latent, srvp_amd.model, srvp_amd.train, srvp_amd.optim
enq, tot = [], []
for _ in range(5):
    torch.cuda.synchronize(); n = 0
    t0 = time.perf_counter()
    optim.zero_grad(); acc = fused_step(model, x, opt); optim.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print('L.call per step', n, 'enqueue ms', [round(v, 1) for v in enq], 'total ms', [round(v, 1) for v in tot])
t0 = time.perf_counter()
for _ in range(10):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
print('train() loop ms/step', (time.perf_counter() - t0) * 100)
