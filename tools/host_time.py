"""Host enqueue time per training step vs GPU time per step (is the step launch-bound?): B = env B (default 24), headline architecture."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd.train import train
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'bair')]
B = int(os.environ.get('B', 24)); T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
model.init(res_gain=cfg['res_gain'])
model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
for _ in range(8):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
K = 40
# time the host spends WAITING inside the step's one sync (the forward's ELBO event): what is left of the loop time is host work
waited = [0.0]
_orig = torch.cuda.Event.synchronize


def _timed(self):
    a = time.perf_counter()
    _orig(self)
    waited[0] += time.perf_counter() - a


torch.cuda.Event.synchronize = _timed
t0 = time.perf_counter()
for _ in range(K):
    train(model, optim, None, x, dev, opt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
torch.cuda.Event.synchronize = _orig
print(f'B={B}: host loop {1e3 * (t1 - t0) / K:.2f} ms/step of which {1e3 * waited[0] / K:.2f} ms waiting for the forward\'s ELBO event = '
      f'{1e3 * (t1 - t0 - waited[0]) / K:.2f} ms of host work per step; with GPU {1e3 * (t2 - t0) / K:.2f} ms/step')
