"""cProfile of the host side of the training step (where do the ~4 ms of Python per step go?): B = env B (default 24), 60 steps.
    usage: [B=24] python tools/host_profile.py [n_lines]"""
import cProfile, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd.train import train
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'bair')]
B = int(os.environ.get('B', 24)); T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
for _ in range(10):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
K = 60
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    train(model, optim, None, x, dev, opt)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
out = s.getvalue()
print(f'(per step = totals / {K})')
print(out[:12000])
