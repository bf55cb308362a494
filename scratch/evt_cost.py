import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import srvp_amd, bench
from srvp_amd import _lib as L
from srvp_amd.train import train
cfg = bench.CONFIGS[os.environ.get('CFG', 'smmnist')]; B = cfg['batch']; T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
for _ in range(3): train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
prof = {}
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if i % 4 == 0:
        L.PROFILE, L.PROFILE_ONLY = prof, {'srvp_conv_mfma', 'srvp_conv_mfma_multi', 'srvp_wgrad_mfma'}
    train(model, optim, None, x, dev, opt)
    t1 = time.perf_counter()
    L.PROFILE, L.PROFILE_ONLY = None, None
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(i, 'instrumented' if i % 4 == 0 else '', 'host %.2f ms  total %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), {k: len(v) for k, v in prof.items()})
for mode in ('plain', 'events', 'plain', 'events'):
    prof = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        if mode == 'events' and i % 4 == 0:
            L.PROFILE, L.PROFILE_ONLY = prof, {'srvp_conv_mfma', 'srvp_conv_mfma_multi', 'srvp_wgrad_mfma'}
        train(model, optim, None, x, dev, opt)
        L.PROFILE, L.PROFILE_ONLY = None, None
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, 'ms/step %.2f' % ((t2 - t0) * 100))
