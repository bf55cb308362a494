"""Feasibility probe (round 6): the whole training step (zero_grad + forward + ELBO + backward + Adam) captured into ONE hipGraph through
torch.cuda.graph and replayed, against the eager step on the same box.  B = env B (default 24), CFG = env CFG (default bair)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd.train import train, fused_step
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
    loss = train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K):
    loss = train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / K * 1e3
print(f'B={B} eager {eager:.3f} ms/step loss {loss[0]:.2f}', flush=True)
nt_inf, ny, nz = cfg['ctor'][7], cfg['ctor'][4], cfg['ctor'][5]
tape = dict(t_w=torch.stack([torch.randperm(T)[:nt_inf] for _ in range(B)], 1), eps_y0=torch.randn(B, ny, device=dev), eps_z=torch.randn(T - 1, B, nz, device=dev))
if cfg['ctor'][6]:
    tape['t_skip'] = torch.randint(T, (B,))
for _ in range(2):
    optim.zero_grad(); acc = fused_step(model, x, opt, tape=tape); optim.step()
torch.cuda.synchronize()
print('eager with tape: acc', acc.tolist(), flush=True)
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.graph(g, stream=st, capture_error_mode=os.environ.get('MODE', 'thread_local')):
        optim.zero_grad(); acc = fused_step(model, x, opt, tape=tape); optim.step()
except Exception as e:
    print('CAPTURE FAILED:', repr(e)[:2000], flush=True)
    raise
torch.cuda.synchronize()
print('captured', flush=True)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print('replayed: acc', acc.tolist(), flush=True)
t0 = time.perf_counter()
for _ in range(K):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'B={B} graph replay {(t2 - t0) / K * 1e3:.3f} ms/step (host {(t1 - t0) / K * 1e3:.3f} ms/step) vs eager {eager:.3f}', flush=True)
