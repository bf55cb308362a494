"""Untraced placement of the second stream's work inside a training step (HIP events only, no profiler: a profiler slows the host enough
to move it): when do the decoder's weight gradients start / end relative to the decoder chain's end, the latent backward and the step's
end?  B = env B (default 24), headline architecture."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd.train import train
from srvp_amd import convnet
import bench
cfg = bench.CONFIGS['bair']
B = int(os.environ.get('B', 24)); T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
model.init(res_gain=cfg['res_gain'])
model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, 3, 64, 64).to(dev)
for _ in range(6):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
marks = []
def ev(name, stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream) if stream is not None else e.record()
    marks.append((name, e))
orig_dw = convnet.DecoderNet.deferred_wgrads
def dw(self, grads, st):
    ev('side: decoder wgrads start', torch.cuda.current_stream())
    r = orig_dw(self, grads, st)
    ev('side: decoder wgrads end', torch.cuda.current_stream())
    return r
convnet.DecoderNet.deferred_wgrads = dw
orig_db = convnet.DecoderNet.backward
def db(self, *a, **k):
    ev('main: decoder backward start')
    r = orig_db(self, *a, **k)
    ev('main: decoder backward end')
    return r
convnet.DecoderNet.backward = db
orig_eb = convnet.EncoderNet.backward
def eb(self, *a, **k):
    ev('main: encoder backward start')
    r = orig_eb(self, *a, **k)
    ev('main: encoder backward end (main-stream launches)')
    return r
convnet.EncoderNet.backward = eb
res = {}
for it in range(5):
    marks.clear()
    ev('step start')
    train(model, optim, None, x, dev, opt)
    ev('step end (after adam)')
    torch.cuda.synchronize()
    t0 = marks[0][1]
    for name, e in marks[1:]:
        res.setdefault(name, []).append(t0.elapsed_time(e))
print(f'B={B}: ms after the step start (median of 5 steps)')
for name, v in res.items():
    v = sorted(v)
    print(f'  {v[len(v) // 2]:8.3f}  {name}')
