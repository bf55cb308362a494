"""Per-layer timing of the MFMA conv / wgrad launches at the headline config (scratch diagnostic)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.train import train
import bench

cfg = bench.CONFIGS['bair']
B = int(os.environ.get('B', 192))
T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
model.init(res_gain=cfg['res_gain'])
model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, 3, 64, 64).to(dev)
for _ in range(2):
    train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
pl = list(model._plans.values())[0]
st = L.stream()


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = dict(fwd=0., dg=0., wg=0.)
print(f'{"layer":28s} {"M":>9s} {"K":>6s} {"Cout":>5s} | fwd ms   TF | dgrad ms  TF | wgrad ms  TF')
for name, net in (('enc', pl['enc']), ('dec', pl['dec'])):
    for i, blk in enumerate(net.blocks):
        if blk.role == 'in':
            continue
        if os.environ.get('ONLY') and f'{name}{i:02d}' not in os.environ['ONLY'].split(','):
            continue
        row = f'{name}{i:02d} {blk.geom:6s}{"*" if blk.split else " "} {blk.Hin:2d}->{blk.OH:2d} c{blk.ctot}->{blk.cout}'
        res = []
        for kind, descs in (('fwd', blk._fwd), ('dg', blk._dg)):
            ms, fl = 0., 0.
            for d in descs:
                ms += timeit(lambda: L.call('srvp_conv_mfma', C.byref(d), st))
                fl += 2.0 * d.N * d.OH * d.OW * d.Cout * d.ntaps * (d.C0 + d.C1)
            res.append((ms, fl / ms / 1e9))
            tot[kind] += ms
        wg = blk._wg if isinstance(blk._wg, list) else [blk._wg]
        ms, fl = 0., 0.
        for d in wg:
            ms += timeit(lambda: L.call('srvp_wgrad_mfma', C.byref(d), st))
            fl += 2.0 * d.N * d.OH * d.OW * d.Cout * d.ntaps * (d.C0 + d.C1)
        res.append((ms, fl / ms / 1e9))
        tot['wg'] += ms
        d0 = blk._fwd[-1]
        print(f'{row:28s} {d0.N * d0.OH * d0.OW:9d} {d0.ntaps * (d0.C0 + d0.C1):6d} {d0.Cout:5d} | ' +
              ' | '.join(f'{m:6.3f} {t:5.0f}' for m, t in res))
print('totals ms', tot)
