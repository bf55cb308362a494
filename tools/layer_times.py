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

cfg = bench.CONFIGS[os.environ.get('CFG', 'bair')]
B = int(os.environ.get('B', cfg['batch']))
T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
model.init(res_gain=cfg['res_gain'])
model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
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


tot = dict(fwd=0., dg=0., wg=0., skip=0.)
print('# launch forms AS THE STEP ISSUES THEM (round 6): the four sub-pixel / transposed phases of a block as ONE grid (srvp_conv_mfma_multi), the hoisted')
print('# skip half conv_s(skip) and its data gradient (once per SAMPLE, on the second stream in the step) in their own columns; TF = executed FLOPs / time.')
print('# (the 64x64 layers run on the streaming kernels from 96 frames up -- conv_stream64 / conv_stream_sub64 -- through the same entry point)')
print(f'{"layer":28s} {"M":>9s} {"K":>6s} {"Cout":>5s} | fwd ms   TF | dgrad ms  TF | wgrad ms  TF | skip fwd+dgrad ms')
fl_of = lambda ds: sum(2.0 * d.N * d.OH * d.OW * d.Cout * d.ntaps * (d.C0 + d.C1) for d in ds)


def launch(ds):
    """the launch form model / convnet use for this descriptor list"""
    if len(ds) == 4:
        arr = (L.ConvDesc * 4)(*ds)
        return lambda: L.call('srvp_conv_mfma_multi', arr, 4, st)
    return lambda: [L.call('srvp_conv_mfma', C.byref(d), st) for d in ds]


for name, net in (('enc', pl['enc']), ('dec', pl['dec'])):
    for i, blk in enumerate(net.blocks):
        if blk.role == 'in':
            continue
        if os.environ.get('ONLY') and f'{name}{i:02d}' not in os.environ['ONLY'].split(','):
            continue
        row = f'{name}{i:02d} {blk.geom:6s}{"*" if blk.split else " "} {blk.Hin:2d}->{blk.OH:2d} c{blk.ctot}->{blk.cout}'
        res = []
        fwd_main = list(blk._fwd[1:]) if blk.split else list(blk._fwd)
        dg_main = list(blk._dg[:1]) if blk.split else list(blk._dg)
        skip = ([blk._fwd[0]] if blk.split else []) + (list(blk._dg[1:]) if blk.split else [])
        for kind, descs in (('fwd', fwd_main), ('dg', dg_main)):
            ms = timeit(launch(descs))
            res.append((ms, fl_of(descs) / ms / 1e9))
            tot[kind] += ms
        wg = blk._wg if isinstance(blk._wg, list) else [blk._wg]
        ms = timeit(lambda: [L.call('srvp_wgrad_mfma', C.byref(d), st) for d in wg])
        fl = sum(2.0 * d.N * d.OH * d.OW * d.Cout * d.ntaps * (d.C0 + d.C1) for d in wg)
        res.append((ms, fl / ms / 1e9))
        tot['wg'] += ms
        sk = timeit(launch(skip)) if skip else 0.
        tot['skip'] += sk
        d0 = fwd_main[-1]
        print(f'{row:28s} {d0.N * d0.OH * d0.OW * len(fwd_main):9d} {d0.ntaps * (d0.C0 + d0.C1):6d} {d0.Cout:5d} | ' +
              ' | '.join(f'{m:6.3f} {t:5.0f}' for m, t in res) + (f' | {sk:6.3f}' if skip else ' |'))
print('totals ms', {k: round(v, 3) for k, v in tot.items()})
