"""What-if timings of the sub-pixel forward launches (scratch)."""
import ctypes as C, os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd, bench
from srvp_amd import _lib as L
from srvp_amd.train import train
cfg = bench.CONFIGS['bair']; B = 192; T = cfg['T']
dev = torch.device('cuda', 0); torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, 3, 64, 64).to(dev)
for _ in range(2): train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
pl = list(model._plans.values())[0]; st = L.stream()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def clone(ds):
    arr = (L.ConvDesc * len(ds))()
    for i, d in enumerate(ds): C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(L.ConvDesc))
    return arr
for i, blk in enumerate(pl['dec'].blocks):
    if not getattr(blk, 'subpix', False): continue
    ds = blk._fwd[-4:]
    fl = sum(2.0 * d.N * d.OH * d.OW * d.Cout * d.ntaps * d.C0 for d in ds)
    res = {}
    def run(arr): return timeit(lambda: L.call('srvp_conv_mfma_multi', arr, 4, st))
    a = clone(ds); res['base'] = run(a)
    a = clone(ds)
    for d in a: d.stats = None
    res['nostats'] = run(a)
    a = clone(ds)
    for d in a: d.add_f32 = None; d.add_mod = 0
    res['noS'] = run(a)
    a = clone(ds)
    for d in a: d.add_f32 = None; d.add_mod = 0; d.stats = None
    res['noS_nostats'] = run(a)
    if blk.split:
        res['Sconv'] = timeit(lambda: L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st))
    print(f'dec{i:02d} {blk.Hin}->{blk.OH} C0={ds[0].C0} cout={blk.cout} GF={fl/1e9:.0f} ' + ' '.join(f'{k}={v:.3f}ms({fl/v/1e9:.0f}TF)' if k != 'Sconv' else f'{k}={v:.3f}ms' for k, v in res.items()))
