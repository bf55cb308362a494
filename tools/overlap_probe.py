"""How much does an HBM-bound BatchNorm pass gain from running beside an MFMA-bound convolution of the next layer?  (sizing the idea of
pipelining half batches through the forward; DESIGN.md section 3: nothing -- concurrent = serial under the power cap.)
usage: python tools/overlap_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd, bench
from srvp_amd import _lib as L
from srvp_amd.train import train
cfg = bench.CONFIGS['bair']; B = 192; T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, 3, 64, 64).to(dev)
for _ in range(2): train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
pl = list(model._plans.values())[0]
enc = pl['enc']
params = model._named_tensors()
s2 = torch.cuda.Stream()
def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for bi_conv, bi_bn in ((3, 2), (5, 4), (8, 7), (2, 1), (1, 0)):
    cb, bb = enc.blocks[bi_conv], enc.blocks[bi_bn]
    def conv():
        for d in cb._fwd: L.call('srvp_conv_mfma', C.byref(d), L.stream())
    def bn():
        enc._bn_forward(bb, params, L.stream(), None, keep=pl['keep'])
    def both():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            bn()
            done = torch.cuda.Event(); done.record()
        conv()
        torch.cuda.current_stream().wait_event(done)
    t1, t2, t12 = timeit(conv), timeit(bn), timeit(both)
    print(f'conv enc{bi_conv:02d} {t1:.3f} ms  bn_act enc{bi_bn:02d} {t2:.3f} ms  serial {t1 + t2:.3f}  concurrent {t12:.3f}  gain {t1 + t2 - t12:.3f}')
