"""Weight pack / gradient unpack launches of a whole network in isolation: time per launch and a checksum of what they wrote.
usage: SRVP_PACK_TILED=<bits> python tools/pack_time.py [config] -> one JSON line (tests/test_gpu_blocks.py compares the checksums of
the kernel paths; profiles/ keeps the timings)."""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.train import train
import bench

name = sys.argv[1] if len(sys.argv) > 1 else 'bair'
cfg = bench.CONFIGS[name]
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
model.init(res_gain=cfg['res_gain'])
model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(cfg['T'], 2, cfg['ctor'][1], 64, 64).to(dev)
train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
model._flat[0].copy_(torch.randn(model._flat[0].shape, generator=torch.Generator().manual_seed(3)).to(dev) * 0.05)   # deterministic weights again (the step's atomics are not)
torch.cuda.synchronize()
pl = list(model._plans.values())[0]
st = L.stream()


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


MODE = os.environ.get('SRVP_PACK_TILES', '1') + '/' + os.environ.get('SRVP_PACK_TILED', 'default')
out = {'config': name, 'mode': MODE}
g = torch.Generator().manual_seed(7)
params, grads = model._named_tensors(), model._grads()
for nm in ('enc', 'dec'):
    net = pl[nm]
    blocks = [b for b in net.blocks if b.role in ('mfma', 'out')]
    packed = [t for b in blocks for t in (getattr(b, 'wt_f', None), getattr(b, 'wt_d', None), getattr(b, 'wt_f_s', None), getattr(b, 'wt_d_s', None)) if t is not None]
    for t in packed:
        t.zero_()
    net.pack_weights(params, st)
    torch.cuda.synchronize()
    out[nm + '_pack_digest'] = digest(packed)
    out[nm + '_pack_us'] = timeit(lambda: net.pack_weights(params, st))
    for b in blocks:
        for t in (getattr(b, 'dw', None), getattr(b, 'dw_s', None)):
            if t is not None:
                t.copy_(torch.randn(t.shape, generator=g).to(dev))
    flat_g = model._flat[1]
    flat_g.zero_()
    net.unpack_wgrads(grads, st)
    torch.cuda.synchronize()
    out[nm + '_unpack_digest'] = digest([flat_g])
    out[nm + '_unpack_us'] = timeit(lambda: net.unpack_wgrads(grads, st))
    pc, uc = net.__dict__['_pack_cache'], net.__dict__['_unpack_cache']
    out[nm + '_jobs'] = {k: (c[k][1], c[k][2]) if c[k] else None for c in (pc, uc) for k in ('tiles', 'multi')} | {'unpack_tiles': uc['tiles'][1:] if uc['tiles'] else None, 'unpack_multi': uc['multi'][1:] if uc['multi'] else None}
print(json.dumps(out))
