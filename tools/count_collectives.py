"""Collectives of one training step at 24 sequences with SRVP_FORCE_COLLECTIVES=1 on one rank, counted by hooks on Sync.allreduce_stats and
Sync.reduce_slice: 42 statistics all-reduces (21 BatchNorm layers x forward / backward) + the gradient slices for the VGG recipes (round 6:
decoder tail, decoder head, encoder deep stages, latent networks under the backward; the encoder's first stages at the step's end).  Prints
one JSON line: counts, the slices in issue order (elements of the flat gradient buffer), how many were issued after the last weight-gradient
launch, and whether they tile the buffer exactly once.
    usage: python tools/count_collectives.py [config] [batch]"""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ['SRVP_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import srvp_amd
from srvp_amd import distributed as D, _lib as L
from srvp_amd.train import train
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'bair']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
torch.cuda.set_device(0)
sync = D.init_process_group()
torch.manual_seed(1)
m = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); m.init(cfg['res_gain']); m.cuda().train()
dp = D.DataParallel(m, sync)
optim = srvp_amd.FusedAdam(m, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(cfg['T'], B, cfg['ctor'][1], 64, 64).cuda()
cnt = dict(stats=0, grads=0)
events = []                                   # ('wgrad', name) / ('slice', lo, hi) in HOST issue order
o1 = D.Sync.allreduce_stats
def a1(self, t, count, site=None):
    cnt['stats'] += 1
    return o1(self, t, count, site)
D.Sync.allreduce_stats = a1
o2 = D.Sync.reduce_slice
def a2(self, model, lo, hi):
    cnt['grads'] += 1
    events.append(('slice', lo, hi))
    return o2(self, model, lo, hi)
D.Sync.reduce_slice = a2
o3 = L.call
WG = {'srvp_wgrad_mfma', 'srvp_conv_in_wgrad', 'srvp_conv_in_wgrad_bn', 'srvp_linear_wgrad_f32'}
def a3(name, *a):
    if name in WG:
        events.append(('wgrad', name))
    return o3(name, *a)
L.call = a3
for mod in (srvp_amd.convnet, srvp_amd.latent, srvp_amd.model):
    mod.L.call = a3
for i in range(3):
    cnt['stats'] = cnt['grads'] = 0
    events.clear()
    train(dp, optim, None, x, torch.device('cuda'), opt)
torch.cuda.synchronize()
slices = [(e[1], e[2]) for e in events if e[0] == 'slice']
last_wg = max(i for i, e in enumerate(events) if e[0] == 'wgrad')
after = [(e[1], e[2]) for e in events[last_wg + 1:] if e[0] == 'slice']
total = m._flat[1].numel() if sum(p.numel() for p in m.parameters()) == m._flat[1].numel() else sum(p.numel() for p in m.parameters())
srt = sorted(slices)
tiles = bool(srt) and srt[0][0] == 0 and all(a[1] == b[0] for a, b in zip(srt, srt[1:])) and srt[-1][1] == sum(p.numel() for p in m.parameters())
gnorm = float(m._flat[1][:sum(p.numel() for p in m.parameters())].double().norm().item())
print(json.dumps(dict(grad_norm_after_exchange=gnorm, config=sys.argv[1] if len(sys.argv) > 1 else 'bair', batch=B, statistics_allreduces=cnt['stats'], gradient_allreduces=cnt['grads'],
                      slices_in_issue_order=slices, slice_mbytes=[round((b - a) * 4 / 1e6, 2) for a, b in slices],
                      after_last_weight_gradient=after, after_last_weight_gradient_mbytes=[round((b - a) * 4 / 1e6, 2) for a, b in after],
                      tiles_buffer_exactly_once=tiles, transport=sync.describe())))
