"""Collectives of one training step at 24 sequences with SRVP_FORCE_COLLECTIVES=1 on one rank, counted by hooks on Sync.allreduce_stats and the
native gradient communicator: 42 statistics all-reduces (21 BatchNorm layers x forward / backward) + 3 gradient slices for the VGG recipes.
    usage: python tools/count_collectives.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ['SRVP_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import srvp_amd
from srvp_amd import distributed as D
from srvp_amd.train import train
import bench
cfg = bench.CONFIGS['bair']
torch.cuda.set_device(0)
sync = D.init_process_group()
torch.manual_seed(1)
m = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); m.init(cfg['res_gain']); m.cuda().train()
dp = D.DataParallel(m, sync)
optim = srvp_amd.FusedAdam(m, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=2, obs_scale=0.71, beta_y=1.0, beta_z=1.0, l2_res=1.0))
x = torch.rand(12, 24, 3, 64, 64).cuda()
cnt = dict(stats=0, grads=0)
o1 = D.Sync.allreduce_stats
def a1(self, t, count, site=None):
    cnt['stats'] += 1
    return o1(self, t, count, site)
D.Sync.allreduce_stats = a1
if sync.native_grads is not None:
    o2 = sync.native_grads.allreduce
    def a2(t):
        cnt['grads'] += 1
        return o2(t)
    sync.native_grads.allreduce = a2
for i in range(3):
    cnt['stats'] = cnt['grads'] = 0
    train(dp, optim, None, x, torch.device('cuda'), opt)
torch.cuda.synchronize()
print('collectives per step:', cnt)
