import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from golden_util import Fixture
from oracle import srvp_oracle as O
import srvp_amd
name = sys.argv[1] if len(sys.argv)>1 else 'tiny_vgg_nc3_skip0_e1'
fx = Fixture(name)
m = srvp_amd.StochasticLatentResidualVideoPredictor(*fx.meta['ctor']); m.load_state_dict(fx.state('sd0')); m=m.cuda().train()
x = fx.t('x').cuda(); tape = fx.tape()
outs = m._forward_impl(x, x.shape[0], fx.meta['n_euler'], tape, training=True)
pl = m._last_plan
# oracle with recording
rec = []
orig = O._conv_block_bf16
def recblock(h, sd, spec, training):
    r = orig(h, sd, spec, training); rec.append((spec['key'], r.detach().clone())); return r
O._conv_block_bf16 = recblock; O.PRECISION='bf16'
sd = fx.state('sd0')
with torch.no_grad():
    o = O.forward(sd, fx.cfg, fx.t('x'), x.shape[0], fx.meta['n_euler'], tape, True)
O.PRECISION='fp32'
blocks = pl['enc'].blocks + pl['dec'].blocks
for (key, ref), blk in zip(rec, blocks):
    if blk.out is not None:
        got = blk.out.interior().permute(0,3,1,2).float().cpu()
    elif blk.out_f32 is not None:
        got = blk.out_f32[:, :blk.cout_r].float().cpu().view(ref.shape)
    else:
        got = pl['dec'].x_out.cpu()
    d = (got-ref).abs()
    print(f"{key:28s} shape {tuple(ref.shape)} maxabs {d.max():.3e} refmax {ref.abs().max():.2e} frac_mismatch {(d>0).float().mean():.4f} rel_l2 {(d.norm()/ref.norm()):.2e}")
