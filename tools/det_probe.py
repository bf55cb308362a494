"""Which parameter gradients differ between two identical fp32 deterministic-mode steps (diagnostic for model.set_deterministic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import srvp_amd
from srvp_amd.train import fused_step
from make_golden import synth_video
T, B, NE = 8, 8, 2
CTOR = (64, 1, 16, 32, 8, 8, True, 2, 32, 3, 64, 4, 'vgg') if len(sys.argv) < 2 else (64, 1, 64, 128, 20, 20, False, 5, 256, 3, 512, 4, 'dcgan')
dev = torch.device('cuda')
x = torch.from_numpy(synth_video(T, B, 1, seed=11)).to(dev)
g = torch.Generator().manual_seed(99)
tape = dict(t_skip=torch.randint(T, (B,), generator=g), t_w=torch.stack([torch.randperm(T, generator=g)[:CTOR[7]] for _ in range(B)], 1),
            eps_y0=torch.randn(B, CTOR[4], generator=g), eps_z=torch.randn(T - 1, B, CTOR[5], generator=g))
opt = srvp_amd.DotDict(dict(n_euler_steps=NE, obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0))
outs = []
for rep in range(3):
    torch.manual_seed(4)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*CTOR)
    m.init(1.2)
    m.to(dev).train().set_precision('fp32').set_deterministic(True)
    optim = srvp_amd.FusedAdam(m, lr=3e-4)
    optim.zero_grad()
    acc = fused_step(m, x, opt, tape=tape)
    torch.cuda.synchronize()
    outs.append((acc.cpu().clone(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()},
                 {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if 'running' in k}))
for r in (1, 2):
    print('run 0 vs', r, 'elbo acc equal:', torch.equal(outs[0][0], outs[r][0]))
    bad = [(k, (outs[0][1][k] - outs[r][1][k]).abs().max().item()) for k in outs[0][1] if not torch.equal(outs[0][1][k], outs[r][1][k])]
    print('  differing gradients:', len(bad), bad[:40])
    badb = [k for k in outs[0][2] if not torch.equal(outs[0][2][k], outs[r][2][k])]
    print('  differing running statistics:', badb[:10])
