import sys, torch
sys.path.insert(0, '/root/repo')
import srvp_amd
from oracle import srvp_oracle as O
from srvp_amd.train import elbo_terms_and_grads
archi, nc, skipco, T, B, ne, nt_inf = 'vgg', 3, True, 3, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 2
ctor = (64, nc, 8, 16, 4, 5, skipco, nt_inf, 16, 3, 32, 3, archi)
torch.manual_seed(3)
m = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor); m.init(1.2)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(5)
x = torch.rand(T, B, nc, 64, 64, generator=g)
tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
            eps_y0=torch.randn(B, 4, generator=g), eps_z=torch.randn(T - 1, B, 5, generator=g))
tape['t_skip'] = torch.randint(T, (B,), generator=g)
hp = dict(obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0)
f64 = lambda v: v.double() if v.is_floating_point() else v.clone()
scal, outs_ref, grads_ref = O.train_step({k: f64(v) for k, v in sd.items()}, O.make_cfg(*ctor), x.double(), ne, {k: f64(v) for k, v in tape.items()}, hp)
scal32, _, grads32 = O.train_step({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, ne, tape, hp)
m = m.cuda().train().set_precision('fp32')
m.flatten_parameters_(); m._grads(); m._flat[1].zero_()
xg = x.cuda()
outs = m._forward_impl(xg, T, ne, tape, training=True)
opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
acc, gr = elbo_terms_and_grads(m, xg, outs, opt)
m._backward_impl(gr[0], None, None, gr[1], gr[2], gr[3], gr[4])
for k, p in m.named_parameters():
    if k.startswith(('encoder', 'decoder')): continue
    r = grads_ref[k]
    e = (p.grad.double().cpu() - r).norm().item() / (r.norm().item() + 1e-30)
    e32 = (grads32[k].double() - r).norm().item() / (r.norm().item() + 1e-30)
    print(f'{k:28s} hip-vs-64 {e:.2e}   oracle32-vs-64 {e32:.2e}  norm {r.norm().item():.3e}')
