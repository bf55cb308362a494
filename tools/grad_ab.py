"""Same training step twice with one environment switch flipped (child processes): relative difference of the loss and of every parameter gradient.
    usage: python tools/grad_ab.py smmnist|bair|kth|human SWITCH=VALUE [batch]      (e.g. smmnist SRVP_UP_FUSED_REDUCE=0)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cfgname, B):
    sys.path.insert(0, ROOT)
    import torch
    import srvp_amd
    from bench import CONFIGS
    from srvp_amd.train import fused_step
    cfg = CONFIGS[cfgname]
    torch.manual_seed(1)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
    m.init(res_gain=cfg['res_gain'])
    m.cuda().train()
    opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
    T = cfg['T']
    g = torch.Generator().manual_seed(5)
    x = torch.rand(T, B, cfg['ctor'][1], 64, 64, generator=g).cuda()
    ny, nz, nt_inf, skip = cfg['ctor'][4], cfg['ctor'][5], cfg['ctor'][7], cfg['ctor'][6]
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1), eps_y0=torch.randn(B, ny, generator=g),
                eps_z=torch.randn(T - 1, B, nz, generator=g))
    if skip:
        tape['t_skip'] = torch.randint(T, (B,), generator=g)
    optim = srvp_amd.FusedAdam(m, lr=3e-4)
    optim.zero_grad()
    acc = fused_step(m, x, opt, tape=tape)
    torch.cuda.synchronize()
    out = dict(acc=acc.cpu().tolist(), grads={k: p.grad.detach().float().cpu() for k, p in m.named_parameters()})
    torch.save(out, sys.argv[-1])


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    import torch
    cfgname, switch = sys.argv[1], sys.argv[2]
    B = sys.argv[3] if len(sys.argv) > 3 else '16'
    k, v = switch.split('=')
    res = []
    for env in ({}, {k: v}):
        path = f'/tmp/grad_ab_{len(res)}.pt'
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', cfgname, B, path], env=dict(os.environ, **env))
        res.append(torch.load(path, weights_only=False))
    a, b = res
    worst = max(((a['grads'][n] - b['grads'][n]).norm() / (a['grads'][n].norm() + 1e-30)).item() for n in a['grads'])
    which = max(a['grads'], key=lambda n: ((a['grads'][n] - b['grads'][n]).norm() / (a['grads'][n].norm() + 1e-30)).item())
    print(json.dumps(dict(config=cfgname, switch=switch, batch=int(B), elbo_terms=[a['acc'], b['acc']], worst_grad_rel_diff=worst, tensor=which)))
