"""North-star ELBO gate (1e-4 relative vs the fp32 CPU oracle) of the PRODUCTION precision at states TRAINING visits.

The 400-frame gates of tests/test_gpu_parity_gate.py are taken at the INITIAL weights, where the KTH / Human3.6M recipes are
ill-conditioned (untrained residual MLP at res_gain 1.2).  Here the full-width recipe is trained for N Adam steps in fp32 parity mode on
moving-blob videos (tests/make_golden.py::synth_video, a fresh seed per step), and every `every` steps the SAME held-out 400-frame batch
and noise tape go through (a) the HIP path in bf16, (b) the HIP path in fp32 mode, and -- at the first and last checkpoint -- (c) the
fp32 CPU oracle.  Prints one JSON line per checkpoint.   usage: python tools/gate_after_training.py kth|human [steps] [every] [seed]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import srvp_amd
from make_golden import synth_video
from srvp_amd.train import train, elbo_terms_and_grads

RECIPES = {'kth': dict(nc=1, T=20, B=20), 'human': dict(nc=3, T=16, B=26)}


def run(name, steps=300, every=100, oracle_at=None, log=print, seed=0, grads_at=()):
    from oracle import srvp_oracle as O
    r = RECIPES[name]
    nc, T, B, ne = r['nc'], r['T'], r['B'], 2
    ctor = (64, nc, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg')
    hp = dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    dev = torch.device('cuda')
    model.to(dev).train().set_precision('fp32')
    optim = srvp_amd.FusedAdam(model, lr=3e-4)
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    # seed = 0 is the run of round 4 (profiles/r04_gate_after_training_*.jsonl); another seed changes the held-out batch, its noise tape AND the
    # training videos (the initial weights stay the recipe's torch.manual_seed(1))
    g = torch.Generator().manual_seed(321 + 1000 * seed)
    x_eval = torch.from_numpy(synth_video(T, B, nc, seed=77 + 1000 * seed))
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:3] for _ in range(B)], 1), eps_y0=torch.randn(B, 50, generator=g),
                eps_z=torch.randn(T - 1, B, 50, generator=g), t_skip=torch.randint(T, (B,), generator=g))
    xg = x_eval.to(dev)
    oracle_at = set(oracle_at if oracle_at is not None else (0, steps))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(cores, 16)))

    def hip(precision):
        model.set_precision(precision)
        with torch.no_grad():
            outs = model._forward_impl(xg, T, ne, tape, training=True)
            acc, _ = elbo_terms_and_grads(model, xg, outs, opt, want_grads=False)
        nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
        return (nll + kl_y0 + kl_z + l2) / B, nll / B, kl_z / B

    def oracle():
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            outs = O.forward(sd, O.make_cfg(*ctor), x_eval, T, ne, tape, training=True)
            res = O.elbo(x_eval, outs, hp['obs_scale'], hp['beta_y'], hp['beta_z'], hp['l2_res'])
        return float(res['loss'])

    def grad_gate():
        """(round 6) every parameter gradient of the PRODUCTION (bf16) path on the held-out batch against the fp32 CPU
        oracle's autograd at the same weights: per tensor cosine and norm ratio.  The model's flat gradient buffer / Adam state are left
        as they were (the training run goes on unchanged)."""
        from srvp_amd.train import fused_step
        model.set_precision('bf16')
        model.flatten_parameters_()
        saved = model._flat[1].clone()
        model._flat[1].zero_()
        for p_, g_ in zip(model.parameters(), model._flat[3]):
            p_.grad = g_
        fused_step(model, xg, opt, tape=tape)
        torch.cuda.synchronize()
        got = {k: p_.grad.detach().cpu().double().clone() for k, p_ in model.named_parameters()}
        model._flat[1].copy_(saved)
        model.set_precision('fp32')
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        _, _, ref = O.train_step(sd, O.make_cfg(*ctor), x_eval, ne, tape, hp)
        out = {}
        for k, g_ in got.items():
            r_ = ref[k].double()
            nr = r_.norm().item()
            out[k] = dict(cos=(torch.dot(g_.flatten(), r_.flatten()) / (g_.norm() * r_.norm() + 1e-300)).item(), ratio=g_.norm().item() / max(nr, 1e-300),
                          norm=nr, numel=r_.numel())
        return out

    rows = []
    rel = lambda a, b: abs(a - b) / abs(b)
    for it in range(steps + 1):
        if it % every == 0 or it == steps:
            # (the evaluation forwards run in training mode -- batch statistics, as the gate is defined -- and therefore also move the
            # BatchNorm running statistics; those do not enter a training-mode loss)
            l16, n16, kz16 = hip('bf16')
            l32, n32, kz32 = hip('fp32')
            row = dict(recipe=name, seed=seed, step=it, frames=T * B, loss_fp32_mode=l32, nll=n32, kl_z=kz32, bf16_vs_fp32_mode=rel(l16, l32), bf16_nll_vs_fp32_mode=rel(n16, n32))
            if it in oracle_at:
                t0 = time.time()
                lo = oracle()
                row.update(loss_oracle=lo, fp32_mode_vs_oracle=rel(l32, lo), bf16_vs_oracle=rel(l16, lo), oracle_s=round(time.time() - t0, 1))
            if it in grads_at:
                gg = grad_gate()
                worst_c = min(gg, key=lambda k: gg[k]['cos'])
                worst_r = max(gg, key=lambda k: abs(gg[k]['ratio'] - 1))
                row.update(grad_gate=dict(tensors=len(gg), worst_cos=gg[worst_c]['cos'], worst_cos_tensor=worst_c, worst_ratio=gg[worst_r]['ratio'],
                                          worst_ratio_tensor=worst_r, below_0p99=sorted(k for k in gg if gg[k]['cos'] < 0.99),
                                          outside_5pct=sorted(k for k in gg if abs(gg[k]['ratio'] - 1) > 0.05)),
                           grad_gate_all={k: (round(v['cos'], 5), round(v['ratio'], 4), float('%.3g' % v['norm'])) for k, v in gg.items()})
            rows.append(row)
            log(json.dumps(row))
            model.set_precision('fp32')
        if it < steps:
            xb = torch.from_numpy(synth_video(T, B, nc, seed=1000 + it + 100000 * seed)).to(dev)
            train(model, optim, None, xb, dev, opt)
    return rows


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'kth'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    run(name, steps, every, oracle_at=range(0, steps + 1, every), seed=seed, grads_at=(100,) if os.environ.get('GRAD_GATE', '1') == '1' else ())
