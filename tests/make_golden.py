"""
Golden-vector generator (build container only).

Imports the REAL reference from /root/reference (read-only, never copied), runs its training step
(`train.train`) / eval forward / test.py-style rollout on tiny seeded configurations, records the RNG tape
(torch.randint, torch.randperm, Normal.rsample draws) and dumps inputs + expected outputs as small .npz
fixtures under tests/golden/.  The fixtures are data only; the oracle (oracle/srvp_oracle.py) and the HIP path
are both checked against them.  /root/reference does not exist on the GPU box, so this script is never run there.

    python tests/make_golden.py            # regenerate every fixture
    python tests/make_golden.py --full     # only the full-width fixtures; --dense: only the remove_intermediate=False ones;
                                           # --roll2: add the well-conditioned 53-frame leg to the config-5 fixture
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
REF = '/root/reference'


def import_reference():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # configargparse is not installed: argparse subclass with `.add` (args.py:16,43)
    if 'configargparse' not in sys.modules:
        m = types.ModuleType('configargparse')

        class ArgumentParser(argparse.ArgumentParser):
            def add(self, *a, **k):
                return self.add_argument(*a, **k)
        m.ArgumentParser = ArgumentParser
        m.ArgumentDefaultsHelpFormatter = argparse.ArgumentDefaultsHelpFormatter
        sys.modules['configargparse'] = m
        # argument groups need .add too
        argparse._ArgumentGroup.add = argparse._ArgumentGroup.add_argument
        argparse._MutuallyExclusiveGroup.add = argparse._MutuallyExclusiveGroup.add_argument
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tvd = types.ModuleType('torchvision.datasets')
        tv.datasets = tvd
        sys.modules['torchvision'] = tv
        sys.modules['torchvision.datasets'] = tvd
    import module.srvp as srvp
    import train as ref_train
    import helper
    return srvp, ref_train, helper


class Tape:
    """Records the reference's random draws in call order."""

    def __init__(self):
        self.randint, self.randperm, self.normal = [], [], []
        self._orig = {}

    def __enter__(self):
        import torch.distributions.normal as tdn
        self._orig = dict(randint=torch.randint, randperm=torch.randperm, sn=tdn._standard_normal)
        tape = self

        def randint(*a, **k):
            r = tape._orig['randint'](*a, **k)
            tape.randint.append(r.clone())
            return r

        def randperm(*a, **k):
            r = tape._orig['randperm'](*a, **k)
            tape.randperm.append(r.clone())
            return r

        def sn(shape, dtype, device):
            r = tape._orig['sn'](shape, dtype, device)
            tape.normal.append(r.clone())
            return r
        torch.randint, torch.randperm, tdn._standard_normal = randint, randperm, sn
        return self

    def __exit__(self, *exc):
        import torch.distributions.normal as tdn
        torch.randint, torch.randperm, tdn._standard_normal = self._orig['randint'], self._orig['randperm'], self._orig['sn']


def tape_arrays(tape, cfg, training):
    out = {}
    if training:
        if cfg['skipco']:
            out['tape.t_skip'] = tape.randint[0].numpy()
        out['tape.t_w'] = torch.stack([p[:cfg['nt_inf']] for p in tape.randperm], 1).numpy()
    out['tape.eps_y0'] = tape.normal[0].numpy()
    if len(tape.normal) > 1:
        out['tape.eps_z'] = torch.stack(tape.normal[1:]).numpy()
    return out


OUT_NAMES = ['x_', 'y', 'z', 'w', 'q_y_0_params', 'q_z_params', 'p_z_params', 'res']

# name: constructor args (reference positional order), T, B, n_euler, hyper-parameters
TINY = {}
for archi, nc in (('dcgan', 1), ('vgg', 3)):
    for skipco in (False, True):
        for n_euler in (1, 2):
            TINY[f'tiny_{archi}_nc{nc}_skip{int(skipco)}_e{n_euler}'] = dict(
                ctor=(64, nc, 4, 8, 3, 3, skipco, 2, 8, 3, 16, 4, archi), T=4, B=3, n_euler=n_euler,
                hp=dict(obs_scale=0.2 if archi == 'vgg' else 1.0, beta_y=1.0, beta_z=2.0 if archi == 'dcgan' else 1.0,
                        l2_res=1.0), res_gain=1.41 if archi == 'dcgan' else 1.2, lr=3e-4)
# cross cases (channel count vs archi) and a wider/odd-dimension case
TINY['tiny_dcgan_nc3_skip1_e2'] = dict(ctor=(64, 3, 4, 8, 3, 3, True, 2, 8, 3, 16, 4, 'dcgan'), T=4, B=3, n_euler=2,
                                       hp=dict(obs_scale=0.71, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.41, lr=3e-4)
TINY['tiny_vgg_nc1_skip1_e2'] = dict(ctor=(64, 1, 4, 8, 3, 3, True, 3, 8, 3, 16, 4, 'vgg'), T=5, B=2, n_euler=2,
                                     hp=dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.2, lr=3e-4)
TINY['small_vgg_nc3_skip1_e2'] = dict(ctor=(64, 3, 8, 16, 5, 7, True, 2, 24, 3, 40, 4, 'vgg'), T=3, B=2, n_euler=2,
                                      hp=dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.2, lr=3e-4)
TINY['small_dcgan_nc1_skip0_e1'] = dict(ctor=(64, 1, 8, 16, 5, 7, False, 3, 24, 2, 40, 3, 'dcgan'), T=4, B=2, n_euler=1,
                                        hp=dict(obs_scale=1.0, beta_y=1.0, beta_z=2.0, l2_res=1.0), res_gain=1.41, lr=3e-4)


def synth_video(T, B, C, seed):
    """Seeded smooth-ish synthetic frames in [0,1] (moving blobs) -- pure numpy, no dataset needed."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:64, 0:64].astype(np.float32)
    x = np.zeros((T, B, C, 64, 64), np.float32)
    for b in range(B):
        for c in range(C):
            p = rng.uniform(12, 52, 2)
            v = rng.uniform(-4, 4, 2)
            s = rng.uniform(3, 8)
            for t in range(T):
                q = p + v * t
                x[t, b, c] = np.exp(-((yy - q[0]) ** 2 + (xx - q[1]) ** 2) / (2 * s * s))
    x += rng.uniform(0, 0.05, x.shape).astype(np.float32)
    return np.clip(x, 0, 1)


def gen_one(name, spec, srvp, ref_train, helper):
    torch.set_num_threads(1)
    ctor, T, B, n_euler, hp = spec['ctor'], spec['T'], spec['B'], spec['n_euler'], spec['hp']
    cfg_keys = ['nx', 'nc', 'nf', 'nhx', 'ny', 'nz', 'skipco', 'nt_inf', 'nh_inf', 'nlayers_inf', 'nh_res',
                'nlayers_res', 'archi']
    cfg = dict(zip(cfg_keys, ctor))
    torch.manual_seed(1)
    model = srvp.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(res_gain=spec['res_gain'])
    # perturb BN affine / running stats so eval-mode and bias paths are non-trivial
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_mean'):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif k.endswith('running_var'):
                v.copy_(1 + 0.3 * torch.rand(v.shape, generator=g))
            elif k.endswith('.1.bias') or k.endswith('upconv.1.bias') or k.endswith('last_conv.1.bias'):
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
    x = torch.from_numpy(synth_video(T, B, cfg['nc'], seed=123))
    out = {}
    out['x'] = x.numpy()
    for k, v in model.state_dict().items():
        out['sd0.' + k] = v.detach().numpy().copy()

    # ---------------- train step through the reference's own train.train (train.py:49-129) ----------------
    opt = helper.DotDict(dict(n_euler_steps=n_euler, torch_amp=False, apex_amp=False, **hp))
    optimizer = torch.optim.Adam(model.parameters(), lr=spec['lr'])
    model.train()
    grads = {}
    orig_step = optimizer.step

    def step_and_capture(*a, **k):
        for kk, p in model.named_parameters():
            grads[kk] = p.grad.detach().clone()
        return orig_step(*a, **k)
    optimizer.step = step_and_capture
    fwd_outs = {}

    def forward_fn(xx, nt, dt):
        o = model(xx, nt, dt=dt)
        for n_, v in zip(OUT_NAMES, o):
            fwd_outs[n_] = v.detach().clone()
        return o
    with Tape() as tape:
        loss, nll, kl_y_0, kl_z = ref_train.train(forward_fn, optimizer, None, x, torch.device('cpu'), opt)
    out.update(tape_arrays(tape, cfg, True))
    for n_, v in fwd_outs.items():
        out['train.' + n_] = v.numpy()
    out['train.scalars'] = np.array([loss, nll, kl_y_0, kl_z], np.float64)
    out['train.l2_res'] = np.array(torch.norm(fwd_outs['res'], p=2, dim=2).sum().item(), np.float64)
    for k, v in grads.items():
        out['grad.' + k] = v.numpy()
    for k, v in model.state_dict().items():
        out['sd1.' + k] = v.detach().numpy().copy()

    # ---------------- eval forward with prediction beyond the data (train.py:165-173) ----------------
    model.eval()
    nt_cond = max(cfg['nt_inf'], T - 1)
    nt_pred = T + 2
    with torch.no_grad(), Tape() as tape:
        o = model(x[:nt_cond], nt_pred, dt=1 / n_euler)
    ev = tape_arrays(tape, cfg, False)
    out.update({k.replace('tape.', 'eval.tape.'): v for k, v in ev.items()})
    out['eval.nt_cond'] = np.array(nt_cond)
    out['eval.nt'] = np.array(nt_pred)
    for n_, v in zip(OUT_NAMES, o):
        if v is not None:
            out['eval.' + n_] = v.numpy()

    # ---------------- test.py-style rollout (test.py:235-246) ----------------
    nt_gen = 6
    with torch.no_grad(), Tape() as tape:
        skip = model.encode(x[:nt_cond])[1] if cfg['skipco'] else None
        x_rec, y, _, w, _, _, _, _ = model(x[:nt_cond], nt_cond, dt=1 / n_euler)
        n_fwd = len(tape.normal)
        y_os = model.generate(y[-1], [], nt_gen - nt_cond + 1 + 2, 1 / n_euler)[0]
        y2 = y_os[1:].contiguous()
        x_pred = model.decode(w, y2, skip).clamp(0, 1)
    out['roll.eps_y0'] = tape.normal[0].numpy()
    out['roll.eps_z_fwd'] = torch.stack(tape.normal[1:n_fwd]).numpy()
    out['roll.eps_z_gen'] = torch.stack(tape.normal[n_fwd:]).numpy()
    out['roll.x_rec'] = x_rec.numpy()
    out['roll.y_gen'] = y_os.numpy()
    out['roll.x_pred'] = x_pred.numpy()

    meta = dict(ctor=list(ctor), T=T, B=B, n_euler=n_euler, hp=hp, res_gain=spec['res_gain'], lr=spec['lr'])
    out['meta'] = np.array(repr(meta))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    return loss


# ---------------------------------------------------------------------------------------------------------------------
# FULL-WIDTH reference runs (nf = 64, nhx = 128, nh_inf = 256, nh_res = 512) of the C2..C5 shapes of
# BASELINE.json at batch 2.  Full tensors would be tens of MB, so the fixture holds what pins the computation without them:
# the noise tape, per-tensor weight checksums (weights / inputs are re-created on the target from their seeds), the ELBO
# scalars, the small latent outputs in full, strided samples of the decoded frames, and for every parameter gradient its norm,
# its projection on a seeded random direction, and (for tensors of <= 1024 elements) the gradient itself.
FULL = {
    'full_c2_smmnist_dcgan': dict(ctor=(64, 1, 64, 128, 20, 20, False, 5, 256, 3, 512, 4, 'dcgan'), T=15, B=2, n_euler=1,
                                  hp=dict(obs_scale=1.0, beta_y=1.0, beta_z=2.0, l2_res=1.0), res_gain=1.41, lr=3e-4),
    'full_c3_kth_vgg': dict(ctor=(64, 1, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg'), T=20, B=2, n_euler=2,
                            hp=dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.2, lr=3e-4),
    'full_c4_bair_vgg': dict(ctor=(64, 3, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg'), T=12, B=2, n_euler=2,
                             hp=dict(obs_scale=0.71, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.41, lr=3e-4),
    'full_c5_human_vgg': dict(ctor=(64, 3, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg'), T=16, B=2, n_euler=2,
                              hp=dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.2, lr=3e-4,
                              rollout=dict(nt_cond=8, nt=53)),
    # round 5: the same three VGG recipes at >= 96 frames, the size from which the product's STREAMING kernels take the 64x64-resolution
    # layers (csrc/conv_stream.hip: conv_stream64 / conv_stream_sub64 and their fused BatchNorm-backward sums), so that reference-made
    # numbers pass through them and not only through the tile kernels
    'full_w3_kth_vgg_b5': dict(ctor=(64, 1, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg'), T=20, B=5, n_euler=2,
                               hp=dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.2, lr=3e-4),
    'full_w4_bair_vgg_b8': dict(ctor=(64, 3, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg'), T=12, B=8, n_euler=2,
                                hp=dict(obs_scale=0.71, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.41, lr=3e-4),
    'full_w5_human_vgg_b6': dict(ctor=(64, 3, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg'), T=16, B=6, n_euler=2,
                                 hp=dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0), res_gain=1.2, lr=3e-4),
}


def frame_samples(x_):
    """Strided pixel samples of (nt, B, C, 64, 64) frames: every 5th row / column with a per-frame phase."""
    return x_[:, :, :, 1::5, 2::5].contiguous()


def grad_direction(name, shape):
    g = torch.Generator().manual_seed(abs(hash_name(name)) % (2 ** 31))
    return torch.randn(shape, generator=g)


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000000007
    return h


def gen_full(name, spec, srvp, ref_train, helper):
    torch.set_num_threads(8)
    ctor, T, B, n_euler, hp = spec['ctor'], spec['T'], spec['B'], spec['n_euler'], spec['hp']
    cfg_keys = ['nx', 'nc', 'nf', 'nhx', 'ny', 'nz', 'skipco', 'nt_inf', 'nh_inf', 'nlayers_inf', 'nh_res',
                'nlayers_res', 'archi']
    cfg = dict(zip(cfg_keys, ctor))
    torch.manual_seed(1)
    model = srvp.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(res_gain=spec['res_gain'])
    x = torch.from_numpy(synth_video(T, B, cfg['nc'], seed=321))
    out = {'x.checksum': np.array([x.double().sum().item(), x.double().abs().max().item()])}
    out['sd0.checksums'] = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in model.state_dict().values()])
    opt = helper.DotDict(dict(n_euler_steps=n_euler, torch_amp=False, apex_amp=False, **hp))
    optimizer = torch.optim.Adam(model.parameters(), lr=spec['lr'])
    model.train()
    grads = {}
    orig_step = optimizer.step

    def step_and_capture(*a, **k):
        for kk, p in model.named_parameters():
            grads[kk] = p.grad.detach().clone()
        return orig_step(*a, **k)
    optimizer.step = step_and_capture
    fwd_outs = {}

    def forward_fn(xx, nt, dt):
        o = model(xx, nt, dt=dt)
        for n_, v in zip(OUT_NAMES, o):
            fwd_outs[n_] = v.detach().clone()
        return o
    with Tape() as tape:
        loss, nll, kl_y_0, kl_z = ref_train.train(forward_fn, optimizer, None, x, torch.device('cpu'), opt)
    out.update(tape_arrays(tape, cfg, True))
    for n_, v in fwd_outs.items():
        out['train.' + n_] = (frame_samples(v) if n_ == 'x_' else v).numpy()
    out['train.scalars'] = np.array([loss, nll, kl_y_0, kl_z], np.float64)
    out['train.l2_res'] = np.array(torch.norm(fwd_outs['res'], p=2, dim=2).sum().item(), np.float64)
    names = list(grads)
    out['grad.norm'] = np.array([grads[k].double().norm().item() for k in names])
    out['grad.dot'] = np.array([(grads[k].double() * grad_direction(k, grads[k].shape).double()).sum().item() for k in names])
    for k in names:
        if grads[k].numel() <= 1024:
            out['grad.' + k] = grads[k].numpy()
    # BN running statistics after the step (conv.py:104)
    for k, v in model.state_dict().items():
        if k.endswith(('running_mean', 'running_var')):
            out['sd1.' + k] = v.detach().numpy().copy()
    # ---------------- long-horizon prediction (config 5: test.py / train.evaluate call pattern, seq_len_test = 53) ----------------
    ro = spec.get('rollout')
    if ro is not None:
        # inference weights: the initial ones (BN running statistics at their defaults), so that the target needs no optimizer step
        torch.manual_seed(1)
        model = srvp.StochasticLatentResidualVideoPredictor(*ctor)
        model.init(res_gain=spec['res_gain'])
        model.eval()
        xr = torch.from_numpy(synth_video(ro['nt_cond'], B, cfg['nc'], seed=322))
        with torch.no_grad(), Tape() as tape:
            o = model(xr, ro['nt'], dt=1 / n_euler)
        ev = tape_arrays(tape, cfg, False)
        out.update({k.replace('tape.', 'roll.tape.'): v for k, v in ev.items()})
        out['roll.y'] = o[1].numpy()
        out['roll.x_'] = frame_samples(o[0]).numpy()
        out['roll.cfg'] = np.array([ro['nt_cond'], ro['nt']])
    meta = dict(ctor=list(ctor), T=T, B=B, n_euler=n_euler, hp=hp, res_gain=spec['res_gain'], lr=spec['lr'], grad_names=names)
    out['meta'] = np.array(repr(meta))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    return loss


def gen_known_answers():
    """Small known-answer vectors produced by the reference's utils / schedule code."""
    import math
    import module.utils as rutils
    import torch.distributions as distrib
    out = {}
    g = torch.Generator().manual_seed(5)
    raw = torch.randn(4, 10, generator=g) * 3
    raw[0, 5] = 25.0   # softplus threshold branch
    raw[1, 6] = -30.0
    raw2 = torch.randn(4, 10, generator=g)
    q = rutils.make_normal_from_raw_params(raw)
    p = rutils.make_normal_from_raw_params(raw2)
    out['ka.raw_q'], out['ka.raw_p'] = raw.numpy(), raw2.numpy()
    out['ka.q_scale'] = q.scale.numpy()
    out['ka.kl_qp'] = distrib.kl_divergence(q, p).numpy()
    out['ka.kl_q0'] = distrib.kl_divergence(q, distrib.Normal(0, 1)).numpy()
    loc = torch.rand(3, 7, generator=g)
    data = torch.rand(3, 7, generator=g)
    for s in (1.0, 0.2, 0.71):
        out[f'ka.nll_{s}'] = rutils.neg_logprob(loc, data, scale=s).numpy()
    out['ka.nll_loc'], out['ka.nll_data'] = loc.numpy(), data.numpy()
    # Euler grid exactly as srvp.py:377-402 walks it
    for n in (1, 2, 4):
        for nt in (12, 15, 16, 20, 53):
            dt = 1 / n
            rows, t_data = [], 0
            for t in np.linspace(dt, nt - 1, n * (nt - 1)):
                prev = t_data
                t_data = int(math.ceil(t))
                rows.append((t_data, int(t_data != prev), int(float(t).is_integer())))
            out[f'ka.euler_n{n}_nt{nt}'] = np.array(rows, np.int64)
    # LR schedule (train.py:292-293)
    n_iter = 1000
    out['ka.lr_lambda'] = np.array([max(0, (n_iter - i) / n_iter) for i in range(0, 1200, 100)])
    np.savez_compressed(os.path.join(GOLDEN, 'known_answers.npz'), **out)


def metric_videos():
    """Seeded (prediction, ground truth) video pairs for the metric fixtures: (nt=3, B=2, C, 64, 64) in [0, 1] -- smooth
    structured frames (sums of random sinusoids) plus noise of three strengths, so that SSIM covers ~0.1 .. ~1."""
    out = {}
    for C, seed in ((1, 21), (3, 22)):
        g = torch.Generator().manual_seed(seed)
        yy, xx = torch.meshgrid(torch.arange(64.), torch.arange(64.), indexing='ij')
        gt = torch.zeros(3, 2, C, 64, 64)
        for _ in range(6):
            f = torch.rand(3, 2, C, 2, generator=g) * 0.5
            ph = torch.rand(3, 2, C, 1, 1, generator=g) * 6.28
            gt += torch.sin(f[..., 0, None, None] * yy + f[..., 1, None, None] * xx + ph) / 6
        gt = (gt * 0.5 + 0.5).clamp(0, 1)
        noise = torch.randn(3, 2, C, 64, 64, generator=g) * torch.tensor([0.005, 0.05, 0.3]).view(3, 1, 1, 1, 1)
        pred = (gt + noise).clamp(0, 1)
        out[C] = (pred.contiguous(), gt.contiguous())
    return out


def gen_metrics():
    """PSNR / SSIM of the reference (test.py:249-253 through metrics/ssim.py) on seeded videos."""
    from metrics.ssim import ssim_loss
    out = {}
    for C, (pred, gt) in metric_videos().items():
        nt, bsz = pred.shape[:2]
        img = pred.shape[2:]
        ssim = ssim_loss(pred.view(nt * bsz, *img), gt.view(nt * bsz, *img), max_val=1., reduction='none')   # test.py:56
        ssim = ssim.mean(dim=[2, 3]).view(nt, bsz, img[0])                                                   # test.py:57
        mse = torch.mean((pred - gt) ** 2, dim=[3, 4])                                                       # test.py:249
        out[f'c{C}.pred'], out[f'c{C}.gt'] = pred.numpy(), gt.numpy()
        out[f'c{C}.ssim'], out[f'c{C}.mse'] = ssim.numpy(), mse.numpy()
        out[f'c{C}.psnr'] = (10 * torch.log10(1 / mse)).numpy()                                              # test.py:251
        print(f'metrics C={C}: ssim', ssim.mean(dim=(1, 2)).tolist())
    np.savez_compressed(os.path.join(GOLDEN, 'metrics.npz'), **out)


def synthetic_digits(n=6, seed=31):
    """28x28 uint8 'digits' (no MNIST download in the container): bright random strokes on black, values up to 255 so that
    overlapping objects exercise the clamp of mmnist.py:123."""
    rng = np.random.RandomState(seed)
    d = np.zeros((n, 28, 28), np.uint8)
    for i in range(n):
        for _ in range(3):
            r0, c0 = rng.randint(2, 20, 2)
            h, w = rng.randint(3, 9, 2)
            d[i, r0:r0 + h, c0:c0 + w] = np.maximum(d[i, r0:r0 + h, c0:c0 + w], rng.randint(100, 256, (h, w)).astype(np.uint8))
    return d


def gen_mmnist():
    """Videos and trajectories of the reference's stochastic / deterministic Moving-MNIST training generator
    (data/mmnist.py) under np.random.seed, on synthetic digits."""
    import data.mmnist as rmm
    digits = synthetic_digits()
    out = {'digits': digits}
    cases = {'s15': (15, 4, False, 2, 123, 8), 's30': (30, 4, False, 2, 7, 6), 'd20': (20, 4, True, 2, 11, 6),
             'fast3': (25, 9, False, 3, 5, 6)}
    for name, (T, ms, det, nd, seed, B) in cases.items():
        ds = rmm.MovingMNIST(list(digits), 64, T, ms, det, nd, True)
        np.random.seed(seed)
        out[f'{name}.videos'] = np.stack([ds[i] for i in range(B)], 0)
        out[f'{name}.cfg'] = np.array([T, ms, int(det), nd, seed, B])
        np.random.seed(seed + 1000)
        out[f'{name}.traj'] = np.array([ds._compute_trajectory(28, 28) for _ in range(20)], dtype=np.int64)
        out[f'{name}.traj_init'] = np.array(ds._compute_trajectory(28, 28, init_cond=(30, 3, -ms, 3)), dtype=np.int64)
    # the collated float batch of data/base.py:71-84 for the first case
    from data.base import collate_fn as ref_collate
    ds = rmm.MovingMNIST(list(digits), 64, 15, 4, False, 2, True)
    np.random.seed(123)
    out['s15.batch'] = ref_collate([ds[i] for i in range(8)]).numpy()
    # the 95 % / 5 % item split of data/base.py:96-132 on 1000 (scalar) items
    ds = rmm.MovingMNIST(list(range(1000)), 64, 15, 4, False, 2, True)
    out['fold.val_1000'] = np.array(ds.get_fold('val').data)
    out['fold.train_1000_head'] = np.array(ds.get_fold('train').data[:50])
    np.savez_compressed(os.path.join(GOLDEN, 'mmnist.npz'), **out)
    print('mmnist fixture:', {k: v.shape for k, v in out.items() if k.endswith('videos')})


def gen_dense(name, spec, srvp):
    """remove_intermediate=False (srvp.py:402,415-470): eval forward that keeps and decodes every Euler sub-step, conditioning on
    nt_cond frames and predicting beyond them.  Same seeded model / video as the fixture `name`; written to dense_<name>.npz."""
    torch.set_num_threads(1)
    ctor, T, B, n_euler = spec['ctor'], spec['T'], spec['B'], spec['n_euler']
    cfg_keys = ['nx', 'nc', 'nf', 'nhx', 'ny', 'nz', 'skipco', 'nt_inf', 'nh_inf', 'nlayers_inf', 'nh_res', 'nlayers_res', 'archi']
    cfg = dict(zip(cfg_keys, ctor))
    torch.manual_seed(1)
    model = srvp.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(res_gain=spec['res_gain'])
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_mean'):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif k.endswith('running_var'):
                v.copy_(1 + 0.3 * torch.rand(v.shape, generator=g))
            elif k.endswith('.1.bias') or k.endswith('upconv.1.bias') or k.endswith('last_conv.1.bias'):
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
    x = torch.from_numpy(synth_video(T, B, cfg['nc'], seed=123))
    out = {'x': x.numpy()}
    for k, v in model.state_dict().items():
        out['sd0.' + k] = v.detach().numpy().copy()
    model.eval()
    nt_cond, nt_pred = max(cfg['nt_inf'], T - 1), T + 2
    with torch.no_grad(), Tape() as tape:
        o = model(x[:nt_cond], nt_pred, dt=1 / n_euler, remove_intermediate=False)
    out.update(tape_arrays(tape, cfg, False))
    out['nt_cond'], out['nt'], out['n_euler'] = np.array(nt_cond), np.array(nt_pred), np.array(n_euler)
    for n_, v in zip(OUT_NAMES, o):
        if v is not None:
            out['out.' + n_] = v.numpy()
    out['meta'] = np.array(repr(dict(ctor=list(ctor), T=T, B=B, n_euler=n_euler)))
    np.savez_compressed(os.path.join(GOLDEN, 'dense_' + name + '.npz'), **out)
    return tuple(o[0].shape)


# ---------------------------------------------------------------------------------------------------------------------
# Second long-horizon leg of config 5: the 53-frame prediction from a WELL-CONDITIONED state, so that every
# one of the 53 frames can be held to a tight tolerance.  At the recipe's res_gain = 1.2 the untrained residual MLP expands |y| by
# ~1.35x per frame (7 -> 3e5 over the horizon) and with it every arithmetic difference; here: same seed, init(res_gain = 0.8)
# (|y| 7 -> 46, the reference's own fp32 vs float64 runs agree to 3e-7 at EVERY frame), BatchNorm running statistics settled by
# three training-mode forwards of the reference (stored in the fixture), and the output layer's weight scaled by 30 so that the
# decoded frames of this untrained network span 0.39 .. 0.68 and move by up to 1.4e-2 per frame instead of sitting at 0.5.
ROLL2 = {'full_c5_human_vgg': dict(nt_cond=8, nt=53, res_gain=0.8, out_scale=30.0, settle=3)}


def roll2_state(model, ro, T, B, nc):
    """The three training-mode forwards that settle the BatchNorm statistics + the output-layer scaling (reference model)."""
    model.train()
    xs = torch.from_numpy(synth_video(T, B, nc, seed=323))
    with torch.no_grad():
        for i in range(ro['settle']):
            torch.manual_seed(100 + i)
            model(xs, T, dt=0.5)
        model.state_dict()['decoder.conv.3.1.weight'].mul_(ro['out_scale'])
    model.eval()


def gen_roll2(name, spec, ro, srvp):
    torch.set_num_threads(8)
    ctor, T, B, n_euler = spec['ctor'], spec['T'], spec['B'], spec['n_euler']
    cfg_keys = ['nx', 'nc', 'nf', 'nhx', 'ny', 'nz', 'skipco', 'nt_inf', 'nh_inf', 'nlayers_inf', 'nh_res', 'nlayers_res', 'archi']
    cfg = dict(zip(cfg_keys, ctor))
    path = os.path.join(GOLDEN, name + '.npz')
    old = np.load(path, allow_pickle=False)
    out = {k: old[k] for k in old.files if not k.startswith('roll2.')}
    torch.manual_seed(1)
    model = srvp.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(res_gain=ro['res_gain'])
    roll2_state(model, ro, T, B, cfg['nc'])
    for k, v in model.state_dict().items():
        if k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
            out['roll2.bn.' + k] = v.detach().numpy().copy()
    out['roll2.sd.checksums'] = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in model.state_dict().values()])
    xr = torch.from_numpy(synth_video(ro['nt_cond'], B, cfg['nc'], seed=322))
    with torch.no_grad(), Tape() as tape:
        o = model(xr, ro['nt'], dt=1 / n_euler)
    ev = tape_arrays(tape, cfg, False)
    out.update({k.replace('tape.', 'roll2.tape.'): v for k, v in ev.items()})
    out['roll2.y'] = o[1].numpy()
    out['roll2.x_'] = frame_samples(o[0]).numpy()
    out['roll2.cfg'] = np.array([ro['nt_cond'], ro['nt']])
    out['roll2.recipe'] = np.array([ro['res_gain'], ro['out_scale'], ro['settle']], np.float64)
    np.savez_compressed(path, **out)
    return float(o[1][-1].norm()), float(o[0].min()), float(o[0].max())


DENSE = ['tiny_vgg_nc3_skip1_e2', 'tiny_dcgan_nc1_skip0_e2']


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    srvp, ref_train, helper = import_reference()
    if '--dense' in sys.argv:                              # only the remove_intermediate=False fixtures
        for name in DENSE:
            print(f'dense_{name}: frames', gen_dense(name, TINY[name], srvp), flush=True)
        return
    if '--roll2' in sys.argv:                              # only the second long-horizon leg, added to the existing full-width fixture
        for name, ro in ROLL2.items():
            print(f'{name} roll2: |y_52|, min / max decoded frame', gen_roll2(name, FULL[name], ro, srvp), flush=True)
        return
    if '--full' in sys.argv:                               # only the full-width fixtures (minutes of CPU time)
        only = [a for a in sys.argv[1:] if a.startswith('full_')]     # (optionally: just the named fixtures)
        for name, spec in FULL.items():
            if only and name not in only:
                continue
            loss = gen_full(name, spec, srvp, ref_train, helper)
            print(f'{name}: loss {loss:.6f}', flush=True)
            if name in ROLL2:
                gen_roll2(name, spec, ROLL2[name], srvp)
        return
    for name, spec in TINY.items():
        loss = gen_one(name, spec, srvp, ref_train, helper)
        print(f'{name}: loss {loss:.6f}')
    gen_known_answers()
    gen_metrics()
    gen_mmnist()
    print('done')


if __name__ == '__main__':
    main()
