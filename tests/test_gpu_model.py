"""
GPU parity of the full hot path (model forward, ELBO, backward, Adam, eval prediction, rollout API) against
(a) the golden fixtures generated from the real reference (tests/golden/*.npz) and (b) the CPU oracle at full layer
width.  The HIP path computes convolutions with bf16 operands / fp32 accumulation (DESIGN.md "precision"), so the
tolerances below are the bf16 ones, stated per quantity.  The ELBO error of bf16 storage is rounding noise that averages
as 1/sqrt(frames): <= 5e-4 on the 12-frame tiny fixtures, <= 2e-4 at full width on 16-40 frames here, and the north_star gate
(<= 1e-4) is asserted at 384 frames in tests/test_gpu_parity_gate.py (measured 3e-6).  The same fixtures are held to 1e-5
(ELBO) / 2e-3 (every gradient) in fp32 mode: tests/test_gpu_fp32_mode.py.
"""
import json
import os

import pytest
import torch

from golden_util import Fixture, OUT_NAMES, fixture_names

pytestmark = pytest.mark.gpu
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_report.jsonl')


def report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, 'a') as f:
        f.write(json.dumps(kw) + '\n')


def max_abs(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def build(fx, sd_key='sd0'):
    import srvp_amd
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*fx.meta['ctor'])
    missing = m.load_state_dict(fx.state(sd_key), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.cuda()


def opt_of(fx):
    import srvp_amd
    return srvp_amd.DotDict(dict(n_euler_steps=fx.meta['n_euler'], **fx.meta['hp']))


@pytest.mark.parametrize('name', fixture_names())
def test_train_step_vs_reference_fixture(name):
    from srvp_amd import FusedAdam
    from srvp_amd.train import elbo_terms_and_grads
    fx = Fixture(name)
    model = build(fx)
    model.train()
    x = fx.t('x').cuda()
    tape = fx.tape()
    opt = opt_of(fx)
    optim = FusedAdam(model, lr=fx.meta['lr'])
    optim.zero_grad()
    outs = model._forward_impl(x, x.shape[0], fx.meta['n_euler'], tape, training=True)
    outs_c = [o.clone() if o is not None else None for o in outs]
    acc, g = elbo_terms_and_grads(model, x, outs, opt)
    model._backward_impl(g[0], None, None, g[1], g[2], g[3], g[4])
    nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
    B = x.shape[1]
    hp = fx.meta['hp']
    loss = (nll + hp['beta_y'] * kl_y0 + hp['beta_z'] * kl_z + hp['l2_res'] * l2) / B
    ref = fx.z['train.scalars']
    errs = dict(loss=abs(loss - ref[0]) / abs(ref[0]), nll=abs(nll / B - ref[1]) / abs(ref[1]),
                kl_y0=abs(kl_y0 / B - ref[2]) / max(abs(ref[2]), 1e-6), kl_z=abs(kl_z / B - ref[3]) / max(abs(ref[3]), 1e-6),
                l2=abs(l2 - float(fx.z['train.l2_res'])) / abs(float(fx.z['train.l2_res'])))
    # (a) against the CPU oracle evaluated under the product's numerics model (same rounding points): tight
    from oracle import srvp_oracle as O
    O.PRECISION = 'bf16'
    try:
        sd_m = fx.state('sd0')
        scal_m, outs_m, grads_m = O.train_step(sd_m, fx.cfg, fx.t('x'), fx.meta['n_euler'], tape, hp)
    finally:
        O.PRECISION = 'fp32'
    e_model = abs(loss - scal_m['loss']) / abs(scal_m['loss'])
    out_m = {n: (max_abs(o, r), rel_l2(o, r)) for n, o, r in zip(OUT_NAMES, outs_c, outs_m)}
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    gerr_m = {k: rel_l2(grads[k], grads_m[k]) for k in grads_m}
    worst_m = sorted(gerr_m.items(), key=lambda kv: -kv[1])[:5]
    med_m = sorted(gerr_m.values())[len(gerr_m) // 2]
    # (b) against the fp32 fixture of the real reference: ELBO tight, the rest within the bf16 bands
    out_err = {}
    for n, o in zip(OUT_NAMES, outs_c):
        r = fx.t('train.' + n)
        out_err[n] = (max_abs(o, r), rel_l2(o, r))
    gref = fx.group('grad.')
    gerr = {k: rel_l2(grads[k], gref[k]) for k in gref}
    cos = {k: torch.nn.functional.cosine_similarity(grads[k].flatten().double().cpu(), gref[k].flatten().double(), dim=0).item()
           for k in gref}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:5]
    report(test='train_fixture', name=name, scalars=errs, outs=out_err, worst_grads=worst,
           median_grad=sorted(gerr.values())[len(gerr) // 2], e_model=e_model, outs_model=out_m, worst_grads_model=worst_m,
           median_grad_model=med_m, min_cos=min(cos.values()))
    # DCGAN (10 conv layers, strided): the two implementations take the same bf16 rounding decisions almost everywhere
    # -> tight.  VGG (22 conv layers at full resolution): isolated 1-ulp bf16 rounding flips caused by the different
    # fp32 summation order grow through the stack (measured: 0.04 % of the elements after layer 1, 70 % after
    # layer 10), so two *correct* bf16 implementations agree only to the bf16 band; per-layer exactness is asserted
    # separately in test_gpu_blocks.py and by test_vgg_backward_teacher_forced below.
    vgg = fx.cfg['archi'] == 'vgg'
    assert e_model < (2e-4 if vgg else 1e-6), e_model
    assert out_m['x_'][0] < (3e-2 if vgg else 2e-3), out_m
    for n in OUT_NAMES[1:]:
        assert out_m[n][1] < (6e-2 if vgg else 2e-3), (n, out_m[n])
    assert max(gerr_m.values()) < (0.7 if vgg else 0.08) and med_m < (0.2 if vgg else 0.03), (worst_m, med_m)
    assert errs['loss'] < 5e-4 and errs['nll'] < 5e-4, errs
    assert errs['kl_y0'] < 3e-2 and errs['kl_z'] < 3e-2 and errs['l2'] < 2e-2, errs
    assert out_err['x_'][0] < 5e-2, out_err
    # ---- Adam + BN running statistics (train.py:120; conv.py:104)
    optim.step()
    sd0, sd1 = fx.state('sd0'), fx.state('sd1')
    lr = fx.meta['lr']
    agree_all = []
    for k, v in model.state_dict().items():
        v = v.detach().cpu()
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(sd1[k]), k
        elif k.endswith(('running_mean', 'running_var')):
            assert max_abs(v, sd1[k]) < 2e-3 + 2e-2 * sd1[k].abs().max().item(), k
        else:
            assert max_abs(v, sd1[k]) < 2.5 * lr, k
            moved = (sd1[k] - sd0[k]).abs() > 0.5 * lr
            if moved.any():
                agree_all.append((((v - sd0[k]).sign() == (sd1[k] - sd0[k]).sign()) | ~moved).float().mean().item())
    # Adam's first step is lr * sign(g): a bf16-perturbed small gradient may flip -- most update directions agree
    assert sum(agree_all) / len(agree_all) > 0.80, sum(agree_all) / len(agree_all)


@pytest.mark.parametrize('name', fixture_names())
def test_eval_prediction_and_rollout_vs_reference_fixture(name):
    fx = Fixture(name)
    model = build(fx, 'sd1')
    model.eval()
    x = fx.t('x').cuda()
    ne = fx.meta['n_euler']
    nt_cond, nt = int(fx.z['eval.nt_cond']), int(fx.z['eval.nt'])
    tape = fx.tape('eval.tape.')
    outs = model(x[:nt_cond], nt, dt=1 / ne, tape=tape)
    errs = {}
    for n, o in zip(OUT_NAMES, outs):
        if o is None:
            assert not fx.has('eval.' + n)
            continue
        r = fx.t('eval.' + n)
        errs[n] = (max_abs(o, r), rel_l2(o, r))
    report(test='eval_fixture', name=name, outs=errs)
    assert errs['x_'][0] < 3e-2, errs
    for n in errs:
        if n != 'x_':
            assert errs[n][1] < 3e-2, (n, errs[n])
    # test.py:235-246 call pattern through the granular API
    skip = model.encode(x[:nt_cond])[1] if model.skipco else None
    tape2 = {'eps_y0': fx.t('roll.eps_y0'), 'eps_z': fx.t('roll.eps_z_fwd')}
    x_rec, y, _, w, _, _, _, _ = model(x[:nt_cond], nt_cond, dt=1 / ne, tape=tape2)
    x_rec, y, w = x_rec.clone(), y.clone(), w.clone()
    eps_gen = fx.t('roll.eps_z_gen').cuda()
    y_os = model.generate(y[-1], [], eps_gen.shape[0] + 1, 1 / ne, eps_z=eps_gen)[0]
    x_pred = model.decode(w, y_os[1:].contiguous(), skip).clamp(0, 1)
    e = dict(x_rec=max_abs(x_rec, fx.t('roll.x_rec')), y_gen=rel_l2(y_os, fx.t('roll.y_gen')),
             x_pred=max_abs(x_pred, fx.t('roll.x_pred')))
    report(test='rollout_fixture', name=name, errs=e)
    assert e['x_rec'] < 3e-2 and e['x_pred'] < 3e-2 and e['y_gen'] < 3e-2, e


@pytest.mark.parametrize('archi,nc,skipco,ne,B,T', [('vgg', 3, True, 2, 4, 4), ('dcgan', 1, False, 1, 8, 5), ('dcgan', 1, True, 2, 4, 4)])
def test_full_width_vs_oracle(archi, nc, skipco, ne, B, T):
    """Full layer widths (nf=64, nhx=128, nh_res=512) on 16-40 frames: ELBO within 2e-4 of the CPU oracle in bf16 mode (the
    north_star 1e-4 gate is asserted on 384 frames in test_gpu_parity_gate.py, and 1e-5 in fp32 mode in test_gpu_fp32_mode.py)."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import elbo_terms_and_grads
    torch.manual_seed(1)
    ny = nz = 50 if archi == 'vgg' else 20
    nt_inf = 2
    ctor = (64, nc, 64, 128, ny, nz, skipco, nt_inf, 256, 3, 512, 4, archi)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(123)
    x = torch.rand(T, B, nc, 64, 64, generator=g)
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                eps_y0=torch.randn(B, ny, generator=g), eps_z=torch.randn(T - 1, B, nz, generator=g))
    if skipco:
        tape['t_skip'] = torch.randint(T, (B,), generator=g)
    hp = dict(obs_scale=0.2 if archi == 'vgg' else 1.0, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    torch.set_num_threads(8)          # NOT os.cpu_count(): hundreds of threads oversubscribe the small CPU convolutions
    scal, outs_ref, grads_ref = O.train_step({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, ne, tape, hp)
    O.PRECISION = 'bf16'
    try:
        scal_m, outs_m, grads_m = O.train_step({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, ne, tape, hp)
    finally:
        O.PRECISION = 'fp32'
    model = model.cuda().train()
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    model.flatten_parameters_()
    model._grads()
    model._flat[1].zero_()
    xg = x.cuda()
    outs = model._forward_impl(xg, T, ne, tape, training=True)
    x_ = outs[0].clone()
    acc, gr = elbo_terms_and_grads(model, xg, outs, opt)
    model._backward_impl(gr[0], None, None, gr[1], gr[2], gr[3], gr[4])
    nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
    loss = (nll + kl_y0 + kl_z + l2) / B
    e_loss = abs(loss - scal['loss']) / abs(scal['loss'])
    gerr = {k: rel_l2(p.grad, grads_ref[k]) for k, p in model.named_parameters()}
    gerr_m = {k: rel_l2(p.grad, grads_m[k]) for k, p in model.named_parameters()}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:6]
    worst_m = sorted(gerr_m.items(), key=lambda kv: -kv[1])[:6]
    e_model = abs(loss - scal_m['loss']) / abs(scal_m['loss'])
    report(test='full_width', archi=archi, skipco=skipco, loss=loss, loss_ref=scal['loss'], e_loss=e_loss, e_model=e_model,
           x_maxabs=max_abs(x_, outs_ref[0]), x_maxabs_model=max_abs(x_, outs_m[0]), worst_grads=worst,
           median_grad=sorted(gerr.values())[len(gerr) // 2], worst_grads_model=worst_m,
           median_grad_model=sorted(gerr_m.values())[len(gerr_m) // 2])
    # vs the reference arithmetic (fp32 oracle): the north_star ELBO tolerance, measured on 16-40 frames (the BN batch of
    # the benchmark configuration is 2304 frames, where the rounding noise averages further down)
    assert e_loss < (1.4e-4 if archi == 'vgg' else 5e-6), (loss, scal['loss'])       # measured 1.07e-4 (vgg, 16 frames) / 1e-6 (dcgan): + 25 % / x5
    assert max_abs(x_, outs_ref[0]) < 3e-2
    # vs the same algorithm under the product's numerics model: the kernels implement the algorithm
    vgg = archi == 'vgg'
    assert e_model < (2e-4 if vgg else 2e-6), (loss, scal_m['loss'])
    assert max_abs(x_, outs_m[0]) < (2e-2 if vgg else 5e-3)
    assert max(gerr_m.values()) < (0.5 if vgg else 0.1), worst_m


@pytest.mark.parametrize('name', ['tiny_vgg_nc3_skip1_e2', 'tiny_vgg_nc1_skip1_e2', 'small_vgg_nc3_skip1_e2', 'tiny_dcgan_nc3_skip1_e2'])
def test_conv_backward_teacher_forced(name):
    """
    Backward of the conv encoder / decoder with the FORWARD STATE PINNED to the oracle's (numerics-model) values:
    every block's raw output, BatchNorm coefficients and activation are overwritten with the oracle's bf16-exact
    tensors, then the HIP backward runs on the oracle's output gradient.  What is left is the backward arithmetic
    itself (+ its own bf16 rounding of the propagated gradients, which is linear noise and does not grow chaotically),
    so VGG can be checked as tightly as DCGAN: every parameter gradient within 3 % relative L2 (median < 1 %).
    """
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd import _lib as L
    fx = Fixture(name)
    cfg, hp, ne = fx.cfg, fx.meta['hp'], fx.meta['n_euler']
    x, tape = fx.t('x'), fx.tape()
    T, B = x.shape[0], x.shape[1]
    # ---- oracle forward under the numerics model, recording every block
    rec = {}
    orig = O._conv_block_bf16

    def recording(h, sd, spec, training):
        w = sd[spec['key'] + '.weight']
        role = spec.get('role', 'mfma')
        hh, ww = (O._bf(h), O._bf(w)) if role in ('mfma', 'out') else (h, w)
        r = torch.nn.functional.conv2d(hh, ww, None, spec['s'], spec['p']) if spec['kind'] == 'conv' else \
            torch.nn.functional.conv_transpose2d(hh, ww, None, spec['s'], spec['p'])
        e = dict(raw=O._bf(r).detach() if role != 'out' else None)
        if spec['bnkey'] is not None:
            mean, var = r.mean(dim=(0, 2, 3)), r.var(dim=(0, 2, 3), unbiased=False)
            inv = torch.rsqrt(var + O.BN_EPS)
            scale = sd[spec['bnkey'] + '.weight'] * inv
            e.update(scale=scale.detach(), shift=(sd[spec['bnkey'] + '.bias'] - mean * scale).detach(), mean=mean.detach(), inv=inv.detach())
        out = orig(h, sd, spec, training)
        e['out'] = out.detach()
        rec[spec['key']] = e
        return out
    sd = fx.state('sd0')
    pkeys, _ = O.split_state(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in pkeys}
    work = dict(sd)
    work.update(leaves)
    O._conv_block_bf16, O.PRECISION = recording, 'bf16'
    try:
        outs = O.forward(work, cfg, x, T, ne, tape, True)
        terms = O.elbo(x, outs, hp['obs_scale'], hp['beta_y'], hp['beta_z'], hp['l2_res'])
        conv_keys = [k for k in pkeys if k.startswith(('encoder.', 'decoder.'))]
        gref = dict(zip(conv_keys, torch.autograd.grad(terms['loss'], [leaves[k] for k in conv_keys], retain_graph=True)))
        d_x_ref = torch.autograd.grad(terms['loss'], outs[0], retain_graph=True)[0]
    finally:
        O._conv_block_bf16, O.PRECISION = orig, 'fp32'
    # ---- HIP forward (allocates the plan), then pin the conv state to the oracle's
    model = build(fx).train()
    model.flatten_parameters_()
    xg = x.cuda()
    model._forward_impl(xg, T, ne, tape, training=True)
    pl = model._last_plan

    def put_nhwc(dst, src_nchw):              # dst: [N][H][W][Cp] view, src: (N, C, H, W)
        dst.zero_()
        dst[..., :src_nchw.shape[1]].copy_(src_nchw.permute(0, 2, 3, 1))
    for net in (pl['enc'], pl['dec']):
        for blk in net.blocks:
            e = rec[blk.spec['key']]
            if blk.role != 'out':
                put_nhwc(blk.raw, e['raw'].cuda())
                if blk.has_bn:
                    blk.coef.zero_()
                    for i, nme in enumerate(('scale', 'shift', 'mean', 'inv')):
                        blk.coef[i, :blk.cout_r] = e[nme].cuda()
            if blk.out is not None:
                f = blk.out
                put_nhwc(f.t[:, f.b:f.b + f.H, f.b:f.b + f.W, :], e['out'].cuda())
            if blk.pool is not None:
                f = blk.pool
                put_nhwc(f.t[:, f.b:f.b + f.H, f.b:f.b + f.W, :], torch.nn.functional.max_pool2d(e['out'], 2, 2).cuda())
                if getattr(blk, 'raw_pool', None) is not None:
                    # the raw value behind every pooled activation (srvp_bn_finalize_act raw_pool: first maximum of the stored activations)
                    ob = e['out'].to(torch.bfloat16).float()
                    _, ix = torch.nn.functional.max_pool2d(ob, 2, 2, return_indices=True)
                    put_nhwc(blk.raw_pool, e['raw'].to(torch.bfloat16).float().flatten(2).gather(2, ix.flatten(2)).view_as(ix).cuda())
            if blk.role == 'out':
                blk.x_out.copy_(torch.sigmoid(e['out']).cuda())
    # ---- HIP backward of decoder and encoder on the oracle's gradients
    grads = model._grads()
    model._flat[1].zero_()
    params = model._named_tensors()
    st = L.stream()
    dec, enc = pl['dec'], pl['enc']
    dec.backward(d_x_ref.reshape(T * B, *d_x_ref.shape[2:]).cuda().contiguous(), params, grads, st, None)
    torch.cuda.synchronize()
    dec_err = {k: rel_l2(grads[k], gref[k]) for k in conv_keys if k.startswith('decoder.')}
    # encoder: gradient wrt hx and the skips from the oracle graph
    hx_ref = rec[[b.spec['key'] for b in enc.blocks][-1]]['out']
    # (d_hx / skip gradients are re-derived by autograd through the recorded oracle graph)
    enc_out_tensors = []
    O._conv_block_bf16, O.PRECISION = orig, 'bf16'
    try:
        work2 = dict(fx.state('sd0'))
        leaves2 = {k: work2[k].clone().requires_grad_(True) for k in pkeys}
        work2.update(leaves2)
        hx2, skips2 = O.encode(work2, cfg, x, True, tape.get('t_skip'))
        hx2.retain_grad()
        for s_ in (skips2 or []):
            s_.retain_grad()
        w2 = O.infer_w(work2, cfg, hx2, True, tape.get('t_w'))
        y0, qy0 = O.infer_y(work2, cfg, hx2[:cfg['nt_inf']], tape['eps_y0'])
        y2, z2, qz2, pz2, res2 = O.generate(work2, cfg, y0, hx2, T, ne, tape['eps_z'], True)
        x2 = O.decode(work2, cfg, w2, y2, skips2, True)
        t2 = O.elbo(x, (x2, y2, z2, w2, qy0, qz2, pz2, res2), hp['obs_scale'], hp['beta_y'], hp['beta_z'], hp['l2_res'])
        t2['loss'].backward()
    finally:
        O.PRECISION = 'fp32'
    d_hx = torch.zeros(T * B, srvp_amd.convnet.cpad(cfg['nhx']), device='cuda')
    d_hx[:, :cfg['nhx']] = hx2.grad.reshape(T * B, -1).cuda()
    skip_grads = None
    if cfg['skipco']:
        idx = torch.full((T * B,), -1, dtype=torch.int32, device='cuda')
        idx[pl['skip_sel'].long()] = torch.arange(B, dtype=torch.int32, device='cuda')
        skip_grads = {}
        for i, s_ in enumerate(skips2):
            g_ = s_.grad                                         # (B, C, H, W)
            Cp = srvp_amd.convnet.cpad(g_.shape[1])
            t_ = torch.zeros(B, g_.shape[2], g_.shape[3], Cp, dtype=torch.bfloat16, device='cuda')
            t_[..., :g_.shape[1]] = g_.permute(0, 2, 3, 1).cuda()
            skip_grads[i] = (t_, idx)
    model._flat[1].zero_()
    enc.backward(xg.view(T * B, *xg.shape[2:]), d_hx, skip_grads, params, grads, st, None)
    torch.cuda.synchronize()
    gref2 = {k: leaves2[k].grad for k in conv_keys if k.startswith('encoder.')}
    enc_err = {k: rel_l2(grads[k], gref2[k]) for k in gref2}
    errs = {**dec_err, **enc_err}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    med = sorted(errs.values())[len(errs) // 2]
    report(test='teacher_forced_backward', name=name, worst=worst, median=med)
    assert max(errs.values()) < 0.03 and med < 0.01, (worst, med)


def test_single_rank_collectives(tmp_path):
    """bench.py under torch.distributed.run with ONE rank and every collective forced on (RCCL all-reduce of the BN
    statistics and of the flat gradient slices, rank-0 broadcast): same loss as the plain single-process run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '8',
            '--no-cpu-baseline', '--no-kernel-timing']
    plain = subprocess.run(base, capture_output=True, text=True, timeout=600, cwd=root)
    assert plain.returncode == 0, plain.stderr[-2000:]
    env = dict(os.environ, SRVP_FORCE_COLLECTIVES='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29517'] + base[1:]
    dist = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert dist.returncode == 0, dist.stderr[-2000:]
    l0 = json.loads(plain.stdout.strip().splitlines()[-1])['loss']
    l1 = json.loads(dist.stdout.strip().splitlines()[-1])['loss']
    # (not bit-equal: the fp64 statistics atomics retire in a different order from run to run)
    assert abs(l0 - l1) <= 1e-4 * abs(l0), (l0, l1)
    # the native in-stream RCCL path was the one taken, and the line says what RCCL itself reports about its two communicators
    c = json.loads(dist.stdout.strip().splitlines()[-1])['comm']
    assert 'rccl' in c['statistics'] and 'rccl' in c['gradients'], c
    assert c['rccl_statistics_comm']['ranks'] == 1 and c['rccl_gradients_comm']['ranks'] == 1 and c['rccl_statistics_comm']['version'] > 0, c
    assert c['statistics_allreduce_us']['samples'] == 200 and c['gradient_allreduce']['algbw_GBps'] > 0, c


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it: no WORLD_SIZE in the environment) starts its two
    ranks itself (torch.distributed.run on 127.0.0.1), runs the data-parallel step -- SyncBN statistics + gradient all-reduce -- and
    rank 0 prints the ONE JSON line.  On a 1-GPU box the two ranks share the device, which RCCL refuses: SRVP_DIST_BACKEND=gloo is
    the diagnostic transport for that (the measured configuration is RCCL, one rank per GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    if torch.cuda.device_count() < 2:
        env['SRVP_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--batch', '8', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak'
    assert d['config']['collectives'] and d['config']['global_batch'] == 16 and d['config']['parallelism'] == 'dp2'
    assert 'strong_scaling' in d and d['strong_scaling']['per_gpu_batch'] == 96
    assert abs(d['value'] - 16 * 12 * 3 / (d['ms_per_step'] * 3 / 1e3)) < 1e-6 * d['value']
    assert d['loss'] == d['loss']
    # the line explains its own scaling: transport per exchange, what RCCL reports per communicator (native path only),
    # one statistics all-reduce in stream order, one 95 MB gradient all-reduce; and the step watchdog is armed
    c = d['comm']
    assert c['world'] == 2 and c['statistics'] and c['gradients'] and 'error' not in c, c
    assert c['statistics_allreduce_us']['samples'] == 200 and c['statistics_allreduce_us']['median'] > 0
    assert c['gradient_allreduce']['bytes'] == 95_000_000 and c['gradient_allreduce']['algbw_GBps'] > 0
    assert c['step_watchdog_s'] == 30.0
    if 'rccl' in c['statistics']:
        assert c['rccl_statistics_comm']['ranks'] == 2 and c['rccl_gradients_comm']['ranks'] == 2


def test_bench_multi_rank_headline_is_the_reference_split():
    """`bench.py --gpus N` without --batch (how the driver's scaling runs call it): the headline `value` is BASELINE.json's configuration AS THE
    REFERENCE RUNS IT on N GPUs -- ONE global batch split over the ranks (train.py:218-219), scaling 'strong' -- and the per-GPU-work-fixed line
    is reported beside it as `weak_scaling`.  Two ranks sharing the test box's GPU over gloo, SM-MNIST recipe (128 = 2 x 64)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    if torch.cuda.device_count() < 2:
        env['SRVP_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'smmnist', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-kernel-timing'], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['config']['global_batch'] == 128 and d['config']['per_gpu_batch'] == 64, d['config']
    assert abs(d['value'] - 128 * 15 * 2 / (d['ms_per_step'] * 2 / 1e3)) < 1e-6 * d['value']
    w = d['weak_scaling']
    assert w['scaling'] == 'weak' and w['per_gpu_batch'] == 128 and w['global_batch'] == 256 and w['value'] > 0, w
    assert 'strong_scaling' not in d


def test_resume_train_state(tmp_path):
    """save_train_state / load_train_state continue a run -- same losses as the uninterrupted run (up to the
    order of the fp64 statistics atomics), optimizer moments and LR schedule included; config.json is written as JSON."""
    import json
    import srvp_amd
    from srvp_amd.train import train, save_train_state, load_train_state, write_config
    dev = torch.device('cuda')
    ctor = (64, 1, 8, 16, 4, 4, True, 2, 16, 3, 32, 4, 'vgg')
    opt = srvp_amd.DotDict(dict(n_euler_steps=2, obs_scale=1.0, beta_y=1.0, beta_z=1.0, l2_res=1.0, lr=1e-3))

    def make():
        torch.manual_seed(3)
        m = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
        m.init(1.41)
        m.to(dev).train()
        o = srvp_amd.FusedAdam(m, lr=opt.lr)
        sch = torch.optim.lr_scheduler.LambdaLR(o, lr_lambda=lambda i: max(0, (10 - i) / 10))
        return m, o, sch
    x = torch.rand(4, 3, 1, 64, 64, generator=torch.Generator().manual_seed(5)).to(dev)
    m, o, sch = make()
    torch.manual_seed(11)
    for _ in range(2):
        train(m, o, None, x, dev, opt); sch.step()
    path = str(tmp_path / 'train_state.pt')
    save_train_state(path, m, o, sch, 2, -12.5)
    ref = []
    for _ in range(2):
        ref.append(train(m, o, None, x, dev, opt)[0]); sch.step()
    m2, o2, sch2 = make()
    itr, best = load_train_state(path, m2, o2, sch2, dev)
    assert (itr, best) == (2, -12.5)
    assert sch2.get_last_lr() == pytest.approx(torch.optim.lr_scheduler.LambdaLR.get_last_lr(sch2)) and o2.step_count == 2
    got = []
    for _ in range(2):
        got.append(train(m2, o2, None, x, dev, opt)[0]); sch2.step()
    for a, b in zip(ref, got):
        assert abs(a - b) <= 1e-4 * abs(a), (ref, got)
    write_config(srvp_amd.DotDict(dict(nx=64, archi='vgg', device=[0], lr=3e-4, skipco=True)), str(tmp_path / 'config.json'))
    assert json.load(open(tmp_path / 'config.json'))['archi'] == 'vgg'


@pytest.mark.parametrize('nc', [1, 3])
def test_u8_collate_matches_reference_collate(nc):
    """collate_u8 + srvp_frames_u8_to_f32 == the reference's CPU collate_fn (data/base.py:54-84), bit for bit."""
    import numpy as np
    from srvp_amd.data import collate_u8, frames_from_u8
    rng = np.random.RandomState(7)
    T, B = 5, 3
    videos = [rng.randint(0, 256, size=(T, 64, 64) if nc == 1 else (T, 64, 64, 3)).astype(np.uint8) for _ in range(B)]
    ref = torch.zeros((T, B, nc, 64, 64), dtype=torch.uint8)            # restatement of the reference collate
    for i, v in enumerate(videos):
        if nc == 1:
            ref[:, i, 0] += torch.from_numpy(v)
        else:
            ref[:, i] += torch.from_numpy(np.moveaxis(v, 3, 1))
    ref = ref.float() / 255
    got = frames_from_u8(collate_u8(videos), torch.device('cuda'))
    assert got.shape == ref.shape and torch.equal(got.cpu(), ref)


def test_prefetcher_stages_batches_one_step_ahead():
    """srvp_amd.data.Prefetcher (reference train.py:84,262: `batch.to(device)` at the top of the step): host batches -- stacked uint8 videos of
    collate_u8, float32 (T, B, C, H, W), pinned or not -- come out as device float32 batches in order, equal to the in-line conversion, staged
    on a copy stream while the consumer works on the previous one; device batches pass through; a training loop over it gives the losses of
    the loop over resident batches."""
    import numpy as np
    import srvp_amd
    from srvp_amd.data import Prefetcher, collate_u8, frames_from_u8
    from srvp_amd.train import train
    dev = torch.device('cuda')
    rng = np.random.RandomState(11)
    T, B = 4, 3
    u8s = [collate_u8([rng.randint(0, 256, size=(T, 64, 64, 3)).astype(np.uint8) for _ in range(B)]) for _ in range(5)]
    f32s = [torch.rand(T, B, 3, 64, 64, generator=torch.Generator().manual_seed(i)) for i in range(3)]          # (not pinned)
    want = [frames_from_u8(u, dev) for u in u8s] + [f.to(dev) for f in f32s]
    resident = torch.rand(T, B, 3, 64, 64, device=dev)
    busy = torch.randn(4096, 4096, device=dev)
    got = []
    for i, xb in enumerate(Prefetcher(u8s + f32s + [resident], dev)):
        busy @ busy                                             # the consumer's stream has work queued while the next batch is staged
        assert xb.is_cuda and xb.dtype == torch.float32 and xb.shape == (T, B, 3, 64, 64)
        got.append(xb.clone())
    assert len(got) == 9 and got[-1].data_ptr() != resident.data_ptr() and torch.equal(got[-1], resident)
    for a, b in zip(got[:-1], want):
        assert torch.equal(a, b)
    assert list(Prefetcher([], dev)) == [] and len(Prefetcher(u8s, dev)) == 5

    def losses(feed):
        torch.manual_seed(2)
        m = srvp_amd.StochasticLatentResidualVideoPredictor(64, 3, 8, 16, 4, 4, True, 2, 16, 3, 32, 4, 'vgg')
        m.init(1.41)
        m.to(dev).train()
        o = srvp_amd.FusedAdam(m, lr=1e-3)
        opt = srvp_amd.DotDict(dict(n_euler_steps=2, obs_scale=1.0, beta_y=1.0, beta_z=1.0, l2_res=1.0))
        torch.manual_seed(5)
        return [train(m, o, None, xb, dev, opt)[0] for xb in feed]
    a = losses(Prefetcher(u8s, dev))
    b = losses([frames_from_u8(u, dev) for u in u8s])
    assert all(abs(x - y) <= 1e-4 * abs(y) for x, y in zip(a, b)), (a, b)       # (fp64 statistics atomics: not bitwise)


@pytest.mark.parametrize('archi,nc,skipco', [('vgg', 3, True), ('dcgan', 1, False)])
def test_batched_samples_match_per_sample_forward(archi, nc, skipco):
    """model.sample (one encoding, S futures fanned into the batch dimension of the latent path and the
    decoder) == S separate inference forward passes (reference train.py:170-174 / test.py:237-246) fed the same draws."""
    import srvp_amd
    dev = torch.device('cuda')
    torch.manual_seed(2)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(64, nc, 16, 32, 8, 8, skipco, 2, 32, 2, 32, 2, archi)
    m.init(1.41)
    m.to(dev).train()
    T, B, nt, S = 3, 4, 7, 3
    x = torch.rand(T, B, nc, 64, 64, generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():                                       # settle the BN running statistics, so that in eval mode the
        for _ in range(25):                                     # frames really depend on the latent draws
            m(x, T, 0.5)
    m.eval()
    g = torch.Generator().manual_seed(9)
    eps_y0 = torch.randn(S * B, 8, generator=g).to(dev)
    eps_z = torch.randn(nt - 1, S * B, 8, generator=g).to(dev)
    xs = m.sample(x, nt, S, dt=0.5, tape=dict(eps_y0=eps_y0, eps_z=eps_z))
    assert xs.shape == (nt, S, B, nc, 64, 64)
    for s in range(S):
        tape = dict(eps_y0=eps_y0[s * B:(s + 1) * B].contiguous(), eps_z=eps_z[:, s * B:(s + 1) * B].contiguous())
        ref = m(x, nt, 0.5, tape=tape)[0]
        assert torch.allclose(xs[:, s], ref, atol=2e-3, rtol=0), (s, (xs[:, s] - ref).abs().max().item())
    assert (xs[:, 0] - xs[:, 1]).abs().max() > 2e-2             # the samples do differ
    # decoder chunking (chunk = 2 of the 3 samples per pass; the encoder and the latent path still run once): same frames, bit for bit
    # in the samples the two passes do not share and in the ones they do
    xc = m.sample(x, nt, S, dt=0.5, tape=dict(eps_y0=eps_y0, eps_z=eps_z), chunk=2)
    assert torch.equal(xc, xs)
    assert torch.equal(m.sample(x, nt, S, dt=0.5, tape=dict(eps_y0=eps_y0, eps_z=eps_z), chunk=1), xs)


def test_sample_more_rows_than_one_persistent_launch():
    """model.sample with more (video, future) rows than ONE co-resident generation launch holds (16 clusters of 32 rows at 512 hidden units: 600
    rows = 19 tiles run as two launches, tile0 > 0 in the second) with and without decoder chunking, and against the per-layer launch sequence."""
    import srvp_amd
    from srvp_amd import latent as LT
    dev = torch.device('cuda')
    torch.manual_seed(3)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(64, 1, 8, 32, 20, 20, False, 2, 64, 2, 512, 4, 'dcgan')
    m.init(0.8)
    m.to(dev).train()
    T, B, nt, S = 3, 2, 5, 300
    x = torch.rand(T, B, 1, 64, 64, generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():
        for _ in range(10):
            m(x, T, 0.5)
    m.eval()
    g = torch.Generator().manual_seed(9)
    tape = dict(eps_y0=torch.randn(S * B, 20, generator=g).to(dev), eps_z=torch.randn(nt - 1, S * B, 20, generator=g).to(dev))
    whole = m.sample(x, nt, S, dt=0.5, tape=tape)
    assert m._last_sample_lat._rd.fused_ws                      # the persistent generation launches were taken
    chunked = m.sample(x, nt, S, dt=0.5, tape=tape, chunk=60)
    assert torch.equal(chunked, whole)
    LT.ROLLOUT_GEN_FUSED = False
    try:
        m.drop_sample_plans()
        launches = m.sample(x, nt, S, dt=0.5, tape=tape, chunk=60)
        assert not m._last_sample_lat._rd.fused_ws
    finally:
        LT.ROLLOUT_GEN_FUSED = True
        m.drop_sample_plans()
    assert (launches - chunked).abs().max().item() < 2e-3
    assert (whole[:, 0] - whole[:, 299]).abs().max() > 1e-5     # the futures do differ (an untrained decoder barely shows it)


def test_evaluate_best_of_n_psnr():
    """train.evaluate (reference train.py:132-189): best-of-n_samples_test PSNR per video over the predicted frames, with the
    samples drawn by model.sample and the PSNR by the device metrics kernel -- against the same selection done with the
    float64 oracle metrics on the same samples."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import evaluate
    dev = torch.device('cuda')
    torch.manual_seed(2)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(64, 1, 16, 32, 8, 8, True, 2, 32, 2, 32, 2, 'vgg')
    m.init(1.41)
    m.to(dev).train()
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(6, 3, 1, 64, 64, generator=g) for _ in range(2)]
    with torch.no_grad():
        for _ in range(25):
            m(batches[0].to(dev), 6, 0.5)
    m.eval()
    opt = srvp_amd.DotDict(dict(nt_cond=2, n_iter_test=2, n_samples_test=3, n_euler_steps=2))
    torch.manual_seed(77)
    got = evaluate(m, batches, dev, opt)
    torch.manual_seed(77)
    tot = 0.0
    for x in batches:
        xs = m.sample(x[:2].to(dev), 6, 3, dt=0.5).cpu()                      # (nt, S, B, C, H, W)
        ps = torch.stack([O.video_psnr(xs[:, s], x).mean(dim=(0, 2)) for s in range(3)])      # (S, B)
        best = ps.argmax(0)
        bx = torch.stack([xs[:, best[b], b] for b in range(3)], 1)
        tot += O.video_psnr(bx, x)[2:].mean().item() * 3
    want = -tot / 6
    assert abs(got - want) <= 1e-4 * abs(want), (got, want)


def test_cli_trains_on_smmnist(tmp_path):
    """`python -m srvp_amd.train` (the reference's train.py CLI) end to end on the device-side Stochastic Moving-MNIST
    generator: fake MNIST IDX file -> batches -> train steps -> validation (model.sample + device PSNR) ->
    model.pt / model_best.pt / model_<itr>.pt / config.json / train_state.pt, exit status 0."""
    import gzip
    import json
    import struct
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(0)
    imgs = np.zeros((200, 28, 28), np.uint8)
    for i in range(200):
        r, c = rng.randint(4, 16, 2)
        imgs[i, r:r + 8, c:c + 6] = rng.randint(120, 256)
    os.makedirs(tmp_path / 'MNIST' / 'raw')
    with gzip.open(tmp_path / 'MNIST' / 'raw' / 'train-images-idx3-ubyte.gz', 'wb') as f:
        f.write(struct.pack('>iiii', 2051, 200, 28, 28) + imgs.tobytes())
    save = tmp_path / 'run'
    cmd = [sys.executable, '-m', 'srvp_amd.train', '--device', '0', '--seed', '3', '--dataset', 'smmnist', '--data_dir', str(tmp_path),
           '--save_path', str(save), '--nc', '1', '--seq_len', '6', '--nt_cond', '3', '--nt_inf', '3', '--ny', '8', '--nz', '8',
           '--archi', 'dcgan', '--nf', '16', '--nhx', '32', '--nh_inf', '32', '--nh_res', '32', '--nlayers_inf', '2', '--nlayers_res', '2',
           '--batch_size', '8', '--batch_size_test', '4', '--n_iter_test', '2', '--n_samples_test', '3', '--seq_len_test', '9',
           '--lr_scheduling_burnin', '6', '--lr_scheduling_n_iter', '4', '--val_interval', '5', '--chkpt_interval', '5',
           '--n_euler_steps', '2', '--beta_z', '2']
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for name in ('model.pt', 'model_best.pt', 'model_5.pt', 'model_10.pt', 'config.json', 'train_state.pt'):
        assert (save / name).exists(), (name, os.listdir(save))
    assert json.load(open(save / 'config.json'))['dataset'] == 'smmnist'
    sd = torch.load(save / 'model.pt', map_location='cpu')
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


@pytest.mark.parametrize('label,nc,nt_cond,nt,nt_inf,ne', [('kth', 1, 10, 20, 3, 2), ('human', 3, 8, 53, 3, 2)])
def test_full_width_eval_rollout_vs_oracle(label, nc, nt_cond, nt, nt_inf, ne):
    """SURVEY §8 C3 / C5 shapes, inference: conditioning frames -> posterior steps -> prior rollout far past the data (53 frames =
    104 Euler steps for Human3.6M) at full layer widths, BN running statistics settled by a few training-mode passes first;
    frames and latent states against the CPU oracle (fp32 reference arithmetic) on the same draws, and model.sample on the same
    draws against forward."""
    import srvp_amd
    from oracle import srvp_oracle as O
    torch.manual_seed(1)
    ctor = (64, nc, 64, 128, 50, 50, True, nt_inf, 256, 3, 512, 4, 'vgg')
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    # contractive residual function: with the training-time gain (1.2) an UNTRAINED dynamics MLP grows |y| ~1.3x per frame
    # (1e6 by frame 53), the pre-sigmoid logits become differences of huge terms and bf16 rounding alone flips saturated pixels --
    # the fp32 and the bf16-model oracles then disagree with each other just as much (tools/eval_shape_probe.py)
    model.init(0.6)
    model.cuda().train()
    g = torch.Generator().manual_seed(321)
    B = 2
    xw = torch.rand(nt_cond, 6, nc, 64, 64, generator=g)
    with torch.no_grad():
        for _ in range(12):
            model(xw.cuda(), nt_cond, 1 / ne)
    model.eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = xw[:, :B].contiguous()
    tape = dict(eps_y0=torch.randn(B, 50, generator=g), eps_z=torch.randn(nt - 1, B, 50, generator=g))
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = O.forward(sd, O.make_cfg(*ctor), x, nt, ne, tape, training=False)
        out = model(x.cuda(), nt, 1 / ne, tape=tape)
    e = dict(x=max_abs(out[0], ref[0]), y=rel_l2(out[1], ref[1]), z=rel_l2(out[2], ref[2]), w=rel_l2(out[3], ref[3]),
             res=rel_l2(out[7], ref[7]))
    report(test='full_width_eval_rollout', label=label, errs=e)
    assert out[0].shape == (nt, B, nc, 64, 64) and out[7].shape[0] == ne * (nt - 1)
    assert e['x'] < 3e-2 and e['y'] < 2e-2 and e['w'] < 2e-2 and e['res'] < 5e-2, e
    xs = model.sample(x.cuda(), nt, 1, dt=1 / ne, tape=tape)
    assert max_abs(xs[:, 0], out[0]) < 2e-3


@pytest.mark.parametrize('archi,nc,skipco', [('vgg', 3, True), ('dcgan', 1, False)])
def test_training_reduces_the_elbo(archi, nc, skipco):
    """End to end: 60 Adam steps on one fixed batch through train.train (forward + ELBO + backward + Adam, all on the HIP
    path) must drive the loss down -- a whole-pipeline check on top of the per-step parity tests (a sign error in any
    gradient path or in the optimizer shows up here)."""
    import srvp_amd
    from srvp_amd.train import train
    dev = torch.device('cuda')
    torch.manual_seed(0)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(64, nc, 16, 32, 8, 8, skipco, 2, 32, 2, 32, 2, archi)
    m.init(1.2)
    m.to(dev).train()
    o = srvp_amd.FusedAdam(m, lr=1e-3)
    opt = srvp_amd.DotDict(dict(n_euler_steps=2, obs_scale=1.0, beta_y=1.0, beta_z=1.0, l2_res=1.0))
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.arange(64.), torch.arange(64.), indexing='ij')
    c = torch.rand(6, 4, 1, 2, generator=g) * 40 + 12
    x = torch.exp(-((yy - c[..., 0, None, None]) ** 2 + (xx - c[..., 1, None, None]) ** 2) / 60).expand(6, 4, nc, 64, 64).contiguous().to(dev)
    losses = [train(m, o, None, x, dev, opt)[0] for _ in range(60)]
    assert all(l == l for l in losses), losses[:5]
    # the Gaussian NLL carries the constant 0.5 log(2 pi) per pixel (obs_scale = 1): compare what can actually be reduced
    const = 6 * nc * 64 * 64 * 0.9189385332
    first, last = sum(losses[:5]) / 5 - const, sum(losses[-5:]) / 5 - const
    assert last < 0.5 * first, (first, last)


def test_bench_contract():
    """bench.py prints exactly ONE line on stdout, a JSON object with the driver's fields plus `roofline` (and, without
    --no-cpu-baseline, `cpu_baseline`); everything else (library banners included) goes to stderr."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '4', '--warmup', '1', '--batch', '8', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in d, k
    assert d['unit'] == 'frames/s' and d['n_gpus'] == 1 and d['steps'] == 4 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert abs(d['value'] - 8 * 12 * 4 / (d['ms_per_step'] * 4 / 1e3)) < 1e-6 * d['value']
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('B', [3, 8])
def test_fused_bn_backward_reduction_equals_separate_pass(B):
    """srvp_conv_desc.bnr_* (BatchNorm-backward sums of the producer accumulated in the epilogue of the consumer's data-gradient
    launch) against the separate srvp_bn_bwd_reduce pass, at full VGG widths:
      * sharp: after a training step, for every producer whose reduction was fused, the separate kernel is run on the very tensors
        the step left behind (dA = the consumer's data gradient, raw, coefficients) -- the two pairs of per-channel sums agree to
        1e-5 (same terms g = bf16(dA) f'(.), g xhat; fp32 partial sums in another order);
      * end to end: the same step with the fusion switched off gives the same loss and parameter gradients within the band a 1-ulp
        change of a BatchNorm coefficient opens on 12-32 frames of an untrained bf16 network."""
    import ctypes as C
    import srvp_amd
    from srvp_amd import convnet, _lib as L
    from srvp_amd.train import fused_step
    dev = torch.device('cuda')
    ctor = (64, 3, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg')
    T, ne = 4, 2
    g = torch.Generator().manual_seed(31)
    x = torch.rand(T, B, 3, 64, 64, generator=g).to(dev)
    tape = dict(t_skip=torch.randint(T, (B,), generator=g), t_w=torch.stack([torch.randperm(T, generator=g)[:2] for _ in range(B)], 1),
                eps_y0=torch.randn(B, 50, generator=g), eps_z=torch.randn(T - 1, B, 50, generator=g))
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0))
    grads, fused_layers = {}, {}
    old = convnet.BN_FUSED_REDUCE
    try:
        for mode in (True, False):
            convnet.BN_FUSED_REDUCE = mode
            torch.manual_seed(1)
            m = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
            m.init(1.41)
            m.to(dev).train()
            o = srvp_amd.FusedAdam(m, lr=1e-3)
            o.zero_grad()
            acc = fused_step(m, x, opt, tape=tape)
            torch.cuda.synchronize()
            grads[mode] = ({k: p.grad.detach().clone() for k, p in m.named_parameters()}, acc.cpu())
            pl = m._last_plan
            nets = (pl['enc'], pl['dec'])
            fused_layers[mode] = sum(1 for net in nets for b in net.blocks if getattr(b, '_reduce_fused', False))
            if mode:
                for net in nets:
                    for q in net.blocks:
                        p = next((b for b in net.blocks if b.out is not None and q.srcs and b.out is q.srcs[0]), None)
                        if p is None or not getattr(p, '_reduce_fused', False):
                            continue
                        d = net._bnbwd_desc(p, dict(t=q.dcat, mode=0, cstride=q.dcat_c, coff=0, border=0))
                        ref = torch.zeros_like(p.red)
                        L.call('srvp_bn_bwd_reduce', C.byref(d), L.ptr(ref), L.stream())
                        torch.cuda.synchronize()
                        err = ((p.red - ref).abs().max() / (ref.abs().max() + 1e-30)).item()
                        assert err < 1e-5, (p.spec['key'], err)
    finally:
        convnet.BN_FUSED_REDUCE = old
    assert fused_layers[True] >= 12 and fused_layers[False] == 0, fused_layers     # 6 encoder (+ pooled stages) + 5 + 4 (sub-pixel consumers) decoder layers
    assert torch.allclose(grads[True][1], grads[False][1], rtol=1e-12)              # the forward is the same code
    worst = max(((grads[True][0][k] - grads[False][0][k]).norm() / (grads[False][0][k].norm() + 1e-30)).item() for k in grads[True][0])
    assert worst < 5e-2, worst


@pytest.mark.gpu
@pytest.mark.parametrize('B', [3, 8])
def test_pooled_bn_backward_reduction_rides_the_consumer(B):
    """Pooled VGG stages (conv.py:204-222): the forward stores the raw value at every window's arg-max (srvp_bn_finalize_act raw_pool), the
    consumer's data-gradient launch accumulates the BatchNorm-backward sums from (pooled dA, raw_pool) in its epilogue, and the skip-connection
    gradient of the B selected frames is added by a da_mode 3 reduction.  Sharp: after a training step, the separate pass (da_mode 2: arg-max
    re-derived from four raw pixels per window, + da2) run on the tensors the step left behind gives the same two sums per channel to 1e-5.
    End to end: the step with SRVP_POOL_FUSED_REDUCE off gives the same loss and the same gradients within the band a 1-ulp change of a
    BatchNorm coefficient opens on an untrained bf16 network."""
    import ctypes as C
    import srvp_amd
    from srvp_amd import convnet, _lib as L
    from srvp_amd.train import fused_step
    dev = torch.device('cuda')
    ctor = (64, 3, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg')
    T, ne = 4, 2
    g = torch.Generator().manual_seed(33)
    x = torch.rand(T, B, 3, 64, 64, generator=g).to(dev)
    tape = dict(t_skip=torch.randint(T, (B,), generator=g), t_w=torch.stack([torch.randperm(T, generator=g)[:2] for _ in range(B)], 1),
                eps_y0=torch.randn(B, 50, generator=g), eps_z=torch.randn(T - 1, B, 50, generator=g))
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0))
    grads, npooled = {}, {}
    old = convnet.POOL_FUSED_REDUCE
    try:
        for mode in (True, False):
            convnet.POOL_FUSED_REDUCE = mode
            torch.manual_seed(1)
            m = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
            m.init(1.41)
            m.to(dev).train()
            o = srvp_amd.FusedAdam(m, lr=1e-3)
            o.zero_grad()
            acc = fused_step(m, x, opt, tape=tape)
            torch.cuda.synchronize()
            grads[mode] = ({k: p.grad.detach().clone() for k, p in m.named_parameters()}, acc.cpu())
            pl = m._last_plan
            enc, dec = pl['enc'], pl['dec']
            pooled = [b for b in enc.blocks if b.pool is not None and getattr(b, '_reduce_fused', False)]
            npooled[mode] = len(pooled)
            if mode:
                sk = {stage: (dsel, pl['skip_idx']) for stage, dsel in dec.skip_grads(T, B, L.stream()).items()}
                for p in pooled:
                    q = next(b for b in enc.blocks if b.srcs and b.srcs[0] is p.pool)
                    assert p.raw_pool is not None and tuple(p.raw_pool.shape) == tuple(q.dcat.shape)
                    da = dict(t=q.dcat, mode=2, cstride=q.dcat_c, coff=0, border=0)
                    if p.spec['skip_out'] is not None:
                        da['da2'], da['da2_idx'] = sk[3 - p.spec['skip_out']]
                    d = enc._bnbwd_desc(p, da)
                    ref = torch.zeros_like(p.red)
                    L.call('srvp_bn_bwd_reduce', C.byref(d), L.ptr(ref), L.stream())
                    torch.cuda.synchronize()
                    err = ((p.red - ref).abs().max() / (ref.abs().max() + 1e-30)).item()
                    assert err < 1e-5, (p.spec['key'], err)
                    # and raw_pool is the raw value behind every pooled activation
                    sc, sh = p.coef[0].float(), p.coef[1].float()
                    a = torch.nn.functional.leaky_relu(p.raw_pool.float() * sc + sh, 0.2).to(torch.bfloat16).float()
                    diff = (a - p.pool.interior().float()).abs()              # (the kernel's fused multiply-add vs two roundings here: a bf16 ulp, rarely)
                    assert (diff > 0).float().mean().item() < 1e-3 and diff.max().item() <= 2 ** -7 * max(1.0, a.abs().max().item()), p.spec['key']
    finally:
        convnet.POOL_FUSED_REDUCE = old
    assert npooled[True] >= 3 and npooled[False] == 0, npooled          # the 64x64, 32x32 and 16x16 stages (the 8x8 stage feeds the 4x4 -> 1x1 split-K launch)
    assert torch.allclose(grads[True][1], grads[False][1], rtol=1e-12)   # the forward computes the same values
    worst = max(((grads[True][0][k] - grads[False][0][k]).norm() / (grads[False][0][k].norm() + 1e-30)).item() for k in grads[True][0])
    assert worst < 5e-2, worst


@pytest.mark.gpu
@pytest.mark.parametrize('nc', [3, 1])
def test_first_block_weight_gradient_without_stored_output_gradient(nc):
    """srvp_conv_in_wgrad_bn: the first block's BatchNorm-backward apply and its weight gradient as one launch -- the gradient wrt the
    block's pre-BatchNorm output feeds nothing else, so it is formed on the way into LDS from dA and raw and never stored -- against the
    two launches (srvp_bn_bwd_finalize_apply writing it, srvp_conv_in_wgrad reading it back).  Same expressions, same bf16 rounding of the
    gradient tile, same MFMA schedule: the first block's weight / BatchNorm gradients agree to 1e-5, every other gradient is untouched."""
    import srvp_amd
    from srvp_amd import convnet
    from srvp_amd.train import fused_step
    dev = torch.device('cuda')
    ctor = (64, nc, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg')
    T, B, ne = 4, 6, 2
    g = torch.Generator().manual_seed(35)
    x = torch.rand(T, B, nc, 64, 64, generator=g).to(dev)
    tape = dict(t_skip=torch.randint(T, (B,), generator=g), t_w=torch.stack([torch.randperm(T, generator=g)[:2] for _ in range(B)], 1),
                eps_y0=torch.randn(B, 50, generator=g), eps_z=torch.randn(T - 1, B, 50, generator=g))
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0))
    grads = {}
    old = convnet.IN_WGRAD_BN
    try:
        for mode in (True, False):
            convnet.IN_WGRAD_BN = mode
            torch.manual_seed(1)
            m = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
            m.init(1.41)
            m.to(dev).train()
            o = srvp_amd.FusedAdam(m, lr=1e-3)
            o.zero_grad()
            fused_step(m, x, opt, tape=tape)
            torch.cuda.synchronize()
            grads[mode] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    finally:
        convnet.IN_WGRAD_BN = old
    first = [k for k in grads[True] if k.startswith('encoder.conv.0.0.')]
    assert len(first) == 3, first                                       # conv weight, BatchNorm weight and bias
    for k in grads[True]:
        err = ((grads[True][k] - grads[False][k]).norm() / (grads[False][k].norm() + 1e-30)).item()
        if k in first:
            assert err < 1e-5, (k, err)
        else:
            # (everything else is computed by the same launches in both runs; fp64 / fp32 atomics in another order)
            assert err < 2e-3, (k, err)
