"""
Edge cases of the hot path against the oracle (fp32 mode, tight): the smallest problem sizes the reference accepts --
one video, the minimum number of frames (T = nt_inf), four Euler sub-steps per frame, and prediction from a conditioning window
exactly nt_inf long (one frame interval, and far beyond the data).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _model(ctor, seed=3, gain=1.2):
    import srvp_amd
    torch.manual_seed(seed)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    m.init(gain)
    return m


@pytest.mark.parametrize('archi,nc,skipco,T,B,ne,nt_inf', [('vgg', 3, True, 3, 1, 4, 2), ('dcgan', 1, False, 3, 1, 1, 3), ('vgg', 1, True, 3, 2, 2, 1)])
def test_minimal_training_step(archi, nc, skipco, T, B, ne, nt_inf):
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import elbo_terms_and_grads
    ctor = (64, nc, 8, 16, 4, 5, skipco, nt_inf, 16, 3, 32, 3, archi)
    m = _model(ctor)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.rand(T, B, nc, 64, 64, generator=g)
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                eps_y0=torch.randn(B, 4, generator=g), eps_z=torch.randn(T - 1, B, 5, generator=g))
    if skipco:
        tape['t_skip'] = torch.randint(T, (B,), generator=g)
    hp = dict(obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    scal, outs_ref, grads_ref = O.train_step(sd, O.make_cfg(*ctor), x, ne, tape, hp)
    m = m.cuda().train().set_precision('fp32')
    m.flatten_parameters_(); m._grads(); m._flat[1].zero_()
    xg = x.cuda()
    outs = m._forward_impl(xg, T, ne, tape, training=True)
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    acc, gr = elbo_terms_and_grads(m, xg, outs, opt)
    m._backward_impl(gr[0], None, None, gr[1], gr[2], gr[3], gr[4])
    nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
    loss = (nll + kl_y0 + kl_z + l2) / B
    assert abs(loss - scal['loss']) <= 1e-5 * abs(scal['loss']), (loss, scal['loss'])
    assert outs[7].shape[0] == ne * (T - 1)
    # a batch of one or two frames per BatchNorm layer is as ill-conditioned as it gets (with two samples at 1x1 resolution the
    # normalised values are +-1 whatever the input: the gradient of the convolution in front is zero up to eps effects, so its
    # "relative" error is noise): gradients to 2e-2 of max(own norm, 1 % of the median tensor norm)
    norms = sorted(v.double().norm().item() for v in grads_ref.values())
    floor = 1e-2 * norms[len(norms) // 2]
    bad = {}
    for k, p in m.named_parameters():
        e = (p.grad.double().cpu() - grads_ref[k].double()).norm().item()
        if e > 2e-2 * max(grads_ref[k].double().norm().item(), floor):
            bad[k] = e / grads_ref[k].double().norm().item()
    assert not bad, bad


@pytest.mark.parametrize('B', [1, 3])
@pytest.mark.parametrize('nt', [2, 7])
def test_eval_nt_edges(nt, B):
    """Inference from exactly nt_inf conditioning frames: nt = 2 (one frame interval) and nt far beyond the data.  (nt = 1 is not
    a valid call of the reference: module/srvp.py:412 stacks an empty list of residuals.)"""
    from oracle import srvp_oracle as O
    ctor = (64, 1, 8, 16, 4, 5, True, 2, 16, 3, 32, 3, 'vgg')
    m = _model(ctor)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, B, 1, 64, 64, generator=g)                      # exactly nt_inf conditioning frames
    tape = dict(eps_y0=torch.randn(B, 4, generator=g), eps_z=torch.randn(max(nt - 1, 1), B, 5, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = O.forward(sd, O.make_cfg(*ctor), x, nt, 2, dict(eps_y0=tape['eps_y0'], eps_z=tape['eps_z'][:max(nt - 1, 0)]), training=False)
    m = m.cuda().eval().set_precision('fp32')
    out = m(x.cuda(), nt, 0.5, tape=tape)
    assert out[0].shape == (nt, B, 1, 64, 64)
    assert (out[0].cpu() - ref[0]).abs().max().item() <= 2e-5
    assert rel_l2(out[1], ref[1]) <= 1e-5
    # one future from one encoding == the forward
    xs = m.sample(x.cuda(), nt, 1, dt=0.5, tape=tape)
    assert (xs[:, 0] - out[0]).abs().max().item() <= 1e-5
    # two futures: the second one on other draws, against its own inference forward
    g2 = torch.Generator().manual_seed(9)
    e2y, e2z = torch.randn(B, 4, generator=g2), torch.randn(max(nt - 1, 1), B, 5, generator=g2)
    xs2 = m.sample(x.cuda(), nt, 2, dt=0.5, tape=dict(eps_y0=torch.cat([tape['eps_y0'], e2y]), eps_z=torch.cat([tape['eps_z'], e2z], 1)))
    out2 = m(x.cuda(), nt, 0.5, tape=dict(eps_y0=e2y, eps_z=e2z))
    assert (xs2[:, 0] - out[0]).abs().max().item() <= 1e-5 and (xs2[:, 1] - out2[0]).abs().max().item() <= 1e-5


def _one_train_step(m, x, tape, ne, hp):
    import srvp_amd
    from srvp_amd.train import elbo_terms_and_grads
    m.train()
    m.flatten_parameters_(); m._grads(); m._flat[1].zero_()
    xg = x.cuda()
    outs = m._forward_impl(xg, x.shape[0], ne, tape, training=True)
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    acc, gr = elbo_terms_and_grads(m, xg, outs, opt)
    m._backward_impl(gr[0], None, None, gr[1], gr[2], gr[3], gr[4])
    torch.cuda.synchronize()
    return acc.cpu().clone(), m._flat[1].detach().cpu().clone()


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_interleaved_shapes_modes_and_precisions(precision):
    """Plans, packed weights, descriptor tables and workspaces are cached per problem size: a training step must give the same
    result whether it is the first thing a model does or comes after other batch sizes, inference passes, multi-sample rollouts
    and a precision switch (stale-cache bugs show up here)."""
    ctor = (64, 3, 8, 16, 4, 5, True, 2, 16, 3, 64, 4, 'vgg')
    hp = dict(obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    ne = 2
    g = torch.Generator().manual_seed(21)

    def problem(T, B):
        x = torch.rand(T, B, 3, 64, 64, generator=g)
        tape = dict(t_skip=torch.randint(T, (B,), generator=g), t_w=torch.stack([torch.randperm(T, generator=g)[:2] for _ in range(B)], 1),
                    eps_y0=torch.randn(B, 4, generator=g), eps_z=torch.randn(T - 1, B, 5, generator=g))
        return x, tape
    pA, pB = problem(4, 5), problem(3, 34)
    # reference: a fresh model per problem (BN running statistics do not enter a training-mode step)
    want = {}
    for name, (x, tape) in (('A', pA), ('B', pB)):
        m = _model(ctor).cuda().set_precision(precision)
        want[name] = _one_train_step(m, x, tape, ne, hp)
    m = _model(ctor).cuda().set_precision(precision)
    seq = ['A', 'B', 'eval', 'A', 'sample', 'B', 'switch', 'A', 'B']
    other = 'fp32' if precision == 'bf16' else 'bf16'
    for step in seq:
        if step in ('A', 'B'):
            x, tape = pA if step == 'A' else pB
            acc, grad = _one_train_step(m, x, tape, ne, hp)
            wa, wg = want[step]
            # (the fp64 BatchNorm atomics retire in a different order from run to run: equal to summation order, not bit for bit)
            assert torch.allclose(acc, wa, rtol=1e-6, atol=0), (step, acc, wa)
            e = ((grad.double() - wg.double()).norm() / wg.double().norm()).item()
            assert e <= (1e-4 if precision == 'fp32' else 0.12), (step, e)
        elif step == 'eval':
            m.eval()
            m(pA[0][:2].cuda(), 6, 1 / ne)
        elif step == 'sample':
            m.eval()
            m.sample(pB[0][:2, :3].cuda(), 5, 3, dt=1 / ne)
        else:
            m.set_precision(other)
            _one_train_step(m, pA[0], pA[1], ne, hp)
            m.set_precision(precision)


SWEEP = [
    # archi, nc, nf, nhx, ny, nz, skipco, nt_inf, nh_inf, nl_inf, nh_res, nl_res, T, B, ne
    ('vgg', 3, 16, 20, 6, 3, True, 3, 40, 2, 96, 2, 5, 3, 4),
    ('vgg', 1, 24, 33, 7, 9, False, 1, 24, 1, 64, 3, 3, 4, 1),
    ('dcgan', 3, 16, 48, 10, 10, True, 2, 32, 3, 128, 4, 4, 5, 2),
    ('dcgan', 1, 20, 17, 3, 8, False, 4, 20, 2, 32, 5, 6, 2, 2),
    ('vgg', 3, 8, 64, 12, 4, True, 2, 16, 4, 160, 3, 3, 7, 1),
    ('dcgan', 3, 32, 32, 5, 5, False, 2, 64, 3, 64, 2, 3, 9, 4),
]


@pytest.mark.parametrize('cfg', SWEEP, ids=[f'{c[0]}_nc{c[1]}_nf{c[2]}_nhx{c[3]}_y{c[4]}z{c[5]}_s{int(c[6])}_res{c[10]}x{c[11]}_T{c[12]}B{c[13]}e{c[14]}' for c in SWEEP])
def test_config_sweep_vs_oracle_fp32(cfg):
    """Unusual widths / depths / paddings the fixtures do not reach (channel counts that are not multiples of 32, odd latent sizes,
    1..5-layer MLPs, 1..4 Euler sub-steps, nt_inf from 1 to T-...): one fp32-mode training step against the oracle."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import elbo_terms_and_grads
    archi, nc, nf, nhx, ny, nz, skipco, nt_inf, nh_inf, nl_inf, nh_res, nl_res, T, B, ne = cfg
    ctor = (64, nc, nf, nhx, ny, nz, skipco, nt_inf, nh_inf, nl_inf, nh_res, nl_res, archi)
    m = _model(ctor, seed=11)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    x = torch.rand(T, B, nc, 64, 64, generator=g)
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                eps_y0=torch.randn(B, ny, generator=g), eps_z=torch.randn(T - 1, B, nz, generator=g))
    if skipco:
        tape['t_skip'] = torch.randint(T, (B,), generator=g)
    hp = dict(obs_scale=0.6, beta_y=1.0, beta_z=2.0, l2_res=0.5)
    torch.set_num_threads(8)
    scal, outs_ref, grads_ref = O.train_step(sd, O.make_cfg(*ctor), x, ne, tape, hp)
    m = m.cuda().train().set_precision('fp32')
    m.flatten_parameters_(); m._grads(); m._flat[1].zero_()
    xg = x.cuda()
    outs = m._forward_impl(xg, T, ne, tape, training=True)
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    acc, gr = elbo_terms_and_grads(m, xg, outs, opt)
    m._backward_impl(gr[0], None, None, gr[1], gr[2], gr[3], gr[4])
    nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
    loss = (nll + hp['beta_y'] * kl_y0 + hp['beta_z'] * kl_z + hp['l2_res'] * l2) / B
    assert abs(loss - scal['loss']) <= 1e-5 * abs(scal['loss']), (loss, scal['loss'])
    for o, r in zip(outs, outs_ref):
        assert rel_l2(o, r) <= 1e-4
    norms = sorted(v.double().norm().item() for v in grads_ref.values())
    floor = 1e-2 * norms[len(norms) // 2]
    bad = {}
    for k, p in m.named_parameters():
        e = (p.grad.double().cpu() - grads_ref[k].double()).norm().item()
        if e > 1e-2 * max(grads_ref[k].double().norm().item(), floor):
            bad[k] = e / (grads_ref[k].double().norm().item() + 1e-30)
    assert not bad, bad


@pytest.mark.parametrize('archi,nc,skipco,ne', [('vgg', 3, True, 2), ('dcgan', 1, False, 4)])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_remove_intermediate_false_decodes_every_euler_substep(archi, nc, skipco, ne, precision):
    """forward(..., remove_intermediate=False) (srvp.py:402,415-470: generation at 1/dt times the frame rate): (nt - 1) / dt + 1 states
    and frames, conditioning on T frames and predicting beyond them, against the oracle on the same draws; the integer-time rows
    equal the default call."""
    from oracle import srvp_oracle as O
    T, B, nt = 4, 3, 6
    ctor = (64, nc, 8, 16, 6, 5, skipco, 2, 16, 3, 32, 3, archi)
    m = _model(ctor, seed=5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    x = torch.rand(T, B, nc, 64, 64, generator=g)
    tape = dict(eps_y0=torch.randn(B, 6, generator=g), eps_z=torch.randn(nt - 1, B, 5, generator=g))
    ref = O.forward(sd, O.make_cfg(*ctor), x, nt, ne, tape, False, remove_intermediate=False)
    m = m.cuda().eval().set_precision(precision)
    tg = {k: v.cuda() for k, v in tape.items()}
    outs = m(x.cuda(), nt, 1.0 / ne, remove_intermediate=False, tape=tg)
    S = (nt - 1) * ne
    assert outs[0].shape == (S + 1, B, nc, 64, 64) and outs[1].shape == (S + 1, B, 6) and outs[7].shape == (S, B, 6)
    tol = 2e-5 if precision == 'fp32' else 3e-2
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert (o is None) == (r is None)
        if o is not None:
            assert o.shape == r.shape, i
            assert rel_l2(o, r) <= tol, (i, rel_l2(o, r))
    dflt = m(x.cuda(), nt, 1.0 / ne, tape=tg)
    assert torch.equal(dflt[1], outs[1][::ne])
    assert rel_l2(dflt[0], outs[0][::ne]) <= (1e-6 if precision == 'fp32' else 2e-2)     # (other batch statistics? no: eval mode; other tile shapes)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_remove_intermediate_false_matches_reference_fixture(precision):
    """The same call against outputs of the REAL reference (tests/golden/dense_*.npz, tests/make_golden.py --dense)."""
    import srvp_amd
    from golden_util import Fixture, OUT_NAMES, dense_fixture_names
    for name in dense_fixture_names():
        fx = Fixture(name)
        nt_cond, nt, ne = int(fx.z['nt_cond']), int(fx.z['nt']), int(fx.z['n_euler'])
        m = srvp_amd.StochasticLatentResidualVideoPredictor(*fx.meta['ctor'])
        m.load_state_dict(fx.state('sd0'))
        m = m.cuda().eval().set_precision(precision)
        tape = {k: v.cuda() for k, v in fx.tape().items()}
        outs = m(fx.t('x')[:nt_cond].cuda(), nt, 1.0 / ne, remove_intermediate=False, tape=tape)
        for n, o in zip(OUT_NAMES, outs):
            if o is None:
                assert not fx.has('out.' + n)
                continue
            e = rel_l2(o, fx.t('out.' + n))
            assert e <= (2e-5 if precision == 'fp32' else 3e-2), (name, n, e)


@pytest.mark.parametrize('archi,nc,skipco', [('vgg', 3, True), ('dcgan', 1, False)])
def test_four_training_steps_track_the_oracle(archi, nc, skipco):
    """train.train four times in a row (fp32 mode; the weight packing, the weight-gradient unpacking and the skip convolutions run on
    a second stream): the draws of every step are replayed through the oracle's train_step + Adam -- the loss of every step, the
    BatchNorm running statistics and the parameters after the last step must agree (a stale packed weight, a late unpack or a
    lost running-statistics update would show from step 2 on)."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import train
    T, B, ne, lr = 4, 5, 2, 1e-3
    ctor = (64, nc, 8, 16, 6, 5, skipco, 2, 16, 3, 32, 3, archi)
    m = _model(ctor, seed=9)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.make_cfg(*ctor)
    hp = dict(obs_scale=0.7, beta_y=1.0, beta_z=1.5, l2_res=1.0)
    m = m.cuda().train().set_precision('fp32')
    optim = srvp_amd.FusedAdam(m, lr=lr)
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    g = torch.Generator().manual_seed(4)
    adam = {}
    torch.set_num_threads(8)
    for step in range(4):
        x = torch.rand(T, B, nc, 64, 64, generator=g)
        loss, nll, kl_y0, kl_z = train(m, optim, None, x, torch.device('cuda'), opt)
        tape = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in m.last_tape.items()}
        scal, _, _ = O.train_step(sd, cfg, x, ne, tape, hp, adam, lr)
        # (from the second step on the two parameter trajectories differ where a gradient is ~0: Adam's first updates are lr * sign(g),
        # whatever |g| -- the small KL term feels that at the percent level, the loss does not)
        assert abs(loss - scal['loss']) <= (2e-5 if step == 0 else 1e-4) * abs(scal['loss']), (step, loss, scal['loss'])
        assert abs(nll - scal['nll']) <= (2e-5 if step == 0 else 1e-4) * abs(scal['nll']), (step, nll, scal['nll'])
        assert abs(kl_z - scal['kl_z']) <= (1e-4 if step == 0 else 5e-2) * abs(scal['kl_z']) + 1e-6, (step, kl_z, scal['kl_z'])
    torch.cuda.synchronize()
    mine = m.state_dict()
    moved = 0
    for k, v in sd.items():
        a = mine[k].detach().cpu()
        if k.endswith('num_batches_tracked'):
            assert int(a) == int(v) == 4, k
        elif k.endswith('running_var'):
            assert rel_l2(a, v) <= 3e-2, (k, rel_l2(a, v))      # (they follow the O(lr) differences of the weights, see above)
        elif k.endswith('running_mean'):                        # (means of ~0: measured against the layer's standard deviation)
            assert (a - v).abs().max().item() <= 3e-2 * sd[k[:-4] + 'var'].sqrt().max().item(), k
        else:
            # four Adam steps of size <= lr each: the parameters must have moved the same way (sign-sensitive where a gradient is ~0)
            # (a quarter of the elements at most, and never fewer than two allowed: on a 6-element bias two sign flips are 0.33 -- measured in
            # round 5 on dynamics.module.2.1.bias with either form of the rollout kernels, 1e-5 apart at step 0 and both 3e-3 from the oracle)
            d = (a - v).abs()
            assert d.max().item() <= 8 * lr and (d > lr).sum().item() <= max(2, 0.25 * d.numel()), (k, d.max().item(), (d > lr).float().mean().item())
            moved += 1
    assert moved > 10
