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
