"""
Pins the CPU oracle (oracle/srvp_oracle.py) against the fixtures generated from the real reference
(tests/make_golden.py): forward outputs, ELBO scalars, every parameter gradient, post-Adam parameters and BN
running statistics, the eval-mode prediction path and the test.py-style rollout, plus known-answer vectors.
"""
import numpy as np
import pytest
import torch

from golden_util import Fixture, OUT_NAMES, fixture_names, dense_fixture_names, GOLDEN
from oracle import srvp_oracle as O

torch.set_num_threads(1)


def close(a, b, rtol, atol, what=''):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f'{what}: max err {err.max().item():.3e} (max ref {b.abs().max().item():.3e})'


@pytest.mark.parametrize('name', fixture_names())
def test_train_step_matches_reference(name):
    fx = Fixture(name)
    sd = fx.state('sd0')
    x = fx.t('x')
    tape = fx.tape()
    adam = {}
    scal, outs, grads = O.train_step(sd, fx.cfg, x, fx.meta['n_euler'], tape, fx.meta['hp'], adam, fx.meta['lr'])
    ref = fx.z['train.scalars']
    for i, k in enumerate(['loss', 'nll', 'kl_y_0', 'kl_z']):
        assert abs(scal[k] - ref[i]) <= 2e-6 * abs(ref[i]) + 1e-6, (k, scal[k], ref[i])
    assert abs(scal['l2_res'] - float(fx.z['train.l2_res'])) <= 1e-5 * abs(float(fx.z['train.l2_res'])) + 1e-6
    for n, o in zip(OUT_NAMES, outs):
        close(o, fx.t('train.' + n), 1e-4, 2e-5, n)
    gref = fx.group('grad.')
    assert set(gref) == set(grads)
    for k, g in grads.items():
        rel = (g - gref[k]).norm() / (gref[k].norm() + 1e-12)
        assert rel < 2e-3, (k, rel.item())
    sd1 = fx.state('sd1')
    for k, v in sd.items():
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(sd1[k])
        elif k.endswith(('running_mean', 'running_var')):
            close(v, sd1[k], 1e-5, 1e-6, k)
        else:
            # Adam's first step moves every weight by ~lr*sign(g): compare the update direction where it is defined
            close(v, sd1[k], 0, 2.5 * fx.meta['lr'], k)
            moved = (sd1[k] - fx.state('sd0')[k]).abs() > 0.5 * fx.meta['lr']
            agree = ((v - fx.state('sd0')[k]).sign() == (sd1[k] - fx.state('sd0')[k]).sign()) | ~moved
            assert agree.float().mean() > 0.995, k


@pytest.mark.parametrize('name', fixture_names())
def test_eval_prediction_matches_reference(name):
    fx = Fixture(name)
    sd = fx.state('sd1')
    x = fx.t('x')
    nt_cond, nt = int(fx.z['eval.nt_cond']), int(fx.z['eval.nt'])
    tape = fx.tape('eval.tape.')
    with torch.no_grad():
        outs = O.forward(sd, fx.cfg, x[:nt_cond], nt, fx.meta['n_euler'], tape, training=False)
    for n, o in zip(OUT_NAMES, outs):
        if o is None:
            assert not fx.has('eval.' + n)
            continue
        close(o, fx.t('eval.' + n), 1e-4, 2e-5, n)


@pytest.mark.parametrize('name', dense_fixture_names())
def test_dense_forward_matches_reference(name):
    """remove_intermediate=False (srvp.py:402): every Euler sub-step kept and decoded, prediction beyond the conditioning frames."""
    fx = Fixture(name)
    nt_cond, nt, ne = int(fx.z['nt_cond']), int(fx.z['nt']), int(fx.z['n_euler'])
    with torch.no_grad():
        outs = O.forward(fx.state('sd0'), fx.cfg, fx.t('x')[:nt_cond], nt, ne, fx.tape(), training=False, remove_intermediate=False)
    assert outs[0].shape[0] == (nt - 1) * ne + 1
    for n, o in zip(OUT_NAMES, outs):
        if o is None:
            assert not fx.has('out.' + n)
            continue
        close(o, fx.t('out.' + n), 1e-4, 2e-5, n)


@pytest.mark.parametrize('name', fixture_names())
def test_rollout_matches_reference(name):
    """test.py:235-246: encode -> forward(nt_cond) -> generate(y[-1], [], n) -> decode."""
    fx = Fixture(name)
    sd, cfg = fx.state('sd1'), fx.cfg
    x = fx.t('x')
    nt_cond = int(fx.z['eval.nt_cond'])
    ne = fx.meta['n_euler']
    with torch.no_grad():
        skip = O.encode(sd, cfg, x[:nt_cond], False, None)[1]
        tape = {'eps_y0': fx.t('roll.eps_y0'), 'eps_z': fx.t('roll.eps_z_fwd')}
        x_rec, y, _, w, _, _, _, _ = O.forward(sd, cfg, x[:nt_cond], nt_cond, ne, tape, training=False)
        eps_gen = fx.t('roll.eps_z_gen')
        y_os = O.generate(sd, cfg, y[-1], [], eps_gen.shape[0] + 1, ne, eps_gen, training=False)[0]
        x_pred = O.decode(sd, cfg, w, y_os[1:], skip, training=False).clamp(0, 1)
    close(x_rec, fx.t('roll.x_rec'), 1e-4, 2e-5, 'x_rec')
    close(y_os, fx.t('roll.y_gen'), 1e-4, 2e-5, 'y_gen')
    close(x_pred, fx.t('roll.x_pred'), 1e-4, 2e-5, 'x_pred')


def test_known_answers():
    z = np.load(GOLDEN + '/known_answers.npz')
    raw_q, raw_p = torch.from_numpy(z['ka.raw_q']), torch.from_numpy(z['ka.raw_p'])
    close(O.normal_params(raw_q)[1], z['ka.q_scale'], 1e-6, 1e-12)
    close(O.kl_normal(raw_q, raw_p), z['ka.kl_qp'], 1e-5, 1e-6)
    close(O.kl_normal(raw_q, None), z['ka.kl_q0'], 1e-5, 1e-6)
    loc, data = torch.from_numpy(z['ka.nll_loc']), torch.from_numpy(z['ka.nll_data'])
    for s in (1.0, 0.2, 0.71):
        close(O.neg_logprob(loc, data, s), z[f'ka.nll_{s}'], 1e-6, 1e-6)
    for n in (1, 2, 4):
        for nt in (12, 15, 16, 20, 53):
            ref = z[f'ka.euler_n{n}_nt{nt}']
            mine = np.array([(f, int(new), int(keep)) for f, new, keep in O.euler_schedule(nt, n)])
            assert (ref == mine).all(), (n, nt)
    assert np.allclose([O.lr_lambda(i, 1000) for i in range(0, 1200, 100)], z['ka.lr_lambda'])


def test_metrics_vs_reference_fixture():
    """the oracle's SSIM / MSE / PSNR restatement against the reference's own metrics/ssim.py + test.py:249-253
    outputs (fixture made by tests/make_golden.py:gen_metrics; the reference computes in float32)."""
    z = np.load(GOLDEN + '/metrics.npz')
    for C in (1, 3):
        pred, gt = torch.from_numpy(z[f'c{C}.pred']), torch.from_numpy(z[f'c{C}.gt'])
        close(O.video_ssim(pred, gt), z[f'c{C}.ssim'], 2e-5, 2e-6, 'ssim')
        close(O.video_mse(pred, gt), z[f'c{C}.mse'], 1e-5, 1e-9, 'mse')
        close(O.video_psnr(pred, gt), z[f'c{C}.psnr'], 1e-5, 1e-5, 'psnr')
    w = O.gaussian_window(11, 1.5)
    assert abs(w.sum().item() - 1) < 1e-12 and torch.allclose(w, w.t()) and w.argmax().item() == 60
