"""
SURVEY §8c item 2: FULL-WIDTH fixtures made by running the real reference (tests/make_golden.py --full; reference
train.py:49-129 through module/srvp.py:415-470) on the C2..C5 shapes of BASELINE.json at batch 2 -- SM-MNIST DCGAN (T=15,
nt_inf=5), KTH VGG (T=20, nt_inf=3), BAIR VGG (T=12), Human3.6M VGG (T=16, nt_inf=3) + its 53-frame prediction at the
recipe's res_gain = 1.2.  Weights and inputs are re-created from their seeds (checked against the fixture's checksums); the
fixture pins the ELBO scalars, all latent outputs, strided samples of the decoded frames, and for every parameter gradient
its norm and its projection on a seeded random direction (full tensors for the small ones).
Round 5 (`full_w*`): the three VGG recipes again at >= 96 frames (KTH T=20 B=5, BAIR T=12 B=8, Human3.6M T=16 B=6), the size from which the
product's STREAMING kernels (csrc/conv_stream.hip) take the 64x64-resolution layers -- the GPU test asserts that they did (srvp_conv_stream_count).

  * CPU: the oracle against these fixtures (pins the oracle at full width, not only at nf in {4, 8}).
  * GPU: the HIP path in fp32 mode (tight) and in bf16 mode (ELBO band) against the same fixtures.
Gradient tolerances: the reference's own fp32 autograd on these untrained small-batch BatchNorm networks is reproducible to
3e-3 .. 1e-2 between summation orders (fp32 oracle vs float64 oracle, tests/test_gpu_fp32_mode.py; the oracle's hand-written
BatchNorm against the reference's fused one moves the KTH-shape BN gradients by 1.1e-2), hence 3e-2 on norms / projections / tensors.
"""
import ast
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, OUT_NAMES, full_fixture_names
from make_golden import frame_samples, grad_direction, synth_video


class Full:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
        self.meta = ast.literal_eval(str(self.z['meta']))
        self.ctor = tuple(self.meta['ctor'])

    def t(self, k):
        return torch.from_numpy(np.array(self.z[k]))

    def tape(self, prefix='tape.'):
        return {k[len(prefix):]: self.t(k) for k in self.z.files if k.startswith(prefix)}

    def model_and_input(self):
        """Re-creates the reference's initial weights (same-seed construction, tests/test_host.py) and input, and checks both
        against the checksums taken from the reference run."""
        import srvp_amd
        torch.manual_seed(1)
        m = srvp_amd.StochasticLatentResidualVideoPredictor(*self.ctor)
        m.init(res_gain=self.meta['res_gain'])
        cs = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in m.state_dict().values()])
        # (orthogonal init runs a LAPACK QR: its rounding differs between host CPUs at the 1e-7 level -- far below what a wrong
        # RNG order or a wrong initialiser would show)
        assert np.allclose(cs, self.z['sd0.checksums'], rtol=1e-5, atol=1e-5)
        x = torch.from_numpy(synth_video(self.meta['T'], self.meta['B'], self.ctor[1], seed=321))
        assert np.allclose([x.double().sum().item(), x.double().abs().max().item()], self.z['x.checksum'], rtol=1e-9)
        return m, x


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check_step(fx, loss_terms, outs, grads, tol_loss, tol_x, tol_out, tol_grad):
    """loss_terms: (loss, nll/B, kl_y0/B, kl_z/B, l2_res); outs: the 8 forward outputs; grads: name -> tensor."""
    ref = fx.z['train.scalars']
    assert abs(loss_terms[0] - ref[0]) <= tol_loss * abs(ref[0]), (loss_terms[0], ref[0])
    assert abs(loss_terms[1] - ref[1]) <= tol_loss * abs(ref[1])
    assert abs(loss_terms[4] - float(fx.z['train.l2_res'])) <= max(tol_out, 10 * tol_loss) * abs(float(fx.z['train.l2_res']))
    for n, o in zip(OUT_NAMES, outs):
        r = fx.t('train.' + n)
        if n == 'x_':
            assert (frame_samples(o.detach().cpu()) - r).abs().max().item() <= tol_x, n
        else:
            assert rel_l2(o, r) <= tol_out, (n, rel_l2(o, r))
    names = fx.meta['grad_names']
    bad = []
    for i, k in enumerate(names):
        g = grads[k].detach().double().cpu()
        nref, dref = float(fx.z['grad.norm'][i]), float(fx.z['grad.dot'][i])
        d = (g * grad_direction(k, g.shape).double()).sum().item()
        if abs(g.norm().item() - nref) > tol_grad * nref or abs(d - dref) > 2 * tol_grad * nref:
            bad.append((k, g.norm().item() / nref, (d - dref) / nref))
        if 'grad.' + k in fx.z.files and rel_l2(g, fx.t('grad.' + k)) > tol_grad:
            bad.append((k, 'full', rel_l2(g, fx.t('grad.' + k))))
    assert not bad, bad[:8]


@pytest.mark.parametrize('name', full_fixture_names())
def test_oracle_vs_full_width_reference_fixture(name):
    from oracle import srvp_oracle as O
    fx = Full(name)
    m, x = fx.model_and_input()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.set_num_threads(8)
    scal, outs, grads = O.train_step(sd, O.make_cfg(*fx.ctor), x, fx.meta['n_euler'], fx.tape(), fx.meta['hp'])
    check_step(fx, (scal['loss'], scal['nll'], scal['kl_y_0'], scal['kl_z'], scal['l2_res']), outs, grads, 2e-6, 1e-5, 1e-4, 3e-2)
    for k in fx.z.files:
        if k.startswith('sd1.'):
            assert (sd[k[4:]] - fx.t(k)).abs().max().item() <= 1e-5 * (1 + fx.t(k).abs().max().item()), k
    if 'roll.cfg' in fx.z.files:
        nt_cond, nt = (int(v) for v in fx.z['roll.cfg'])
        m2, _ = fx.model_and_input()
        sd2 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
        xr = torch.from_numpy(synth_video(nt_cond, fx.meta['B'], fx.ctor[1], seed=322))
        with torch.no_grad():
            o = O.forward(sd2, O.make_cfg(*fx.ctor), xr, nt, fx.meta['n_euler'], fx.tape('roll.tape.'), training=False)
        check_rollout(fx, o[1], o[0], 1e-6)


def roll2_model(fx):
    """The well-conditioned long-horizon state of tests/make_golden.py (ROLL2): same-seed weights at init(res_gain = 0.8), the
    reference's settled BatchNorm statistics from the fixture, output-layer weight x 30; checked against the reference's checksums."""
    import srvp_amd
    rg, out_scale, _ = (float(v) for v in fx.z['roll2.recipe'])
    torch.manual_seed(1)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*fx.ctor)
    m.init(res_gain=rg)
    sd = m.state_dict()
    with torch.no_grad():
        for k in fx.z.files:
            if k.startswith('roll2.bn.'):
                sd[k[len('roll2.bn.'):]].copy_(fx.t(k))
        sd['decoder.conv.3.1.weight'].mul_(out_scale)
    cs = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in m.state_dict().values()])
    assert np.allclose(cs, fx.z['roll2.sd.checksums'], rtol=1e-5, atol=1e-5)
    nt_cond, nt = (int(v) for v in fx.z['roll2.cfg'])
    xr = torch.from_numpy(synth_video(nt_cond, fx.meta['B'], fx.ctor[1], seed=322))
    return m, xr, nt


def check_rollout2(fx, y, x_, tol_y, tol_x, what):
    """EVERY one of the 53 frames at ONE tolerance (no growth with t): latent states relative, decoded-frame samples absolute."""
    yr, xr = fx.t('roll2.y'), fx.t('roll2.x_')
    xs = frame_samples(x_.detach().float().cpu())
    ey = [rel_l2(y[t], yr[t]) for t in range(yr.shape[0])]
    ex = [(xs[t] - xr[t]).abs().max().item() for t in range(yr.shape[0])]
    try:
        from test_gpu_parity_gate import report
        report(test='c5_rollout_53_frames_well_conditioned', what=what, max_rel_y=max(ey), max_abs_x=max(ex), y_last=ey[-1], x_last=ex[-1])
    except Exception:
        pass
    assert max(ey) <= tol_y, (what, 'y', int(np.argmax(ey)), max(ey))
    assert max(ex) <= tol_x, (what, 'x_', int(np.argmax(ex)), max(ex))
    # the leg is not degenerate: the decoded frames span a range and move over the horizon
    assert xr.max() - xr.min() > 0.2 and (xr[-1] - xr[8]).abs().max() > 0.02


def test_oracle_vs_c5_rollout_every_frame():
    """Config 5's reason to exist is the 53-frame horizon (reference test.py:237-246): the oracle reproduces the reference's prediction
    from the well-conditioned state at 1e-5 (latent states, relative) / 1e-5 (decoded frames) at every frame."""
    from oracle import srvp_oracle as O
    fx = Full('full_c5_human_vgg')
    m, xr, nt = roll2_model(fx)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.set_num_threads(8)
    with torch.no_grad():
        o = O.forward(sd, O.make_cfg(*fx.ctor), xr, nt, fx.meta['n_euler'], fx.tape('roll2.tape.'), training=False)
    check_rollout2(fx, o[1], o[0], 1e-5, 1e-5, 'oracle')


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_hip_c5_rollout_every_frame(precision):
    """The HIP path on the same leg, EVERY frame at one tolerance (VERDICT r3 item 6 asked for <= 1e-3 / <= 3e-2): fp32 mode 2e-5
    (latent states, relative) / 1e-5 (decoded frames) -- measured 2.3e-6 / 5.4e-7 --, production bf16 1e-3 / 5e-3 -- measured
    1.5e-4 / 1.3e-3.  No tolerance grows with the frame index."""
    fx = Full('full_c5_human_vgg')
    m, xr, nt = roll2_model(fx)
    m = m.cuda().eval().set_precision(precision)
    o = m(xr.cuda(), nt, dt=1 / fx.meta['n_euler'], tape=fx.tape('roll2.tape.'))
    if precision == 'fp32':
        check_rollout2(fx, o[1], o[0], 2e-5, 1e-5, 'hip fp32')
    else:
        check_rollout2(fx, o[1], o[0], 1e-3, 5e-3, 'hip bf16')


def check_rollout(fx, y, x_, eps0):
    """(The documented ILL-CONDITIONED leg; the every-frame check is check_rollout2 on the well-conditioned leg above.)
    53-frame prediction at res_gain = 1.2: the untrained residual MLP expands |y| by ~1.3x per frame, and so it expands any
    arithmetic difference: the fp32 reference run twice with different summation orders agrees to eps0 * 1.35^t at frame t.
    The latent states are therefore held to a tolerance that grows at that rate, the decoded frames (saturating sigmoid of
    logits that scale with |y|) to the matching absolute band."""
    yr, xr = fx.t('roll.y'), fx.t('roll.x_')
    nt = yr.shape[0]
    for t in range(nt):
        tol = min(0.5, eps0 * 1.35 ** t)
        e = rel_l2(y[t], yr[t])
        assert e <= tol, (t, e, tol)
    xs = frame_samples(x_.detach().cpu())
    for t in range(nt):
        tol = min(1.0, 20 * eps0 * 1.35 ** t + 1e-5)
        assert (xs[t] - xr[t]).abs().max().item() <= tol, (t, (xs[t] - xr[t]).abs().max().item(), tol)


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('name', full_fixture_names())
def test_hip_vs_full_width_reference_fixture(name, precision):
    import srvp_amd
    from srvp_amd.train import elbo_terms_and_grads
    fx = Full(name)
    m, x = fx.model_and_input()
    m = m.cuda().train().set_precision(precision)
    hp, ne = fx.meta['hp'], fx.meta['n_euler']
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    m.flatten_parameters_()
    m._grads()
    m._flat[1].zero_()
    xg = x.cuda()
    from srvp_amd import _lib as L
    cnt = [L.load().srvp_conv_stream_count(i) for i in range(3)]
    outs = m._forward_impl(xg, x.shape[0], ne, fx.tape(), training=True)
    outs_c = [o.clone() for o in outs]
    acc, gr = elbo_terms_and_grads(m, xg, outs, opt)
    m._backward_impl(gr[0], None, None, gr[1], gr[2], gr[3], gr[4])
    if name.startswith('full_w') and precision == 'bf16':
        # the >= 96-frame fixtures exist so that reference-made numbers pass through the STREAMING kernels (csrc/conv_stream.hip): forward +
        # data gradient of the 64 -> 64 channel layer at 64x64, the data gradient with fused BatchNorm-backward sums, the sub-pixel stage entry
        took = [L.load().srvp_conv_stream_count(i) - cnt[i] for i in range(3)]
        assert took[0] >= 1 and took[1] >= 1 and took[2] >= 1, took
    nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
    B = x.shape[1]
    loss = (nll + hp['beta_y'] * kl_y0 + hp['beta_z'] * kl_z + hp['l2_res'] * l2) / B
    grads = {k: p.grad for k, p in m.named_parameters()}
    if precision == 'fp32':
        check_step(fx, (loss, nll / B, kl_y0 / B, kl_z / B, l2), outs_c, grads, 1e-5, 2e-5, 2e-4, 3e-2)
    else:
        # bf16 storage on 24-40 frames: the ELBO band of tests/test_gpu_model.py (1e-4 is asserted at 384 frames), frames 3e-2
        ref = fx.z['train.scalars']
        try:
            from test_gpu_parity_gate import report
            report(test='full_width_golden_bf16', name=name, frames=int(x.shape[0] * x.shape[1]),
                   e_loss=abs(loss - ref[0]) / abs(ref[0]))
        except Exception:
            pass
        # Two classes of recipe.  WELL-CONDITIONED at the initial weights (C2 SM-MNIST, C4 BAIR = the north_star's benchmarked config): measured
        # 8.9e-6 (30 frames), 1.1-4.1e-5 (24 frames), 1.06e-5 (96 frames) over rounds 2-5 -- bound 5e-5 / the north_star's 1e-4 at 24 frames.
        # ILL-CONDITIONED at the initial weights (C3 KTH, C5 Human3.6M: obs_scale 0.2, untrained residual MLP at res_gain 1.2 -- |y| grows ~1.3x per
        # frame): the ELBO error there is one draw of amplified rounding noise, not a property of a kernel.  Measured in round 5
        # (tools/conv_in_ab.py -> profiles/r05_conv_in_ab.jsonl): the SAME 40-frame KTH fixture reads 9.4e-6 with the image-side tile kernel
        # and 1.16e-4 with the streaming kernel, while the two kernels' BatchNorm statistics agree to 1.2e-8 (each within 1.2e-8 of float64 sums) and
        # their raw outputs differ on 6.9e-5 of the elements by one bf16 ulp (both 5-7e-5 off the bf16-rounded float64 convolution): flipping one
        # rounding in 14 000 moves this ELBO by 1e-4.  400 frames of the same recipes: 1.5e-3 .. 3.4e-3 (KTH), 2.4e-4 (Human) at step 0, and
        # <= 3.5e-5 at steps 100 / 200 / 300 of training on two seeds (test_gpu_parity_gate.py, profiles/r05_gate_after_training_*.jsonl) -- the
        # north_star's 1e-4 is asserted THERE for these recipes.  Bounds here = 4 x measured, a regression tripwire only.
        bound = {'full_c2_smmnist_dcgan': 4e-5, 'full_c4_bair_vgg': 1e-4, 'full_w4_bair_vgg_b8': 5e-5,
                 'full_c3_kth_vgg': 5e-4, 'full_w3_kth_vgg_b5': 2.2e-3, 'full_c5_human_vgg': 1.2e-3, 'full_w5_human_vgg_b6': 7.5e-4}.get(name, 3e-4)
        grad_report = {}
        if name.startswith('full_w'):
            # parameter gradients of the bf16 path on >= 96 frames against the reference's: the worst norm ratio / projection error over all
            # tensors (reported; the bound is the one the 400-frame gate of test_gpu_parity_gate.py states relative to the numerics model)
            names = fx.meta['grad_names']
            rat = [grads[k].detach().double().norm().item() / max(float(fx.z['grad.norm'][i]), 1e-30) for i, k in enumerate(names)]
            grad_report = dict(grad_norm_ratio_min=min(rat), grad_norm_ratio_max=max(rat), grad_norm_ratio_median=float(np.median(rat)))
            try:
                report(test='full_width_golden_bf16_grads', name=name, **grad_report)
            except Exception:
                pass
            # measured (round 5): norm ratios 0.946 .. 1.082 over all tensors of the three recipes, median 0.998 .. 1.003
            assert 0.99 < grad_report['grad_norm_ratio_median'] < 1.01 and grad_report['grad_norm_ratio_min'] > 0.9 and grad_report['grad_norm_ratio_max'] < 1.15, grad_report
        assert abs(loss - ref[0]) <= bound * abs(ref[0]), (loss, ref[0], bound)
        assert (frame_samples(outs_c[0].cpu()) - fx.t('train.x_')).abs().max().item() <= 3e-2
        for n, o in zip(OUT_NAMES[1:], outs_c[1:]):
            assert rel_l2(o, fx.t('train.' + n)) <= 6e-2, n
    if 'roll.cfg' in fx.z.files and precision == 'fp32':
        nt_cond, nt = (int(v) for v in fx.z['roll.cfg'])
        m2, _ = fx.model_and_input()
        m2 = m2.cuda().eval().set_precision('fp32')
        xr = torch.from_numpy(synth_video(nt_cond, fx.meta['B'], fx.ctor[1], seed=322)).cuda()
        o = m2(xr, nt, dt=1 / ne, tape=fx.tape('roll.tape.'))
        check_rollout(fx, o[1], o[0], 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['full_w3_kth_vgg_b5', 'full_w5_human_vgg_b6'])
def test_streaming_kernels_tight_per_layer_on_reference_fixtures(name):
    """the bf16 ELBO bounds of the KTH / Human3.6M fixtures above are tripwires (4 x measured: the step-0 ELBO of these
    recipes amplifies single roundings), so a regression of a few 1e-4 inside conv_stream64 / conv_stream_sub64 / the image-side streaming
    kernels would pass there.  Here the SAME reference-made inputs and weights (>= 96 frames) go through the training forward twice -- the
    streaming kernels on (default) and off (the halo / tile kernels) -- and every layer the streaming kernels produce is held tightly,
    layer by layer: raw conv outputs equal up to single bf16 ulps (on < 0.5 % of the elements where the inputs are identical -- the first block --,
    < 15 % behind it, never more than 2^-6 of the layer's range), BatchNorm statistics to 1e-5 (first block: measured 2.3e-6) / 1e-4 (second) relative, decoded frames to 2e-2; both forms then sit inside the fixture's bounds (test above)."""
    from srvp_amd import _lib as L
    fx = Full(name)
    ne = fx.meta['n_euler']
    res = {}
    for on in (1, 0):
        m, x = fx.model_and_input()
        m = m.cuda().train().set_precision('bf16')
        m.flatten_parameters_()
        for sw in ('srvp_conv_set_stream64', 'srvp_conv_set_in_stream'):
            L.call(sw, on)
        try:
            cnt = [L.load().srvp_conv_stream_count(i) for i in range(3)]
            outs = m._forward_impl(x.cuda(), x.shape[0], ne, fx.tape(), training=True)
            torch.cuda.synchronize()
            took = [L.load().srvp_conv_stream_count(i) - cnt[i] for i in range(3)]
        finally:
            for sw in ('srvp_conv_set_stream64', 'srvp_conv_set_in_stream'):
                L.call(sw, 1)
        pl = m._last_plan
        layers = {f'enc{i}': b for i, b in enumerate(pl['enc'].blocks[:2])}
        layers.update({f'dec{len(pl["dec"].blocks) - 2}': pl['dec'].blocks[-2]})
        res[on] = dict(took=took, x_=outs[0].float().clone(),
                       raw={k: b.raw.float().clone() for k, b in layers.items()},
                       stats={k: b.stats.clone() for k, b in layers.items() if getattr(b, 'stats', None) is not None})
    assert res[1]['took'][0] >= 1 and res[0]['took'] == [0, 0, 0], (res[1]['took'], res[0]['took'])
    for k in res[1]['raw']:
        a, b = res[1]['raw'][k], res[0]['raw'][k]
        diff = (a - b).abs()
        scale = max(1.0, b.abs().max().item())
        assert diff.max().item() <= 2 ** -6 * scale, (k, diff.max().item())                    # isolated roundings, never a wrong value
        frac = (diff > 0).float().mean().item()
        # the first layer sees identical inputs: isolated one-ulp flips only (measured 7e-5 of the elements in round 5); the layers behind it see
        # inputs that already differ by those flips and the one-ulp BatchNorm coefficients they cause (measured 4.4 % at the second block)
        # (the decoder's stage entry sits behind the whole latent path of an ill-conditioned recipe: bounded in size above, not in count)
        assert frac < (5e-3 if k == 'enc0' else (0.15 if k == 'enc1' else 1.01)), (k, frac)
        try:
            from test_gpu_parity_gate import report
            report(test='streaming_vs_tile_on_fixture', name=name, layer=k, frac_differing=frac, max_diff=diff.max().item())
        except Exception:
            pass
    for k, tol in (('enc0', 1e-5), ('enc1', 1e-4)):            # (statistics: fp64 sums of fp32 accumulators; the second block's inputs differ by the flips above)
        a, b = res[1]['stats'][k], res[0]['stats'][k]
        assert ((a - b).abs().max() / b.abs().max()).item() < tol, (k, ((a - b).abs().max() / b.abs().max()).item())
    assert (res[1]['x_'] - res[0]['x_']).abs().max().item() <= 2e-2          # (both within 3e-2 of the reference's frames: test above; measured 1.1e-2)
