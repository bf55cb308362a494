"""
Does bf16 activation storage TRAIN to the same place as fp32 arithmetic?  (the per-tensor gradients of an
untrained network move by up to 45 % under bf16 storage; the single-step tests bound the kernels against a numerics model, they
do not answer what 300 optimisation steps make of it.)

The same 300-step Adam trajectory -- same initial weights, same mini-batches, same noise tape at every step -- is run twice
through the product: production precision ('bf16': bf16 MFMA operands / activation storage, fp32 accumulation and masters) and
parity mode ('fp32': the mode held to 1e-5 against the reference's fixtures).  Asserted: the loss curves agree step by step
(within 1 % of the range the curve covers), both runs learn, and the validation PSNR of the bf16-trained model (deterministic
protocol: same draws) agrees with the fp32-mode one to a few tenths of a dB (the fp32-mode run repeated -- its atomics are not bitwise
reproducible, 300 Adam steps amplify that -- lands up to 0.43 dB from itself; bf16 sits 0.0-0.33 dB from the mean of such a pair).
Reference: train.py:49-129 (the step), 132-189 (validation PSNR).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

STEPS, T, B, NE = 300, 8, 8, 2
CTOR = (64, 1, 16, 32, 8, 8, True, 2, 32, 3, 64, 4, 'vgg')      # reduced width, full depth (every layer type of the VGG recipe)
HP = dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0)


def _run(precision, data, val, report, steps=STEPS, deterministic=False, return_params=False):
    import srvp_amd
    from srvp_amd.train import fused_step
    from srvp_amd import metrics as M
    dev = torch.device('cuda')
    torch.manual_seed(4)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*CTOR)
    model.init(1.2)
    model.to(dev).train().set_precision(precision)
    if deterministic:
        model.set_deterministic(True)
    optim = srvp_amd.FusedAdam(model, lr=3e-4)                           # the recipes' learning rate (args.py: --lr 3e-4)
    opt = srvp_amd.DotDict(dict(n_euler_steps=NE, **HP))
    g = torch.Generator().manual_seed(99)
    losses = []
    import time
    t_start = None
    for it in range(steps):
        if it == 5:
            torch.cuda.synchronize(); t_start = time.perf_counter()
        idx = torch.randperm(data.shape[1], generator=g)[:B]
        x = data[:, idx].contiguous().to(dev)
        tape = dict(t_skip=torch.randint(T, (B,), generator=g),
                    t_w=torch.stack([torch.randperm(T, generator=g)[:CTOR[7]] for _ in range(B)], 1),
                    eps_y0=torch.randn(B, CTOR[4], generator=g), eps_z=torch.randn(T - 1, B, CTOR[5], generator=g))
        optim.zero_grad()
        acc = fused_step(model, x, opt, tape=tape)
        optim.step()
        nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
        losses.append((nll + kl_y0 + kl_z + l2) / B)
    # validation (train.py:132-189 with one sample and fixed draws): condition on 3 frames, predict the remaining 5
    model.eval()
    nt_cond = 3
    gv = torch.Generator().manual_seed(7)
    Bv = val.shape[1]
    tape = dict(eps_y0=torch.randn(Bv, CTOR[4], generator=gv), eps_z=torch.randn(T - 1, Bv, CTOR[5], generator=gv))
    with torch.no_grad():
        x_ = model(val[:nt_cond].to(dev), T, 1 / NE, tape=tape)[0]
        psnr = M.psnr(x_, val.to(dev))[nt_cond:].mean().item()
    if return_params:
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t_start) / max(1, steps - 5) * 1e3 if t_start is not None else float('nan')
        flat = model._flat[0].detach().clone().cpu()
        if deterministic:
            model.set_deterministic(False)
        return losses, psnr, flat, x_.detach().cpu(), ms
    return losses, psnr


def test_bf16_trains_like_fp32_over_300_steps():
    from make_golden import synth_video
    from test_gpu_parity_gate import report
    data = torch.from_numpy(synth_video(T, 64, 1, seed=11))
    val = torch.from_numpy(synth_video(T, 16, 1, seed=12))
    l32, p32 = _run('fp32', data, val, report)
    l32b, p32b = _run('fp32', data, val, report)        # the same run again: split-K / statistics atomics make it non-bitwise; training amplifies
    l16, p16 = _run('bf16', data, val, report)
    # the ELBO of this recipe falls from +6e4 through zero to -2e4 (obs_scale 0.2: the NLL constant is negative), so differences are
    # measured against the range the curve covers, not against its current value
    scale = max(l32) - min(l32)
    rel = [abs(a - b) / scale for a, b in zip(l16, l32)]
    rel_self = [abs(a - b) / scale for a, b in zip(l32b, l32)]
    k = 20
    sm = lambda v: [sum(v[i:i + k]) / k for i in range(0, len(v) - k + 1, k)]
    rel_sm = [abs(a - b) / scale for a, b in zip(sm(l16), sm(l32))]
    report(test='trajectory_300', first=(l16[0], l32[0]), last=(l16[-1], l32[-1]), loss_range=scale, max_rel=max(rel), median_rel=sorted(rel)[len(rel) // 2],
           max_rel_smoothed=max(rel_sm), max_rel_fp32_rerun=max(rel_self), psnr_bf16=p16, psnr_fp32=p32, psnr_fp32_rerun=p32b)
    assert sum(l32[-20:]) / 20 < sum(l32[:5]) / 5 - 0.5 * scale, 'the fp32-mode run must learn'
    assert sum(l16[-20:]) / 20 < sum(l16[:5]) / 5 - 0.5 * scale, 'the bf16 run must learn'
    # Bands: twice what the worst of ~10 runs on different boxes showed.  The yardstick is the fp32-mode run REPEATED: its split-K / statistics
    # atomics are not bitwise reproducible and 300 Adam steps amplify that to 0.6 % of the curve's range and 0.02-0.10 dB of PSNR; bf16 vs fp32
    # measured 0.3-0.7 % and -0.15 ... +0.25 dB (no systematic sign: a change of summation order inside the bf16 path moves it as much).
    assert max(rel) <= 2e-2, (max(rel), rel.index(max(rel)), max(rel_self))
    assert max(rel_sm) <= 1e-2, max(rel_sm)
    # PSNR (round 5, eight repeats of this test on two boxes, with the round's kernels and with every round-5 switch off alike): the fp32-mode run
    # and its repeat land anywhere in 19.00-19.45 dB (up to 0.43 dB apart: the earlier 0.02-0.10 dB came from fewer repeats), bf16 in 18.97-19.24,
    # 0.0-0.33 dB from the mean of the fp32 pair (0.17 dB lower on average).  bf16 is held against that mean, the pair against its own spread.
    assert abs(p16 - 0.5 * (p32 + p32b)) <= 0.6, (p16, p32, p32b)
    assert abs(p32b - p32) <= 0.8, (p32, p32b)


def test_fp32_deterministic_mode_is_bit_reproducible():
    """the reference's CPU path is bit-reproducible run to run; the parity mode is too once
    model.set_deterministic(True) (or SRVP_DETERMINISTIC=1) replaces every arrival-order atomic sum by a fixed-order one (BatchNorm
    statistics from the stored fp32 raw output, two-launch BatchNorm-backward / image-side weight-gradient sums, single-split weight
    gradients, single-workgroup ELBO accumulators, one stream).  Two identical 300-step training runs: EVERY loss, the final
    parameters and the validation frames are bit-equal.  The default fp32 mode on the same run is reported beside it (it differs from
    run to run in the last bits and from the deterministic mode by summation order only)."""
    from make_golden import synth_video
    from test_gpu_parity_gate import report
    data = torch.from_numpy(synth_video(T, 64, 1, seed=11))
    val = torch.from_numpy(synth_video(T, 16, 1, seed=12))
    n, nd = STEPS, 60                                  # the 300 steps of the trajectory test above, twice; the default mode beside it on 60
    a = _run('fp32', data, val, report, steps=n, deterministic=True, return_params=True)
    b = _run('fp32', data, val, report, steps=n, deterministic=True, return_params=True)
    c = _run('fp32', data, val, report, steps=nd, deterministic=False, return_params=True)
    d = _run('fp32', data, val, report, steps=nd, deterministic=False, return_params=True)
    scale = max(a[0]) - min(a[0])
    report(test='deterministic_fp32', steps=n, ms_per_step_deterministic=a[4], ms_per_step_default=c[4],
           max_loss_diff_det_vs_det=max(abs(x - y) for x, y in zip(a[0], b[0])),
           max_rel_loss_diff_default_vs_default=max(abs(x - y) for x, y in zip(c[0], d[0])) / scale,
           max_rel_loss_diff_det_vs_default=max(abs(x - y) for x, y in zip(a[0], c[0])) / scale,
           params_equal_det=bool(torch.equal(a[2], b[2])), params_equal_default=bool(torch.equal(c[2], d[2])))
    assert a[0] == b[0], [(i, x, y) for i, (x, y) in enumerate(zip(a[0], b[0])) if x != y][:3]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[1] == b[1]
    # same algorithm, another summation order: the deterministic run stays inside the band two default runs span
    assert max(abs(x - y) for x, y in zip(a[0], c[0])) / scale <= 2e-2


def test_deterministic_mode_is_shared_by_models_and_left_by_the_last():
    """the process-wide part of the deterministic mode (overlap switches, the library switch and its workspace) is held at module
    level with a count of the models in the mode -- the second model entering must not save the already-disabled switches, the first one
    leaving must not switch the library off under the second, and an explicit set_deterministic(False) is not undone by the environment
    switch."""
    import srvp_amd
    from srvp_amd import convnet as cn, model as M
    dev = torch.device('cuda')
    ctor = (64, 1, 8, 16, 5, 7, True, 2, 24, 3, 40, 4, 'vgg')
    before = (M.OVERLAP_WGRAD, M.OVERLAP_SKIP, M.OVERLAP_PACK, cn.ENC_WGRAD_SIDE_MAXN)
    a = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor).to(dev).train().set_precision('fp32')
    b = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor).to(dev).train().set_precision('fp32')
    with pytest.raises(ValueError):
        srvp_amd.StochasticLatentResidualVideoPredictor(*ctor).to(dev).set_deterministic(True)           # bf16: refused
    a.set_deterministic(True)
    b.set_deterministic(True)
    a.set_deterministic(True)                                   # idempotent
    assert M._DET['count'] == 2 and cn.DETERMINISTIC and not M.OVERLAP_WGRAD
    a.set_deterministic(False)
    assert M._DET['count'] == 1 and cn.DETERMINISTIC and M._DET['ws'] is not None and not M.OVERLAP_WGRAD      # b is still in the mode
    x = torch.rand(3, 2, 1, 64, 64, device=dev)
    torch.manual_seed(3)
    o1 = b(x, 3, 1.0)[0].clone()
    torch.manual_seed(3)
    o2 = b(x, 3, 1.0)[0].clone()
    assert torch.equal(o1, o2)
    b.set_deterministic(False)
    assert M._DET['count'] == 0 and not cn.DETERMINISTIC and M._DET['ws'] is None
    assert (M.OVERLAP_WGRAD, M.OVERLAP_SKIP, M.OVERLAP_PACK, cn.ENC_WGRAD_SIDE_MAXN) == before
    os.environ['SRVP_DETERMINISTIC'] = '1'
    try:
        a(x, 3, 1.0)                                            # explicit off above: the environment switch does not re-enter
        assert not a.deterministic and M._DET['count'] == 0
    finally:
        del os.environ['SRVP_DETERMINISTIC']
