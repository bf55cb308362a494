"""
Does bf16 activation storage TRAIN to the same place as fp32 arithmetic?  (the per-tensor gradients of an
untrained network move by up to 45 % under bf16 storage; the single-step tests bound the kernels against a numerics model, they
do not answer what 300 optimisation steps make of it.)

The same 300-step Adam trajectory -- same initial weights, same mini-batches, same noise tape at every step -- is run twice
through the product: production precision ('bf16': bf16 MFMA operands / activation storage, fp32 accumulation and masters) and
parity mode ('fp32': the mode held to 1e-5 against the reference's fixtures).  Asserted: the loss curves agree step by step
(within 1 % of the range the curve covers), both runs learn, and the validation PSNR of the bf16-trained model (deterministic
protocol: same draws) agrees with the fp32-mode one to a few tenths of a dB with no systematic sign (+0.25 / -0.15 dB measured with two
summation orders of the bf16 path; the fp32-mode run repeated -- its atomics are not bitwise reproducible -- stays within 0.05 dB).
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


def _run(precision, data, val, report):
    import srvp_amd
    from srvp_amd.train import fused_step
    from srvp_amd import metrics as M
    dev = torch.device('cuda')
    torch.manual_seed(4)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*CTOR)
    model.init(1.2)
    model.to(dev).train().set_precision(precision)
    optim = srvp_amd.FusedAdam(model, lr=3e-4)                           # the recipes' learning rate (args.py: --lr 3e-4)
    opt = srvp_amd.DotDict(dict(n_euler_steps=NE, **HP))
    g = torch.Generator().manual_seed(99)
    losses = []
    for it in range(STEPS):
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
    assert abs(p16 - p32) <= 0.6, (p16, p32, p32b)
    assert abs(p32b - p32) <= 0.4, (p32, p32b)
