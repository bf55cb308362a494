"""
The north_star parity gate at the headline architecture (BASELINE.json: "ELBO within 1e-4 relative of the CPU reference"):
BAIR shape -- VGG-64, nc=3, skip connections, T=12, n_euler=2 -- at full layer widths on 384 frames (B=32), HIP path in
its production precision (bf16 MFMA operands / bf16 activation storage) against the fp32 CPU oracle (= the reference's
arithmetic, oracle pinned by tests/golden/*.npz), forward AND gradients.  Plus SURVEY §8f-1 against the oracle: the
batched multi-sample rollout `model.sample` and `train.evaluate`'s best-of-N selection vs per-sample oracle forwards.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_report.jsonl')


def report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, 'a') as f:
        f.write(json.dumps(kw) + '\n')


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def cosine(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def _train_tape(T, B, nt_inf, ny, nz, skipco, g):
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                eps_y0=torch.randn(B, ny, generator=g), eps_z=torch.randn(T - 1, B, nz, generator=g))
    if skipco:
        tape['t_skip'] = torch.randint(T, (B,), generator=g)
    return tape


def test_bf16_elbo_gate_bair_384_frames():
    """ELBO <= 1e-4 relative (measured: 3e-6), decoded frames, and gradient quality vs the fp32 oracle at 384 frames.
    Gradients: the whole-step gradient direction must agree (cosine >= 0.99 over all 23.8 M parameters).  Per tensor, bf16
    activation storage alone moves the encoder gradients of this untrained network by up to 45 % (cosine 0.90): the oracle's
    numerics model -- the reference algorithm with the product's FORWARD rounding points and exact fp32 gradient arithmetic --
    shows the same deviation from the fp32 oracle (median 8.7 %, worst cosine 0.90) as the HIP path does (8.4 %, 0.90).  So
    the per-tensor bounds are stated against that model: the kernels (bf16 gradient storage included) may not be worse than
    what bf16 forward storage costs by itself.  The tight per-tensor statement (<= 2e-3) is made in fp32 mode
    (tests/test_gpu_fp32_mode.py)."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import elbo_terms_and_grads
    T, B, ne = 12, 32, 2
    ctor = (64, 3, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg')
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.41)                                   # BAIR recipe (README.md:124-128: default res_gain)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(123)
    x = torch.rand(T, B, 3, 64, 64, generator=g)
    tape = _train_tape(T, B, 2, 50, 50, True, g)
    hp = dict(obs_scale=0.71, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(cores, 16)))
    scal, outs_ref, grads_ref = O.train_step({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, ne, tape, hp)
    # the same algorithm with the product's FORWARD rounding points only (bf16 operands / stored activations, exact fp32
    # gradient arithmetic): how much of the gradient difference is inherent to bf16 activation storage at this batch size
    O.PRECISION = 'bf16'
    try:
        _, _, grads_m = O.train_step({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, ne, tape, hp)
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
    e_nll = abs(nll / B - scal['nll']) / abs(scal['nll'])
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    gerr = {k: rel_l2(grads[k], grads_ref[k]) for k in grads}
    gcos = {k: cosine(grads[k], grads_ref[k]) for k in grads}
    flat = torch.cat([grads[k].flatten().double().cpu() for k in grads])
    flat_ref = torch.cat([grads_ref[k].flatten().double() for k in grads])
    med = sorted(gerr.values())[len(gerr) // 2]
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:6]
    wcos = sorted(gcos.items(), key=lambda kv: kv[1])[:6]
    x_err = (x_.double().cpu() - outs_ref[0].double()).abs().max().item()
    mcos = {k: cosine(grads_m[k], grads_ref[k]) for k in grads}
    merr = {k: rel_l2(grads_m[k], grads_ref[k]) for k in grads}
    hm_err = {k: rel_l2(grads[k], grads_m[k]) for k in grads}
    report(test='bf16_gate_384', frames=T * B, loss=loss, loss_ref=scal['loss'], e_loss=e_loss, e_nll=e_nll, x_maxabs=x_err,
           median_grad=med, worst_grads=worst, worst_cos=wcos, flat_rel=rel_l2(flat, flat_ref), flat_cos=cosine(flat, flat_ref),
           model_vs_fp32=dict(median=sorted(merr.values())[len(merr) // 2], worst=sorted(merr.items(), key=lambda kv: -kv[1])[:6],
                              worst_cos=sorted(mcos.items(), key=lambda kv: kv[1])[:6]),
           hip_vs_model=dict(median=sorted(hm_err.values())[len(hm_err) // 2], worst=sorted(hm_err.items(), key=lambda kv: -kv[1])[:6]))
    assert e_loss <= 1e-4, (loss, scal['loss'], e_loss)            # BASELINE.json north_star
    assert e_nll <= 1e-4, e_nll
    assert x_err < 3e-2, x_err
    # gradients of the whole step in bf16 storage against the fp32 reference arithmetic
    assert cosine(flat, flat_ref) >= 0.99, cosine(flat, flat_ref)
    m_med = sorted(merr.values())[len(merr) // 2]
    assert med <= 1.25 * m_med + 0.01, (med, m_med, worst)
    assert min(gcos.values()) >= min(mcos.values()) - 0.03, (wcos, sorted(mcos.items(), key=lambda kv: kv[1])[:3])
    assert max(gerr.values()) <= 1.25 * max(merr.values()) + 0.02, (worst, max(merr.values()))
    # the decoder tensors (short backward chain, no dependence on the encoder's rounded activations) are tight in absolute terms
    short = [k for k in gerr if k.startswith('decoder.')]
    assert min(gcos[k] for k in short) >= 0.98, sorted(((k, gcos[k]) for k in short), key=lambda kv: kv[1])[:4]


@pytest.mark.parametrize('name,nc,T,B', [('kth', 1, 20, 20), ('human', 3, 16, 26), ('smmnist', 1, 15, 26)])
def test_elbo_gate_undiluted_recipes_400_frames(name, nc, T, B):
    """The north_star gate on the recipes whose ELBO is NOT dominated by the model-independent NLL constant: KTH (config 3: nc=1,
    T=20) and Human3.6M (config 5: nc=3, T=16), both nt_inf=3, obs_scale 0.2, res_gain 1.2 (README.md:111-122), full layer widths,
    >= 400 STRUCTURED frames (moving blobs, tests/make_golden.py::synth_video), forward + ELBO terms.

    What holds and what does not (measured here instead of argued):
      * precision='fp32' (parity mode) vs the fp32 CPU oracle: ELBO <= 1e-4 (north_star), asserted at 1e-5;
      * precision='bf16' (production): at these INITIAL weights the ELBO is ill-conditioned -- the untrained residual MLP at
        res_gain 1.2 grows |y| from 23 to 2100 over the 38 Euler steps and KL(q_z || p_z(y_t)) is ~half of the loss, so the
        0.3 % relative error that bf16 activation STORAGE leaves on the encoder features (13 layers x 2 roundings) reaches the
        loss undamped.  The oracle itself with nothing but the product's forward rounding points inserted (O.PRECISION = 'bf16',
        exact fp32 arithmetic otherwise) is 2.4e-3 (KTH) off the fp32 oracle: no implementation that stores bf16 activations
        meets 1e-4 HERE, whereas it does on the BAIR recipe (test above: 3e-6).  Asserted: the HIP path is no further from fp32
        than 2x that numerics model, and within 1.5e-3 of the model itself (the kernels implement the algorithm).
    'smmnist' runs the same comparison on config 2's recipe (DCGAN, no skip connections, ny = nz = 20, nt_inf = 5, n_euler = 1, beta_z = 2,
    obs_scale 1, 390 frames): there the production bf16 path itself is held to the north_star 1e-4."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from make_golden import synth_video
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import elbo_terms_and_grads
    dcgan = name == 'smmnist'
    ne = 1 if dcgan else 2
    ctor = (64, nc, 64, 128, 20, 20, False, 5, 256, 3, 512, 4, 'dcgan') if dcgan else (64, nc, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg')
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.41 if dcgan else 1.2)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(321)
    x = torch.from_numpy(synth_video(T, B, nc, seed=77))
    tape = _train_tape(T, B, ctor[7], ctor[4], ctor[5], ctor[6], g)
    hp = dict(obs_scale=1.0, beta_y=1.0, beta_z=2.0, l2_res=1.0) if dcgan else dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(cores, 16)))

    def oracle(prec):
        O.PRECISION = prec
        try:
            with torch.no_grad():
                outs = O.forward({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, T, ne, tape, training=True)
                r = O.elbo(x, outs, hp['obs_scale'], hp['beta_y'], hp['beta_z'], hp['l2_res'])
        finally:
            O.PRECISION = 'fp32'
        return float(r['loss']), float(r['nll']) / B, outs
    ref_loss, ref_nll, outs_ref = oracle('fp32')
    mod_loss, mod_nll, _ = oracle('bf16')
    model = model.cuda().train()
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    xg = x.cuda()

    def hip(precision):
        model.set_precision(precision)
        with torch.no_grad():
            outs = model._forward_impl(xg, T, ne, tape, training=True)
            x_ = outs[0].clone()
            acc, _ = elbo_terms_and_grads(model, xg, outs, opt, want_grads=False)
        nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
        return (nll + hp['beta_y'] * kl_y0 + hp['beta_z'] * kl_z + hp['l2_res'] * l2) / B, nll / B, x_
    l32, n32, x32 = hip('fp32')
    l16, n16, x16 = hip('bf16')
    rel = lambda a, b: abs(a - b) / abs(b)
    const = T * nc * 64 * 64 * (torch.log(torch.tensor(hp['obs_scale'])).item() + 0.9189385332)   # the model-independent part of the NLL per video
    res = dict(fp32_mode_vs_fp32=rel(l32, ref_loss), fp32_mode_nll=rel(n32, ref_nll), bf16_vs_fp32=rel(l16, ref_loss), bf16_nll_vs_fp32=rel(n16, ref_nll),
               model_vs_fp32=rel(mod_loss, ref_loss), model_nll_vs_fp32=rel(mod_nll, ref_nll), bf16_vs_model=rel(l16, mod_loss),
               bf16_nll_vs_model=rel(n16, mod_nll), x_maxabs_fp32_mode=(x32.double().cpu() - outs_ref[0].double()).abs().max().item(),
               x_maxabs_bf16=(x16.double().cpu() - outs_ref[0].double()).abs().max().item())
    report(test=f'elbo_gate_{name}', frames=T * B, loss_ref=ref_loss, nll_constant_per_video=const, data_term_per_video=ref_nll - const, **res)
    if dcgan:
        assert res['bf16_vs_fp32'] <= 1e-4 and res['bf16_nll_vs_fp32'] <= 1e-4, res          # north_star, production precision
    else:
        assert abs(ref_nll - const) > 0.5 * abs(ref_nll), 'the data term should carry this NLL'
    assert res['fp32_mode_vs_fp32'] <= 1e-5 and res['fp32_mode_nll'] <= 1e-5, res      # north_star (1e-4) with margin, parity mode
    assert res['x_maxabs_fp32_mode'] <= 1e-4, res
    assert res['bf16_vs_fp32'] <= 2 * res['model_vs_fp32'] + 1e-4, res                  # bf16 storage: bounded by the numerics model
    assert res['bf16_vs_model'] <= 1.5e-3 and res['bf16_nll_vs_model'] <= 5e-4, res
    assert res['x_maxabs_bf16'] < 3e-2, res


# per-tensor gradient gate at a trained state: cosine >= 0.99 (two BatchNorm bias vectors may sit between the floor and 0.99), norm within 5 %
GRAD_COS_FLOOR, GRAD_RATIO_TOL = 0.985, 0.05


@pytest.mark.parametrize('seed', [0, 1])
@pytest.mark.parametrize('name', ['kth', 'human'])
def test_elbo_gate_production_precision_once_training_started(name, seed):
    """north_star "ELBO within 1e-4 relative of the CPU reference" in the BENCHMARKED precision on configs 3 and 5, at states TRAINING
    VISITS.  At the initial weights of these recipes bf16 storage is 3.4e-3 / 2.4e-4 off (test above: the
    untrained residual MLP at res_gain 1.2 is ill-conditioned); that is a property of step 0 only.  tools/gate_after_training.py trains the
    full-width recipe in fp32 parity mode on moving-blob videos and puts the SAME held-out 400-frame batch + noise tape through the
    HIP path in bf16, in fp32 mode, and through the fp32 CPU oracle.  Round 4 measured seed 0 (profiles/r04_gate_after_training_*.jsonl)
        KTH       step 0: 3.4e-3   100: 8.3e-6   200: 6.5e-5   300: 2.2e-5
        Human3.6M step 0: 2.3e-4   100: 2.2e-5   200: 4.2e-6   300: 2.0e-5
    and asserted step 100 only.  Asserted now at steps 100, 200 AND 300, on two seeds (held-out batch, noise tape and training videos all
    change with the seed): production bf16 <= 1e-4 vs the oracle, fp32 mode <= 1e-5 (profiles/r05_gate_after_training_*.jsonl)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import gate_after_training as G
    rows = G.run(name, steps=300, every=100, oracle_at=(100, 200, 300), log=lambda s: None, seed=seed, grads_at=(100,) if seed == 0 else ())
    first = rows[0]
    for row in rows:
        report(test='elbo_gate_after_training', **{k: v for k, v in row.items() if k != 'grad_gate_all'})
    assert [r['step'] for r in rows] == [0, 100, 200, 300]
    assert rows[1]['loss_oracle'] < 0.0 < first['loss_fp32_mode']            # it did train (the NLL went from +1e5 to -1e5)
    for row in rows[1:]:
        assert row['fp32_mode_vs_oracle'] <= 1e-5, row
        assert row['bf16_vs_oracle'] <= 1e-4, row                             # north_star, production precision
    if seed == 0:
        # (round 6) the benchmarked path's GRADIENTS at a trained state (step 100, held-out 400-frame batch): every
        # parameter tensor's bf16-path gradient against the fp32 oracle's autograd -- direction and length per tensor.  This replaces
        # "bounded relative to the numerics model on an untrained network" as the statement about the production path's gradients.
        # Measured (profiles/r06_grad_gate_*.jsonl): Human3.6M worst cosine 0.99926, norms within 0.4 %; KTH median 0.99994, every conv /
        # linear weight >= 0.9908, two BatchNorm BIAS vectors of the 16x16 encoder stage (256 elements each: one long cancelling sum over
        # 102400 positions per channel) at 0.9885 / 0.9886, norms within 1.7 %.
        gg = rows[1]['grad_gate']
        assert gg['tensors'] == len(rows[1]['grad_gate_all']) >= 90, gg
        assert gg['worst_cos'] >= GRAD_COS_FLOOR, gg
        assert len(gg['below_0p99']) <= 2 and all(k.endswith('.bias') and '.conv.' in k for k in gg['below_0p99']), gg     # BatchNorm biases only
        assert abs(gg['worst_ratio'] - 1) <= GRAD_RATIO_TOL and not gg['outside_5pct'], gg


def _settled_full_width_model(nc, nt_inf, gain, seed, x_warm, ne):
    import srvp_amd
    torch.manual_seed(seed)
    ctor = (64, nc, 64, 128, 50, 50, True, nt_inf, 256, 3, 512, 4, 'vgg')
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(gain)
    model.cuda().train()
    with torch.no_grad():                     # settle the BN running statistics (eval mode uses them)
        for _ in range(12):
            model(x_warm.cuda(), x_warm.shape[0], 1 / ne)
    model.eval()
    return model, ctor


def test_batched_samples_vs_oracle_full_width():
    """SURVEY §8f-1 against the ORACLE: model.sample(x, nt, S=3) on a given tape == three oracle inference forwards
    (reference train.py:170-174 / test.py:237-246: one forward per sample, each re-encoding the conditioning frames)."""
    from oracle import srvp_oracle as O
    nc, nt_cond, nt, ne, S, B = 3, 4, 10, 2, 3, 2
    g = torch.Generator().manual_seed(99)
    xw = torch.rand(nt_cond, 6, nc, 64, 64, generator=g)
    model, ctor = _settled_full_width_model(nc, 2, 1.2, 5, xw, ne)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = xw[:, :B].contiguous()
    eps_y0 = torch.randn(S * B, 50, generator=g)
    eps_z = torch.randn(nt - 1, S * B, 50, generator=g)
    xs = model.sample(x.cuda(), nt, S, dt=1 / ne, tape=dict(eps_y0=eps_y0, eps_z=eps_z)).cpu()
    assert xs.shape == (nt, S, B, nc, 64, 64)
    torch.set_num_threads(8)
    errs = []
    with torch.no_grad():
        for s in range(S):
            tape = dict(eps_y0=eps_y0[s * B:(s + 1) * B], eps_z=eps_z[:, s * B:(s + 1) * B])
            ref = O.forward({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x, nt, ne, tape, training=False)[0]
            errs.append((xs[:, s] - ref).abs().max().item())
    report(test='sample_vs_oracle', errs=errs)
    assert max(errs) < 3e-2, errs
    assert (xs[:, 0] - xs[:, 1]).abs().max() > 1e-3                 # the futures do differ


def test_evaluate_best_of_n_vs_oracle_full_width():
    """train.evaluate (reference train.py:132-189) at full width: best-of-N PSNR selection from model.sample + the device
    metrics kernel, against the same protocol done entirely by the oracle (per-sample inference forwards on the same
    draws, float64 PSNR, arg-max per video)."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd.train import evaluate
    nc, nt_cond, nt, ne, S, B = 1, 3, 8, 2, 3, 2
    g = torch.Generator().manual_seed(17)
    xw = torch.rand(nt, 6, nc, 64, 64, generator=g)
    model, ctor = _settled_full_width_model(nc, 2, 1.2, 6, xw[:nt_cond], ne)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = xw[:, :B].contiguous()
    dev = torch.device('cuda')
    opt = srvp_amd.DotDict(dict(nt_cond=nt_cond, n_iter_test=1, n_samples_test=S, n_euler_steps=ne))
    torch.manual_seed(77)
    got = evaluate(model, [x], dev, opt)
    # the draws evaluate() made: model.sample draws eps_y0 then eps_z from the device generator (model._draw_tape)
    torch.manual_seed(77)
    eps_y0 = torch.randn(B * S, 50, device=dev).cpu()
    eps_z = torch.randn(nt - 1, B * S, 50, device=dev).cpu()
    torch.set_num_threads(8)
    ps, xs = [], []
    with torch.no_grad():
        for s in range(S):
            tape = dict(eps_y0=eps_y0[s * B:(s + 1) * B], eps_z=eps_z[:, s * B:(s + 1) * B])
            x_s = O.forward({k: v.clone() for k, v in sd.items()}, O.make_cfg(*ctor), x[:nt_cond], nt, ne, tape, training=False)[0]
            xs.append(x_s)
            ps.append(O.video_psnr(x_s, x).mean(dim=(0, 2)))                      # (B,)  train.py:175-176
    best = torch.stack(ps).argmax(0)
    bx = torch.stack([xs[best[b]][:, b] for b in range(B)], 1)
    want = -O.video_psnr(bx, x)[nt_cond:].mean().item()
    report(test='evaluate_vs_oracle', got=got, want=want)
    # PSNR of bf16-path frames (<= 3e-2 max-abs, ~1e-3 typical) against fp32-path frames on noise-like targets
    assert abs(got - want) <= 2e-3 * abs(want), (got, want)
