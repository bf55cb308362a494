"""
GPU parity of the fp32 latent path in isolation (content variable, y_0, LSTM + q_z, prior, residual Euler rollout and
their backward) against the CPU oracle's autograd on the same frame encodings and noise tape.  fp32 on both sides:
tolerance 1e-4 relative (accumulation order only).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()


# (the persistent fused rollout kernels take the chains whose hidden width is a multiple of 32: cases 2 and 4-7 -- headline
# dimensions, SM-MNIST dimensions on three ragged row tiles, 2- and 3-layer MLPs, one Euler step per frame)
FUSED = {1: False, 2: True, 3: False, 4: True, 5: True, 6: True, 7: True, 8: True, 9: True}


@pytest.mark.parametrize('case,ne,T,B,dims', [(1, 1, 4, 3, (8, 3, 3, 8, 16, 3, 4, 2)), (2, 2, 5, 6, (128, 50, 50, 256, 512, 3, 4, 2)),
                                               (3, 2, 4, 5, (16, 5, 7, 24, 40, 2, 3, 3)), (4, 1, 6, 70, (32, 20, 20, 64, 512, 2, 4, 3)),
                                               (5, 2, 4, 33, (16, 10, 6, 32, 64, 2, 2, 2)), (6, 2, 4, 40, (16, 12, 9, 32, 96, 2, 3, 2)),
                                               (7, 4, 3, 192, (32, 50, 50, 64, 512, 2, 4, 2)),
                                               # 19 row tiles at 16 workgroups per cluster: more clusters than fit on the chip at once,
                                               # so the chain runs as two co-resident launches (tile0 > 0 in the second)
                                               (8, 1, 3, 600, (16, 20, 20, 32, 512, 2, 3, 2)),
                                               # (round 6) 19 sixteen-row tiles at nh_inf = 256 / nh_res = 512: the persistent LSTM (forward AND
                                               # backward) and both rollout kernels run as two co-resident launches each
                                               (9, 1, 3, 300, (16, 20, 20, 256, 512, 2, 3, 2))])
def test_latent_forward_backward(case, ne, T, B, dims):
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd import _lib as L
    from srvp_amd.latent import LatentNet
    nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = dims
    ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
    torch.manual_seed(5)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    cfg = O.make_cfg(*ctor)
    g = torch.Generator().manual_seed(9)
    hx = torch.tanh(torch.randn(T, B, nhx, generator=g))
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                eps_y0=torch.randn(B, ny, generator=g), eps_z=torch.randn(T - 1, B, nz, generator=g))
    # ---- oracle (the large-batch cases in float64: the CPU's own fp32 autograd moves by 1.5e-4 on d_hx / 2.5e-3 on a bias gradient between
    # 8 and 128 host threads at B = 300 -- more than the tolerances below -- while the HIP result is the same to 1e-9 either way; measured
    # round 6, when the case passed alone and failed behind any test file that had called torch.set_num_threads(8))
    odt = torch.float64 if B >= 300 else torch.float32
    cast = lambda v: v.to(odt) if v.is_floating_point() else v
    sd = {k: cast(v.detach().clone()) for k, v in model.state_dict().items()}
    lat_keys = [k for k in sd if not k.startswith(('encoder.', 'decoder.'))]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in lat_keys}
    work = dict(sd)
    work.update(leaves)
    hxr = cast(hx).clone().requires_grad_(True)
    w = O.infer_w(work, cfg, hxr, True, tape['t_w'])
    y0, qy0 = O.infer_y(work, cfg, hxr[:nt_inf], cast(tape['eps_y0']))
    y, z, qz, pz, res = O.generate(work, cfg, y0, hxr, T, ne, cast(tape['eps_z']), True)
    cot = {n: torch.randn(t.shape, generator=g) for n, t in dict(y=y, w=w, qy0=qy0, qz=qz, pz=pz, res=res).items()}
    total = (y * cast(cot['y'])).sum() + (w * cast(cot['w'])).sum() + (qy0 * cast(cot['qy0'])).sum() + (qz * cast(cot['qz'])).sum() + \
        (pz * cast(cot['pz'])).sum() + (res * cast(cot['res'])).sum()
    gl = torch.autograd.grad(total, [hxr] + [leaves[k] for k in lat_keys])
    g_hx, g_par = gl[0], dict(zip(lat_keys, gl[1:]))
    # ---- HIP
    model = model.cuda()
    model.flatten_parameters_()
    grads = model._grads()
    model._flat[1].zero_()
    params = model._named_tensors()
    st = L.stream()
    lat = LatentNet(model._cfg(), T, B, T, ne, torch.device('cuda'), True)
    hxg = hx.cuda()
    tg = {k: v.cuda() for k, v in tape.items()}
    w_g = lat.infer_w(hxg, params, tg['t_w'], st)
    y0_g, qy0_g = lat.infer_y(hxg[:nt_inf], params, tg['eps_y0'], st)
    lat.posterior(hxg, params, st)
    y_g, z_g, qz_g, pz_g, res_g = lat.generate(y0_g, T, params, tg['eps_z'], st)
    torch.cuda.synchronize()
    assert bool(lat._rd.fused_ws) == FUSED[case], 'fused-rollout eligibility changed'
    if FUSED[case]:
        # the same chain through the per-layer launch sequence: identical algorithm, different fp32 summation order
        keep = {n: t.clone() for n, t in dict(y=y_g, res=res_g, hid=lat.hid_dyn, inp=lat.inp_all).items()}
        lat._rd.fused_ws = None
        L.call('srvp_rollout_fwd', __import__('ctypes').byref(lat._rd), st)
        torch.cuda.synchronize()
        for n, t in dict(y=lat.y_all[::ne], res=lat.res[:lat.S], hid=lat.hid_dyn, inp=lat.inp_all).items():
            assert rel(keep[n], t) < 2e-5, (n, rel(keep[n], t))
        lat._rd.fused_ws = L.ptr(lat._fused_ws)
    for n, a, b in (('w', w_g, w), ('qy0', qy0_g, qy0), ('y', y_g, y), ('z', z_g, z), ('qz', qz_g, qz), ('pz', pz_g, pz),
                    ('res', res_g, res)):
        assert rel(a, b.detach()) < 1e-4, (n, rel(a, b.detach()))
    c = {k: v.cuda().contiguous() for k, v in cot.items()}
    d_hx = lat.backward(hxg, params, grads, tg['eps_y0'], tg['eps_z'], c['y'], c['w'], c['qy0'], c['qz'], c['pz'], c['res'],
                        None, st)
    torch.cuda.synchronize()
    assert rel(d_hx.view(T, B, nhx), g_hx) < 2e-4, rel(d_hx.view(T, B, nhx), g_hx)
    for k in lat_keys:
        assert rel(grads[k], g_par[k]) < 5e-4, (k, rel(grads[k], g_par[k]))


def test_fused_rollout_under_load_is_deterministic():
    """The persistent rollout kernels exchange activations between workgroups inside one launch (agent-scope stores / loads +
    a counter barrier, csrc/rollout_fused.hip).  Such hand-offs fail under UNEVEN load with warm caches, not on an idle chip:
    run forward + backward 25 times while a second stream keeps every CU busy with unrelated streaming work, on buffers that
    are re-used every time, and require results BIT-IDENTICAL to the idle-chip run (the kernels sum in a fixed order) -- which
    itself matches the per-layer launch sequence (test_latent_forward_backward)."""
    import ctypes
    import srvp_amd
    from srvp_amd import _lib as L
    from srvp_amd.latent import LatentNet
    ne, T, B = 2, 12, 200
    nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = 128, 50, 50, 256, 512, 3, 4, 2
    ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
    torch.manual_seed(5)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    model = model.cuda()
    model.flatten_parameters_()
    params = model._named_tensors()
    g = torch.Generator().manual_seed(9)
    dev = torch.device('cuda')
    st = L.stream()
    lat = LatentNet(model._cfg(), T, B, T, ne, dev, True)
    hx = torch.tanh(torch.randn(T, B, nhx, generator=g)).to(dev)
    t_w = torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1).to(dev)
    eps_y0, eps_z = torch.randn(B, ny, generator=g).to(dev), torch.randn(T - 1, B, nz, generator=g).to(dev)
    lat.infer_w(hx, params, t_w, st)
    y0, _ = lat.infer_y(hx[:nt_inf], params, eps_y0, st)
    lat.posterior(hx, params, st)
    d_res = torch.randn(lat.S, B, ny, generator=g).to(dev)
    lat.d_y_all.copy_(torch.randn(lat.S + 1, B, ny, generator=g))

    def run():
        lat.generate(y0, T, params, eps_z, st)
        assert lat._rd.fused_ws
        bd = L.RolloutBwdDesc()
        bd.f = lat._rd
        bd.d_y_all, bd.d_z, bd.d_pz, bd.d_res = L.ptr(lat.d_y_all), None, None, L.ptr(d_res)
        bd.d_y0, bd.d_qz, bd.dhid_dyn, bd.dhid_pz, bd.work = (L.ptr(lat.d_y0), L.ptr(lat.d_qz_samp), L.ptr(lat.dhid_dyn),
                                                               L.ptr(lat.dhid_pz), L.ptr(lat.work))
        bd.dinp_all = L.ptr(lat.dinp_all)
        L.call('srvp_rollout_bwd', ctypes.byref(bd), st)
        return [t.clone() for t in (lat.y_all, lat.res, lat.hid_dyn, lat.dhid_dyn[:nl_res - 1], lat.dinp_all, lat.d_y0)]
    ref = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=dev)
    big = torch.empty(1 << 28, device=dev)
    for rep in range(25):
        with torch.cuda.stream(side):                     # uneven background load: a GEMM and a 1 GiB streaming fill
            for _ in range(2):
                a @ a
                big.fill_(float(rep))
        got = run()
        torch.cuda.synchronize()
        for k, (x, r) in enumerate(zip(got, ref)):
            assert torch.equal(x, r), (rep, k, (x - r).abs().max().item())


@pytest.mark.parametrize('M,N,K', [(512, 100, 4224), (50, 512, 2112), (37, 129, 515), (1024, 128, 2304), (6, 7, 130)])
def test_gemm_tn_weight_gradient_shape(M, N, K):
    """srvp_gemm_f32 in the weight-gradient form C += A^T B (both operands row-major over K: the LDS-staged split-K kernel),
    aligned and ragged sizes, against float64."""
    from srvp_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    dev = torch.device('cuda')
    lda, ldb = M + (3 if M % 4 else 0), N + (5 if N % 4 else 8)
    A = torch.randn(K, lda, generator=g).to(dev)
    B = torch.randn(K, ldb, generator=g).to(dev)
    C0 = torch.randn(M, N, generator=g).to(dev)
    Cm = C0.clone()
    # A element (m, k) at A[m * 1 + k * lda], B element (k, n) at B[k * ldb + n]
    L.call('srvp_gemm_f32', L.ptr(A), 1, lda, L.ptr(B), ldb, 1, None, L.ptr(Cm), N, M, N, K, L.ACT_NONE, 1, L.stream())
    ref = C0.double() + A[:, :M].double().t() @ B[:, :N].double()
    assert rel(Cm, ref) < 2e-6, rel(Cm, ref)


@pytest.mark.parametrize('M,N,K,bt,act,acc,bias', [(2112, 512, 512, True, 3, 0, True), (2304, 1024, 128, True, 0, 0, True), (2112, 100, 512, True, 0, 0, True),
                                                  (2112, 512, 100, False, 0, 0, False), (2304, 128, 1024, False, 0, 1, False), (1792, 512, 40, True, 3, 0, True),
                                                  (515, 100, 36, True, 2, 1, True), (1000, 260, 68, False, 0, 0, False)])
def test_gemm_tiled_rows_in_the_thousands(M, N, K, bt, act, acc, bias):
    """srvp_gemm_f32 on the shapes of the batched inference MLPs and their data gradients (round 6: 64x64 LDS-tiled exact-fp32 MFMA kernel,
    csrc/latent.hip gemm_f32_tiled_kernel): y = act(x W^T + b) with W given as [N][K] (bt) and dx (+)= dy W with W row-major [K][N], ragged
    M / N / K, against float64 -- and against the 32x32 kernel the small shapes keep (same exact-fp32 products, another summation order)."""
    from srvp_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    dev = torch.device('cuda')
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) if bt else torch.randn(K, N, generator=g)).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    C0 = torch.randn(M, N, generator=g).to(dev)
    Cm = C0.clone()
    if bt:      # B[k][n] = W[n][k]: b_rs = 1, b_cs = K
        L.call('srvp_gemm_f32', L.ptr(A), K, 1, L.ptr(W), 1, K, L.ptr(b), L.ptr(Cm), N, M, N, K, act, acc, L.stream())
        pre = A.double() @ W.double().t()
    else:       # B[k][n] = W[k][n]: b_rs = N, b_cs = 1
        L.call('srvp_gemm_f32', L.ptr(A), K, 1, L.ptr(W), N, 1, L.ptr(b), L.ptr(Cm), N, M, N, K, act, acc, L.stream())
        pre = A.double() @ W.double()
    if bias:
        pre = pre + b.double()
    ref = {0: lambda v: v, 2: torch.tanh, 3: torch.relu}[act](pre)
    if acc:
        ref = ref + C0.double()
    assert rel(Cm, ref) < 2e-6, rel(Cm, ref)


def test_elbo_and_adam_kernels():
    """srvp_nll / srvp_kl / srvp_l2rows values + gradients and srvp_adam against the oracle formulas (train.py:90-106,289)."""
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    dev = torch.device('cuda')
    st = L.stream()
    x_ = torch.rand(3, 2, 3, 64, 64, generator=g).requires_grad_(True)
    x = torch.rand(3, 2, 3, 64, 64, generator=g)
    q = (torch.randn(4, 6, 10, generator=g) * 2).requires_grad_(True)
    p = torch.randn(4, 6, 10, generator=g).requires_grad_(True)
    with torch.no_grad():
        q[0, 0, 7] = 25.0
    res = torch.randn(7, 6, 5, generator=g)
    res[2, 3] = 0
    res.requires_grad_(True)
    nll = O.neg_logprob(x_, x, 0.2).sum()
    klq = O.kl_normal(q, p).sum()
    kl0 = O.kl_normal(q, None).sum()
    l2 = torch.norm(res, p=2, dim=2).sum()
    (0.5 * nll + 0.25 * klq + 2.0 * l2).backward()
    acc = torch.zeros(4, dtype=torch.float64, device=dev)
    xd, x_d, qd, pd, rd = x.to(dev), x_.detach().to(dev), q.detach().to(dev), p.detach().to(dev), res.detach().to(dev)
    d_x, dq, dp, dr = torch.empty_like(x_d), torch.empty_like(qd), torch.empty_like(pd), torch.empty_like(rd)
    L.call('srvp_nll', L.ptr(x_d), L.ptr(xd), L.ptr(d_x), x_d.numel(), 0.2, 0.5, L.ptr(acc[0:1]), st)
    L.call('srvp_kl', L.ptr(qd), L.ptr(pd), L.ptr(dq), L.ptr(dp), 24, 5, 0.25, L.ptr(acc[1:2]), st)
    L.call('srvp_kl', L.ptr(qd), None, None, None, 24, 5, 1.0, L.ptr(acc[2:3]), st)
    L.call('srvp_l2rows', L.ptr(rd), L.ptr(dr), 42, 5, 2.0, L.ptr(acc[3:4]), st)
    a = acc.cpu()
    for got, ref in zip(a.tolist(), (nll.item(), klq.item(), kl0.item(), l2.item())):
        assert abs(got - ref) < 1e-5 * abs(ref), (got, ref)
    assert rel(d_x, x_.grad) < 1e-5 and rel(dq, q.grad) < 1e-4 and rel(dp, p.grad) < 1e-4 and rel(dr, res.grad) < 1e-5
    # Adam: 3 steps against the oracle's update
    n = 1003
    p0 = torch.randn(n, generator=g)
    sd = {'p': p0.clone()}
    state = {}
    pd_, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        O.adam_step(sd, {'p': gr}, state, 3e-4)
        grd = gr.to(dev)
        L.call('srvp_adam', L.ptr(pd_), L.ptr(grd), L.ptr(m), L.ptr(v), n, 3e-4, 0.9, 0.999, 1e-8, step, 1.0, st)
    assert (pd_.cpu() - sd['p']).abs().max().item() < 1e-6


@pytest.mark.parametrize('T,B,nh', [(12, 192, 256), (5, 37, 256), (1, 3, 128), (20, 100, 64), (7, 300, 128), (4, 300, 256)])
def test_persistent_lstm_forward(T, B, nh):
    rel_l2 = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
    """srvp_lstm_fwd_fused (one persistent launch over the T steps) against the per-step launch sequence and against torch's
    nn.LSTM recurrence on the CPU (srvp.py:132,366): hidden / cell states and the saved gate activations."""
    import ctypes as C
    from srvp_amd import _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(T * 1000 + B + nh)
    gx = torch.randn(T, B, 4 * nh, generator=g)                       # x W_ih^T + b_ih + b_hh, precomputed
    whh = torch.randn(4 * nh, nh, generator=g) * (1.0 / nh ** 0.5)
    gxd, wd = gx.to(dev), whh.to(dev)
    st = L.stream()
    outs = {}
    for fused in (0, 1):
        h = torch.full((T, B, nh), 7.0, device=dev); c = torch.full((T, B, nh), 7.0, device=dev); ga = torch.full((T, B, 4 * nh), 7.0, device=dev)
        if fused:
            need = int(L.load().srvp_lstm_fused_ws_bytes(T, B, nh))
            assert need > 0
            ws = torch.zeros(need, dtype=torch.uint8, device=dev)
            L.call('srvp_lstm_fwd_fused', L.ptr(gxd), L.ptr(wd), L.ptr(h), L.ptr(c), L.ptr(ga), T, B, nh, L.ptr(ws), need, st)
        else:
            L.call('srvp_lstm_fwd', L.ptr(gxd), L.ptr(wd), L.ptr(h), L.ptr(c), L.ptr(ga), T, B, nh, st)
        torch.cuda.synchronize()
        outs[fused] = (h.cpu(), c.cpu(), ga.cpu())
    # CPU recurrence (gate order i, f, g, o)
    hp, cp = torch.zeros(B, nh, dtype=torch.float64), torch.zeros(B, nh, dtype=torch.float64)
    hs, cs = [], []
    w64 = whh.double()
    for t in range(T):
        a = gx[t].double() + hp @ w64.t()
        i, f, gg, o = torch.sigmoid(a[:, :nh]), torch.sigmoid(a[:, nh:2 * nh]), torch.tanh(a[:, 2 * nh:3 * nh]), torch.sigmoid(a[:, 3 * nh:])
        cp = f * cp + i * gg
        hp = o * torch.tanh(cp)
        hs.append(hp); cs.append(cp)
    href, cref = torch.stack(hs), torch.stack(cs)
    for fused in (0, 1):
        h, c, ga = outs[fused]
        assert rel_l2(h, href) < 2e-6 and rel_l2(c, cref) < 2e-6, (fused, rel_l2(h, href), rel_l2(c, cref))
    assert rel_l2(outs[1][2], outs[0][2]) < 2e-6
    assert (gxd.cpu() == gx).all()                                     # the fused entry point leaves gates_x untouched


# (B, T observed frames, nt generated frames, n_euler, dims): headline dimensions on one ragged and on three tiles; SM-MNIST dimensions (ny = nz = 20);
# 27 tiles = more clusters than fit on the chip at once (two co-resident launches, tile0 > 0 in the second); posterior only up to frame 1
@pytest.mark.parametrize('B,T,nt,ne,dims', [(5, 4, 9, 2, (128, 50, 50, 256, 512, 3, 4, 3)), (70, 3, 7, 2, (128, 50, 50, 256, 512, 3, 4, 2)),
                                            (40, 5, 8, 1, (32, 20, 20, 64, 512, 2, 4, 3)), (850, 2, 5, 2, (16, 50, 50, 32, 512, 2, 4, 2)),
                                            (33, 1, 6, 4, (16, 12, 12, 32, 96, 2, 3, 1))])
def test_generation_chain_persistent_vs_launches_and_oracle(B, T, nt, ne, dims):
    """The INFERENCE rollout (reference module/srvp.py:377-405: posterior samples while the observed frames last, prior samples p_z(y) afterwards;
    test.py:237-246) as persistent launches (csrc/rollout_fused.hip rollout_gen_ks_kernel: p_z, the sample and the n_euler residual steps of every
    frame inside one kernel) against (a) the per-layer launch sequence on the same buffers -- same algorithm, another fp32 summation order -- and
    (b) the CPU oracle: states y, samples z, prior parameters p_z and residuals at every step."""
    import ctypes
    import srvp_amd
    from oracle import srvp_oracle as O
    from srvp_amd import _lib as L
    from srvp_amd import latent as LT
    nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = dims
    ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
    torch.manual_seed(7)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(0.8)                                            # (contractive residual function: the chain does not amplify rounding differences)
    cfg = O.make_cfg(*ctor)
    g = torch.Generator().manual_seed(13)
    hx = torch.tanh(torch.randn(T, B, nhx, generator=g))
    eps_y0, eps_z = torch.randn(B, ny, generator=g), torch.randn(nt - 1, B, nz, generator=g)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        y0_r, _ = O.infer_y(sd, cfg, hx[:nt_inf], eps_y0)
        y_r, z_r, qz_r, pz_r, res_r = O.generate(sd, cfg, y0_r, hx, nt, ne, eps_z, False)
    model = model.cuda().eval()
    model.flatten_parameters_()
    params = model._named_tensors()
    st = L.stream()
    dev = torch.device('cuda')
    lat = LT.LatentNet(model._cfg(), T, B, nt, ne, dev, False)
    hxg = hx.to(dev)
    y0_g, _ = lat.infer_y(hxg[:nt_inf], params, eps_y0.to(dev), st)
    lat.posterior(hxg, params, st)
    host = torch.zeros(2, dtype=torch.int32).pin_memory()
    L.call('srvp_cluster_stats_read', host.data_ptr(), st)
    torch.cuda.synchronize()
    fails0 = int(host[0])
    out = {}
    try:
        for fused in (True, False):
            LT.ROLLOUT_GEN_FUSED = fused
            for t in (lat.y_all, lat.z, lat.p_z, lat.res):
                t.fill_(7.0)
            y, z, qz, pz, res = lat.generate(y0_g, T, params, eps_z.to(dev), st)
            torch.cuda.synchronize()
            # (the persistent form takes nh_res = 512 with <= two hidden layers per network -- every recipe of the reference; other widths
            # keep the per-layer launch sequence: the last case)
            assert bool(lat._rd.fused_ws) == (fused and nh_res == 512 and nl_res - 2 <= 2), 'generation-chain eligibility changed'
            out[fused] = [t.clone() for t in (y, z, pz, res)]
    finally:
        LT.ROLLOUT_GEN_FUSED = True
    L.call('srvp_cluster_stats_read', host.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(host[0]) == fails0, 'a generation launch found its cluster spread over several XCCs (or a barrier timed out)'
    for n, a, b in zip(('y', 'z', 'pz', 'res'), out[True], out[False]):
        assert a.shape == b.shape and rel(a, b) < 2e-5, (n, rel(a, b))
    for n, a, b in zip(('y', 'z', 'pz', 'res'), out[True], (y_r, z_r, pz_r, res_r)):
        assert a.shape == b.shape and rel(a, b) < 1e-4, (n, rel(a, b))
    assert (out[True][1][-1] - out[True][1][0]).abs().max() > 1e-3           # the samples do differ from frame to frame
