"""
GPU parity of the evaluation metrics kernel (`srvp_frame_metrics` through srvp_amd.metrics) against the reference
fixture (tests/golden/metrics.npz, made from metrics/ssim.py + test.py:249-253) and against the float64 oracle.
fp32 on the device: tolerance 2e-5 relative on SSIM / 1e-5 on MSE and PSNR, written below.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def close(a, b, rtol, atol, what):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    err = (a - b).abs()
    assert (err <= atol + rtol * b.abs()).all(), (what, err.max().item())


@pytest.mark.parametrize('C', [1, 3])
def test_metrics_vs_reference_fixture(C):
    from srvp_amd import metrics
    z = np.load(GOLDEN + '/metrics.npz')
    pred, gt = torch.from_numpy(z[f'c{C}.pred']).cuda(), torch.from_numpy(z[f'c{C}.gt']).cuda()
    close(metrics.ssim(pred, gt), z[f'c{C}.ssim'], 2e-5, 2e-6, 'ssim')
    close(metrics.mse(pred, gt), z[f'c{C}.mse'], 1e-5, 1e-9, 'mse')
    close(metrics.psnr(pred, gt), z[f'c{C}.psnr'], 1e-5, 1e-5, 'psnr')
    assert metrics.ssim(pred, gt).shape == (3, 2, C)


@pytest.mark.parametrize('shape,F,sigma', [((2, 3, 1, 64, 64), 11, 1.5), ((1, 5, 3, 32, 48), 7, 1.0), ((4, 1, 2, 11, 11), 11, 1.5),
                                           ((1, 2, 1, 20, 64), 15, 2.5)])
def test_metrics_vs_oracle_shapes(shape, F, sigma):
    """Other plane sizes / windows (incl. the single-window 11x11 plane) against the float64 oracle."""
    from oracle import srvp_oracle as O
    from srvp_amd import metrics
    g = torch.Generator().manual_seed(sum(shape) + F)
    gt = torch.rand(*shape, generator=g)
    gt = torch.nn.functional.avg_pool2d(gt.view(-1, 1, *shape[3:]), 5, 1, 2).view(shape)      # some structure
    pred = (gt + 0.05 * torch.randn(*shape, generator=g)).clamp(0, 1)
    mse, ssim = metrics.frame_metrics(pred.cuda(), gt.cuda(), max_val=1.0, filter_size=F, sigma=sigma)
    ref = O.ssim_map(pred, gt, 1.0, F, 0.01, 0.03, sigma).mean(dim=(-1, -2))
    close(ssim, ref, 2e-5, 2e-6, 'ssim')
    close(mse, O.video_mse(pred, gt), 1e-5, 1e-9, 'mse')


def test_metrics_errors_and_identity():
    from srvp_amd import metrics
    x = torch.rand(2, 2, 1, 64, 64).cuda()
    mse, ssim = metrics.frame_metrics(x, x)
    assert (mse == 0).all() and (ssim - 1).abs().max() < 1e-6
    with pytest.raises(ValueError):
        metrics.ssim(x, x[:1])
    with pytest.raises(RuntimeError):
        metrics.frame_metrics(torch.rand(1, 1, 1, 65, 64).cuda(), torch.rand(1, 1, 1, 65, 64).cuda())
    with pytest.raises(RuntimeError):
        metrics.frame_metrics(x, x, filter_size=10)


@pytest.mark.parametrize('case', ['s15', 's30', 'd20', 'fast3'])
def test_mmnist_render_vs_reference_fixture(case):
    """srvp_mmnist_render (whole batch assembled on the device) on the reference's trajectories (oracle restatement
    under the reference's np.random stream) == the reference generator's uint8 videos, and the float batch == its collate_fn
    output, bit for bit."""
    from srvp_amd import mmnist as MM
    from oracle import mmnist_ref as R
    z = np.load(GOLDEN + '/mmnist.npz')
    T, ms, det, nd, seed, B = [int(v) for v in z[f'{case}.cfg']]
    gen = MM.MovingMNISTBatches(list(z['digits']), 64, T, ms, bool(det), nd)
    np.random.seed(seed)
    idx, pos = R.draw(len(z['digits']), 28, 28, 64, T, ms, bool(det), nd, B)
    x, u8 = gen.render(idx, pos, True, True)
    assert u8.dtype == torch.uint8 and (u8.cpu().numpy() == z[f'{case}.videos']).all()
    assert x.shape == (T, B, 1, 64, 64)
    ref = torch.from_numpy(z[f'{case}.videos']).float().div(255).permute(1, 0, 2, 3).unsqueeze(2)
    assert torch.equal(x.cpu(), ref)
    if case == 's15':
        assert torch.equal(x.cpu(), torch.from_numpy(z['s15.batch']))


@pytest.mark.parametrize('T,ms,det,nd,B', [(15, 4, False, 2, 16), (30, 9, False, 3, 5), (20, 4, True, 2, 8), (12, 1, False, 1, 3)])
def test_mmnist_device_trajectories_equal_cpu_restatement(T, ms, det, nd, B):
    """The trajectory kernel against the CPU restatement of the same algorithm (oracle/mmnist_ref.philox_trajectories: same
    Philox4x32-10 stream, same float64 ray / box walk): digit indices, every rounded position and the contact counts, bit for bit,
    for two batch counters; batches differ from each other and are reproducible."""
    from srvp_amd import mmnist as MM
    from oracle import mmnist_ref as R
    z = np.load(GOLDEN + '/mmnist.npz')
    gen = MM.MovingMNISTBatches(list(z['digits']), 64, T, ms, det, nd, seed=1234567890123)
    outs = []
    for k in range(2):
        idx, pos, con = (t.cpu().numpy().copy() for t in gen.trajectories(B, want_contacts=True))
        ri, rp, rc = R.philox_trajectories(1234567890123, k, B, nd, T, 64, 28, 28, ms, det, len(z['digits']))
        assert (idx == ri).all() and (pos == rp).all() and (con == rc).all(), k
        assert pos.min() >= 0 and pos.max() <= 64 - 28
        outs.append(pos)
    assert (outs[0] != outs[1]).any()
    gen.counter = 0
    assert (gen.trajectories(B)[1].cpu().numpy() == outs[0]).all()


@pytest.mark.parametrize('ms,det', [(4, False), (9, False), (4, True)])
def test_mmnist_device_generator_matches_reference_in_distribution(ms, det):
    """The device generator draws from the reference's process (data/mmnist.py:113-237): 8192 device trajectories against 3000 of
    the reference restatement (oracle/mmnist_ref.trajectory, itself bit-equal to the reference under np.random.seed): start
    position marginals, per-frame position mean and spread, displacement-magnitude distribution late in the sequence (after
    redraws at the walls) and the fraction of objects touching a wall per frame.  Thresholds = several standard errors."""
    from srvp_amd import mmnist as MM
    from oracle import mmnist_ref as R
    z = np.load(GOLDEN + '/mmnist.npz')
    T, nx, d = 15, 64, 28
    gen = MM.MovingMNISTBatches(list(z['digits']), nx, T, ms, det, 2, seed=99)
    dev = np.concatenate([gen.trajectories(1024)[1].cpu().numpy().reshape(-1, T, 2) for _ in range(4)]).astype(np.float64)   # 8192
    np.random.seed(4242)
    ref = np.array([[(r, c) for r, c, _, _ in R.trajectory(d, d, nx, T, ms, det)] for _ in range(3000)], dtype=np.float64)
    n = min(len(dev), len(ref))
    se = lambda v: v.std() / np.sqrt(n)
    hi = nx - d
    for ax in (0, 1):
        hd = np.bincount(dev[:, 0, ax].astype(int), minlength=hi + 1) / len(dev)
        hr = np.bincount(ref[:, 0, ax].astype(int), minlength=hi + 1) / len(ref)
        assert np.abs(hd - hr).max() < 0.012, np.abs(hd - hr).max()                      # start positions: uniform on {0..36}
        for t in range(T):
            assert abs(dev[:, t, ax].mean() - ref[:, t, ax].mean()) < 5 * se(ref[:, t, ax]) + 0.05, (ax, t)
            assert abs(dev[:, t, ax].std() - ref[:, t, ax].std()) < 0.06 * ref[:, t, ax].std() + 0.05, (ax, t)
        wd = ((dev[:, :, ax] == 0) | (dev[:, :, ax] == hi)).mean(0)                       # on-wall fraction per frame
        wr = ((ref[:, :, ax] == 0) | (ref[:, :, ax] == hi)).mean(0)
        assert np.abs(wd - wr).max() < 0.03, (wd, wr)
        dd, dr = np.abs(np.diff(dev[:, -6:, ax], axis=1)).ravel(), np.abs(np.diff(ref[:, -6:, ax], axis=1)).ravel()
        hd = np.bincount(dd.astype(int), minlength=ms + 2)[:ms + 2] / len(dd)
        hr = np.bincount(dr.astype(int), minlength=ms + 2)[:ms + 2] / len(dr)
        assert np.abs(hd - hr).max() < 0.02, (hd, hr)                                      # speeds after redraws
