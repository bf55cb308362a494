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
def test_mmnist_batches_vs_reference_fixture(case):
    """srvp_mmnist_render (whole batch assembled on the device) under the reference's np.random stream == the
    reference generator's uint8 videos, and the float batch == its collate_fn output, bit for bit."""
    from srvp_amd import mmnist as MM
    z = np.load(GOLDEN + '/mmnist.npz')
    T, ms, det, nd, seed, B = [int(v) for v in z[f'{case}.cfg']]
    gen = MM.MovingMNISTBatches(list(z['digits']), 64, T, ms, bool(det), nd)
    np.random.seed(seed)
    u8 = gen.videos_u8(B)
    assert u8.dtype == torch.uint8 and (u8.cpu().numpy() == z[f'{case}.videos']).all()
    np.random.seed(seed)
    x = gen.batch(B)
    assert x.shape == (T, B, 1, 64, 64)
    ref = torch.from_numpy(z[f'{case}.videos']).float().div(255).permute(1, 0, 2, 3).unsqueeze(2)
    assert torch.equal(x.cpu(), ref)
    if case == 's15':
        assert torch.equal(x.cpu(), torch.from_numpy(z['s15.batch']))
