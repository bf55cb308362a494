"""
On-device evaluation metrics: MSE / PSNR (reference test.py:249-251, train.py:175-176) and the
pixel-averaged SSIM of reference test.py:36-57 (`_ssim_wrapper`, metrics/ssim.py:81-149), one `srvp_frame_metrics` launch for
all (frame, channel) planes of a video batch.  Same argument meaning and output shapes as the reference helpers.
"""
import torch

from . import _lib as L


def _planes(sample, gt):
    if sample.shape != gt.shape:
        raise ValueError('Expected input size ({}) to match target size ({}).'.format(tuple(sample.shape), tuple(gt.shape)))
    assert sample.dim() == 5, 'videos are (length, batch, channels, width, height) (test.py:43-46)'
    assert sample.is_cuda and gt.is_cuda, 'srvp_amd.metrics runs on the MI355X: move the videos to the device'
    return sample.contiguous().float(), gt.contiguous().float()


def frame_metrics(sample, gt, max_val=1.0, filter_size=11, sigma=1.5, k1=0.01, k2=0.03, want_mse=True, want_ssim=True):
    """(mse, ssim), each (length, batch, channels) float32 (or None when not wanted)."""
    x, y = _planes(sample, gt)
    nt, bsz, C, H, W = x.shape
    mse = torch.empty(nt, bsz, C, device=x.device) if want_mse else None
    ssim = torch.empty(nt, bsz, C, device=x.device) if want_ssim else None
    L.call('srvp_frame_metrics', L.ptr(x), L.ptr(y), nt * bsz * C, H, W, float(max_val), int(filter_size), float(sigma),
           float(k1), float(k2), L.ptr(mse), L.ptr(ssim), L.stream())
    return mse, ssim


def mse(sample, gt):
    """torch.mean((sample - gt)**2, dim=[3, 4]) (test.py:249) -> (length, batch, channels)."""
    return frame_metrics(sample, gt, want_ssim=False)[0]


def psnr(sample, gt):
    """10 log10(1 / mse) per (frame, video, channel) (test.py:251, train.py:175-176)."""
    return 10 * torch.log10(1 / mse(sample, gt))


def ssim(sample, gt, max_val=1.0):
    """Pixel-averaged SSIM between two videos, (length, batch, channels) (test.py:36-57)."""
    return frame_metrics(sample, gt, max_val=max_val, want_mse=False)[1]
