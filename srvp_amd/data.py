"""
Input side of the hot path.  The reference's dataset readers / generators (data/*.py) are out of scope for this
round; what the training loop needs is the batch contract of data/base.py:54-84 --
float32 (T, B, C, H, W) in [0, 1], time-major -- which `SyntheticVideos` produces from seeded procedural blobs.
"""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


class SyntheticVideos(Dataset):
    def __init__(self, n, seq_len, nc, nx=64, seed=0):
        self.n, self.seq_len, self.nc, self.nx, self.seed = n, seq_len, nc, nx, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.RandomState(self.seed * 1000003 + i)
        yy, xx = np.mgrid[0:self.nx, 0:self.nx].astype(np.float32)
        v = np.zeros((self.seq_len, self.nc, self.nx, self.nx), np.float32)
        for c in range(self.nc):
            p, d, s = rng.uniform(12, 52, 2), rng.uniform(-3, 3, 2), rng.uniform(3, 7)
            for t in range(self.seq_len):
                q = p + d * t
                v[t, c] = np.exp(-((yy - q[0]) ** 2 + (xx - q[1]) ** 2) / (2 * s * s))
        return torch.from_numpy(np.clip(v, 0, 1))


def collate_fn(videos):
    """(B x (T, C, H, W)) -> (T, B, C, H, W) float32, the layout of reference data/base.py:76-84."""
    return torch.stack(videos, 1)


def collate_u8(videos):
    """Device-side variant of reference data/base.py:54-84: stack the per-video uint8 arrays -- (T, H, W) or
    (T, H, W, 3) each -- into one pinned uint8 tensor [B][T][H][W][C]; `frames_from_u8` finishes the job on the GPU (one
    quarter of the PCIe bytes of the float32 batch, no CPU transpose / divide)."""
    arrs = [torch.as_tensor(v) for v in videos]
    u8 = torch.stack([a if a.ndim == 4 else a.unsqueeze(-1) for a in arrs], 0).contiguous()
    assert u8.dtype == torch.uint8
    return u8.pin_memory() if torch.cuda.is_available() else u8


def frames_from_u8(u8, device):
    """uint8 [B][T][H][W][C] (host or device) -> float32 (T, B, C, H, W) in [0, 1] on `device` (srvp_frames_u8_to_f32)."""
    from . import _lib as L
    u8 = u8.to(device, non_blocking=True)
    B, T, H, W, C_ = u8.shape
    out = torch.empty(T, B, C_, H, W, dtype=torch.float32, device=device)
    L.call('srvp_frames_u8_to_f32', L.ptr(u8), L.ptr(out), B, T, H, W, C_, L.stream())
    return out


def make_loaders(opt, local_rank):
    if opt.dataset != 'synthetic':
        raise NotImplementedError(
            f"dataset '{opt.dataset}': the reference's dataset readers are outside the hot path rebuilt here "
            "(SURVEY.md §8f-2); use --dataset synthetic or feed (T, B, C, 64, 64) batches to srvp_amd.train.train")
    sampler = None
    trainset = SyntheticVideos(4096, opt.seq_len, opt.nc, opt.nx, seed=opt.seed + local_rank)
    if opt.n_gpu > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(trainset)
    train_loader = DataLoader(trainset, batch_size=opt.batch_size, collate_fn=collate_fn, sampler=sampler,
                              shuffle=sampler is None, drop_last=True, num_workers=opt.n_workers, pin_memory=True)
    val_loader = None
    if local_rank == 0:
        valset = SyntheticVideos(max(opt.batch_size_test * opt.n_iter_test, 1), opt.seq_len_test or opt.seq_len, opt.nc,
                                 opt.nx, seed=opt.seed + 7919)
        val_loader = DataLoader(valset, batch_size=opt.batch_size_test, collate_fn=collate_fn, shuffle=True, drop_last=True,
                                num_workers=0)
    return train_loader, val_loader, sampler
