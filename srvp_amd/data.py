"""
Input side of the hot path.  The batch contract is data/base.py:54-84 -- float32 (T, B, C, H, W) in [0, 1], time-major.
Built here: the device-side uint8 collate (`collate_u8` / `frames_from_u8`) and the Stochastic Moving-MNIST
training generator (`srvp_amd.mmnist`, `--dataset smmnist`); `SyntheticVideos` (seeded procedural blobs) serves bench / smoke.
The KTH / Human3.6M / BAIR file readers of the reference (data/{kth,human,bair}.py) stay out of scope.
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


class Prefetcher:
    """Asynchronous input (reference train.py:84,262: `batch.to(device)` at the top of the step, DataLoader(pin_memory=True); SURVEY §2a
    "async H2D on copy stream"): wraps an iterable of HOST batches -- float32 (T, B, C, H, W), or the stacked uint8 videos of `collate_u8`
    -- and yields DEVICE batches.  Batch i + 1 is staged (pinned -> HBM copy, and for uint8 the device-side collate) on a copy stream of
    its own right before batch i is handed out, i.e. under step i; the consumer's stream waits on the staging event, so by the time the
    step's first kernel needs the frames they are resident and the step pays neither the PCIe transfer nor the collate.
    Iterables that already yield device tensors (the device Moving-MNIST generator) pass through untouched.
    Lifetime of a yielded batch: it is record_stream'ed for the consumer's CURRENT stream only, while the step also reads it from the
    model's side / auxiliary streams (weight gradient of the image-side layer).  That is safe because srvp_amd.train.train holds the
    batch until it returns and every stream of a step is joined into the current one before the step's last launch (model._backward_impl):
    the allocator's free marker on the current stream lies behind every reader.  A caller that drops the batch earlier must keep it alive
    itself.  Create ONE Prefetcher per run (one copy stream) and iterate it once per epoch."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.copy_stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        if batch.is_cuda:
            return batch, None
        if not batch.is_pinned():
            batch = batch.pin_memory()                       # (an unpinned source would make the copy synchronous)
        ready = torch.cuda.Event()
        ready.record()                                       # the staging buffers are allocated after everything queued so far on the consumer
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            x = frames_from_u8(batch, self.device) if batch.dtype == torch.uint8 else batch.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return (x, batch), ev                                # (the pinned source stays referenced until the copy has been waited for)

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))                  # batch i + 1 goes out BEFORE step i is queued: it travels under step i
            except StopIteration:
                nxt = None
            if ev is None:
                yield cur
                continue
            x, _src = cur
            torch.cuda.current_stream().wait_event(ev)
            x.record_stream(torch.cuda.current_stream())     # allocated on the copy stream, consumed on the compute stream
            yield x


def mnist_digits(data_dir):
    """The 60000 training digits (n, 28, 28) uint8 that reference data/mmnist.py:337-340 takes from torchvision's MNIST,
    read from the raw IDX file torchvision stores under <data_dir>/MNIST/raw (no torchvision, no download here)."""
    import gzip
    import os
    for name in ('MNIST/raw/train-images-idx3-ubyte', 'MNIST/raw/train-images-idx3-ubyte.gz', 'train-images-idx3-ubyte',
                 'train-images-idx3-ubyte.gz'):
        path = os.path.join(data_dir, name)
        if os.path.exists(path):
            with (gzip.open(path, 'rb') if path.endswith('.gz') else open(path, 'rb')) as f:
                raw = f.read()
            magic, n, h, w = np.frombuffer(raw[:16], dtype='>i4')
            assert magic == 2051, f'{path}: not an IDX image file'
            return np.frombuffer(raw, dtype=np.uint8, offset=16).reshape(n, h, w)
    raise FileNotFoundError(f'MNIST training images (train-images-idx3-ubyte[.gz]) not found under {data_dir}[/MNIST/raw]')


def fold_ids(n, fold):
    """reference data/base.py:115-127: 95 % / 5 % split of the training items under the fixed RandomState(42) shuffle."""
    ids = list(range(n))
    np.random.RandomState(42).shuffle(ids)
    n_train = int(0.95 * n)
    keep = set(ids[:n_train] if fold == 'train' else ids[n_train:])
    return [i for i in range(n) if i in keep]


class MovingMNISTLoader:
    """Iterable of device-resident (T, B, 1, nx, nx) training batches (srvp_amd.mmnist.MovingMNISTBatches); the length mirrors
    the reference's arbitrary 500000-item epoch (data/mmnist.py:107-111)."""

    def __init__(self, gen, batch_size, n_items=500000):
        self.gen, self.batch_size, self.n = gen, batch_size, n_items // batch_size

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            yield self.gen.batch(self.batch_size)


def make_loaders(opt, local_rank):
    if opt.dataset == 'smmnist':
        # trajectories (Philox, one thread per object) and frames both generated on the device
        from .mmnist import MovingMNISTBatches
        digits = mnist_digits(opt.data_dir)
        dev = torch.device('cuda', torch.cuda.current_device())
        # distinct streams per rank (the reference seeds numpy with seed + local_rank, train.py:226-228) and for validation
        mk = lambda fold, T: MovingMNISTBatches(digits[fold_ids(len(digits), fold)], opt.nx, T, opt.max_speed, opt.deterministic,
                                                opt.ndigits, device=dev,
                                                seed=(int(opt.seed or 0) + local_rank) * 2 + (1 if fold == 'val' else 0))
        train_loader = MovingMNISTLoader(mk('train', opt.seq_len), opt.batch_size)
        val_loader = None
        if local_rank == 0:
            val_loader = MovingMNISTLoader(mk('val', opt.seq_len_test or opt.seq_len), opt.batch_size_test,
                                           n_items=max(opt.batch_size_test * opt.n_iter_test, 1))
        return train_loader, val_loader, None
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
