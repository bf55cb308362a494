"""
Stochastic Moving-MNIST batches generated ON the device (second half).

The reference's training set (data/mmnist.py:106-124) builds every video on the CPU: per object a random digit, a random start and
speed, bounce physics with a fresh random speed at every wall contact (mmnist.py:126-237), then `seq_len` slice-adds of the 28x28
digit into float frames, clamp, uint8 -- and the DataLoader's collate turns B such videos into the float batch (data/base.py:71-84).
Here a batch is two launches and no host work: `srvp_mmnist_trajectories` (one thread per object, the bounce walk in registers,
counter-based Philox random numbers keyed by (seed, batch index, object)) and `srvp_mmnist_render` (digit stamping, clamp, /255,
time-major layout), writing the (T, B, 1, nx, nx) float batch in HBM.

Equality with the reference generator is DISTRIBUTIONAL (a sequential Mersenne-twister stream with a data-dependent number of draws
per object cannot be reproduced by parallel threads): start / speed / digit marginals, per-frame position moments and wall-contact
rates are tested against the reference restatement oracle/mmnist_ref.py, which itself reproduces the reference's videos bit for bit
under `np.random.seed` (tests/golden/mmnist.npz); the kernels are pinned exactly by a CPU restatement of the same Philox stream.
Batch k of a run depends on (seed, k) only, so a resumed or re-sharded run sees the same data.
"""
import numpy as np
import torch

from . import _lib as L


class MovingMNISTBatches:
    """Training batches of reference `MovingMNIST(digits, nx, seq_len, max_speed, deterministic, num_digits, train=True)`
    (same positional meaning), produced B videos at a time straight into device memory.  seed / counter: the Philox key and the
    index of the next batch (set `counter` to the iteration number to resume a run on the same data)."""

    def __init__(self, data, nx, seq_len, max_speed, deterministic, num_digits, device=None, seed=0):
        digits = np.ascontiguousarray(np.array(data), dtype=np.uint8)
        assert digits.ndim == 3, 'digits: (n, h, w) uint8 (all of one shape, as MNIST is)'
        self.nx, self.seq_len, self.max_speed = nx, seq_len, max_speed
        self.deterministic, self.num_digits = deterministic, num_digits
        self.n_digits, self.dh, self.dw = digits.shape
        self.device = torch.device(device if device is not None else 'cuda')
        self.digits = torch.from_numpy(digits).to(self.device)
        self.seed, self.counter = int(seed) & (2 ** 64 - 1), 0
        self._bufs = {}

    def change_seq_len(self, seq_len):
        self.seq_len = seq_len

    def trajectories(self, B, want_contacts=False):
        """-> idx int32 (B, num_digits), pos int32 (B, num_digits, T, 2) [, contacts int32 (B, num_digits)] on the device;
        advances the batch counter."""
        key = (B, self.seq_len)
        if key not in self._bufs:
            self._bufs[key] = (torch.empty(B, self.num_digits, dtype=torch.int32, device=self.device),
                               torch.empty(B, self.num_digits, self.seq_len, 2, dtype=torch.int32, device=self.device),
                               torch.empty(B, self.num_digits, dtype=torch.int32, device=self.device))
        idx, pos, con = self._bufs[key]
        L.call('srvp_mmnist_trajectories', self.seed, self.counter, B, self.num_digits, self.seq_len, self.nx, self.dh, self.dw,
               self.max_speed, 1 if self.deterministic else 0, self.n_digits, L.ptr(idx), L.ptr(pos), L.ptr(con) if want_contacts else None,
               L.stream())
        self.counter += 1
        return (idx, pos, con) if want_contacts else (idx, pos)

    def render(self, idx, pos, want_f32=True, want_u8=False):
        """idx / pos (device or host int32 arrays) -> (float batch (T, B, 1, nx, nx) or None, uint8 videos (B, T, nx, nx) or None)."""
        if not torch.is_tensor(idx):
            idx, pos = torch.from_numpy(np.ascontiguousarray(idx, np.int32)), torch.from_numpy(np.ascontiguousarray(pos, np.int32))
        idx_d, pos_d = idx.to(self.device).contiguous(), pos.to(self.device).contiguous()
        B, T = idx_d.shape[0], pos_d.shape[2]
        out = torch.empty(T, B, 1, self.nx, self.nx, device=self.device) if want_f32 else None
        u8 = torch.empty(B, T, self.nx, self.nx, dtype=torch.uint8, device=self.device) if want_u8 else None
        L.call('srvp_mmnist_render', L.ptr(self.digits), self.n_digits, self.dh, self.dw, L.ptr(idx_d), L.ptr(pos_d), B,
               T, self.num_digits, self.nx, L.ptr(out), L.ptr(u8), L.stream())
        return out, u8

    def batch(self, B):
        """(T, B, 1, nx, nx) float32 in [0, 1] on the device: the reference's collate_fn([dataset[i] for i in range(B)]) in law."""
        return self.render(*self.trajectories(B), True, False)[0]

    def videos_u8(self, B):
        """uint8 (B, T, nx, nx) on the device: the reference's per-video arrays, stacked, in law."""
        return self.render(*self.trajectories(B), False, True)[1]
