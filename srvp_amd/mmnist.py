"""
Stochastic Moving-MNIST batches generated for the device (second half).

The reference's training set (data/mmnist.py:106-124) builds every video on the CPU: per object a random digit, a random
start and speed, bounce physics with a fresh random speed at every wall contact (mmnist.py:126-237), then `seq_len` slice-adds of
the 28x28 digit into float frames, clamp, uint8 -- and the DataLoader's collate turns B such videos into the float batch
(data/base.py:71-84).  Here the split is: the trajectory code (a few dozen scalar operations per object, with a data-dependent
number of `np.random` draws, so inherently sequential) stays on the host and consumes `np.random` in exactly the reference's
order; the frame assembly for the whole batch is one `srvp_mmnist_render` launch that writes the (T, B, 1, nx, nx) float batch
(and/or the uint8 videos) directly in HBM.  With the same `np.random.seed`, `videos_u8` equals the reference's
`[dataset[i] for i in range(B)]` bit for bit (tests/golden/mmnist.npz).
"""
import numpy as np
import torch

from . import _lib as L

EPS = 1e-8          # mmnist.py:53


def _outside(sx, sy, x_max, y_max):
    """Which walls the position lies beyond (mmnist.py:177-180, 233-236): (left, upper, right, bottom)."""
    return sx < -EPS, sy < -EPS, sx > x_max + EPS, sy > y_max + EPS


def bounce(sx, sy, dx, dy, x_max, y_max, max_speed, deterministic, randint=None):
    """mmnist.py:171-237 (`_process_collision`) for the box [0, x_max] x [0, y_max]: while the object is outside, find the
    contact point with the wall it crossed, draw a new speed (stochastic variant), point it back inside and spend the
    rest of the time step with it.  Arithmetic and draw order as in the reference."""
    randint = randint or np.random.randint
    left, upper, right, bottom = _outside(sx, sy, x_max, y_max)
    cx = cy = None
    while left or right or upper or bottom:
        if dx == 0:                                     # vertical motion: contact on the upper / bottom wall
            cx, cy = sx, (0 if upper else y_max)
        elif dy == 0:                                   # horizontal motion
            cx, cy = (0 if left else x_max), sy
        else:
            a = dy / dx
            b = sy - a * sx
            # candidate walls in the reference's order; a wall stays flagged only if the line meets it inside the frame
            if left:
                yi = a * 0 + b
                left = (yi >= 0 - EPS) and (yi <= y_max + EPS)
                if left:
                    cx, cy = 0, yi
            if right:
                yi = a * x_max + b
                right = (yi >= 0 - EPS) and (yi <= y_max + EPS)
                if right:
                    cx, cy = x_max, yi
            if upper:
                xi = (0 - b) / a
                upper = (xi >= 0 - EPS) and (xi <= x_max + EPS)
                if upper:
                    cx, cy = xi, 0
            if bottom:
                xi = (y_max - b) / a
                bottom = (xi >= 0 - EPS) and (xi <= x_max + EPS)
                if bottom:
                    cx, cy = xi, y_max
        p = ((sx - cx) / dx) if dx != 0 else ((sy - cy) / dy)       # part of the step spent beyond the wall
        if not deterministic:
            dx = randint(-max_speed, max_speed + 1)
            dy = randint(-max_speed, max_speed + 1)
        if left:
            dx = abs(dx)
        if right:
            dx = -abs(dx)
        if upper:
            dy = abs(dy)
        if bottom:
            dy = -abs(dy)
        sx = cx + dx * p
        sy = cy + dy * p
        left, upper, right, bottom = _outside(sx, sy, x_max, y_max)
    return sx, sy, dx, dy


def trajectory(dh, dw, nx, seq_len, max_speed, deterministic, init_cond=None, randint=None):
    """mmnist.py:126-169 (`_compute_trajectory`): [(row, col, dx, dy)] * seq_len for an object of dh x dw pixels."""
    randint = randint or np.random.randint
    x_max, y_max = nx - dh, nx - dw
    if init_cond is None:
        sx = randint(0, x_max + 1)
        sy = randint(0, y_max + 1)
        dx = randint(-max_speed, max_speed + 1)
        dy = randint(-max_speed, max_speed + 1)
    else:
        sx, sy, dx, dy = init_cond
    out = []
    for _ in range(seq_len):
        sx, sy, dx, dy = bounce(sx, sy, dx, dy, x_max, y_max, max_speed, deterministic, randint)
        out.append((int(round(sx)), int(round(sy)), dx, dy))
        sy += dy
        sx += dx
    return out


class MovingMNISTBatches:
    """Training batches of reference `MovingMNIST(digits, nx, seq_len, max_speed, deterministic, num_digits, train=True)`
    (same positional meaning), produced B videos at a time straight into device memory."""

    def __init__(self, data, nx, seq_len, max_speed, deterministic, num_digits, device=None):
        digits = np.ascontiguousarray(np.array(data), dtype=np.uint8)
        assert digits.ndim == 3, 'digits: (n, h, w) uint8 (all of one shape, as MNIST is)'
        self.nx, self.seq_len, self.max_speed = nx, seq_len, max_speed
        self.deterministic, self.num_digits = deterministic, num_digits
        self.n_digits, self.dh, self.dw = digits.shape
        self.device = torch.device(device if device is not None else 'cuda')
        self.digits = torch.from_numpy(digits).to(self.device)

    def change_seq_len(self, seq_len):
        self.seq_len = seq_len

    def draw(self, B):
        """Host part: per video, per object -- digit index then trajectory, consuming np.random as mmnist.py:116-120 does.
        -> idx int32 (B, num_digits), pos int32 (B, num_digits, T, 2)."""
        idx = np.empty((B, self.num_digits), np.int32)
        pos = np.empty((B, self.num_digits, self.seq_len, 2), np.int32)
        for b in range(B):
            for n in range(self.num_digits):
                idx[b, n] = np.random.randint(self.n_digits)
                tr = trajectory(self.dh, self.dw, self.nx, self.seq_len, self.max_speed, self.deterministic)
                pos[b, n] = [(r, c) for r, c, _, _ in tr]
        return idx, pos

    def _render(self, idx, pos, want_f32, want_u8):
        B = idx.shape[0]
        idx_d = torch.from_numpy(idx).to(self.device)
        pos_d = torch.from_numpy(pos).to(self.device)
        out = torch.empty(self.seq_len, B, 1, self.nx, self.nx, device=self.device) if want_f32 else None
        u8 = torch.empty(B, self.seq_len, self.nx, self.nx, dtype=torch.uint8, device=self.device) if want_u8 else None
        L.call('srvp_mmnist_render', L.ptr(self.digits), self.n_digits, self.dh, self.dw, L.ptr(idx_d), L.ptr(pos_d), B,
               self.seq_len, self.num_digits, self.nx, L.ptr(out), L.ptr(u8), L.stream())
        return out, u8

    def batch(self, B):
        """(T, B, 1, nx, nx) float32 in [0, 1] on the device = collate_fn([dataset[i] for i in range(B)])."""
        return self._render(*self.draw(B), True, False)[0]

    def videos_u8(self, B):
        """uint8 (B, T, nx, nx) on the device = the reference's per-video arrays, stacked."""
        return self._render(*self.draw(B), False, True)[1]
