"""
TEST INFRASTRUCTURE (oracle/): CPU restatements for the Stochastic Moving-MNIST input pipeline.  Only tests/ may
import this; the product (srvp_amd/mmnist.py) generates trajectories on the device.

 1. `bounce` / `trajectory` / `draw`: the reference generator restated (data/mmnist.py:113-237, `_compute_trajectory`,
    `_process_collision`), consuming the GLOBAL `np.random` stream in the reference's order and with its float expressions, so that
    under the same `np.random.seed` it reproduces the reference's trajectories and videos bit for bit -- pinned by
    tests/golden/mmnist.npz (made from the real reference by tests/make_golden.py).  It is what the device generator is compared
    with in distribution.
 2. `philox_trajectories`: the DEVICE algorithm (csrc/util.hip: mmnist_traj_kernel) restated in numpy / python floats -- same
    Philox4x32-10 stream, same ray / box walk in float64 -- which pins the kernel bit-exactly.
"""
import numpy as np

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




def draw(n_digits, dh, dw, nx, seq_len, max_speed, deterministic, num_digits, B):
    """Per video, per object -- digit index then trajectory, consuming np.random as mmnist.py:116-120 does.
    -> idx int32 (B, num_digits), pos int32 (B, num_digits, T, 2)."""
    idx = np.empty((B, num_digits), np.int32)
    pos = np.empty((B, num_digits, seq_len, 2), np.int32)
    for b in range(B):
        for n in range(num_digits):
            idx[b, n] = np.random.randint(n_digits)
            tr = trajectory(dh, dw, nx, seq_len, max_speed, deterministic)
            pos[b, n] = [(r, c) for r, c, _, _ in tr]
    return idx, pos


# ---------------------------------------------------------------------------------------------------------------------
# the device generator, restated (csrc/util.hip)
# ---------------------------------------------------------------------------------------------------------------------
M32 = 0xFFFFFFFF


class Philox:
    """Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = (block, object, batch lo, batch hi); four draws per block."""

    def __init__(self, seed, obj, batch):
        self.k = (seed & M32, (seed >> 32) & M32)
        self.c = [0, obj & M32, batch & M32, (batch >> 32) & M32]
        self.buf = []

    def _block(self):
        a0, a1, a2, a3 = self.c
        x0, x1 = self.k
        for _ in range(10):
            p0, p1 = 0xD2511F53 * a0, 0xCD9E8D57 * a2
            a0, a1, a2, a3 = ((p1 >> 32) ^ a1 ^ x0) & M32, p1 & M32, ((p0 >> 32) ^ a3 ^ x1) & M32, p0 & M32
            x0, x1 = (x0 + 0x9E3779B9) & M32, (x1 + 0xBB67AE85) & M32
        self.c[0] = (self.c[0] + 1) & M32
        self.buf = [a0, a1, a2, a3]

    def next(self):
        if not self.buf:
            self._block()
        return self.buf.pop(0)

    def below(self, n):
        """uniform integer in [0, n): Lemire's multiply-shift with rejection."""
        m = self.next() * n
        lo = m & M32
        if lo < n:
            t = ((1 << 32) - n) % n
            while lo < t:
                m = self.next() * n
                lo = m & M32
        return m >> 32


def philox_trajectories(seed, batch, B, num_digits, T, nx, dh, dw, max_speed, deterministic, n_digits):
    """-> idx (B, nd), pos (B, nd, T, 2), contacts (B, nd): what srvp_mmnist_trajectories writes."""
    x_max, y_max = nx - dh, nx - dw
    idx = np.zeros((B, num_digits), np.int32)
    pos = np.zeros((B, num_digits, T, 2), np.int32)
    con = np.zeros((B, num_digits), np.int32)
    span = 2 * max_speed + 1
    for o in range(B * num_digits):
        g = Philox(seed, o, batch)
        b, n = divmod(o, num_digits)
        idx[b, n] = g.below(n_digits)
        sx, sy = float(g.below(x_max + 1)), float(g.below(y_max + 1))
        vx, vy = g.below(span) - max_speed, g.below(span) - max_speed
        nc = 0
        for t in range(T):
            pos[b, n, t] = (int(np.rint(sx)), int(np.rint(sy)))
            tau = 1.0
            for _ in range(64):
                if not tau > 0.0:
                    break
                tx = (x_max - sx) / vx if vx > 0 else ((0.0 - sx) / vx if vx < 0 else 1e300)
                ty = (y_max - sy) / vy if vy > 0 else ((0.0 - sy) / vy if vy < 0 else 1e300)
                th = tx if tx < ty else ty
                if th >= tau:
                    sx += vx * tau
                    sy += vy * tau
                    break
                hx, hy = tx <= th + 1e-12, ty <= th + 1e-12
                sx += vx * th
                sy += vy * th
                tau -= th
                wx = (1 if vx > 0 else -1) if hx else 0
                wy = (1 if vy > 0 else -1) if hy else 0
                if hx:
                    sx = float(x_max) if vx > 0 else 0.0
                if hy:
                    sy = float(y_max) if vy > 0 else 0.0
                if not deterministic:
                    vx, vy = g.below(span) - max_speed, g.below(span) - max_speed
                if wx:
                    vx = -abs(vx) if wx > 0 else abs(vx)
                if wy:
                    vy = -abs(vy) if wy > 0 else abs(vy)
                nc += 1
            sx = min(max(sx, 0.0), float(x_max))
            sy = min(max(sy, 0.0), float(y_max))
        con[b, n] = nc
    return idx, pos, con
