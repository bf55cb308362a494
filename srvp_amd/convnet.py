"""
HIP launch plans for the convolutional encoder / decoder (reference module/conv.py:129-154, 249-275 and their
autograd backward), built from the layer tables in arch.py.

Data layout: every activation is an NHWC bfloat16 tensor with channels padded to a multiple of 32 and a 1-pixel
zero border physically present (class Feat), so the MFMA convolutions gather without bounds checks; MaxPool /
nearest-Upsample / skip-concat / the per-sample skip gather never materialise a tensor -- they are index arithmetic
inside the consuming kernel.  BatchNorm is split around its grid-wide reduction: statistics in the conv epilogue,
`bn_finalize`, then `bn_act` writes the activated tensor.  All compute is in libsrvp_hip.so (see _lib.py); torch is
used for allocation and streams only.
"""
import ctypes as C
import os

import torch

from . import _lib as L

CP = 32                      # channel padding granule
SUBPIX = os.environ.get('SRVP_SUBPIX', '1') != '0'
S2D = os.environ.get('SRVP_SUBPIX_S2D', '1') != '0'
WGRAD_HALO2 = os.environ.get('SRVP_WGRAD_HALO2', '1') != '0'  # sub-pixel weight gradients: one 16-tap launch, two phases per workgroup on the halo kernel
DETERMINISTIC = False       # set by model.set_deterministic (fp32 parity mode only): fixed-order BatchNorm statistics / weight-gradient sums
EVAL_FOLD = os.environ.get('SRVP_EVAL_FOLD', '1') != '0'      # inference: eval-mode BatchNorm + activation in the conv epilogue (no raw tensor, no bn_act pass)
PACK_TILES = os.environ.get('SRVP_PACK_TILES', '1') != '0'      # 0: every pack / unpack job through the multi kernels (A/B)
BN_FUSED_FINALIZE = os.environ.get('SRVP_BN_FUSED_FINALIZE', '1') != '0'    # bn_finalize / bn_bwd_finalize folded into bn_act / bn_bwd_apply
# encoder weight gradients on the second stream when the batch is small (<= this many frames): grids of 100-600 workgroups do not
# fill the chip, so the weight gradient of block i runs beside the BatchNorm backward / data gradient of block i - 1
# (same-box A/B, ms per step off / on: 9.40 / 9.02 at 24 sequences per GPU, 14.35 / 13.69 at 48, 22.75 / 22.89 at 96, 42.00 / 41.57 at 192)
ENC_WGRAD_SIDE_MAXN = int(os.environ.get('SRVP_ENC_WGRAD_SIDE_MAXN', '1000000'))
# BatchNorm-backward reduction of a producer layer inside the data-gradient launch of its (plain 3x3) consumer (srvp_conv_desc.bnr_*)
BN_FUSED_REDUCE = os.environ.get('SRVP_BN_FUSED_REDUCE', '1') != '0'
# decoder weight gradients issued on the second stream block by block (each right behind its BatchNorm backward) instead of all after
# the decoder's data-gradient chain: the MFMA-bound weight gradients then run beside the HBM-bound BatchNorm passes of the blocks below
# (measured, same box: 41.70 vs 41.21 ms per step at 192 sequences with / without, 8.88 vs 8.86 at 24 in round 3: off.  Round 5, after the
# persistent latent kernels had become 0.5 ms shorter -- the weight gradients no longer fit under the latent backward and the step ended on
# the second queue: 37.54 vs 37.87 ms at 192 sequences, 7.29 vs 7.42 at 24: on)
DEC_WGRAD_EARLY = os.environ.get('SRVP_DEC_WGRAD_EARLY', '1') != '0'
# with DEC_WGRAD_EARLY: only the first n blocks of the decoder's backward walk (the 64x64 / 32x32 stages) go out early, the rest behind the
# data-gradient chain as before (-1: all, -2: half of the blocks).  Same-box sweep n = 0 / 2 / 4 / 6 / 8 / all, ms per step: 24 sequences 7.63 /
# 7.54 / 7.44 / 7.36 / 7.43 / 7.46, KTH 33.35 / 33.27 / 33.41 / 33.27 / 33.40 / 33.79, Human3.6M 27.97 / 27.84 / 27.75 / 27.62 / 27.75 / 27.89,
# SM-MNIST (5 blocks) 5.95 / 5.95 / 6.08 / ...: half of the blocks
DEC_WGRAD_EARLY_N = int(os.environ.get('SRVP_DEC_WGRAD_EARLY_N', '-2'))
UNPACK_SPLIT = int(os.environ.get('SRVP_UNPACK_SPLIT', '4'))       # encoder: blocks >= this index are unpacked before the weight gradients of the first blocks (0: one unpack at the end)
IN_WGRAD_BN = os.environ.get('SRVP_IN_WGRAD_BN', '1') != '0'                # 0: the first block's output gradient is written and read back by its weight gradient
POOL_FUSED_REDUCE = os.environ.get('SRVP_POOL_FUSED_REDUCE', '1') != '0'    # 0: pooled layers keep their own BatchNorm-backward reduction pass
S_QUAD = os.environ.get('SRVP_S_QUAD', '1') != '0'        # hoisted skip half stored pixel-quad-major (16-byte loads in the consumers)
SPLITK = int(os.environ.get('SRVP_CONV_SPLITK', '16'))    # tiny-M long-K launches: K steps shared over this many workgroups
# Sub-pixel form of "nearest x2 upsample, then 3x3 conv" (conv.py:331-349): output phase a in {0,1} of a row pair reads
# two low-resolution rows with SUMS of the original taps -- R[a][u] = kernel rows folded into effective tap u -- and
# the gradient wrt the low-resolution input is a 4x4 stride-2 conv of the output gradient with D[dy] folded rows.
# 4/9 of the MACs of convolving the materialised (or index-mapped) upsampled tensor, same result up to the rounding
# of the folded weights (summed in fp32 from the fp32 masters, rounded to bf16 once).
SUB_R = {0: [(0,), (1, 2)], 1: [(0, 1), (2,)]}
SUB_D = [(2,), (1, 2), (0, 1), (0,)]
# Transposed 4x4 stride-2 convolution (DCGAN decoder, conv.py:299-304), BACKWARD over the space-to-depth output gradient: input row ih
# collects output rows 2 ih - 1 + kh; output row parity a -> [(kernel index kh, padded row offset dy into the s2d tensor)]
S2D_UP = {0: [(1, 1), (3, 2)], 1: [(0, 0), (2, 1)]}
S2D_UP_ON = os.environ.get('SRVP_UP_S2D', '1') != '0'
WGRAD_S2D_SHIFT_SRC = os.environ.get('SRVP_WGRAD_S2D_SHIFT_SRC', '1') != '0'   # 16-tap weight gradients of the 4x4 stride-2 blocks in the sub-pixel blocks' form (halo kernel)
ENC_WGRAD_AFTER_DGRAD = os.environ.get('SRVP_ENC_WGRAD_AFTER_DGRAD', '0') == '1'     # A/B: encoder weight gradients released after the block's data gradient
PHASES_ONE_GRID = os.environ.get('SRVP_PHASES_ONE_GRID', '1') != '0'    # the four phase launches of a 4x4 stride-2 block (forward of 'up', data gradient of 'down') as one grid
DOWN_S2D = os.environ.get('SRVP_DOWN_S2D', '1') != '0'        # 4x4 stride-2 encoder layers on a space-to-depth source (no skip connections)


def _tapset(rows, cols, k=3):
    m = 0
    for kh in rows:
        for kw in cols:
            m |= 1 << (kh * k + kw)
    return m
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
ACT = {'none': L.ACT_NONE, 'leaky_relu': L.ACT_LRELU, 'tanh': L.ACT_TANH, 'relu': L.ACT_RELU, 'sigmoid': L.ACT_SIGMOID}
# transposed conv 4x4 s2 p1 (and data-gradient of conv 4x4 s2 p1): output parity -> [(kernel index, padded input offset)]
PHASE = {0: [(1, 1), (3, 0)], 1: [(0, 2), (2, 1)]}


def cpad(c):
    return (c + CP - 1) // CP * CP


class Feat:
    """NHWC bf16 tensor [N][H+2b][W+2b][C] with zero border b; Cr = number of real channels."""

    def __init__(self, N, H, W, Cr, device, b=1, C=None, dtype=torch.bfloat16, s2d=False):
        """dtype: torch.bfloat16 (production) or torch.float32 (precision = 'fp32' parity mode).
        s2d: stored SPACE-TO-DEPTH for a 4x4 stride-2 consumer, [N][H/2+2][W/2+2][4 C] with a 1-pixel zero border: pixel (y, x) at position
        (y/2, x/2), channel group (y&1)*2 + (x&1) (written that way by srvp_bn_finalize_act / srvp_bn_act_s2d)."""
        self.N, self.H, self.W, self.Cr, self.b = N, H, W, Cr, b
        self.C = cpad(Cr) if C is None else C
        self.s2d = bool(s2d)
        if self.s2d:
            assert b == 1 and H % 2 == 0 and W % 2 == 0
            self.t = torch.zeros(N, H // 2 + 2, W // 2 + 2, 4 * self.C, dtype=dtype, device=device)
        else:
            self.t = torch.zeros(N, H + 2 * b, W + 2 * b, self.C, dtype=dtype, device=device)

    @property
    def Hp(self):
        return self.H + 2 * self.b

    @property
    def Wp(self):
        return self.W + 2 * self.b

    def interior(self):
        assert not self.s2d, 'space-to-depth tensor: use get_nhwc() / put_nhwc()'
        b = self.b
        return self.t[:, b:b + self.H, b:b + self.W, :self.Cr]

    def _s2d_view(self):
        # [N][H/2][2 (y&1)][W/2][2 (x&1)][C] view of the interior of the space-to-depth tensor
        return self.t[:, 1:-1, 1:-1].reshape(self.N, self.H // 2, self.W // 2, 2, 2, self.C).permute(0, 1, 3, 2, 4, 5)

    def get_nhwc(self):
        """[N][H][W][Cr] copy of the interior, whatever the layout."""
        if self.s2d:
            return self._s2d_view().reshape(self.N, self.H, self.W, self.C)[..., :self.Cr].clone()
        return self.interior().clone()

    def put_nhwc(self, x):
        """x: [N][H][W][Cr] -> interior (border and channel padding stay zero)."""
        if self.s2d:
            v = torch.zeros(self.N, self.H, self.W, self.C, dtype=self.t.dtype, device=self.t.device)
            v[..., :self.Cr] = x
            inner = v.reshape(self.N, self.H // 2, 2, self.W // 2, 2, self.C).permute(0, 1, 3, 2, 4, 5).reshape(self.N, self.H // 2, self.W // 2, 4 * self.C)
            self.t[:, 1:-1, 1:-1].copy_(inner)
        else:
            self.interior().copy_(x)

    def to_nchw(self):
        return self.get_nhwc().permute(0, 3, 1, 2).float().contiguous()

    def load_nchw(self, x):
        self.put_nhwc(x.permute(0, 2, 3, 1))


def _pack_desc(taps_off, J, K, Jsegs, Ksegs, sj, sk, tap_sets=None):
    d = L.PackDesc()
    d.ntaps = len(taps_off)
    d.tap_off = L.taps(taps_off)
    if tap_sets is not None:
        d.tap_set = L.taps(tap_sets)
    d.J, d.K = J, K
    d.J0, d.J0r, d.J1r = Jsegs
    d.K0, d.K0r, d.K1r = Ksegs
    d.sj, d.sk = sj, sk
    return d


class Block:
    """One conv block (conv -> [BN] -> act) of the encoder or decoder: buffers + launch descriptors."""

    def __init__(self, spec, role, srcs, ups, N, device, training, skip_map=None, skip_sel=None, f32=False):
        """
        role: 'in' (fp32 frames in), 'mfma', 'out' (fp32 frames out)
        srcs: list of source Feat (1 or 2) -- for role 'in' empty;  ups: nearest-x2 applied to srcs[0] by the gather
        f32: precision = 'fp32' parity mode -- every activation / gradient / packed-weight tensor of the block is fp32 and the
             launches carry elem_f32 = 1 (exact-fp32 MFMA kernels, csrc/conv_f32.hip)
        """
        self.spec, self.role, self.srcs, self.ups, self.N, self.dev = spec, role, srcs, ups, N, device
        self.training = training
        self.f32 = bool(f32)
        self.adt = torch.float32 if f32 else torch.bfloat16
        self.skip_map = skip_map                      # int32 [N] image index into srcs[1]
        self.skip_sel = skip_sel                      # int32 [B] sample -> image index into srcs[1] (hoisted skip half)
        k, s, p = spec['k'], spec['s'], spec['p']
        self.k, self.s, self.p = k, s, p
        self.kind = spec['kind']
        self.cout_r = spec['cout']
        self.cout = cpad(self.cout_r)
        self.has_bn = spec['bnkey'] is not None
        self.act = ACT[spec['act']]
        if role == 'in':
            self.cin_r = [spec['cin']]
            self.Hin = self.Win = 64
        else:
            self.cin_r = [f.Cr for f in srcs]
            assert sum(self.cin_r) == spec['cin'], (self.cin_r, spec['cin'])
            f0 = srcs[0]
            self.Hin, self.Win = (f0.H * 2, f0.W * 2) if ups else (f0.H, f0.W)
        if self.kind == 'conv':
            self.OH = (self.Hin + 2 * p - k) // s + 1
            self.OW = (self.Win + 2 * p - k) // s + 1
        else:
            self.OH = (self.Hin - 1) * s - 2 * p + k
            self.OW = (self.Win - 1) * s - 2 * p + k
        # geometry class of the MFMA layers
        if role in ('mfma', 'out'):
            if self.kind == 'conv' and self.OH == 1 and p == 0:
                self.geom = 'full'                    # k x k conv covering the whole k x k input -> 1x1 (encoder last_conv)
            elif self.kind == 'conv' and s == 1:
                self.geom = 'same'                    # 3x3 s1 p1
            elif self.kind == 'conv' and s == 2:
                self.geom = 'down'                    # 4x4 s2 p1
            elif self.kind == 'convT' and s == 2:
                self.geom = 'up'                      # transposed 4x4 s2 p1
            elif self.kind == 'convT' and self.Hin == 1:
                self.geom = 'expand'                  # transposed k x k on a 1x1 input (decoder first_upconv)
            elif self.kind == 'convT' and s == 1:
                self.geom = 'sameT'                   # transposed 3x3 s1 p1 (VGG decoder output layer)
            else:
                raise NotImplementedError(spec)
            assert not (self.geom in ('full', 'expand') and len(srcs) != 1)
        # Hoisted skip half: the skip connection of a sample is the same for every time step (module/srvp.py:222-223), so
        # conv([h_t, skip]) = conv_h(h_t) + conv_s(skip) with conv_s -- and its weight / data gradients, through the sum
        # of the output gradient over time -- evaluated once per SAMPLE instead of once per FRAME (1/T of the FLOPs of
        # that half; 21 % of all VGG conv FLOPs at T = 12).  Same arithmetic, different summation order.
        self.split = (role == 'mfma' and len(srcs) == 2 and skip_sel is not None and getattr(self, 'geom', None) == 'same')
        self.B = int(skip_sel.numel()) if self.split else 0
        # 4x4 stride-2 convolution over a SPACE-TO-DEPTH source (DCGAN encoder, conv.py:174-179): forward = one halo launch with the four
        # taps of each 64-channel chunk's phase (K = 16 C exactly), weight gradient = the per-tap kernel with the roles of its two operands
        # swapped (the s2d tensor goes in as the channel-sliced "gradient" operand, srvp_wgrad_desc.dout_phase_taps)
        self.s2d_in = bool(role == 'mfma' and getattr(self, 'geom', None) == 'down' and len(srcs) == 1 and getattr(srcs[0], 's2d', False))
        assert not any(getattr(f, 's2d', False) for f in srcs) or self.s2d_in, 'a space-to-depth source needs a 4x4 stride-2 consumer'
        # sub-pixel evaluation of the upsampled 3x3 conv (main input only; a hoisted skip half stays a plain conv)
        self.subpix = bool(SUBPIX and role == 'mfma' and ups and getattr(self, 'geom', None) == 'same' and
                           (len(srcs) == 1 or self.split) and self.k == 3)
        self.ctot = sum(f.C for f in srcs) if srcs else 0
        self.dcat_c = srcs[0].C if self.split else self.ctot     # channels of the input-gradient tensor `dcat`
        # Sub-pixel BACKWARD on the halo kernels (space-to-depth): the output gradient is stored [N][H/2+2][W/2+2][4 cout] (phase
        # = channel group, written that way by srvp_bn_bwd_apply), the data gradient is ONE halo convolution over it with the four
        # taps of each chunk's phase (K = 16 cout exactly) and the weight gradient four 4-tap halo launches on the channel slices --
        # instead of a 16-tap stride-2 gather and 16 per-tap weight gradients on the generic kernels at half the MFMA rate.
        # Needs what the halo kernels need: 64-multiple channel counts, power-of-two low-resolution grid (>= 8 columns where the
        # weight gradient runs on the halo kernel too, i.e. cout = 64).
        self.s2d = bool(S2D and self.subpix and training and not f32 and self.cout % 64 == 0 and srcs[0].C % 64 == 0
                        and srcs[0].W >= (8 if self.cout == 64 else 4) and srcs[0].H >= 4
                        and (srcs[0].W & (srcs[0].W - 1)) == 0 and (srcs[0].H & (srcs[0].H - 1)) == 0
                        and 4 * N * (srcs[0].H + 2) * (srcs[0].W + 2) * self.cout < 2 ** 32)
        # The same machinery for the transposed 4x4 stride-2 blocks of the DCGAN decoder (conv.py:299-304): with the output gradient
        # stored space-to-depth, the data gradient (a 16-tap stride-2 gather on the generic kernel) is ONE halo convolution with the four
        # taps of each chunk's phase, and the 16 weight-gradient taps read their phase slice at unit stride.
        self.s2d_up = bool(S2D and S2D_UP_ON and role == 'mfma' and getattr(self, 'geom', None) == 'up' and training and not f32
                           and self.cout % 64 == 0 and self.Hin >= 4 and (self.Hin & (self.Hin - 1)) == 0 and self.Win == self.Hin
                           and (self.Hin * self.Win <= 256 or self.Hin % 16 == 0) and all(f.C % 32 == 0 for f in srcs)
                           and 4 * N * (self.Hin + 2) * (self.Win + 2) * self.cout < 2 ** 32)
        self.s2d = self.s2d or self.s2d_up
        if role != 'out':
            self.raw = torch.empty(N, self.OH, self.OW, self.cout, dtype=self.adt, device=device)
            C_ = self.cout
            self.coef = torch.zeros(4, C_, dtype=torch.float32, device=device)      # scale, shift, mean, invstd
            if not self.has_bn:
                self.coef[0, :self.cout_r] = 1.0
            if self.has_bn and training:
                self.stats = torch.zeros(2, C_, dtype=torch.float64, device=device)
        self.out = None           # Feat (activated), set by the net
        self.pool = None          # pooled Feat
        self.out_f32 = None       # fp32 [N][C] (encoder output)
        if role == 'out':
            # fp32 frames out: (N, C, 64, 64), the layout of reference module/srvp.py:226
            self.x_out = torch.empty(N, self.cout_r, self.OH, self.OW, dtype=torch.float32, device=device)
        if role in ('mfma', 'out'):
            self._alloc_weights()
        if training and role != 'in':
            # gradient wrt the (virtual, concatenated) input of this block: [N][Hin][Win][ctot] bf16, unpadded
            # (sub-pixel blocks: gradient wrt the LOW-resolution source, i.e. already summed over each 2x2 upsample cell)
            dh, dw_ = (srcs[0].H, srcs[0].W) if self.subpix else (self.Hin, self.Win)
            self.dcat = torch.empty(N, dh, dw_, self.dcat_c, dtype=self.adt, device=device)
        if training:
            self.draw_b = 0 if (role == 'mfma' and self.geom in ('full', 'expand')) else 1
            bd = self.draw_b
            if self.s2d:
                self.draw = torch.zeros(N, self.OH // 2 + 2, self.OW // 2 + 2, 4 * self.cout, dtype=self.adt, device=device)
            else:
                self.draw = torch.zeros(N, self.OH + 2 * bd, self.OW + 2 * bd, self.cout, dtype=self.adt, device=device)
        if training and role != 'out':
            self.red = torch.zeros(2, self.cout, dtype=torch.float64, device=device)
            self.bcoef = torch.zeros(3, self.cout, dtype=torch.float32, device=device)
        if training and role in ('mfma', 'out'):
            ntaps = 16 if self.subpix else self.k * self.k
            self.dw = torch.zeros(ntaps, self.cout, self.dcat_c, dtype=torch.float32, device=device)
        if self.split:
            f1 = srcs[1]
            self.S = torch.empty(self.B, self.OH, self.OW, self.cout, dtype=torch.float32, device=device)
            if training:
                self.draw_sum = torch.zeros(self.B, self.OH + 2, self.OW + 2, self.cout, dtype=self.adt, device=device)
                self.dsel = torch.empty(self.B, self.Hin, self.Win, f1.C, dtype=self.adt, device=device)
                self.dw_s = torch.zeros(self.k * self.k, self.cout, f1.C, dtype=torch.float32, device=device)

    # ------------------------------------------------------------------ weights
    def _alloc_weights(self):
        self._alloc_weights_impl()
        for name in ('pf', 'pd', 'pu', 'pf_s', 'pd_s'):          # fp32 parity mode: every packed weight tensor is fp32
            pd = getattr(self, name, None)
            if pd is not None:
                pd.dst_f32 = 1 if self.f32 else 0
        if self.s2d_in:
            # forward weights over the space-to-depth source: [4 taps (u, v)][cout][K = 4 phases x C], fragment-major; phase (a, b) holds the
            # kernel taps (kh, kw) = (S2D_UP[a][u], S2D_UP[b][v]) (input row 2 oy - 1 + kh: the same parity table as the transposed blocks)
            c0p = self.srcs[0].C
            self.pf_ph = []
            for ph, (a, b) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
                offs = [kh * self.k + kw for kh, _ in S2D_UP[a] for kw, _ in S2D_UP[b]]
                d = _pack_desc(offs, self.pf.J, self.pf.K, (self.pf.J0, self.pf.J0r, self.pf.J1r), (self.pf.K0, self.pf.K0r, self.pf.K1r),
                               self.pf.sj, self.pf.sk)
                d.layout, d.kc_total, d.kc_off, d.dst_f32 = 1, 4 * c0p // 64, ph * c0p // 64, 0
                self.pf_ph.append(d)
            if self.training:
                # weight gradient comes out TRANSPOSED, [16 taps grouped by phase][Cin][Cout] (operand roles swapped in _wgrad_s2d_in)
                order = [kh * self.k + kw for a, b in [(0, 0), (0, 1), (1, 0), (1, 1)] for kh, _ in S2D_UP[a] for kw, _ in S2D_UP[b]]
                pu = self.pu
                self.pu = _pack_desc(order, pu.K, pu.J, (pu.K0, pu.K0r, pu.K1r), (pu.J0, pu.J0r, pu.J1r), pu.sk, pu.sj)
        if self.s2d_up and self.wt_d is not None:
            # data-gradient weights over the space-to-depth output gradient: [4 taps (u, v)][ctot][K = 4 phases x cout], fragment-major,
            # phase (a, b) = K chunks [ph cout/64, (ph + 1) cout/64) holding kernel taps (kh, kw) = (S2D_UP[a][u], S2D_UP[b][v])
            co_p = self.cout
            self.pd_ph = []
            for ph, (a, b) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
                offs = [kh * self.k + kw for kh, _ in S2D_UP[a] for kw, _ in S2D_UP[b]]
                d = _pack_desc(offs, self.pd.J, self.pd.K, (self.pd.J0, self.pd.J0r, self.pd.J1r), (self.pd.K0, self.pd.K0r, self.pd.K1r),
                               self.pd.sj, self.pd.sk)
                d.layout, d.kc_total, d.kc_off, d.dst_f32 = 1, 4 * co_p // 64, ph * co_p // 64, 0
                self.pd_ph.append(d)
            # weight-gradient taps grouped by phase (tap t' = ph * 4 + u * 2 + v reads the phase-ph slice of the gradient)
            order = [kh * self.k + kw for a, b in [(0, 0), (0, 1), (1, 0), (1, 1)] for kh, _ in S2D_UP[a] for kw, _ in S2D_UP[b]]
            pu = self.pu
            self.pu = _pack_desc(order, pu.J, pu.K, (pu.J0, pu.J0r, pu.J1r), (pu.K0, pu.K0r, pu.K1r), pu.sj, pu.sk)
        elif self.s2d and self.wt_d is not None:
            # data-gradient weights of the space-to-depth form: packed [4 taps (u, v)][cin][K = 4 phases x cout], fragment-major;
            # phase (a, b) occupies the K chunks [ph cout/64, (ph + 1) cout/64) and holds the folded taps R[a][u] x R[b][v]
            # transposed -- one pack job per phase (srvp_pack_desc.kc_off)
            c0p, c0r, co_p, co_r = self.srcs[0].C, self.cin_r[0], self.cout, self.cout_r
            self.pd_ph = []
            for ph, (a, b) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
                sets = [_tapset(SUB_R[a][u], SUB_R[b][v]) for u in (0, 1) for v in (0, 1)]
                d = _pack_desc([0] * 4, c0p, co_p, (c0p, c0r, 0), (co_p, co_r, 0), self.pd.sj, self.pd.sk, sets)
                d.layout, d.kc_total, d.kc_off = 1, 4 * co_p // 64, ph * co_p // 64
                self.pd_ph.append(d)

    def _alloc_weights_impl(self):
        k, kk = self.k, self.k * self.k
        dev = self.dev
        co_p, co_r = self.cout, self.cout_r
        ci_r = sum(self.cin_r)
        c0p = self.srcs[0].C
        c0r = self.cin_r[0]
        c1r = self.cin_r[1] if len(self.cin_r) > 1 else 0
        isegs = (c0p, c0r, c1r)
        osegs = (co_p, co_r, 0)
        nat = list(range(kk))
        if self.kind == 'conv':
            sj_f, sk_f = ci_r * kk, kk               # OIHW: j = co, k = ci
        else:
            sj_f, sk_f = kk, co_r * kk               # IOHW: j = co, k = ci
        if self.geom == 'up':
            order_f = [kh * k + kw for py in (0, 1) for px in (0, 1) for kh, _ in PHASE[py] for kw, _ in PHASE[px]]
        else:
            order_f = nat
        if self.geom == 'down':
            order_d = [kh * k + kw for py in (0, 1) for px in (0, 1) for kh, _ in PHASE[py] for kw, _ in PHASE[px]]
        else:
            order_d = nat
        if self.subpix:
            # forward / wgrad entry t = (a*2+b)*4 + u*2+v; dgrad entry t = dy*4 + dx
            fsets = [_tapset(SUB_R[a][u], SUB_R[b][v]) for a in (0, 1) for b in (0, 1) for u in (0, 1) for v in (0, 1)]
            dsets = [_tapset(SUB_D[dy], SUB_D[dx]) for dy in range(4) for dx in range(4)]
            z16 = [0] * 16
            h = (c0p, c0r, 0)
        if self.split:
            c1p = self.srcs[1].C
            h, sg = (c0p, c0r, 0), (c1p, c1r, 0)
            if self.subpix:
                self.pf = _pack_desc(z16, co_p, c0p, osegs, h, sj_f, sk_f, fsets)
                self.pd = _pack_desc(z16, c0p, co_p, h, osegs, sk_f, sj_f, dsets)
            else:
                self.pf = _pack_desc(nat, co_p, c0p, osegs, h, sj_f, sk_f)
                self.pd = _pack_desc(nat, c0p, co_p, h, osegs, sk_f, sj_f)
            self.pu = self.pf
            self.pf_s = _pack_desc(nat, co_p, c1p, osegs, sg, sj_f, sk_f)
            self.pd_s = _pack_desc(nat, c1p, co_p, sg, osegs, sk_f, sj_f)
            self.s_off = c0r * sk_f                  # element offset of the skip half inside the fp32 weight
            kh_ = 16 if self.subpix else kk
            self.wt_f = torch.empty(kh_, co_p, c0p, dtype=self.adt, device=dev)
            self.wt_f_s = torch.empty(kk, co_p, c1p, dtype=self.adt, device=dev)
            self.wt_d = torch.empty(kh_, c0p, co_p, dtype=self.adt, device=dev) if self.training else None
            self.wt_d_s = torch.empty(kk, c1p, co_p, dtype=self.adt, device=dev) if self.training else None
            return
        if self.subpix:
            self.pf = _pack_desc(z16, co_p, c0p, osegs, h, sj_f, sk_f, fsets)
            self.pd = _pack_desc(z16, c0p, co_p, h, osegs, sk_f, sj_f, dsets)
            self.pu = self.pf
            self.wt_f = torch.empty(16, co_p, c0p, dtype=self.adt, device=dev)
            self.wt_d = torch.empty(16, c0p, co_p, dtype=self.adt, device=dev) if self.training else None
            return
        self.pf = _pack_desc(order_f, co_p, self.ctot, osegs, isegs, sj_f, sk_f)
        self.pd = _pack_desc(order_d, self.ctot, co_p, isegs, osegs, sk_f, sj_f)
        self.pu = _pack_desc(nat, co_p, self.ctot, osegs, isegs, sj_f, sk_f)
        # image-side output layer on the streaming kernel (csrc/conv_out.hip): a tap-major copy [9][32][64] of its forward weights
        # (the kernel hardcodes its source: [N][66][66][64] with a 1-pixel border, read at full resolution -- ADVICE r4)
        self.stream_out = bool(self.role == 'out' and self.geom == 'sameT' and not self.f32 and len(self.srcs) == 1 and co_p == 32 and
                               self.srcs[0].b == 1 and self.srcs[0].C == 64 and not self.ups and not getattr(self.srcs[0], 's2d', False) and
                               L.load().srvp_conv_out_eligible(self.ctot, self.OH, self.OW, co_r, self.k, self.s, self.p))
        # image-side output layer of the DCGAN decoder (transposed 4x4 stride 2, 32x32 -> 64x64) on its streaming kernel (csrc/conv_out.hip,
        # round 6): reads the bordered [N][34][34][64] activation and the fp32 master weight directly
        self.stream_up_out = bool(self.role == 'out' and self.geom == 'up' and not self.f32 and len(self.srcs) == 1 and self.srcs[0].b == 1
                                  and self.srcs[0].C == 64 and not self.ups and not getattr(self.srcs[0], 's2d', False)
                                  and L.load().srvp_conv_up_out_eligible(self.ctot, self.Hin, self.Win, co_r, self.k, self.s, self.p))
        if self.stream_out:
            self.pf_o = _pack_desc(order_f, co_p, self.ctot, osegs, isegs, sj_f, sk_f)
            self.pf_o.layout = 0
            self.wt_o = torch.empty(kk, co_p, self.ctot, dtype=self.adt, device=dev)
        self.wt_f = torch.empty(kk, co_p, self.ctot, dtype=self.adt, device=dev)
        self.wt_d = torch.empty(kk, self.ctot, co_p, dtype=self.adt, device=dev) if self.training else None

    def pack_jobs(self, w):
        """[(fp32 source pointer, packed destination tensor, pack descriptor)] of this block's weight buffers."""
        jobs = [(L.ptr(w), self.wt_f, d) for d in self.pf_ph] if self.s2d_in else [(L.ptr(w), self.wt_f, self.pf)]
        if getattr(self, 'stream_out', False):
            jobs.append((L.ptr(w), self.wt_o, self.pf_o))
        if self.wt_d is not None and self.s2d:
            jobs += [(L.ptr(w), self.wt_d, d) for d in self.pd_ph]
        elif self.wt_d is not None:
            jobs.append((L.ptr(w), self.wt_d, self.pd))
        if self.split:
            ws = L.ptr(w) + 4 * self.s_off
            jobs.append((ws, self.wt_f_s, self.pf_s))
            if self.wt_d_s is not None:
                jobs.append((ws, self.wt_d_s, self.pd_s))
        return jobs

    def unpack_jobs(self, gw):
        """[(packed fp32 gradient tensor, fp32 gradient pointer, descriptor)]: inverse mapping for the weight gradient."""
        if self.split:
            return [(self.dw, L.ptr(gw), self.pf), (self.dw_s, L.ptr(gw) + 4 * self.s_off, self.pf_s)]
        return [(self.dw, L.ptr(gw), self.pu)]

    def pack(self, w, st):
        for d in (self.pf_ph if self.s2d_in else [self.pf]):
            L.call('srvp_pack_weight', L.ptr(w), L.ptr(self.wt_f), C.byref(d), st)
        if getattr(self, 'stream_out', False):
            L.call('srvp_pack_weight', L.ptr(w), L.ptr(self.wt_o), C.byref(self.pf_o), st)
        if self.wt_d is not None and self.s2d:
            for d in self.pd_ph:
                L.call('srvp_pack_weight', L.ptr(w), L.ptr(self.wt_d), C.byref(d), st)
        elif self.wt_d is not None:
            L.call('srvp_pack_weight', L.ptr(w), L.ptr(self.wt_d), C.byref(self.pd), st)
        if self.split:
            ws = L.ptr(w) + 4 * self.s_off
            L.call('srvp_pack_weight', ws, L.ptr(self.wt_f_s), C.byref(self.pf_s), st)
            if self.wt_d_s is not None:
                L.call('srvp_pack_weight', ws, L.ptr(self.wt_d_s), C.byref(self.pd_s), st)

    # ------------------------------------------------------------------ descriptors
    def _src_fields(self, d, which=None):
        """which: None = all sources; 'h' / 's' = main / skip half only (hoisted skip)."""
        if which == 's':
            f1 = self.srcs[1]
            d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(f1.t), f1.C, f1.Hp, f1.Wp, 0
            d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
            d.map0 = L.ptr(self.skip_sel)
            return
        f0 = self.srcs[0]
        d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(f0.t), f0.C, f0.Hp, f0.Wp, 1 if self.ups else 0
        if len(self.srcs) > 1 and which != 'h':
            f1 = self.srcs[1]
            d.src1, d.C1, d.H1p, d.W1p, d.ups1 = L.ptr(f1.t), f1.C, f1.Hp, f1.Wp, 0
            d.map1 = L.ptr(self.skip_map)
        else:
            d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None

    def _set_layout(self, descs, pack_desc):
        """Launches that run on the halo-tiled kernel read their weights MFMA-fragment-major: ask the library, mark the
        descriptors and the pack descriptor of that weight buffer (all launches sharing a buffer agree by construction)."""
        if not descs:
            return
        for d in descs:
            d.elem_f32 = 1 if self.f32 else 0
        want = [int(L.load().srvp_conv_wants_fragmajor(C.byref(d))) for d in descs]
        assert len(set(want)) == 1, want
        if sum(d.ntaps for d in descs) != pack_desc.ntaps:
            # the launch re-reads the packed tensor under another shape (the 4x4 -> 1x1 / 1x1 -> 4x4 layers run as plain
            # GEMMs over [taps * channels]): only the tap-major layout survives that reinterpretation
            want = [0]
        for d in descs:
            d.wt_fragmajor = want[0]
        pack_desc.layout = want[0]

    def _split_k(self, d, tag, dst_ptr, stats_ptr):
        """Tiny-M, long-K launch (4x4 -> 1x1 forward, 1x1 -> 4x4 data gradient: K = 16 * C as 128 dependent K steps on a handful
        of workgroups): share the K steps out over srvp_conv_desc.splitk workgroups per tile, each into its own fp32 slab, and let
        srvp_splitk_finish sum the slabs in a fixed order into `dst_ptr` (+ the BatchNorm statistics).  Returns the finish call's
        arguments, or None when the launch keeps its single pass."""
        ctot = d.C0 + d.C1
        if self.f32 or SPLITK <= 1 or ctot % 64 or d.Cout % 4:
            return None
        steps = d.ntaps * (ctot // 64)
        tiles = -(-(d.N * d.OH * d.OW) // 128) * -(-d.Cout // 128)
        sk = min(SPLITK, steps // 4, max(1, 512 // tiles))
        if sk <= 1:
            return None
        M = d.N * d.OH * d.OW
        parts = torch.empty(sk * M * d.Cout, dtype=torch.float32, device=self.dev)
        setattr(self, 'parts_' + tag, parts)
        d.dst, d.dst_is_f32, d.stats, d.stat_mod, d.splitk = L.ptr(parts), 1, None, 1, sk
        d.Cdst, d.cdst_off = d.Cout, 0
        return (L.ptr(parts), sk, M * d.Cout, M, d.Cout, dst_ptr, stats_ptr, self.cout)

    def finish_fwd(self, st):
        """After the forward launches of this block: sums the split-K slabs (if the forward was split) into raw + statistics."""
        if getattr(self, '_fwd_fin', None):
            L.call('srvp_splitk_finish', *self._fwd_fin, st)

    def finish_dgrad(self, st):
        if getattr(self, '_dg_fin', None):
            L.call('srvp_splitk_finish', *self._dg_fin, st)

    def _set_taps(self, d, taps):
        d.ntaps = len(taps)
        d.dy = L.taps([t[0] for t in taps])
        d.dx = L.taps([t[1] for t in taps])

    def _fwd_s2d_in(self):
        f0 = self.srcs[0]
        d = L.ConvDesc()
        d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(f0.t), 4 * f0.C, f0.H // 2 + 2, f0.W // 2 + 2, 0
        d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
        d.ntaps = 4
        ent = [(dy, dx) for a, b in [(0, 0), (0, 1), (1, 0), (1, 1)] for _, dy in S2D_UP[a] for _, dx in S2D_UP[b]]
        d.dy, d.dx = L.taps([e[0] for e in ent]), L.taps([e[1] for e in ent])
        d.tap_phase_chunks = f0.C // 64
        d.si, d.wt, d.Cout = 1, L.ptr(self.wt_f), self.cout
        d.N, d.OH, d.OW = self.N, self.OH, self.OW
        d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst, d.cdst_off = L.ptr(self.raw), self.OH, self.OW, 1, 0, 0, self.cout, 0
        use_stats = self.has_bn and self.training and not (DETERMINISTIC and self.f32)     # (deterministic mode: srvp_bn_stats_f32_det on the stored raw output)
        d.stats, d.stat_mod = (L.ptr(self.stats) if use_stats else None), self.cout
        d.out_f32, d.out_nc, d.out_sigmoid = None, 0, 0
        d.wt_fragmajor, d.elem_f32 = 1, 0
        assert int(L.load().srvp_conv_runs_on_halo(C.byref(d))) >= 128, 'space-to-depth forward is not eligible for the halo kernel'
        return [d]

    def fwd_descs(self):
        """List of ConvDesc for the forward convolution (4 for the transposed stride-2 phases, else 1)."""
        self.__dict__.pop('_fwd_arr', None)
        if self.s2d_in:
            self._fwd_fin = None
            return self._fwd_s2d_in()
        out = self._fwd_descs_raw()
        if self.split:
            self._set_layout(out[:1], self.pf_s)       # conv_s(skip)
            self._set_layout(out[1:], self.pf)
        else:
            self._set_layout(out, self.pf)
        return out

    def _fwd_descs_raw(self):
        k, N = self.k, self.N
        out = []
        use_stats = self.has_bn and self.training and not (DETERMINISTIC and self.f32)     # (deterministic mode: srvp_bn_stats_f32_det on the stored raw output)
        b_in = self.srcs[0].b
        frame_out = self.role == 'out'
        dst_ptr = None if frame_out else L.ptr(self.raw)

        def finish(d):
            if frame_out:
                d.out_f32, d.out_nc, d.out_sigmoid = L.ptr(self.x_out), self.cout_r, 1
            else:
                d.out_f32, d.out_nc, d.out_sigmoid = None, 0, 0
            out.append(d)
        if self.split:
            off = b_in - self.p
            taps = [(kh + off, kw + off) for kh in range(k) for kw in range(k)]
            ds = L.ConvDesc()                     # conv_s(skip): once per sample, fp32 out
            self._src_fields(ds, 's')
            self._set_taps(ds, taps)
            ds.si, ds.wt, ds.Cout = 1, L.ptr(self.wt_f_s), self.cout
            ds.N, ds.OH, ds.OW = self.B, self.OH, self.OW
            ds.dst, ds.DHp, ds.DWp, ds.so, ds.ooy, ds.oox, ds.Cdst, ds.cdst_off = L.ptr(self.S), self.OH, self.OW, 1, 0, 0, self.cout, 0
            ds.dst_is_f32, ds.stats, ds.stat_mod = 1, None, 1
            # S pixel-quad-major for its consumers (stride 2 for the sub-pixel phases): one 16-byte load per four accumulators
            so_c = 2 if self.subpix else 1
            # (measured per layer, same box: 0.336 -> 0.300, 0.410 -> 0.350, 0.593 -> 0.475 ms on the 512 / 256 / 128-channel stage
            # entries; the 64-channel one, on the 64-column kernel variant, got slower: 0.875 -> 0.938 ms, so it keeps the plain layout)
            quad = so_c if (S_QUAD and not self.f32 and (self.OW // so_c) % 4 == 0 and self.cout % 128 == 0) else 0
            ds.f32_quad = quad
            out.append(ds)
            if self.subpix:
                for d in self._subpix_fwd(dst_ptr, use_stats):
                    d.add_f32, d.add_mod, d.f32_quad = L.ptr(self.S), self.B, quad
                    finish(d)
                return out
            d = L.ConvDesc()                      # conv_h(h_t) + S[sample]
            self._src_fields(d, 'h')
            self._set_taps(d, taps)
            d.si, d.wt, d.Cout = 1, L.ptr(self.wt_f), self.cout
            d.N, d.OH, d.OW = N, self.OH, self.OW
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst, d.cdst_off = dst_ptr, self.OH, self.OW, 1, 0, 0, self.cout, 0
            d.stats, d.stat_mod = (L.ptr(self.stats) if use_stats else None), self.cout
            d.add_f32, d.add_mod, d.f32_quad = L.ptr(self.S), self.B, quad
            finish(d)
        elif self.subpix:
            for d in self._subpix_fwd(dst_ptr, use_stats):
                finish(d)
        elif self.geom in ('same', 'down', 'full', 'sameT'):
            d = L.ConvDesc()
            self._src_fields(d)
            if self.geom == 'sameT':
                # out[o] = sum_kh in[o + p - kh] W[ci][co][kh]  -> padded input coordinate o + p - kh + b
                taps = [(self.p - kh + b_in, self.p - kw + b_in) for kh in range(k) for kw in range(k)]
            else:
                off = b_in - self.p
                taps = [(kh + off, kw + off) for kh in range(k) for kw in range(k)]
            self._set_taps(d, taps)
            d.si, d.wt, d.Cout = self.s, L.ptr(self.wt_f), self.cout
            d.N, d.OH, d.OW = N, self.OH, self.OW
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst, d.cdst_off = dst_ptr, self.OH, self.OW, 1, 0, 0, self.cout, 0
            d.stats, d.stat_mod = (L.ptr(self.stats) if use_stats else None), self.cout
            self._fwd_fin = None
            if self.geom == 'full' and not frame_out:
                self._fwd_fin = self._split_k(d, 'f', dst_ptr, L.ptr(self.stats) if use_stats else None)
            finish(d)
        elif self.geom == 'up':
            assert b_in == 1
            for ph, (py, px) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
                d = L.ConvDesc()
                self._src_fields(d)
                self._set_taps(d, [(dy, dx) for _, dy in PHASE[py] for _, dx in PHASE[px]])
                d.si, d.Cout = 1, self.cout
                d.wt = L.ptr(self.wt_f) + ph * 4 * self.cout * self.ctot * self.wt_f.element_size()
                d.N, d.OH, d.OW = N, self.Hin, self.Win
                d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst, d.cdst_off = dst_ptr, self.OH, self.OW, 2, py, px, self.cout, 0
                d.stats, d.stat_mod = (L.ptr(self.stats) if use_stats else None), self.cout
                finish(d)
        elif self.geom == 'expand':
            d = L.ConvDesc()
            self._src_fields(d)
            self._set_taps(d, [(b_in, b_in)])
            d.si, d.wt, d.Cout = 1, L.ptr(self.wt_f), k * k * self.cout
            d.N, d.OH, d.OW = N, 1, 1
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst, d.cdst_off = L.ptr(self.raw), 1, 1, 1, 0, 0, k * k * self.cout, 0
            d.stats, d.stat_mod = (L.ptr(self.stats) if use_stats else None), self.cout
            finish(d)
        return out

    def _subpix_fwd(self, dst_ptr, use_stats):
        """Four phase convolutions (2x2 folded taps on the low-resolution source, outputs interleaved at stride 2)."""
        f0 = self.srcs[0]
        assert f0.b == 1
        out = []
        for ph, (a, b) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            d = L.ConvDesc()
            d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(f0.t), f0.C, f0.Hp, f0.Wp, 0
            d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
            self._set_taps(d, [(a + u, b + v) for u in (0, 1) for v in (0, 1)])
            d.si, d.Cout = 1, self.cout
            d.wt = L.ptr(self.wt_f) + ph * 4 * self.cout * f0.C * self.wt_f.element_size()
            d.N, d.OH, d.OW = self.N, f0.H, f0.W
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst, d.cdst_off = dst_ptr, self.OH, self.OW, 2, a, b, self.cout, 0
            d.stats, d.stat_mod = (L.ptr(self.stats) if use_stats else None), self.cout
            out.append(d)
        return out

    def dgrad_descs(self):
        """ConvDesc list computing dcat = gradient wrt the block input from draw."""
        self.__dict__.pop('_dg_arr', None)
        out = self._dgrad_descs_raw()
        if self.s2d:
            out[0].wt_fragmajor, out[0].elem_f32 = 1, 0
            want = int(L.load().srvp_conv_wants_fragmajor(C.byref(out[0])))
            assert want == 1, 'space-to-depth data gradient is not eligible for the halo kernel'
            if self.split:
                self._set_layout(out[1:], self.pd_s)
            return out
        if self.split:
            self._set_layout(out[:1], self.pd)
            self._set_layout(out[1:], self.pd_s)       # gradient wrt the skip tensor
        else:
            self._set_layout(out, self.pd)
        return out

    def _dgrad_descs_raw(self):
        k, N, bd = self.k, self.N, self.draw_b
        out = []

        def base():
            d = L.ConvDesc()
            d.src0, d.C0, d.ups0 = L.ptr(self.draw), self.cout, 0
            d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
            d.H0p, d.W0p = self.OH + 2 * bd, self.OW + 2 * bd
            d.Cout = self.ctot
            d.stats, d.stat_mod = None, 1
            d.Cdst, d.cdst_off = self.ctot, 0
            return d
        if self.geom == 'same':
            d = base()
            # dIn[i] = sum_kh dOut[i + p - kh]  -> padded coordinate i + p - kh + bd
            taps = [(self.p - kh + bd, self.p - kw + bd) for kh in range(k) for kw in range(k)]
            if self.s2d:
                # gradient wrt the low-resolution source from the space-to-depth output gradient: phase (a, b), folded tap (u, v) of
                # the forward reads padded input row i + a + u, so the gradient of input row p collects phase-(a, b) output row
                # p + 1 - a - u = padded row p + 2 - a - u of the s2d tensor
                f0 = self.srcs[0]
                d.src0, d.C0, d.H0p, d.W0p = L.ptr(self.draw), 4 * self.cout, f0.H + 2, f0.W + 2
                d.ntaps = 4
                ent = [(2 - a - u, 2 - b - v) for a in (0, 1) for b in (0, 1) for u in (0, 1) for v in (0, 1)]
                d.dy, d.dx = L.taps([e[0] for e in ent]), L.taps([e[1] for e in ent])
                d.tap_phase_chunks = self.cout // 64
                d.si, d.wt = 1, L.ptr(self.wt_d)
                d.N, d.OH, d.OW = N, f0.H, f0.W
                d.Cout, d.Cdst = self.dcat_c, self.dcat_c
                d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), f0.H, f0.W, 1, 0, 0
            elif self.subpix:
                # gradient wrt the low-resolution source: 4x4 stride-2 conv of draw with folded taps (padded row 2 i + dy)
                assert bd == 1
                f0 = self.srcs[0]
                self._set_taps(d, [(dy, dx) for dy in range(4) for dx in range(4)])
                d.si, d.wt = 2, L.ptr(self.wt_d)
                d.N, d.OH, d.OW = N, f0.H, f0.W
                d.Cout, d.Cdst = self.dcat_c, self.dcat_c
                d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), f0.H, f0.W, 1, 0, 0
            else:
                self._set_taps(d, taps)
                d.si, d.wt = 1, L.ptr(self.wt_d)
                d.N, d.OH, d.OW = N, self.Hin, self.Win
                d.Cout, d.Cdst = self.dcat_c, self.dcat_c
                d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), self.Hin, self.Win, 1, 0, 0
            out.append(d)
            if self.split:
                # gradient wrt the skip tensor of each sample = data-gradient of the time-summed output gradient
                d2 = base()
                d2.src0 = L.ptr(self.draw_sum)
                self._set_taps(d2, taps)
                d2.si, d2.wt = 1, L.ptr(self.wt_d_s)
                d2.N, d2.OH, d2.OW = self.B, self.Hin, self.Win
                c1p = self.srcs[1].C
                d2.Cout, d2.Cdst = c1p, c1p
                d2.dst, d2.DHp, d2.DWp, d2.so, d2.ooy, d2.oox = L.ptr(self.dsel), self.Hin, self.Win, 1, 0, 0
                out.append(d2)
        elif self.geom == 'sameT':
            d = base()
            # dIn[i] = sum_kh dOut[i - p + kh] W[ci][co][kh]  -> padded coordinate i - p + kh + bd
            self._set_taps(d, [(kh - self.p + bd, kw - self.p + bd) for kh in range(k) for kw in range(k)])
            d.si, d.wt = 1, L.ptr(self.wt_d)
            d.N, d.OH, d.OW = N, self.Hin, self.Win
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), self.Hin, self.Win, 1, 0, 0
            out.append(d)
        elif self.geom == 'down':
            for ph, (py, px) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
                d = base()
                self._set_taps(d, [(dy, dx) for _, dy in PHASE[py] for _, dx in PHASE[px]])
                d.si = 1
                d.wt = L.ptr(self.wt_d) + ph * 4 * self.ctot * self.cout * self.wt_d.element_size()
                d.N, d.OH, d.OW = N, self.OH, self.OW
                d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), self.Hin, self.Win, 2, py, px
                out.append(d)
        elif self.geom == 'up' and self.s2d_up:
            d = base()
            d.src0, d.C0, d.H0p, d.W0p = L.ptr(self.draw), 4 * self.cout, self.Hin + 2, self.Win + 2
            d.ntaps = 4
            ent = [(dy, dx) for a, b in [(0, 0), (0, 1), (1, 0), (1, 1)] for _, dy in S2D_UP[a] for _, dx in S2D_UP[b]]
            d.dy, d.dx = L.taps([e[0] for e in ent]), L.taps([e[1] for e in ent])
            d.tap_phase_chunks = self.cout // 64
            d.si, d.wt = 1, L.ptr(self.wt_d)
            d.N, d.OH, d.OW = N, self.Hin, self.Win
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), self.Hin, self.Win, 1, 0, 0
            out.append(d)
        elif self.geom == 'up':
            d = base()
            # dIn[ih] = sum_kh dOut[2 ih - 1 + kh] -> padded 2 ih + kh (bd = 1)
            self._set_taps(d, [(kh, kw) for kh in range(k) for kw in range(k)])
            d.si, d.wt = 2, L.ptr(self.wt_d)
            d.N, d.OH, d.OW = N, self.Hin, self.Win
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), self.Hin, self.Win, 1, 0, 0
            out.append(d)
        elif self.geom == 'full':
            d = base()                                 # GEMM: [N][cout] x [cout][(kh,kw,ci)]
            self._set_taps(d, [(0, 0)])
            d.H0p, d.W0p = 1, 1
            d.si, d.wt, d.Cout = 1, L.ptr(self.wt_d), k * k * self.ctot
            d.N, d.OH, d.OW = N, 1, 1
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox, d.Cdst = L.ptr(self.dcat), 1, 1, 1, 0, 0, k * k * self.ctot
            out.append(d)
        elif self.geom == 'expand':
            d = base()                                 # k x k conv of draw [N][k][k][cout] -> [N][1][1][cin]
            self._set_taps(d, [(kh, kw) for kh in range(k) for kw in range(k)])
            d.H0p, d.W0p = k, k
            d.si, d.wt = 1, L.ptr(self.wt_d)
            d.N, d.OH, d.OW = N, 1, 1
            d.dst, d.DHp, d.DWp, d.so, d.ooy, d.oox = L.ptr(self.dcat), 1, 1, 1, 0, 0
            self._dg_fin = self._split_k(d, 'd', L.ptr(self.dcat), None)
            out.append(d)
        return out

    def _wgrad_s2d_up(self):
        """Weight gradient of a transposed 4x4 stride-2 block from its space-to-depth output gradient: ONE 16-tap launch of the per-tap
        kernel, tap t' = ph * 4 + u * 2 + v reading the phase-ph channel slice at offset (dy, dx) of S2D_UP, unit stride (the plain
        form samples the bordered gradient at stride 2)."""
        d = L.WgradDesc()
        self._src_fields(d, None)
        b_in = self.srcs[0].b
        d.ntaps = 16
        ent = [(dy, dx) for a, b in [(0, 0), (0, 1), (1, 0), (1, 1)] for _, dy in S2D_UP[a] for _, dx in S2D_UP[b]]
        d.si, d.so = 1, 1
        if WGRAD_S2D_SHIFT_SRC and b_in == 1:
            # the same sum with the index shifted by the tap's offset (p' = p + o_t - 1): the channel-sliced gradient is read at the FIXED interior
            # position and the tap moves the plain source instead, inside its 3x3 window (both tensors carry zero borders) -- the shape of a
            # sub-pixel block's 16-tap weight gradient, which the halo kernel takes two phases per workgroup over one staged patch
            # (wgrad_halo_kernel<.., 8, 2>; grids narrower than 8 columns stay on the per-tap kernel)
            d.dy, d.dx = L.taps([2 - e[0] for e in ent]), L.taps([2 - e[1] for e in ent])
            d.ooy, d.oox = L.taps([1] * 16), L.taps([1] * 16)
        else:
            d.dy, d.dx = L.taps([b_in] * 16), L.taps([b_in] * 16)
            d.ooy, d.oox = L.taps([e[0] for e in ent]), L.taps([e[1] for e in ent])
        d.dout, d.Cout, d.dout_cstride, d.dout_coff, d.dout_phase_taps = L.ptr(self.draw), self.cout, 4 * self.cout, 0, 4
        d.DHp, d.DWp = self.Hin + 2, self.Win + 2
        d.N, d.OH, d.OW = self.N, self.Hin, self.Win
        d.dw = L.ptr(self.dw)
        bj = 128 if self.cout % 128 == 0 else 64
        bc = 32
        for cand in (128, 64):
            if d.C0 % cand == 0 and (d.C1 == 0 or d.C1 % cand == 0):
                bc = cand
                break
        tiles = (self.cout // bj) * (self.ctot // bc) * 16
        chunks = (self.N * self.Hin * self.Win + 31) // 32
        d.splitk = int(max(1, min((1024 + tiles - 1) // tiles, chunks // 4 if chunks >= 4 else 1)))
        return d

    def _wgrad_s2d_in(self):
        """Weight gradient of a 4x4 stride-2 block whose source is stored space-to-depth, on the per-tap kernel with the operand ROLES
        SWAPPED: the s2d source is the channel-sliced operand (tap t' = ph * 4 + u * 2 + v reads the phase-ph slice at offset (dy, dx) of
        S2D_UP, unit stride), the block's own output gradient the plain one -- so the result is dW transposed, [16][Cin][Cout]
        (self.pu is the matching descriptor)."""
        f0 = self.srcs[0]
        bd = self.draw_b
        d = L.WgradDesc()
        d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(self.draw), self.cout, self.OH + 2 * bd, self.OW + 2 * bd, 0
        d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
        d.ntaps = 16
        ent = [(dy, dx) for a, b in [(0, 0), (0, 1), (1, 0), (1, 1)] for _, dy in S2D_UP[a] for _, dx in S2D_UP[b]]
        d.si, d.so = 1, 1
        if WGRAD_S2D_SHIFT_SRC and bd == 1:
            # (as in _wgrad_s2d_up: the s2d operand at its fixed interior position, the tap moves the plain one inside its 3x3 window)
            d.dy, d.dx = L.taps([2 - e[0] for e in ent]), L.taps([2 - e[1] for e in ent])
            d.ooy, d.oox = L.taps([1] * 16), L.taps([1] * 16)
        else:
            d.dy, d.dx = L.taps([bd] * 16), L.taps([bd] * 16)
            d.ooy, d.oox = L.taps([e[0] for e in ent]), L.taps([e[1] for e in ent])
        d.dout, d.Cout, d.dout_cstride, d.dout_coff, d.dout_phase_taps = L.ptr(f0.t), f0.C, 4 * f0.C, 0, 4
        d.DHp, d.DWp = f0.H // 2 + 2, f0.W // 2 + 2
        d.N, d.OH, d.OW = self.N, self.OH, self.OW
        d.dw = L.ptr(self.dw)
        bj = 128 if f0.C % 128 == 0 else 64
        bc = 128 if self.cout % 128 == 0 else (64 if self.cout % 64 == 0 else 32)
        tiles = (f0.C // bj) * (self.cout // bc) * 16
        chunks = (self.N * self.OH * self.OW + 31) // 32
        d.splitk = int(max(1, min((1024 + tiles - 1) // tiles, chunks // 4 if chunks >= 4 else 1)))
        return d

    def wgrad_desc(self):
        if self.s2d_in:
            return self._wgrad_s2d_in()
        if self.s2d_up:
            return self._wgrad_s2d_up()
        if self.s2d:
            main = self._wgrad_s2d()
            return main + [self._wgrad_one('s')] if self.split else main
        if self.split:
            return [self._wgrad_one('h'), self._wgrad_one('s')]
        return self._wgrad_one(None)

    def _wgrad_s2d(self):
        """Weight gradient of a sub-pixel block from its space-to-depth output gradient; entries (a*2+b)*4 + u*2+v of dw.
        ONE 16-tap launch, tap t reading the slice of phase t / 4 at unit stride: the halo kernel takes it with two phases per workgroup
        (round 4: wgrad_halo_kernel<.., 8, 2>; low-resolution grids narrower than 8 columns stay on the per-tap kernel).
        SRVP_WGRAD_HALO2=0 (round 3): cout = 64: four 4-tap HALO launches, one per output phase; wider layers: the 4-tap halo kernel is
        LDS-DMA bound (measured 0.55-0.60 vs 0.50-0.52 ms), so the 16-tap launch runs on the per-tap kernel."""
        f0 = self.srcs[0]
        out = []
        if self.cout > 64 or WGRAD_HALO2:
            ent = [(a, b, u, v) for a in (0, 1) for b in (0, 1) for u in (0, 1) for v in (0, 1)]
            d = L.WgradDesc()
            d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(f0.t), f0.C, f0.Hp, f0.Wp, 0
            d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
            d.ntaps = 16
            d.dy, d.dx = L.taps([a + u for a, b, u, v in ent]), L.taps([b + v for a, b, u, v in ent])
            d.si, d.so = 1, 1
            d.ooy, d.oox = L.taps([1] * 16), L.taps([1] * 16)
            d.dout, d.Cout, d.dout_cstride, d.dout_coff, d.dout_phase_taps = L.ptr(self.draw), self.cout, 4 * self.cout, 0, 4
            d.DHp, d.DWp = f0.H + 2, f0.W + 2
            d.N, d.OH, d.OW = self.N, f0.H, f0.W
            d.dw = L.ptr(self.dw)
            bj = 128 if self.cout % 128 == 0 else 64
            bc = 128 if f0.C % 128 == 0 else 64
            tiles = (self.cout // bj) * (f0.C // bc) * 16
            chunks = (self.N * f0.H * f0.W + 31) // 32
            d.splitk = int(max(1, min((1024 + tiles - 1) // tiles, chunks // 4 if chunks >= 4 else 1)))
            return [d]
        for ph, (a, b) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            d = L.WgradDesc()
            d.src0, d.C0, d.H0p, d.W0p, d.ups0 = L.ptr(f0.t), f0.C, f0.Hp, f0.Wp, 0
            d.src1, d.C1, d.H1p, d.W1p, d.ups1, d.map1 = None, 0, 1, 1, 0, None
            d.ntaps = 4
            d.dy, d.dx = L.taps([a + u for u in (0, 1) for v in (0, 1)]), L.taps([b + v for u in (0, 1) for v in (0, 1)])
            d.si, d.so = 1, 1
            d.ooy, d.oox = L.taps([1] * 4), L.taps([1] * 4)
            d.dout, d.Cout, d.dout_cstride, d.dout_coff = L.ptr(self.draw), self.cout, 4 * self.cout, ph * self.cout
            d.DHp, d.DWp = f0.H + 2, f0.W + 2
            d.N, d.OH, d.OW = self.N, f0.H, f0.W
            d.dw = L.ptr(self.dw) + 4 * ph * 4 * self.cout * self.dcat_c
            # split-K for the per-tap kernel (the halo kernel picks its own): enough workgroups to fill the chip
            bj = 128 if self.cout % 128 == 0 else 64
            bc = 128 if f0.C % 128 == 0 else 64
            tiles = (self.cout // bj) * (f0.C // bc) * 4
            chunks = (self.N * f0.H * f0.W + 31) // 32
            d.splitk = int(max(1, min((1024 + tiles - 1) // tiles, chunks // 4 if chunks >= 4 else 1)))
            out.append(d)
        return out

    # (tests / tools) the output gradient in the block's own layout
    def put_draw(self, dr):
        """dr: [N][OH][OW][cout_real] -> self.draw (bordered NHWC, or space-to-depth for s2d blocks)."""
        self.draw.zero_()
        cr = dr.shape[-1]
        if self.s2d:
            v = self.draw.view(self.N, self.OH // 2 + 2, self.OW // 2 + 2, 4, self.cout)
            for a in (0, 1):
                for b in (0, 1):
                    v[:, 1:-1, 1:-1, a * 2 + b, :cr].copy_(dr[:, a::2, b::2])
        else:
            bd = self.draw_b
            self.draw[:, bd:bd + self.OH, bd:bd + self.OW, :cr].copy_(dr)

    def get_draw(self):
        """[N][OH][OW][cout] view-independent copy of the stored output gradient (rounded as stored)."""
        if self.s2d:
            v = self.draw.view(self.N, self.OH // 2 + 2, self.OW // 2 + 2, 4, self.cout)
            out = torch.empty(self.N, self.OH, self.OW, self.cout, dtype=self.draw.dtype, device=self.draw.device)
            for a in (0, 1):
                for b in (0, 1):
                    out[:, a::2, b::2] = v[:, 1:-1, 1:-1, a * 2 + b]
            return out
        bd = self.draw_b
        return self.draw[:, bd:bd + self.OH, bd:bd + self.OW].clone()

    def _wgrad_one(self, which):
        k, N, bd = self.k, self.N, self.draw_b
        d = L.WgradDesc()
        d.elem_f32 = 1 if self.f32 else 0
        self._src_fields(d, which)
        b_in = self.srcs[0].b
        nat = [(kh, kw) for kh in range(k) for kw in range(k)]
        d.ntaps = k * k
        d.dout, d.Cout = L.ptr(self.draw), self.cout
        d.DHp, d.DWp = self.OH + 2 * bd, self.OW + 2 * bd
        if self.subpix and which != 's':
            # 16 (phase, folded tap) weight gradients: low-resolution input taps, output gradient sampled at stride 2
            f0 = self.srcs[0]
            ent = [(a, b, u, v) for a in (0, 1) for b in (0, 1) for u in (0, 1) for v in (0, 1)]
            d.ups0 = 0
            d.ntaps = 16
            d.dy, d.dx = L.taps([a + u for a, b, u, v in ent]), L.taps([b + v for a, b, u, v in ent])
            d.si, d.so = 1, 2
            d.ooy, d.oox = L.taps([a + bd for a, b, u, v in ent]), L.taps([b + bd for a, b, u, v in ent])
            d.N, d.OH, d.OW = N, f0.H, f0.W
        elif self.geom in ('same', 'down', 'full'):
            off = b_in - self.p
            d.dy, d.dx = L.taps([kh + off for kh, _ in nat]), L.taps([kw + off for _, kw in nat])
            d.si, d.so = self.s, 1
            d.ooy, d.oox = L.taps([bd] * (k * k)), L.taps([bd] * (k * k))
            d.N, d.OH, d.OW = N, self.OH, self.OW
        elif self.geom == 'sameT':
            d.dy, d.dx = L.taps([self.p - kh + b_in for kh, _ in nat]), L.taps([self.p - kw + b_in for _, kw in nat])
            d.si, d.so = 1, 1
            d.ooy, d.oox = L.taps([bd] * (k * k)), L.taps([bd] * (k * k))
            d.N, d.OH, d.OW = N, self.OH, self.OW
        elif self.geom == 'up':
            d.dy, d.dx = L.taps([b_in] * (k * k)), L.taps([b_in] * (k * k))
            d.si, d.so = 1, 2
            d.ooy, d.oox = L.taps([kh - self.p + bd for kh, _ in nat]), L.taps([kw - self.p + bd for _, kw in nat])
            d.N, d.OH, d.OW = N, self.Hin, self.Win
        elif self.geom == 'expand':
            d.dy, d.dx = L.taps([b_in] * (k * k)), L.taps([b_in] * (k * k))
            d.si, d.so = 1, 1
            d.ooy, d.oox = L.taps([kh for kh, _ in nat]), L.taps([kw for _, kw in nat])
            d.N, d.OH, d.OW = N, 1, 1
        d.dw = L.ptr(self.dw)
        ctot = self.ctot
        if which == 's':
            d.dout, d.N, d.dw, ctot = L.ptr(self.draw_sum), self.B, L.ptr(self.dw_s), self.srcs[1].C
        elif which == 'h':
            ctot = self.srcs[0].C
        # split-K: enough workgroups to fill the chip, but no more than one per 4 pixel chunks
        M = d.N * d.OH * d.OW
        bj = 128 if self.cout % 128 == 0 else (64 if self.cout % 64 == 0 else 32)
        c0, c1 = d.C0, d.C1
        bc = 32
        for cand in (128, 64):
            if c0 % cand == 0 and (c1 == 0 or c1 % cand == 0):
                bc = cand
                break
        tiles = (self.cout // bj) * (ctot // bc) * d.ntaps
        chunks = (M + 31) // 32
        d.splitk = int(max(1, min((1024 + tiles - 1) // tiles, chunks // 4 if chunks >= 4 else 1)))
        if DETERMINISTIC and self.f32:
            d.splitk = 1          # one workgroup per weight tile: a single atomic per element, nothing arrives in a varying order
        return d


class ConvNetBase:
    """Shared launch helpers."""

    def _fold_eval_bn(self, blk):
        """Inference dataflow (reference conv.py:103-104 in eval mode: BatchNorm with running statistics is a per-channel affine map):
        the block's forward launches apply scale / shift / activation to their fp32 accumulators and write the ACTIVATED output where
        the consumer reads it (srvp_conv_desc.ep_*) -- no raw tensor, no srvp_bn_act pass (19 % of the kernel time of a long-horizon
        rollout in round 3).  Plain 3x3 blocks (incl. hoisted-skip and sub-pixel forms) and transposed 4x4 stride-2 blocks."""
        blk._ep = False
        if not (EVAL_FOLD and not blk.training and not blk.f32 and blk.role == 'mfma' and blk.has_bn and blk.out is not None
                and blk.pool is None and blk.geom in ('same', 'up') and not getattr(blk.out, 's2d', False)
                and blk.out.C == blk.cout and getattr(blk, '_fwd_fin', None) is None):
            return
        o = blk.out
        descs = blk._fwd[1:] if blk.split else blk._fwd          # (split: the first launch writes the fp32 hoisted skip half S)
        for d in descs:
            assert d.Cout == blk.cout and not d.dst_is_f32 and not d.out_f32
            assert d.DHp == blk.OH and d.DWp == blk.OW and d.cdst_off == 0
            d.dst, d.Cdst = L.ptr(o.t), o.C
            d.stats, d.stat_mod = None, 1
            d.ep_coef, d.ep_act, d.ep_border = L.ptr(blk.coef), blk.act, o.b
        blk.__dict__.pop('_fwd_arr', None)
        blk._ep = True

    def _eval_coeffs(self, blk, params, st):
        """scale / shift of an eval-mode BatchNorm from its running statistics (recomputed per forward: the statistics are written by
        kernels through raw pointers, so no tensor version tells whether they changed; a 6 us launch off the critical chain)."""
        bk = blk.spec['bnkey']
        g, b, rm, rv = params[bk + '.weight'], params[bk + '.bias'], params[bk + '.running_mean'], params[bk + '.running_var']
        L.call('srvp_bn_eval_coeffs', L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(blk.coef[0]), L.ptr(blk.coef[1]), blk.cout, blk.cout_r, BN_EPS, st)

    def _bn_forward(self, blk, params, st, sync, keep=None):
        N, C_ = blk.N, blk.cout
        scale, shift, mean, invstd = (blk.coef[i] for i in range(4))
        out, pool = blk.out, blk.pool
        act_args = (blk.act, N, blk.OH, blk.OW, C_,
                    L.ptr(out.t) if out is not None else None, out.b if out is not None else 0,
                    L.ptr(pool.t) if pool is not None else None, pool.b if pool is not None else 0,
                    L.ptr(blk.out_f32), L.ptr(keep) if pool is not None else None)
        if blk.has_bn:
            bk = blk.spec['bnkey']
            g, b = params[bk + '.weight'], params[bk + '.bias']
            rm, rv, nbt = params[bk + '.running_mean'], params[bk + '.running_var'], params.get(bk + '.num_batches_tracked')
            if blk.training:
                count = float(N * blk.OH * blk.OW)
                if DETERMINISTIC and blk.f32:
                    # the statistics in a fixed summation order, from the stored fp32 raw output (= the accumulators, unrounded)
                    L.call('srvp_bn_stats_f32_det', L.ptr(blk.raw), blk.raw.numel() // C_, C_, L.ptr(blk.stats), st)
                if sync is not None:
                    count = sync.allreduce_stats(blk.stats, count, site=(bk, 'f'))
                blk.count = count
                if BN_FUSED_FINALIZE:
                    L.call('srvp_bn_finalize_act', L.ptr(blk.raw), L.ptr(blk.stats), count, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
                           L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), blk.cout_r, BN_EPS, BN_MOMENTUM, *act_args,
                           L.ptr(getattr(blk, 'raw_pool', None)) if pool is not None else None,
                           1 if blk.f32 else 0, 1 if (out is not None and out.s2d) else 0, st)
                    return
                L.call('srvp_bn_finalize', L.ptr(blk.stats), count, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
                       L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), C_, blk.cout_r, BN_EPS, BN_MOMENTUM, st)
            else:
                L.call('srvp_bn_eval_coeffs', L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(scale), L.ptr(shift), C_,
                       blk.cout_r, BN_EPS, st)
        if out is not None and out.s2d:
            L.call('srvp_bn_act_s2d', L.ptr(blk.raw), L.ptr(scale), L.ptr(shift), blk.act, N, blk.OH, blk.OW, C_, L.ptr(out.t), st)
            return
        L.call('srvp_bn_act_keep_f32' if blk.f32 else 'srvp_bn_act_keep', L.ptr(blk.raw), L.ptr(scale), L.ptr(shift), *act_args, st)

    def _block_forward(self, blk, params, st, sync, x=None, keep=None):
        if blk.role == 'in':
            w = params[blk.spec['key'] + '.weight']
            L.call('srvp_conv_in_fwd_f32' if blk.f32 else 'srvp_conv_in_fwd', L.ptr(x), L.ptr(w), L.ptr(blk.raw),
                   L.ptr(blk.stats) if (blk.has_bn and blk.training and not (DETERMINISTIC and blk.f32)) else None,
                   blk.N, blk.cin_r[0], 64, 64, blk.cout, blk.cout_r, blk.k, blk.s, blk.p, st)
        elif getattr(blk, '_ep', False):
            skip_s = blk.split and getattr(self, '_skips_done', False)
            if blk.subpix:
                assert len(blk._fwd) == (5 if blk.split else 4), len(blk._fwd)       # [conv_s(skip)] + the four output phases
                if blk.split and not skip_s:
                    L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)
                arr = blk.__dict__.get('_fwd_arr')
                if arr is None:
                    arr = blk._fwd_arr = (L.ConvDesc * 4)(*blk._fwd[-4:])
                L.call('srvp_conv_mfma_multi', arr, 4, st)
            else:
                for d in (blk._fwd[1:] if skip_s else blk._fwd):
                    L.call('srvp_conv_mfma', C.byref(d), st)
            return                                   # (activated output written by the launches above)
        elif blk.subpix:
            # [conv_s(skip) when split], then the four output phases as one grid
            for d in blk._fwd[:-4]:
                if not getattr(self, '_skips_done', False):
                    L.call('srvp_conv_mfma', C.byref(d), st)
            arr = blk.__dict__.get('_fwd_arr')
            if arr is None:
                arr = blk._fwd_arr = (L.ConvDesc * 4)(*blk._fwd[-4:])
            L.call('srvp_conv_mfma_multi', arr, 4, st)
        elif blk.geom == 'up' and len(blk._fwd) == 4 and PHASES_ONE_GRID:
            # transposed 4x4 stride-2 block (conv.py:299-304): its four output phases as ONE grid (same shapes: the library runs them on the
            # halo kernel back to back per tile, or one after the other where it cannot)
            arr = blk.__dict__.get('_fwd_arr')
            if arr is None:
                arr = blk._fwd_arr = (L.ConvDesc * 4)(*blk._fwd)
            L.call('srvp_conv_mfma_multi', arr, 4, st)
            blk.finish_fwd(st)
        else:
            for d in (blk._fwd[1:] if (blk.split and getattr(self, '_skips_done', False)) else blk._fwd):
                L.call('srvp_conv_mfma', C.byref(d), st)
            blk.finish_fwd(st)
        self._bn_forward(blk, params, st, sync, keep)

    @staticmethod
    def _bnbwd_desc(blk, da):
        """BnBwdDesc of (block, gradient source).  Built once per distinct source (the step hands the same plan-owned tensors over every
        time: 24 descriptors of ~25 fields per step otherwise, ~0.25 ms of host work) -- the key holds every field that comes from `da`."""
        key = (da['t'].data_ptr(), da['mode'], da['cstride'], da['coff'], da['border'], bool(da.get('f32')),
               L.ptr(da.get('da2')), L.ptr(da.get('da2_idx')), blk.raw.data_ptr(), blk.out.t.data_ptr() if blk.out is not None else 0)
        cache = blk.__dict__.setdefault('_bnbwd_cache', {})
        d = cache.get(key)
        if d is None:
            if len(cache) > 8:
                cache.clear()
            d = cache[key] = ConvNetBase._bnbwd_desc_build(blk, da)
        return d

    @staticmethod
    def _bnbwd_desc_build(blk, da):
        d = L.BnBwdDesc()
        d.elem_f32 = 1 if blk.f32 else 0
        d.draw_s2d = 1 if getattr(blk, 's2d', False) else 0
        d.raw = L.ptr(blk.raw)
        d.act, d.act_border = (L.ptr(blk.out.t), blk.out.b) if blk.out is not None else (None, 0)
        d.scale, d.shift, d.mean, d.invstd = (L.ptr(blk.coef[i]) for i in range(4))
        d.act_kind = blk.act
        d.da, d.da_mode, d.da_cstride, d.da_coff, d.da_border = L.ptr(da['t']), da['mode'], da['cstride'], da['coff'], da['border']
        d.da_is_f32 = 1 if da.get('f32') else 0
        d.da2, d.da2_idx = L.ptr(da.get('da2')), L.ptr(da.get('da2_idx'))
        d.N, d.H, d.W, d.C = blk.N, blk.OH, blk.OW, blk.cout
        return d

    def _bn_backward(self, blk, params, grads, da, st, sync, in_wgrad=None):
        """da: dict(t, mode, cstride, coff, border, f32, da2, da2_idx).  Produces blk.draw (and BN param grads).
        in_wgrad = (x, dw) (the image-side first block): where the library serves it, the block's weight gradient is formed in the same
        launch from dA and raw and blk.draw is NOT written (srvp_conv_in_wgrad_bn); returns True then."""
        d = self._bnbwd_desc(blk, da)
        if blk.split and blk.draw_b == 1:
            d.tsum, d.tsum_T = L.ptr(blk.draw_sum), blk.N // blk.B      # time-summed gradient for the hoisted skip half
        if blk.has_bn:
            if not getattr(blk, '_reduce_fused', False):       # (else: accumulated by the consumer's data-gradient launch, _fuse_bn_reduce)
                L.call('srvp_bn_bwd_reduce', C.byref(d), L.ptr(blk.red), st)
            elif d.da_mode == 2 and da.get('da2') is not None and not da.get('da2_reduced'):
                # pooled layer whose arg-max terms rode the consumer's data-gradient launch (raw_pool): the skip-connection gradient of its
                # B selected frames is the only term left -- both sums are linear in the gradient
                self._skip_reduce(blk, da, st)
            local = float(blk.N * blk.OH * blk.OW)
            count = local
            if sync is not None:
                count = sync.allreduce_stats(blk.red, count, site=(blk.spec['bnkey'], 'b'))
            bk = blk.spec['bnkey']
            # (count / local = number of ranks whose sums are in `red`: the parameter gradients are formed from the global sums
            # and must come out world times smaller, see srvp_hip.h)
            if BN_FUSED_FINALIZE:
                if (in_wgrad is not None and IN_WGRAD_BN and not blk.f32
                        and L.load().srvp_conv_in_wgrad_bn_ok(C.byref(d), blk.cin_r[0], 64, 64, blk.cout, blk.k, blk.s, blk.p)):
                    L.call('srvp_conv_in_wgrad_bn', L.ptr(in_wgrad[0]), C.byref(d), L.ptr(blk.red), count, L.ptr(grads[bk + '.weight']),
                           L.ptr(grads[bk + '.bias']), L.ptr(blk.bcoef), blk.cout_r, local / count, L.ptr(in_wgrad[1]), blk.N, blk.cin_r[0], blk.cout_r, st)
                    return True
                L.call('srvp_bn_bwd_finalize_apply', C.byref(d), L.ptr(blk.red), count, L.ptr(grads[bk + '.weight']), L.ptr(grads[bk + '.bias']),
                       L.ptr(blk.bcoef), blk.cout_r, local / count, L.ptr(blk.draw), blk.draw_b, st)
                return
            L.call('srvp_bn_bwd_finalize', L.ptr(blk.red), count, L.ptr(blk.coef[0]), L.ptr(blk.coef[2]), L.ptr(blk.coef[3]),
                   L.ptr(grads[bk + '.weight']), L.ptr(grads[bk + '.bias']), L.ptr(blk.bcoef), blk.cout, blk.cout_r, 1, local / count, st)
        elif not getattr(blk, '_bcoef_const', False):
            # block without BatchNorm: constant coefficients (1, 0, 0), written once
            L.call('srvp_bn_bwd_finalize', None, 1.0, None, None, None, None, None, L.ptr(blk.bcoef), blk.cout, blk.cout_r, 0, 1.0, st)
            blk._bcoef_const = True
        L.call('srvp_bn_bwd_apply', C.byref(d), L.ptr(blk.bcoef), L.ptr(blk.draw), blk.draw_b, st)

    @staticmethod
    def _wgrad(blk, st):
        for d in (blk._wg if isinstance(blk._wg, list) else [blk._wg]):
            L.call('srvp_wgrad_mfma', C.byref(d), st)

    def _mfma_backward(self, blk, grads, st, need_dgrad=True, wgrad=True):
        # (split blocks: the time-summed output gradient draw_sum was written by srvp_bn_bwd_apply alongside draw)
        if wgrad:
            self._wgrad(blk, st)
        if need_dgrad:
            if blk.geom == 'down' and len(blk._dg) == 4 and PHASES_ONE_GRID:
                # data gradient of a 4x4 stride-2 block: the four input phases as ONE grid
                arr = blk.__dict__.get('_dg_arr')
                if arr is None:
                    arr = blk._dg_arr = (L.ConvDesc * 4)(*blk._dg)
                L.call('srvp_conv_mfma_multi', arr, 4, st)
            else:
                for d in blk._dg:
                    L.call('srvp_conv_mfma', C.byref(d), st)
            blk.finish_dgrad(st)

    # ---- whole-network launches: every layer's pack / unpack in ONE kernel (device-resident job table, rebuilt only when a
    # pointer or layout changes) and one multi-tensor zero for the accumulators -- ~140 tiny launches per step otherwise
    @staticmethod
    def _job_table(jobs, dev, cache, src_is_tensor):
        """Device-resident job tables of one network: 'tiles' (jobs the lean tile kernels take: srvp_pack_job_tiles > 0) and 'multi'
        (the rest: fp32 parity mode, odd channel counts), each (table, njobs, workgroups) or None."""
        key = tuple((s.data_ptr() if src_is_tensor else s, d if src_is_tensor else d.data_ptr(), pd.layout) for s, d, pd in jobs)
        if cache.get('key') != key:
            lib = L.load()
            parts = {'tiles': [], 'multi': []}
            for s, d, pd in jobs:
                nt = int(lib.srvp_pack_job_tiles(C.byref(pd), 1 if src_is_tensor else 0)) if PACK_TILES else 0
                parts['tiles' if nt > 0 and len(parts['tiles']) < 256 else 'multi'].append((s, d, pd, nt))
            cache.update(key=key, keep=[t for j in jobs for t in j[:2] if torch.is_tensor(t)])
            for name, part in parts.items():
                cache[name] = None
                if not part:
                    continue
                arr = (L.PackJob * len(part))()
                mx = 0
                for i, (s, d, pd, nt) in enumerate(part):
                    arr[i].src = s.data_ptr() if src_is_tensor else s
                    arr[i].dst = d if src_is_tensor else d.data_ptr()
                    arr[i].d = pd
                    mx += nt if name == 'tiles' else lib.srvp_pack_job_wgs(pd.ntaps * pd.J * pd.K)
                raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                cache[name] = (raw.to(dev), len(part), mx)
        return cache

    def pack_weights(self, params, st):
        jobs = [j for blk in self.blocks if blk.role in ('mfma', 'out') for j in blk.pack_jobs(params[blk.spec['key'] + '.weight'])]
        if not jobs:
            return
        c = self._job_table(jobs, self.dev, self.__dict__.setdefault('_pack_cache', {}), False)
        if c['tiles']:
            L.call('srvp_pack_weight_tiles', L.ptr(c['tiles'][0]), c['tiles'][1], c['tiles'][2], st)
        if c['multi']:
            L.call('srvp_pack_weight_multi', L.ptr(c['multi'][0]), c['multi'][1], c['multi'][2], st)

    def unpack_wgrads(self, grads, st, part=None):
        """part = (lo, hi): the blocks lo <= index < hi only (the encoder unpacks its heavy deep blocks while the weight gradients of its
        first blocks are still to come, UNPACK_SPLIT)."""
        f32_out = getattr(self, '_f32_out', lambda: False)()       # (decoder) the image-side layer accumulates into grads directly
        jobs = [j for i, blk in enumerate(self.blocks) if blk.role in ('mfma', 'out') and not (f32_out and blk is self.blocks[-1])
                and (part is None or part[0] <= i < part[1])
                for j in blk.unpack_jobs(grads[blk.spec['key'] + '.weight'])]
        if not jobs:
            return
        c = self._job_table(jobs, self.dev, self.__dict__.setdefault('_unpack_cache' if part is None else '_unpack_cache_%d_%d' % part, {}), True)
        if c['tiles']:
            L.call('srvp_unpack_wgrad_tiles', L.ptr(c['tiles'][0]), c['tiles'][1], c['tiles'][2], st)
        if c['multi']:
            L.call('srvp_unpack_wgrad_multi', L.ptr(c['multi'][0]), c['multi'][1], c['multi'][2], st)

    def _fuse_bn_reduce(self):
        """Pairs (producer P, consumer Q) where Q is a plain 3x3 stride-1 block reading P's activated output at the same resolution
        and nothing else adds to that gradient (no pooling, upsampling or skip connection in between): Q's data-gradient launch then
        carries P's raw output / coefficients and accumulates P's two BatchNorm-backward sums in its epilogue
        (srvp_conv_desc.bnr_*), and P's srvp_bn_bwd_reduce launch -- a second read of (dA, raw) -- is dropped."""
        if not (BN_FUSED_REDUCE and self.training and not self.f32):
            return
        lib = L.load()
        for q in self.blocks:
            # (round 5, measured and not kept: a transposed 4x4 stride-2 consumer in its space-to-depth form -- one stride-1 halo data-gradient launch at
            # the producer's resolution, the DCGAN decoder's chain -- qualifies as well; 6.36 -> 6.31 ms on config 2, docs/experiments.md)
            if q.role != 'mfma' or getattr(q, 'geom', None) != 'same' or not getattr(q, '_dg', None):
                continue
            # an upsampling consumer qualifies in its space-to-depth form only: its data gradient is then ONE launch that writes the
            # gradient wrt the LOW-resolution source, already summed over each 2x2 cell -- the producer's dA
            if (q.ups or q.subpix) and not (q.subpix and q.s2d):
                continue
            p = next((b for b in self.blocks if b.out is not None and b.out is q.srcs[0]), None)
            if p is None:
                # a POOLED producer (round 4): its forward stores the raw value at every window's arg-max (raw_pool, srvp_bn_finalize_act), the
                # pooled gradient q writes lands on exactly that position, so q's launch accumulates the sums from (dA_pooled, raw_pool) as for
                # a same-resolution pair -- the VALU-bound pass that re-derived the arg-max from four raw pixels per window (1.7 ms per step over
                # the four VGG stages) is gone; a skip-connection gradient on top is added by a da_mode 3 reduction of its B frames
                pp = next((b for b in self.blocks if b.pool is not None and b.pool is q.srcs[0]), None)
                if (pp is None or not POOL_FUSED_REDUCE or not BN_FUSED_FINALIZE or not pp.has_bn or pp.act != L.ACT_LRELU or (q.ups or q.subpix)
                        or len(q._dg) != 1):
                    continue
                d = q._dg[0]
                shape = (pp.N, pp.OH // 2, pp.OW // 2, pp.cout)
                if shape != tuple(q.dcat.shape) or d.Cout != pp.cout or int(lib.srvp_conv_runs_on_halo(C.byref(d))) < 256:
                    continue
                pp.raw_pool = torch.empty(shape, dtype=pp.adt, device=self.dev)
                d.bnr_raw, d.bnr_coef, d.bnr_red = L.ptr(pp.raw_pool), L.ptr(pp.coef), L.ptr(pp.red)
                pp._reduce_fused = True
                continue
            if not p.has_bn or p.act != L.ACT_LRELU or p.spec.get('skip_out') is not None or p.pool is not None:
                continue
            d = q._dg[0]
            if tuple(p.raw.shape) != tuple(q.dcat.shape) or d.Cout != p.cout or int(lib.srvp_conv_runs_on_halo(C.byref(d))) < 256:
                continue
            d.bnr_raw, d.bnr_coef, d.bnr_red = L.ptr(p.raw), L.ptr(p.coef), L.ptr(p.red)
            p._reduce_fused = True

    def _pool_accumulators(self):
        """Re-homes the per-step accumulators of all blocks (BN statistics, BN-backward sums, weight-gradient tiles) in three flat
        buffers, so that clearing them is three fills per network instead of one per tensor (torch._foreach_zero_ on a list of
        mixed dtypes falls back to one launch per tensor: ~65 tiny dependent launches per step).  Called between block
        construction and descriptor construction (descriptors capture the pointers)."""
        def pool(names, dtype):
            ts = [(b, n) for b in self.blocks for n in names if getattr(b, n, None) is not None]
            if not ts:
                return None
            sizes = [((getattr(b, n).numel() + 31) // 32) * 32 for b, n in ts]
            flat = torch.zeros(sum(sizes), dtype=dtype, device=self.dev)
            off = 0
            for (b, n), sz in zip(ts, sizes):
                t = getattr(b, n)
                assert t.dtype == dtype
                setattr(b, n, flat[off:off + t.numel()].view(t.shape))
                off += sz
            return flat
        self._acc_fwd = pool(('stats',), torch.float64)
        self._acc_bwd = [t for t in (pool(('red',), torch.float64), pool(('dw', 'dw_s'), torch.float32)) if t is not None]

    def zero_forward_accumulators(self):
        if self._acc_fwd is not None:
            self._acc_fwd.zero_()

    def zero_backward_accumulators(self):
        for t in self._acc_bwd:
            t.zero_()


class EncoderNet(ConvNetBase):
    """conv.py:129-154 on N = T*B frames: x fp32 (N, C, 64, 64) -> hx fp32 (N, nh) and the stage outputs (skips)."""

    def __init__(self, specs, N, device, training, f32=False, use_skips=True):
        """use_skips: the stage outputs feed decoder skip connections (they are then read by decoder convolutions in the plain layout).
        Without them (DCGAN without --skipco: config 2) the activations consumed by the 4x4 stride-2 layers are stored space-to-depth."""
        self.N, self.dev, self.training, self.f32 = N, device, training, bool(f32)
        adt = torch.float32 if f32 else torch.bfloat16
        self.blocks = []
        cur = None
        for i, sp in enumerate(specs):
            role = 'in' if i == 0 else 'mfma'
            blk = Block(sp, role, [] if role == 'in' else [cur], False, N, device, training, f32=f32)
            last = i == len(specs) - 1
            nxt_pool = (not last) and specs[i + 1]['pre'] == 'pool'
            nxt = None if last else specs[i + 1]
            # the activation is stored space-to-depth when its consumer is a 4x4 stride-2 convolution that the halo kernel can take
            # (64-multiple channels, power-of-two output grid of whole images per tile or 16-multiples) and nothing else reads it
            oh2 = blk.OH // 2
            s2d_out = bool(DOWN_S2D and not f32 and not use_skips and nxt is not None and nxt['kind'] == 'conv'
                           and (nxt['k'], nxt['s'], nxt['p']) == (4, 2, 1) and not nxt_pool and cpad(blk.cout_r) % 64 == 0
                           and blk.OH % 2 == 0 and blk.OH == blk.OW and oh2 >= 4 and (oh2 & (oh2 - 1)) == 0 and (oh2 * oh2 <= 256 or oh2 % 16 == 0))
            if last:
                blk.out_f32 = torch.empty(N, blk.cout, dtype=torch.float32, device=device)
                cur = None
            else:
                need_full = (sp['skip_out'] is not None) or not nxt_pool or training
                blk.out = Feat(N, blk.OH, blk.OW, blk.cout_r, device, dtype=adt, s2d=s2d_out) if need_full else None
                if nxt_pool:
                    blk.pool = Feat(N, blk.OH // 2, blk.OW // 2, blk.cout_r, device, dtype=adt)
                    cur = blk.pool
                else:
                    cur = blk.out
            self.blocks.append(blk)
        self._pool_accumulators()
        for blk in self.blocks:
            if blk.role == 'mfma':
                blk._fwd = blk.fwd_descs()
                if training:
                    blk._dg = blk.dgrad_descs()
                    blk._wg = blk.wgrad_desc()
        self._fuse_bn_reduce()
        # keyed by the decoder's skip index (deepest first, conv.py:153): encoder stage s feeds decoder block 3 - s
        self.skips = {3 - sp['skip_out']: b.out for sp, b in zip(specs, self.blocks) if sp['skip_out'] is not None}
        self.nh_r = specs[-1]['cout']

    def forward(self, x, params, st, sync=None, keep=None, packed=None, on_skip=None):
        """on_skip (callable, optional): called with the decoder's skip index right after the block whose output feeds that skip connection
        (the caller may start the decoder's hoisted skip convolution of that stage on another stream at once).
        keep: int32 [N] or None.  The full-resolution activation of a POOLED block is read by nothing but the skip connections
        (the next layer reads the pooled tensor, BN backward recomputes it from the raw output), i.e. for one frame per sample:
        with `keep` given, pooled blocks store it only for the frames with keep[n] != 0 (saves ~2 GB of writes per step at the
        headline config).  packed: event after which the packed MFMA weights are ready (they are packed on another stream while the
        image-side layer, which reads the fp32 weights, runs)."""
        if self.training:
            self.zero_forward_accumulators()
        for blk in self.blocks:
            if packed is not None and blk.role != 'in':
                L.wait(packed)
                packed = None
            self._block_forward(blk, params, st, sync, x=x, keep=keep)
            if on_skip is not None and blk.spec['skip_out'] is not None:
                on_skip(3 - blk.spec['skip_out'])
        return self.blocks[-1].out_f32[:, :self.nh_r]

    def _skip_term(self, blk, sg):
        """da entries for the skip-connection gradient sg = (rows bf16 [B][H][W][C], frame -> row int32 [N][, row -> frame int32 [B]])"""
        out = dict(da2=sg[0], da2_idx=sg[1])
        if len(sg) > 2:
            out['da2_sel'] = sg[2]                                       # row -> frame (the inverse of da2_idx)
        else:
            # (a caller that hands over the frame -> row map only: invert it here, a few tiny launches off the hot path)
            inv = getattr(self, '_da2_sel', None)
            if inv is None or inv.numel() != sg[0].shape[0]:
                inv = self._da2_sel = torch.zeros(sg[0].shape[0], dtype=torch.int32, device=self.dev)
            fr = torch.nonzero(sg[1] >= 0).flatten()
            inv[sg[1][fr].long()] = fr.to(torch.int32)
            out['da2_sel'] = inv
        return out

    def _skip_reduce(self, blk, da, st):
        """srvp_bn_bwd_reduce da_mode 3: the skip-connection term of a pooled stage whose arg-max terms ride its consumer's data gradient"""
        d3 = L.BnBwdDesc.from_buffer_copy(self._bnbwd_desc(blk, da))       # (a copy: the cached descriptor serves _bn_backward unchanged)
        d3.da_mode, d3.N, d3.da2_idx = 3, da['da2'].shape[0], L.ptr(da['da2_sel'])     # rows of da2 and the frame each belongs to
        L.call('srvp_bn_bwd_reduce', C.byref(d3), L.ptr(blk.red), st)

    def backward(self, x, d_hx, skip_grads, params, grads, st, sync=None, side=None, aux=None, on_part=None):
        """
        on_part (callable, optional): on_part(lo, hi) is called, with `side` current, right after the weight gradients of blocks
        lo <= index < hi have been unpacked there (every parameter gradient of those blocks is then final behind that point of `side`: the
        data-parallel exchange of that slice may start while the first stages are still going backward).
        d_hx: fp32 [N][nh_padded] gradient of the encoder output; skip_grads: {stage: (dsel bf16 [B][H][W][C], idx int32 [N])}
        side (a torch stream, optional): the unpacking of the MFMA layers' weight gradients runs there, under the image-side layer's
        backward (its BatchNorm passes and weight gradient, ~1 ms that needs none of it); the caller joins the stream afterwards.
        aux (a torch stream, optional): the skip-connection terms of the pooled stages' BatchNorm-backward sums (da_mode 3) are formed
        there at once -- everything they read exists when this function is entered -- instead of in line on the main stream, where each
        of these 40 us reductions waited 0.25-0.33 ms for workgroup slots beside the second stream's weight gradients.
        """
        nb = len(self.blocks)
        self.zero_backward_accumulators()
        da = dict(t=d_hx, mode=0, cstride=d_hx.shape[1], coff=0, border=0, f32=True)
        unpacked = False
        wg_on_side = side is not None and self.N <= ENC_WGRAD_SIDE_MAXN
        split_at = min(UNPACK_SPLIT, nb - 1) if (wg_on_side and nb > UNPACK_SPLIT > 1 and self.blocks[0].role == 'in') else 0
        skip_red = None
        if aux is not None and skip_grads:
            todo = [(i, b) for i, b in enumerate(self.blocks) if b.pool is not None and getattr(b, '_reduce_fused', False)
                    and b.has_bn and b.spec['skip_out'] is not None and (3 - b.spec['skip_out']) in skip_grads and i + 1 < nb]
            if todo:
                ev = L.record()                                               # accumulators cleared, skip gradients written
                with L.on_stream(aux):
                    aux.wait_event(ev)
                    for i, b in todo:
                        q = self.blocks[i + 1]
                        self._skip_reduce(b, dict(t=q.dcat, mode=2, cstride=q.dcat_c, coff=0, border=0,
                                                  **self._skip_term(b, skip_grads[3 - b.spec['skip_out']])), L.stream())
                    skip_red = L.record()
                self._skip_red_blocks = {id(b) for _, b in todo}
        for i in range(nb - 1, -1, -1):
            blk = self.blocks[i]
            if blk.role == 'in' and side is not None and nb > 1:
                ev = L.record()
                with L.on_stream(side):
                    side.wait_event(ev)
                    self.unpack_wgrads(grads, L.stream(), part=(0, split_at) if split_at else None)
                unpacked = True
            if split_at and i == split_at - 1 and wg_on_side:
                # (round 5) every weight gradient of blocks >= split_at has been issued on `side`: their unpacking (97 % of the encoder's
                # weights: 85 us that ran as the step's tail behind the LAST weight gradient) goes out now, in front of the first blocks'
                with L.on_stream(side):
                    self.unpack_wgrads(grads, L.stream(), part=(split_at, nb))
                    if on_part is not None:
                        on_part(split_at, nb)
            sk = blk.spec['skip_out']
            if sk is not None and skip_grads and (3 - sk) in skip_grads:
                da.update(self._skip_term(blk, skip_grads[3 - sk]))
                if skip_red is not None and id(blk) in self._skip_red_blocks:
                    L.wait(skip_red)      # (its da_mode 3 term was accumulated on `aux`)
                    da['da2_reduced'] = True
            if blk.role == 'in':
                w = blk.spec['key'] + '.weight'
                if self._bn_backward(blk, params, grads, da, st, sync, in_wgrad=(x, grads[w])):
                    continue
                L.call('srvp_conv_in_wgrad_f32' if blk.f32 else 'srvp_conv_in_wgrad', L.ptr(x), L.ptr(blk.draw), L.ptr(grads[w]), blk.N, blk.cin_r[0], 64, 64,
                       blk.cout, blk.cout_r, blk.k, blk.s, blk.p, st)
            else:
                self._bn_backward(blk, params, grads, da, st, sync)
                if side is not None and self.N <= ENC_WGRAD_SIDE_MAXN:
                    if ENC_WGRAD_AFTER_DGRAD:
                        # A/B (round 5): the block's weight gradient released only when its data gradient is done, so that it
                        # runs beside the HBM-bound BatchNorm pass of the block below instead of beside its own MFMA-bound data gradient
                        self._mfma_backward(blk, grads, st, wgrad=False)
                    ev = L.record()
                    with L.on_stream(side):
                        side.wait_event(ev)
                        self._wgrad(blk, L.stream())
                    if not ENC_WGRAD_AFTER_DGRAD:
                        self._mfma_backward(blk, grads, st, wgrad=False)
                else:
                    self._mfma_backward(blk, grads, st)
                pooled = blk.spec['pre'] == 'pool'
                da = dict(t=blk.dcat, mode=2 if pooled else 0, cstride=blk.dcat_c, coff=0, border=0)
        if not unpacked:
            self.unpack_wgrads(grads, st)


class DecoderNet(ConvNetBase):
    """conv.py:249-275 on N = nt*B latent rows: z (N, nz) -> x_ fp32 (N, C, 64, 64); skips come from an EncoderNet."""

    def __init__(self, specs, N, device, training, skip_feats=None, skip_map=None, skip_sel=None, f32=False):
        """skip_map: int32 [N] frame -> image of the skip tensors; skip_sel: int32 [B] sample -> image (enables the
        hoisted skip half; frames are ordered t*B + b)."""
        self.N, self.dev, self.training, self.f32 = N, device, training, bool(f32)
        adt = torch.float32 if f32 else torch.bfloat16
        self.blocks = []
        z_r = specs[0]['cin']
        self.z = Feat(N, 1, 1, z_r, device, b=0, dtype=adt)
        cur, ups = self.z, False
        for i, sp in enumerate(specs):
            last = i == len(specs) - 1
            srcs = [cur]
            if sp['cat'] is not None:
                srcs.append(skip_feats[sp['cat']])
            blk = Block(sp, 'out' if last else 'mfma', srcs, ups, N, device, training,
                        skip_map=skip_map if sp['cat'] is not None else None,
                        skip_sel=skip_sel if sp['cat'] is not None else None, f32=f32)
            if not last:
                blk.out = Feat(N, blk.OH, blk.OW, blk.cout_r, device, dtype=adt)
                cur, ups = blk.out, sp['post_up']
            self.blocks.append(blk)
        self._pool_accumulators()
        for blk in self.blocks:
            blk._fwd = blk.fwd_descs()
            if training:
                blk._dg = blk.dgrad_descs()
                blk._wg = blk.wgrad_desc()
            else:
                self._fold_eval_bn(blk)
        self._fuse_bn_reduce()
        ob = self.blocks[-1]
        self.nc = ob.cout_r
        self.x_out = ob.x_out

    def precompute_skips(self, st, cat=None):
        """conv_s(skip) of every hoisted-skip block (the first descriptor of its forward): it depends on the encoder's skip
        tensors only, so the caller may run it on a second stream under the latency-bound latent forward; the next forward()
        then skips those launches.  cat: the blocks reading skip connection `cat` only (the caller goes through all of them)."""
        for blk in self.blocks:
            if blk.split and (cat is None or blk.spec['cat'] == cat):
                L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)
        if cat is None:
            self._skips_done = True

    def forward(self, z_f32, params, st, sync=None, latent=None, coeffs_current=False):
        """z_f32: fp32 [N][nz_real], or None with latent = (w fp32 [B][nh], y fp32 base pointer tensor, y_tstride, nt, B, nh, ny): the rows
        [w[b] | y[t][b]] are then assembled by the library (srvp.py:216-221) straight into the padded decoder input.
        coeffs_current (inference): the eval-mode BatchNorm coefficients of the folded blocks were formed by an earlier call on the same
        parameters (the later chunks of one model.sample call) -- not recomputed."""
        if latent is not None:
            w, y, y_ts, nt, B, nh, ny = latent
            assert nt * B == self.N
            L.call('srvp_latent_to_z', L.ptr(w), L.ptr(y), y_ts, L.ptr(self.z.t), nt, B, nh, ny, self.z.C, 1 if self.f32 else 0, st)
        else:
            L.call('srvp_pad_f32' if self.f32 else 'srvp_cast_f32_bf16', L.ptr(z_f32), L.ptr(self.z.t), self.N, z_f32.shape[1], self.z.C, st)
        if self.training:
            self.zero_forward_accumulators()
        for blk in self.blocks[:-1]:
            if getattr(blk, '_ep', False) and not coeffs_current:     # inference: every folded block's coefficients up front, off the conv chain
                self._eval_coeffs(blk, params, st)
        for blk in self.blocks[:-1]:
            self._block_forward(blk, params, st, sync)
        self._skips_done = False
        ob = self.blocks[-1]
        if getattr(ob, 'stream_out', False):
            # image-side output layer on the streaming kernel: rolling LDS row window, packed-bf16 dot products, sigmoid + fp32 frames
            L.call('srvp_conv_out_fwd', L.ptr(ob.srcs[0].t), L.ptr(ob.wt_o), L.ptr(self.x_out), self.N, ob.cout_r, 1, st)
            return self.x_out
        if getattr(ob, 'stream_up_out', False):
            # DCGAN output layer (conv.py:304-305: ConvTranspose2d(64, nc, 4, 2, 1) + sigmoid) on its streaming kernel: 4 nc (phase, channel)
            # columns of a 16-wide MFMA tile instead of four phase convolutions with nc padded to 32 (0.31 -> 0.07 ms at 1920 frames)
            L.call('srvp_conv_up_out_fwd', L.ptr(ob.srcs[0].t), L.ptr(params[ob.spec['key'] + '.weight']), L.ptr(self.x_out), self.N, ob.cout_r, 1, st)
            return self.x_out
        # (other geometries: MFMA conv with Cout padded to 32, sigmoid + fp32 frame store in the epilogue)
        if ob.geom == 'up' and len(ob._fwd) == 4 and PHASES_ONE_GRID:
            # DCGAN output layer (conv.py:304: ConvTranspose2d(64, nc, 4, 2, 1)): its four output phases as one grid -- the phases of a tile run
            # back to back, so the 64-channel activation is read from HBM once instead of four times (4 x 94 us for 4 GFLOP at 1920 frames)
            arr = ob.__dict__.get('_fwd_arr')
            if arr is None:
                arr = ob._fwd_arr = (L.ConvDesc * 4)(*ob._fwd)
            L.call('srvp_conv_mfma_multi', arr, 4, st)
            return self.x_out
        for d in ob._fwd:
            L.call('srvp_conv_mfma', C.byref(d), st)
        return self.x_out

    def deferred_wgrads(self, grads, st):
        """The weight gradients of a backward(..., defer_wgrad=True): nothing downstream but the optimizer needs them, so the
        caller runs them on a second stream, concurrently with the latency-bound latent backward that follows.  (With `side` given
        to backward() they were already issued there block by block; only the unpacking is left.)"""
        if not getattr(self, '_wgrads_issued', False):
            for blk in self.blocks:
                if blk is self.blocks[-1] and self._f32_out():
                    self._out_wgrad_f32(grads, st)
                else:
                    self._wgrad(blk, st)
        else:
            for fn in self._wg_deferred:                                     # (SRVP_DEC_WGRAD_EARLY_N: the blocks that were not issued early)
                fn(st)
        self._wg_deferred = []
        self._wgrads_issued = False
        lo = getattr(self, '_unpacked_from', None)                          # blocks >= lo were unpacked early (backward(on_part=...))
        self._unpacked_from = None
        self.unpack_wgrads(grads, st, part=None if lo is None else (0, lo))
        return 0, (len(self.blocks) if lo is None else lo)

    def _f32_out(self):
        """The image-side layer's gradients run on the exact-fp32 MFMA first-layer kernels (see backward)."""
        ob = self.blocks[-1]
        return self.training and (ob.k, ob.s, ob.p) in ((3, 1, 1), (4, 2, 1)) and ob.ctot in (32, 64) and len(ob.srcs) == 1 \
            and ob.cout_r in (1, 3) and ob.OH == 64

    def _out_wgrad_f32(self, grads, st):
        """dW[ci][o][kh][kw] = sum_q act[ci][q] dpre[o][q - p + k]: the first-layer weight-gradient kernel with the gradient frames
        as the image and the layer's (bordered NHWC) input activations as the output gradient -- accumulated straight into the
        ConvTranspose weight's (Cin, nc, k, k) gradient, in fp32."""
        ob = self.blocks[-1]
        f0 = ob.srcs[0]
        L.call('srvp_conv_in_wgrad_f32' if self.f32 else 'srvp_conv_in_wgrad', L.ptr(self.dpre_f32), L.ptr(f0.t), L.ptr(grads[ob.spec['key'] + '.weight']),
               self.N, ob.cout_r, 64, 64, f0.C, ob.cin_r[0], ob.k, ob.s, ob.p, st)

    def backward(self, d_x, params, grads, st, sync=None, defer_wgrad=False, side=None, on_part=None):
        """d_x: fp32 (N, C, 64, 64).  Returns dz bf16 [N][nz_padded]; skip gradients are left in the blocks' dcat.
        side (torch stream, with defer_wgrad): every block's weight gradient is issued there as soon as its output gradient exists.
        on_part (callable, optional; with the early weight gradients): the blocks whose weight gradients went out early are unpacked on `side`
        as soon as the last of them is issued and on_part(lo, hi) is called there (their parameter gradients are final behind that point:
        the data-parallel exchange of that slice starts under the rest of the decoder backward); deferred_wgrads() unpacks the rest."""
        ob = self.blocks[-1]
        self.zero_backward_accumulators()
        self._unpacked_from = None
        early = defer_wgrad and side is not None and DEC_WGRAD_EARLY
        # blocks (in backward order) whose weight gradient goes out early
        n_early = DEC_WGRAD_EARLY_N if DEC_WGRAD_EARLY_N >= 0 else (len(self.blocks) // 2 if DEC_WGRAD_EARLY_N == -2 else len(self.blocks))
        self._wg_deferred = []

        def on_side(fn):
            ev = L.record()
            with L.on_stream(side):
                side.wait_event(ev)
                fn(L.stream())
        # The data-gradient of the image-side layer contracts over nc*k*k <= 48 values per pixel: as an MFMA conv on the
        # padded bf16 gradient it wastes 10x the work.  It IS the first-layer forward kernel with the gradient frames as
        # the "image" and the ConvTranspose weight (Cin, nc, k, k) read as (O, I, k, k): exact fp32 on the matrix cores.
        # The weight gradient is the first-layer weight-gradient kernel in the same way (_out_wgrad_f32), so the padded bf16 copy
        # of the gradient frames (600 MB at the headline config) is never written.
        f32_out = self._f32_out()
        if f32_out and not hasattr(self, 'dpre_f32'):
            self.dpre_f32 = torch.empty(self.N, ob.cout_r, ob.OH, ob.OW, dtype=torch.float32, device=self.dev)
        L.call('srvp_out_dpre_f32' if (self.f32 and not f32_out) else 'srvp_out_dpre', L.ptr(self.x_out), L.ptr(d_x), None if f32_out else L.ptr(ob.draw),
               L.ptr(self.dpre_f32) if f32_out else None, self.N, ob.cout_r, ob.OH, ob.OW, ob.cout, 1, st)
        if f32_out:
            if not defer_wgrad:
                self._out_wgrad_f32(grads, st)
            elif early and n_early > 0:
                on_side(lambda s_: self._out_wgrad_f32(grads, s_))
            elif early:
                self._wg_deferred.append(lambda s_: self._out_wgrad_f32(grads, s_))
            prod = self.blocks[-2] if len(self.blocks) > 1 else None
            fuse = (BN_FUSED_REDUCE and not self.f32 and prod is not None and prod.out is ob.srcs[0] and prod.has_bn and prod.act == L.ACT_LRELU
                    and (ob.k, ob.s, ob.p) == (3, 1, 1) and tuple(prod.raw.shape) == tuple(ob.dcat.shape)
                    and bool(L.load().srvp_conv_in_fwd_bnr_ok(ob.cout_r, 64, 64, ob.ctot, ob.k, ob.s, ob.p)))
            if fuse:
                # ... with the BatchNorm-backward sums of the producer block accumulated in the same launch (_fuse_bn_reduce)
                L.call('srvp_conv_in_fwd_bnr', L.ptr(self.dpre_f32), L.ptr(params[ob.spec['key'] + '.weight']), L.ptr(ob.dcat), self.N, ob.cout_r, 64, 64,
                       ob.ctot, ob.cin_r[0], ob.k, ob.s, ob.p, L.ptr(prod.raw), L.ptr(prod.coef), L.ptr(prod.red), st)
            else:
                L.call('srvp_conv_in_fwd_f32' if self.f32 else 'srvp_conv_in_fwd', L.ptr(self.dpre_f32), L.ptr(params[ob.spec['key'] + '.weight']), L.ptr(ob.dcat), None,
                       self.N, ob.cout_r, 64, 64, ob.ctot, ob.cin_r[0], ob.k, ob.s, ob.p, st)
            if prod is not None:
                prod._reduce_fused = bool(fuse)
        else:
            if early and n_early > 0:
                on_side(lambda s_: self._wgrad(ob, s_))
            elif early:
                self._wg_deferred.append(lambda s_: self._wgrad(ob, s_))
            self._mfma_backward(ob, grads, st, wgrad=not defer_wgrad)
        nxt = ob
        for i in range(len(self.blocks) - 2, -1, -1):
            blk = self.blocks[i]
            # (a sub-pixel consumer hands back the gradient already summed over each 2x2 upsample cell)
            da = dict(t=nxt.dcat, mode=1 if (blk.spec['post_up'] and not nxt.subpix) else 0, cstride=nxt.dcat_c, coff=0, border=0)
            self._bn_backward(blk, params, grads, da, st, sync)
            if early and (len(self.blocks) - 1 - i) < n_early:
                on_side(lambda s_, blk=blk: self._wgrad(blk, s_))
                if on_part is not None and (len(self.blocks) - 1 - i) == n_early - 1 and i > 0:
                    def part(s_, lo=i):
                        self.unpack_wgrads(grads, s_, part=(lo, len(self.blocks)))
                        on_part(lo, len(self.blocks))
                    on_side(part)
                    self._unpacked_from = i
            elif early:
                self._wg_deferred.append(lambda s_, blk=blk: self._wgrad(blk, s_))
            self._mfma_backward(blk, grads, st, wgrad=not defer_wgrad)
            nxt = blk
        self._wgrads_issued = bool(early)
        if not defer_wgrad:
            self.unpack_wgrads(grads, st)
        return self.blocks[0].dcat.view(self.N, -1)

    def skip_grads(self, T, B, st):
        """{decoder skip index: bf16 [B][H*W][C] gradient wrt the skip tensor of each sample} (srvp.py:222-223: the skip
        is expanded over time, so its gradient is the sum over the time steps)."""
        out = {}
        for blk in self.blocks:
            if blk.spec['cat'] is None:
                continue
            f1 = blk.srcs[1]
            if blk.split:
                out[blk.spec['cat']] = blk.dsel.view(B, f1.H * f1.W, f1.C)
            else:
                if not hasattr(blk, 'dsel'):
                    blk.dsel = torch.empty(B, f1.H * f1.W, f1.C, dtype=blk.adt, device=self.dev)
                L.call('srvp_skip_grad_reduce_f32' if blk.f32 else 'srvp_skip_grad_reduce', L.ptr(blk.dcat), blk.ctot, blk.srcs[0].C, f1.C, f1.H * f1.W, T, B, L.ptr(blk.dsel), st)
                out[blk.spec['cat']] = blk.dsel
        return out
