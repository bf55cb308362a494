"""
GPU parity tests of the conv-block kernels at real layer widths (the golden fixtures only reach 32-channel tiles):
srvp_conv_mfma (forward + data-gradient), srvp_wgrad_mfma (transpose-read and fallback paths), BatchNorm statistics,
srvp_bn_act / srvp_bn_bwd_* incl. max-pool / upsample / skip-gradient routing, and the small-channel image layers.
Reference: torch CPU fp32 ops on the same bf16-rounded operands (so only accumulation order and the final bf16
rounding differ): tolerance = 2^-7 relative to the tensor's scale for bf16 outputs, 1e-3 for fp32 reductions.
"""
import ctypes as C
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16).float()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def make_feat(N, H, W, Cr, dev, gen, s2d=False):
    from srvp_amd.convnet import Feat
    f = Feat(N, H, W, Cr, dev, s2d=s2d)
    f.put_nhwc((torch.randn(N, H, W, Cr, generator=gen) * 0.5).to(dev))
    return f


def feat_nchw(f):
    return f.get_nhwc().permute(0, 3, 1, 2).float().cpu()


CASES = [
    # kind, k, s, p, c0r, c1r, ups, Hsrc, cout, N
    ('conv', 3, 1, 1, 64, 0, False, 16, 128, 3),       # BN=128, BK=64
    ('conv', 3, 1, 1, 64, 0, False, 32, 64, 2),        # BN=64
    ('conv', 3, 1, 1, 128, 128, True, 8, 128, 3),      # upsample + skip concat (decoder stage entry)
    ('conv', 3, 1, 1, 24, 24, True, 8, 40, 2),         # padded channels everywhere, BK=32
    ('conv', 3, 1, 1, 64, 0, True, 8, 128, 3),         # halo kernel: upsampled source, whole 16x16 images
    ('conv', 3, 1, 1, 64, 0, False, 64, 64, 1),        # halo kernel: 16x16 interior tiles of a 64x64 image
    ('conv', 3, 1, 1, 128, 0, False, 8, 64, 6),        # halo kernel: four 8x8 images per tile, ragged image count
    ('conv', 3, 1, 1, 64, 0, True, 4, 64, 5),          # halo kernel: 4x4 -> 8x8 upsampled, ragged
    ('conv', 3, 1, 1, 64, 0, False, 32, 24, 2),        # Cout padded to 32: tap-split halo wgrad
    ('conv', 3, 1, 1, 128, 0, True, 16, 20, 1),        # same, upsampled source
    ('conv', 4, 2, 1, 64, 0, False, 16, 128, 2),       # DCGAN encoder stride 2
    ('conv', 4, 1, 0, 64, 0, False, 4, 128, 5),        # encoder last_conv ("full")
    ('convT', 4, 2, 1, 64, 64, False, 8, 64, 2),       # DCGAN decoder (4 phases) with skip
    ('convT', 4, 1, 0, 50, 0, False, 1, 64, 6),        # decoder first_upconv ("expand"), odd K padded to 64
    # ---- the widths that carry the FLOPs of VGG-64 (reference module/conv.py:210-223, 335-346), halo kernels at real K
    ('conv', 3, 1, 1, 512, 0, False, 8, 512, 6),       # encoder.conv.3.2/3.3, decoder.conv.0.1: K = 4608
    ('conv', 3, 1, 1, 512, 0, False, 8, 256, 5),       # decoder.conv.0.2 (512 -> 256 @ 8x8)
    ('conv', 3, 1, 1, 256, 0, False, 16, 256, 3),      # encoder.conv.2.2/2.3, decoder.conv.1.1
    ('conv', 3, 1, 1, 256, 0, False, 16, 512, 2),      # encoder.conv.3.1 geometry (after the pool): 256 -> 512
    ('conv', 3, 1, 1, 128, 0, False, 32, 64, 2),       # decoder.conv.2.1: wgrad_halo<*,64> at K = 1152
    ('conv', 3, 1, 1, 64, 0, False, 64, 64, 2),        # encoder.conv.0.1: 64 -> 64 @ 64x64
    ('conv', 3, 1, 1, 512, 512, True, 4, 512, 4),      # decoder.conv.0.0 as ONE two-source launch (generic kernel, K = 9216)
    ('conv', 3, 1, 1, 256, 256, True, 8, 256, 3),      # decoder.conv.1.0 likewise
    ('conv', 4, 2, 1, 256, 0, False, 8, 512, 4),       # DCGAN encoder.conv.3 (K = 4096)
    ('convT', 4, 2, 1, 512, 0, False, 4, 256, 4),      # DCGAN decoder.conv.0
    ('conv', 4, 1, 0, 512, 0, False, 4, 128, 7),       # encoder.last_conv at full width (K = 8192)
    ('convT', 4, 1, 0, 306, 0, False, 1, 512, 5),      # decoder.first_upconv at full width (nh_inf + ny = 306)
    # ---- 4x4 stride-2 convolutions over a SPACE-TO-DEPTH source (11th field): forward on the halo kernel with per-phase taps, weight
    # gradient with swapped operand roles (DCGAN encoder without skip connections: config 2)
    ('conv', 4, 2, 1, 64, 0, False, 32, 128, 3, True),   # encoder.conv.1: 64 -> 128 @ 32 -> 16 (one whole 16x16 image per tile)
    ('conv', 4, 2, 1, 128, 0, False, 16, 256, 5, True),  # encoder.conv.2: 128 -> 256 @ 16 -> 8, ragged image count
    ('conv', 4, 2, 1, 256, 0, False, 8, 512, 6, True),   # encoder.conv.3: 256 -> 512 @ 8 -> 4 (K = 4096)
]


@pytest.mark.parametrize('case', CASES, ids=[f'{c[0]}{c[1]}s{c[2]}_{c[4]}+{c[5]}_{"up" if c[6] else "id"}_{c[8]}{"_s2d" if len(c) > 10 else ""}' for c in CASES])
@pytest.mark.parametrize('use_tr', [1, 0])
def test_block_conv_fwd_bwd(case, use_tr):
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block, Feat
    kind, k, s, p, c0r, c1r, ups, Hs, cout, N = case[:10]
    s2d_src = len(case) > 10 and case[10]
    dev = torch.device('cuda')
    # (hash() of a tuple holding strings changes from process to process: a fixed seed per case keeps the run reproducible)
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 1000)
    if kind == 'convT' and Hs == 1:
        f0 = Feat(N, 1, 1, c0r, dev, b=0)
        f0.interior().copy_(torch.randn(N, 1, 1, c0r, generator=g))
    else:
        f0 = make_feat(N, Hs, Hs, c0r, dev, g, s2d=s2d_src)
    srcs = [f0]
    skip_map = None
    Hin = Hs * 2 if ups else Hs
    if c1r:
        NB = 2                                           # skip tensor holds NB images, selected through the map
        f1 = make_feat(NB * 2, Hin, Hin, c1r, dev, g)
        srcs.append(f1)
        skip_map = torch.tensor([(n % NB) * 2 + 1 for n in range(N)], dtype=torch.int32, device=dev)
    spec = dict(kind=kind, key='w', bnkey='bn', cin=c0r + c1r, cout=cout, k=k, s=s, p=p, act='leaky_relu')
    blk = Block(spec, 'mfma', srcs, ups, N, dev, True, skip_map=skip_map)
    assert blk.s2d_in == bool(s2d_src)
    blk._fwd, blk._dg, blk._wg = blk.fwd_descs(), blk.dgrad_descs(), blk.wgrad_desc()
    wshape = (cout, c0r + c1r, k, k) if kind == 'conv' else (c0r + c1r, cout, k, k)
    w = (torch.randn(*wshape, generator=g) * 0.1).to(dev)
    st = L.stream()
    L.call('srvp_wgrad_set_tr', use_tr)
    blk.pack(w, st)
    blk.stats.zero_()
    for d in blk._fwd:
        L.call('srvp_conv_mfma', C.byref(d), st)
    blk.finish_fwd(st)                                   # (split-K launches: slabs -> raw + statistics)
    torch.cuda.synchronize()
    # ---- reference forward
    x0 = feat_nchw(f0)
    if ups:
        x0 = F.interpolate(x0, scale_factor=2, mode='nearest')
    xin = x0
    if c1r:
        x1 = feat_nchw(f1)[skip_map.cpu().long()]
        xin = torch.cat([x0, x1], 1)
    xin = xin.clone().requires_grad_(True)
    wr = bf(w.cpu()).clone().requires_grad_(True)
    ref = F.conv2d(xin, wr, None, s, p) if kind == 'conv' else F.conv_transpose2d(xin, wr, None, s, p)
    raw = blk.raw[..., :cout].permute(0, 3, 1, 2).float().cpu()
    assert raw.shape == ref.shape
    assert rel_err(raw, ref) < 2 ** -7, rel_err(raw, ref)
    assert blk.raw[..., cout:].abs().max().item() == 0 if blk.cout > cout else True
    s1 = ref.sum(dim=(0, 2, 3)).double()
    s2 = (ref.double() ** 2).sum(dim=(0, 2, 3))
    # (sub-pixel blocks round the FOLDED weights to bf16, the reference rounds each weight: a systematic 2^-9-level
    # difference of the outputs that does not average out in the sums)
    stol = 4e-3 if blk.subpix else 1e-3
    assert rel_err(blk.stats[0, :cout], s1) < stol * max(1.0, (s2.sqrt().max() / (s1.abs().max() + 1e-9)).item())
    assert rel_err(blk.stats[1, :cout], s2) < stol
    # ---- backward
    dr = torch.randn(N, blk.OH, blk.OW, cout, generator=g) * 0.5
    blk.put_draw(dr.to(dev))                             # (bordered NHWC, or space-to-depth for the sub-pixel halo backward)
    dr_ref = blk.get_draw()[..., :cout].permute(0, 3, 1, 2).float().cpu()
    ref.backward(dr_ref)
    grads = {'w.weight': torch.zeros_like(w)}
    blk.dw.zero_()
    for d in (blk._wg if isinstance(blk._wg, list) else [blk._wg]):
        L.call('srvp_wgrad_mfma', C.byref(d), st)
    L.call('srvp_unpack_wgrad', L.ptr(blk.dw), L.ptr(grads['w.weight']), C.byref(blk.pu), st)
    for d in blk._dg:
        L.call('srvp_conv_mfma', C.byref(d), st)
    blk.finish_dgrad(st)
    torch.cuda.synchronize()
    assert rel_err(grads['w.weight'], wr.grad) < 2e-3, rel_err(grads['w.weight'], wr.grad)
    dcat = blk.dcat.float().cpu()                       # [N][Hin][Win][ctot]
    c0p = f0.C
    d0 = dcat[..., :c0r].permute(0, 3, 1, 2)
    g0 = xin.grad[:, :c0r]
    if blk.subpix:
        g0 = F.avg_pool2d(g0, 2) * 4                    # sub-pixel blocks return the gradient wrt the low-res source
    assert rel_err(d0, g0) < 2 ** -7, rel_err(d0, g0)
    if c1r:
        d1 = dcat[..., c0p:c0p + c1r].permute(0, 3, 1, 2)
        assert rel_err(d1, xin.grad[:, c0r:]) < 2 ** -7


HALO_CASES = [
    # c0r, cout, Hs, ups, N, role
    (64, 64, 64, False, 2, 'mfma'), (128, 128, 32, False, 3, 'mfma'), (64, 256, 16, False, 5, 'mfma'),
    (128, 64, 8, False, 6, 'mfma'), (192, 96, 8, False, 7, 'mfma'),
    (64, 128, 4, True, 5, 'mfma'), (64, 64, 8, True, 3, 'mfma'), (128, 64, 16, True, 2, 'mfma'), (64, 64, 32, True, 1, 'mfma'),
    (64, 3, 64, False, 2, 'out'), (64, 1, 32, False, 3, 'out'),
]


@pytest.mark.parametrize('case', HALO_CASES, ids=[f'{c[0]}to{c[1]}_{c[2]}{"up" if c[3] else ""}_n{c[4]}_{c[5]}' for c in HALO_CASES])
def test_conv_halo_matches_generic(case, monkeypatch):
    """The halo-tiled 3x3 kernel and the generic tap-gather kernel accumulate in the same order: bit-identical raw
    outputs / frames / data-gradients; BN statistics equal up to the fp32 per-tile partial sums."""
    from srvp_amd import _lib as L
    from srvp_amd import convnet
    from srvp_amd.convnet import Block
    monkeypatch.setattr(convnet, 'S2D', False)           # (the space-to-depth sub-pixel backward has no generic-kernel twin)
    c0r, cout, Hs, ups, N, role = case
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(11)
    f0 = make_feat(N, Hs, Hs, c0r, dev, g)
    kind = 'convT' if role == 'out' else 'conv'
    spec = dict(kind=kind, key='w', bnkey=None if role == 'out' else 'bn', cin=c0r, cout=cout, k=3, s=1, p=1,
                act='sigmoid' if role == 'out' else 'leaky_relu')
    blk = Block(spec, role, [f0], ups, N, dev, True)
    wshape = (cout, c0r, 3, 3) if kind == 'conv' else (c0r, cout, 3, 3)
    w = (torch.randn(*wshape, generator=g) * 0.1).to(dev)
    st = L.stream()
    bd = blk.draw_b
    blk.draw[:, bd:bd + blk.OH, bd:bd + blk.OW, :cout].copy_(torch.randn(N, blk.OH, blk.OW, cout, generator=g) * 0.5)
    res = {}
    try:
        for halo in (1, 0):
            L.call('srvp_conv_set_halo', halo)
            blk._fwd, blk._dg = blk.fwd_descs(), blk.dgrad_descs()      # the weight layout follows the kernel choice
            blk.pack(w, st)
            out = blk.x_out if role == 'out' else blk.raw
            out.fill_(7.0)
            blk.dcat.fill_(7.0)
            if role != 'out':
                blk.stats.zero_()
            for d in blk._fwd + blk._dg:
                L.call('srvp_conv_mfma', C.byref(d), st)
            torch.cuda.synchronize()
            res[halo] = (out.clone(), blk.dcat.clone(), blk.stats.clone() if role != 'out' else None)
    finally:
        L.call('srvp_conv_set_halo', 1)
    if blk.cout % 64 == 0:
        assert torch.equal(res[1][0], res[0][0])
    else:
        # Cout = 32 (mod 64) runs the generic kernel with 32-channel K chunks: different fp32 summation order
        assert rel_err(res[1][0].float(), res[0][0].float()) < (1e-5 if role == 'out' else 2 ** -7)
    assert torch.equal(res[1][1], res[0][1])
    assert res[1][0].float().abs().max().item() not in (0.0, 7.0)
    if role != 'out':
        assert rel_err(res[1][2], res[0][2]) < 1e-6          # per-tile partial sums are fp32, tile shapes differ


def test_conv_halo_split_skip(monkeypatch):
    """Hoisted skip half through the halo kernel (map0 sample indirection, fp32 S output, S added in the epilogue)."""
    from srvp_amd import _lib as L
    from srvp_amd import convnet
    from srvp_amd.convnet import Block
    monkeypatch.setattr(convnet, 'S2D', False)
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(12)
    T, B, Hs = 3, 2, 8
    N = T * B
    f0 = make_feat(N, Hs, Hs, 64, dev, g)
    f1 = make_feat(5, 2 * Hs, 2 * Hs, 128, dev, g)
    sel = torch.tensor([3, 1], dtype=torch.int32, device=dev)
    smap = sel.repeat(T)
    spec = dict(kind='conv', key='w', bnkey='bn', cin=192, cout=128, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0, f1], True, N, dev, True, skip_map=smap, skip_sel=sel)
    assert blk.split
    w = (torch.randn(128, 192, 3, 3, generator=g) * 0.1).to(dev)
    st = L.stream()
    blk.draw[:, 1:-1, 1:-1].copy_(torch.randn(N, 16, 16, 128, generator=g) * 0.5)
    blk.draw_sum[:, 1:-1, 1:-1].copy_(torch.randn(B, 16, 16, 128, generator=g) * 0.5)
    res = {}
    try:
        for halo in (1, 0):
            L.call('srvp_conv_set_halo', halo)
            blk._fwd, blk._dg = blk.fwd_descs(), blk.dgrad_descs()
            blk.pack(w, st)
            blk.raw.fill_(7.0); blk.S.fill_(7.0); blk.dcat.fill_(7.0); blk.dsel.fill_(7.0)
            blk.stats.zero_()
            for d in blk._fwd + blk._dg:
                L.call('srvp_conv_mfma', C.byref(d), st)
            torch.cuda.synchronize()
            res[halo] = [t.clone() for t in (blk.raw, blk.S, blk.dcat, blk.dsel)]
    finally:
        L.call('srvp_conv_set_halo', 1)
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b)
    # and against torch on the concatenated input
    x0 = F.interpolate(feat_nchw(f0), scale_factor=2, mode='nearest')
    xin = torch.cat([x0, feat_nchw(f1)[smap.cpu().long()]], 1)
    ref = F.conv2d(xin, bf(w.cpu()), None, 1, 1)
    assert rel_err(blk.raw.permute(0, 3, 1, 2).float().cpu(), ref) < 2 ** -7


@pytest.mark.parametrize('N', [96, 300, 1030])
def test_conv_stream64_matches_tile_kernel(N):
    """csrc/conv_stream.hip (64 -> 64 channels at 64x64: persistent workgroups, rolling LDS row window, register-resident weights) against
    the tile kernels on the same launches: forward with BatchNorm statistics, and the data gradient with the producer's fused
    BatchNorm-backward sums (srvp_conv_desc.bnr_*).  Same products, another fp32 summation order: outputs agree to a bf16 ulp on a few
    elements, the statistics to 1e-6.  N = 96 / 300 / 1030 frames: quarter / quarter / whole images per work item."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(17)
    f0 = make_feat(N, 64, 64, 64, dev, g)
    spec = dict(kind='conv', key='w', bnkey='bn', cin=64, cout=64, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0], False, N, dev, True)
    blk._fwd, blk._dg = blk.fwd_descs(), blk.dgrad_descs()
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    st = L.stream()
    blk.pack(w, st)
    blk.draw[:, 1:-1, 1:-1, :].copy_((torch.randn(N, 64, 64, 64, generator=g) * 0.5).to(torch.bfloat16))
    # a producer layer for the fused reduction: its raw output and coefficients
    praw = (torch.randn(N, 64, 64, 64, generator=g)).to(torch.bfloat16).to(dev)
    coef = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1,
                        torch.rand(64, generator=g) + 0.5]).to(dev).contiguous()
    d = blk._dg[0]
    res = {}
    try:
        for on in (1, 0):
            L.call('srvp_conv_set_stream64', on)
            red = torch.zeros(2, 64, dtype=torch.float64, device=dev)
            d.bnr_raw, d.bnr_coef, d.bnr_red = L.ptr(praw), L.ptr(coef), L.ptr(red)
            blk.raw.fill_(7.0); blk.dcat.fill_(7.0); blk.stats.zero_()
            L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)
            L.call('srvp_conv_mfma', C.byref(d), st)
            torch.cuda.synchronize()
            res[on] = (blk.raw.float().clone(), blk.dcat.float().clone(), blk.stats.clone(), red.clone())
    finally:
        L.call('srvp_conv_set_stream64', 1)
        d.bnr_raw, d.bnr_coef, d.bnr_red = None, None, None
    for i, name in ((0, 'raw'), (1, 'dcat')):
        a, b = res[1][i], res[0][i]
        diff = (a - b).abs()
        assert diff.max().item() <= 2 ** -7 * max(1.0, b.abs().max().item()), name           # at most one bf16 ulp
        assert (diff > 0).float().mean().item() < 5e-3, (name, (diff > 0).float().mean().item())
    # (the fused sums see the data gradient as stored: the few elements that round the other way move them at the 1e-5 level)
    assert rel_err(res[1][2], res[0][2]) < 1e-6 and rel_err(res[1][3], res[0][3]) < 2e-4
    # and against torch on the bf16-rounded operands (forward)
    ref = F.conv2d(feat_nchw(f0)[:4], bf(w.cpu()), None, 1, 1)
    assert rel_err(res[1][0][:4].permute(0, 3, 1, 2).cpu(), ref) < 2 ** -7


def test_conv_stream64_vs_torch_autograd_96_frames():
    """the streaming kernels are only reached from 96 frames on, so no reference-sized fixture used to pass through
    them and their data gradient / fused BatchNorm-backward sums were compared with the tile kernels only.  Here conv_stream64_kernel<false>
    (forward + BatchNorm statistics) and <true> (data gradient + the producer's fused BatchNorm-backward sums) at N = 96 are held directly
    against torch CPU autograd on ALL frames (reference module/conv.py:200-203: Conv2d(64, 64, 3, 1, 1) -> BatchNorm2d -> LeakyReLU)."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(171)
    N = 96
    f0 = make_feat(N, 64, 64, 64, dev, g)
    spec = dict(kind='conv', key='w', bnkey='bn', cin=64, cout=64, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0], False, N, dev, True)
    blk._fwd, blk._dg = blk.fwd_descs(), blk.dgrad_descs()
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    st = L.stream()
    blk.pack(w, st)
    dr = (torch.randn(N, 64, 64, 64, generator=g) * 0.5).to(torch.bfloat16)
    blk.draw[:, 1:-1, 1:-1, :].copy_(dr)
    # the producer layer of the fused reduction: raw output + (scale, shift, mean, inverse std); some negative scales, |mean| up to 3 std
    praw = (torch.randn(N, 64, 64, 64, generator=g) + torch.randn(64, generator=g) * 2.0).to(torch.bfloat16)
    coef = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 2.0,
                        torch.rand(64, generator=g) + 0.5]).contiguous()
    coef[0, ::7] *= -1.0
    praw_d, coef_d = praw.to(dev), coef.to(dev)
    red = torch.zeros(2, 64, dtype=torch.float64, device=dev)
    d = blk._dg[0]
    try:
        d.bnr_raw, d.bnr_coef, d.bnr_red = L.ptr(praw_d), L.ptr(coef_d), L.ptr(red)
        blk.raw.fill_(7.0); blk.dcat.fill_(7.0); blk.stats.zero_()
        cnt = [L.load().srvp_conv_stream_count(i) for i in (0, 1)]
        L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)
        L.call('srvp_conv_mfma', C.byref(d), st)
        torch.cuda.synchronize()
        assert [L.load().srvp_conv_stream_count(i) for i in (0, 1)] == [cnt[0] + 1, cnt[1] + 1]      # both ran on conv_stream64_kernel
    finally:
        d.bnr_raw, d.bnr_coef, d.bnr_red = None, None, None
    torch.set_num_threads(8)
    xin = feat_nchw(f0).clone().requires_grad_(True)
    ref = F.conv2d(xin, bf(w.cpu()), None, 1, 1)
    raw = blk.raw.permute(0, 3, 1, 2).float().cpu()
    assert rel_err(raw, ref) < 2 ** -7, rel_err(raw, ref)
    s1, s2 = ref.double().sum(dim=(0, 2, 3)), (ref.double() ** 2).sum(dim=(0, 2, 3))
    assert rel_err(blk.stats[0], s1) < 1e-4 * max(1.0, (s2.sqrt().max() / (s1.abs().max() + 1e-9)).item()) and rel_err(blk.stats[1], s2) < 1e-4
    ref.backward(dr.float().permute(0, 3, 1, 2))
    dcat = blk.dcat.float().cpu()
    assert rel_err(dcat.permute(0, 3, 1, 2), xin.grad) < 2 ** -7, rel_err(dcat.permute(0, 3, 1, 2), xin.grad)
    # fused sums (reference: autograd through BatchNorm + LeakyReLU of the producer): g = dA * act'(scale raw + shift);
    # red[0] = sum g, red[1] = sum g (raw - mean) inv_std
    pr = praw.double()
    gate = torch.where(torch.addcmul(coef[1], praw.float(), coef[0]) > 0, 1.0, 0.2).double()          # fp32 fma, as the kernels evaluate it

    def sums(dA):
        gg = dA.double() * gate
        return torch.stack([gg.sum(dim=(0, 1, 2)), (gg * (pr - coef[2].double())).sum(dim=(0, 1, 2)) * coef[3].double()])
    # (a) from the data gradient AS STORED by the kernel: only the summation differs
    want = sums(dcat)
    scale = sums(dcat.abs()).abs()                       # cancellation-free size of each sum
    assert ((red.cpu() - want).abs() / scale).max().item() < 1e-6, ((red.cpu() - want).abs() / scale).max().item()
    # (b) from torch's own fp32 data gradient (bf16 rounding of 393 216 elements per channel does not average out below 2^-9 / sqrt(n))
    want_t = sums(xin.grad.permute(0, 2, 3, 1))
    assert ((red.cpu() - want_t).abs() / scale).max().item() < 1e-4, ((red.cpu() - want_t).abs() / scale).max().item()


def test_conv_stream_sub64_vs_torch_96_frames():
    """conv_stream_sub64_kernel (decoder.conv.3.0 forward: nearest x2 upsample + concat(skip) + 3x3 conv, reference module/conv.py:270,331-349) at
    T x B = 12 x 8 = 96 frames against torch CPU on ALL frames, with the BatchNorm statistics."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(291)
    T, B = 12, 8
    N = T * B
    f0 = make_feat(N, 32, 32, 64, dev, g)
    f1 = make_feat(B + 2, 64, 64, 64, dev, g)
    sel = torch.tensor([(3 * b + 1) % (B + 2) for b in range(B)], dtype=torch.int32, device=dev)
    smap = sel.repeat(T)
    spec = dict(kind='conv', key='w', bnkey='bn', cin=128, cout=64, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0, f1], True, N, dev, True, skip_map=smap, skip_sel=sel)
    assert blk.split and blk.subpix
    blk._fwd = blk.fwd_descs()
    w = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).to(dev)
    st = L.stream()
    blk.pack(w, st)
    L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)                   # conv_s(skip) -> S
    arr = (L.ConvDesc * 4)(*blk._fwd[-4:])
    blk.raw.fill_(7.0); blk.stats.zero_()
    cnt = L.load().srvp_conv_stream_count(2)
    L.call('srvp_conv_mfma_multi', arr, 4, st)
    torch.cuda.synchronize()
    assert L.load().srvp_conv_stream_count(2) == cnt + 1                 # ran on conv_stream_sub64_kernel
    torch.set_num_threads(8)
    x0 = F.interpolate(feat_nchw(f0), scale_factor=2, mode='nearest')
    xin = torch.cat([x0, feat_nchw(f1)[smap.cpu().long()]], 1)
    ref = F.conv2d(xin, bf(w.cpu()), None, 1, 1)
    raw = blk.raw.permute(0, 3, 1, 2).float().cpu()
    # (the sub-pixel form rounds the FOLDED weights to bf16: a systematic 2^-9-level difference -- same bounds as test_block_conv_fwd_bwd)
    assert rel_err(raw, ref) < 2 ** -7, rel_err(raw, ref)
    s1, s2 = ref.double().sum(dim=(0, 2, 3)), (ref.double() ** 2).sum(dim=(0, 2, 3))
    assert rel_err(blk.stats[0], s1) < 4e-3 * max(1.0, (s2.sqrt().max() / (s1.abs().max() + 1e-9)).item()) and rel_err(blk.stats[1], s2) < 4e-3


@pytest.mark.parametrize('N,nc', [(5, 3), (600, 3), (1100, 3), (37, 1)])
def test_conv_in_stream_matches_tile_kernel(N, nc):
    """csrc/conv_in_stream.hip (image-side 3x3 layer, nc -> 64 channels on 64x64 fp32 frames: persistent workgroup per CU, frame records of bf16
    high + low parts, weights split the same way and register-resident, stores straight from the accumulators) against the exact-fp32 tile
    kernel on the same launches: the plain forward with BatchNorm statistics (encoder.conv.0.0) and the data gradient of the output layer with
    the producer's fused BatchNorm-backward sums (srvp_conv_in_fwd_bnr).  The split drops terms of 2^-16 relative size: the stored bf16
    outputs differ from the exact kernel's by one ulp on a few elements per thousand.  N = 5 / 37, 600, 1100 frames: quarter, half, whole
    frames per work item."""
    from srvp_amd import _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(41)
    x = torch.rand(N, nc, 64, 64, generator=g)
    x[0, :, :2] = torch.randn(nc, 2, 64, generator=g) * 3.0          # (gradient frames are signed)
    w = torch.randn(64, nc, 3, 3, generator=g) * 0.2
    xd, wd = x.to(dev), w.to(dev)
    praw = torch.randn(N, 64, 64, 64, generator=g).to(torch.bfloat16).to(dev)
    coef = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1,
                        torch.rand(64, generator=g) + 0.5]).to(dev).contiguous()
    coef[0, ::5] *= -1.0
    st = L.stream()
    res = {}
    try:
        for on in (1, 0):
            L.call('srvp_conv_set_in_stream', on)
            raw = torch.full((N, 64, 64, 64), 7.0, dtype=torch.bfloat16, device=dev)
            stats = torch.zeros(2, 64, dtype=torch.float64, device=dev)
            L.call('srvp_conv_in_fwd', L.ptr(xd), L.ptr(wd), L.ptr(raw), L.ptr(stats), N, nc, 64, 64, 64, 64, 3, 1, 1, st)
            raw2 = torch.full((N, 64, 64, 64), 7.0, dtype=torch.bfloat16, device=dev)
            red = torch.zeros(2, 64, dtype=torch.float64, device=dev)
            L.call('srvp_conv_in_fwd_bnr', L.ptr(xd), L.ptr(wd), L.ptr(raw2), N, nc, 64, 64, 64, 64, 3, 1, 1, L.ptr(praw), L.ptr(coef), L.ptr(red), st)
            torch.cuda.synchronize()
            res[on] = (raw.float(), stats.clone(), raw2.float(), red.clone())
    finally:
        L.call('srvp_conv_set_in_stream', 1)
    for i in (0, 2):
        a, b = res[1][i], res[0][i]
        diff = (a - b).abs()
        assert diff.max().item() <= 2 ** -7 * max(1.0, b.abs().max().item()), i           # at most one bf16 ulp
        assert (diff > 0).float().mean().item() < 5e-3, (i, (diff > 0).float().mean().item())
    assert torch.equal(res[1][0], res[1][2])                                              # the two launches compute the same output
    assert rel_err(res[1][1], res[0][1]) < 2e-5                                           # (the 2^-17 split residue is systematic)
    # (the fused sums see the data gradient as stored: the few elements that round the other way move them at the 1e-5 level)
    assert rel_err(res[1][3], res[0][3]) < 2e-4
    # against torch in fp32 on the first frames, before the bf16 rounding matters: the statistics of one frame
    ref = F.conv2d(x[:2].double(), w.double(), None, 1, 1)
    assert rel_err(res[1][0][:2].permute(0, 3, 1, 2).cpu().double(), ref) < 2 ** -8
    if N <= 40:
        full = F.conv2d(x.double(), w.double(), None, 1, 1)
        assert rel_err(res[1][1][0].cpu(), full.sum(dim=(0, 2, 3))) < 1e-5
        assert rel_err(res[1][1][1].cpu(), (full ** 2).sum(dim=(0, 2, 3))) < 1e-5
        # the activation gate of the fused sums is a per-channel threshold on the producer's bf16 value there (fmaf(raw, scale, shift) > 0 in the
        # tile kernel): operands with few mantissa bits make both kernels' outputs exact and equal, the producer's values sit within +- 3 bf16
        # steps of -shift / scale, so one wrong gate would move a sum by 1e-3 of its size
        x2 = (torch.randint(-8, 9, (N, nc, 64, 64), generator=g).float() / 8).to(dev)
        w2 = (torch.randint(-4, 5, (64, nc, 3, 3), generator=g).float() / 4).to(dev)
        t0 = (-coef[1] / coef[0]).to(torch.bfloat16).float()
        ulp = torch.maximum(t0.abs(), torch.tensor(1e-30, device=dev)) * 2.0 ** -7
        praw2 = (t0 + ulp * torch.randint(-3, 4, (N, 64, 64, 64), generator=g).float().to(dev)).to(torch.bfloat16)
        out = {}
        try:
            for on in (1, 0):
                L.call('srvp_conv_set_in_stream', on)
                raw2 = torch.empty(N, 64, 64, 64, dtype=torch.bfloat16, device=dev)
                red = torch.zeros(2, 64, dtype=torch.float64, device=dev)
                L.call('srvp_conv_in_fwd_bnr', L.ptr(x2), L.ptr(w2), L.ptr(raw2), N, nc, 64, 64, 64, 64, 3, 1, 1, L.ptr(praw2), L.ptr(coef), L.ptr(red), st)
                torch.cuda.synchronize()
                out[on] = (raw2.float(), red.clone())
        finally:
            L.call('srvp_conv_set_in_stream', 1)
        assert torch.equal(out[1][0], out[0][0])
        assert rel_err(out[1][1], out[0][1]) < 2e-6, rel_err(out[1][1], out[0][1])


@pytest.mark.parametrize('T,B', [(12, 8), (6, 40)])
def test_conv_stream_sub64_matches_tile_kernel(T, B):
    """csrc/conv_stream.hip, sub-pixel stage entry at 64 channels (decoder.conv.3.0 forward: four output phases + hoisted skip half +
    statistics as ONE streaming launch, the fp32 skip tile held in registers across the T frames of a sample) against the four phase
    launches of the tile kernel: raw output within a bf16 ulp on a few elements, statistics to 1e-6; and against torch."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(29)
    N = T * B
    f0 = make_feat(N, 32, 32, 64, dev, g)
    f1 = make_feat(B + 2, 64, 64, 64, dev, g)
    sel = torch.tensor([(3 * b + 1) % (B + 2) for b in range(B)], dtype=torch.int32, device=dev)
    smap = sel.repeat(T)
    spec = dict(kind='conv', key='w', bnkey='bn', cin=128, cout=64, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0, f1], True, N, dev, True, skip_map=smap, skip_sel=sel)
    assert blk.split and blk.subpix
    blk._fwd = blk.fwd_descs()
    w = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).to(dev)
    st = L.stream()
    blk.pack(w, st)
    L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)                   # conv_s(skip) -> S
    arr = (L.ConvDesc * 4)(*blk._fwd[-4:])
    res = {}
    try:
        for on in (1, 0):
            L.call('srvp_conv_set_stream64', on)
            blk.raw.fill_(7.0); blk.stats.zero_()
            L.call('srvp_conv_mfma_multi', arr, 4, st)
            torch.cuda.synchronize()
            res[on] = (blk.raw.float().clone(), blk.stats.clone())
    finally:
        L.call('srvp_conv_set_stream64', 1)
    diff = (res[1][0] - res[0][0]).abs()
    assert diff.max().item() <= 2 ** -7 * max(1.0, res[0][0].abs().max().item())
    assert (diff > 0).float().mean().item() < 5e-3, (diff > 0).float().mean().item()
    assert rel_err(res[1][1], res[0][1]) < 1e-6
    x0 = F.interpolate(feat_nchw(f0)[:2], scale_factor=2, mode='nearest')
    xin = torch.cat([x0, feat_nchw(f1)[smap[:2].cpu().long()]], 1)
    ref = F.conv2d(xin, bf(w.cpu()), None, 1, 1)
    assert rel_err(res[1][0][:2].permute(0, 3, 1, 2).cpu(), ref) < 2 ** -7


@pytest.mark.parametrize('N,nc', [(5, 3), (300, 3), (700, 1), (1100, 3)])
def test_conv_out_stream_matches_tile_kernel(N, nc):
    """csrc/conv_out.hip (image-side output layer, 64 -> nc channels + sigmoid on a rolling LDS window, 16x16x32 MFMA) against the padded
    32-column tile kernel: fp32 frames within 2e-6."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(23)
    f0 = make_feat(N, 64, 64, 64, dev, g)
    spec = dict(kind='convT', key='w', bnkey=None, cin=64, cout=nc, k=3, s=1, p=1, act='sigmoid')
    blk = Block(spec, 'out', [f0], False, N, dev, False)
    assert blk.stream_out
    blk._fwd = blk.fwd_descs()
    w = (torch.randn(64, nc, 3, 3, generator=g) * 0.1).to(dev)
    st = L.stream()
    blk.pack(w, st)
    for d in blk._fwd:
        L.call('srvp_conv_mfma', C.byref(d), st)
    torch.cuda.synchronize()
    ref = blk.x_out.clone()
    blk.x_out.fill_(7.0)
    L.call('srvp_conv_out_fwd', L.ptr(f0.t), L.ptr(blk.wt_o), L.ptr(blk.x_out), N, nc, 1, st)
    torch.cuda.synchronize()
    assert (blk.x_out - ref).abs().max().item() < 2e-6
    t = torch.sigmoid(F.conv_transpose2d(feat_nchw(f0)[:3], bf(w.cpu()), None, 1, 1))
    assert (blk.x_out[:3].cpu() - t).abs().max().item() < 1e-5


@pytest.mark.parametrize('N,nc', [(5, 1), (300, 1), (700, 3), (1100, 1), (1920, 1)])
def test_conv_up_out_stream_matches_tile_kernel_and_torch(N, nc):
    """csrc/conv_out.hip, second kernel (round 6): the DCGAN decoder's image-side output layer -- ConvTranspose2d(64 -> nc, 4x4, stride 2,
    pad 1) + sigmoid, 32x32 -> 64x64 (reference module/conv.py:304-305) -- as a streaming kernel in sub-pixel form, against the padded
    32-column tile kernel (four phase convolutions as one grid) and, on three frames, against torch's conv_transpose2d on the CPU with
    bf16-rounded operands."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(29)
    f0 = make_feat(N, 32, 32, 64, dev, g)
    spec = dict(kind='convT', key='w', bnkey=None, cin=64, cout=nc, k=4, s=2, p=1, act='sigmoid')
    blk = Block(spec, 'out', [f0], False, N, dev, False)
    assert blk.stream_up_out and blk.geom == 'up'
    blk._fwd = blk.fwd_descs()
    w = (torch.randn(64, nc, 4, 4, generator=g) * 0.1).to(dev)
    st = L.stream()
    blk.pack(w, st)
    arr = (L.ConvDesc * 4)(*blk._fwd)
    L.call('srvp_conv_mfma_multi', arr, 4, st)
    torch.cuda.synchronize()
    ref = blk.x_out.clone()
    blk.x_out.fill_(7.0)
    L.call('srvp_conv_up_out_fwd', L.ptr(f0.t), L.ptr(w), L.ptr(blk.x_out), N, nc, 1, st)
    torch.cuda.synchronize()
    assert (blk.x_out - ref).abs().max().item() < 2e-6, (blk.x_out - ref).abs().max().item()
    t = torch.sigmoid(F.conv_transpose2d(feat_nchw(f0)[:3], bf(w.cpu()), None, 2, 1))
    assert (blk.x_out[:3].cpu() - t).abs().max().item() < 1e-5


def test_bf16_payload_casts_round_trip():
    """srvp_cast_f32_bf16 / srvp_cast_bf16_f32 (the two ends of the opt-in bf16 gradient payload, SRVP_GRAD_BF16=1): fp32 -> bf16 is torch's
    round-to-nearest-even, bf16 -> fp32 * scale is exact, at odd offsets and lengths."""
    from srvp_amd import _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1_000_003, generator=g) * torch.logspace(-6, 3, 1_000_003)).to(dev)
    st = L.stream()
    for lo, hi in ((0, 1_000_003), (1, 4098), (77777, 900001)):
        sl = x[lo:hi].contiguous()
        pay = torch.zeros(hi - lo + 8, dtype=torch.bfloat16, device=dev)[3:3 + hi - lo]
        L.call('srvp_cast_f32_bf16', L.ptr(sl), L.ptr(pay), 1, hi - lo, hi - lo, st)
        torch.cuda.synchronize()
        assert torch.equal(pay, sl.bfloat16())
        out = torch.empty(hi - lo, dtype=torch.float32, device=dev)
        L.call('srvp_cast_bf16_f32', L.ptr(pay), L.ptr(out), hi - lo, 0.5, st)
        torch.cuda.synchronize()
        assert torch.equal(out, pay.float() * 0.5)


SPLIT_CASES = [
    # c0r (low-res main input, upsampled x2), c1r (skip), Hs, cout, T, B
    (512, 512, 4, 512, 3, 2),       # decoder.conv.0.0: 1024 -> 512 @ 8x8 (split skip half + sub-pixel main half)
    (256, 256, 8, 256, 3, 2),       # decoder.conv.1.0: 512 -> 256 @ 16x16
    (128, 128, 16, 128, 2, 2),      # decoder.conv.2.0: 256 -> 128 @ 32x32
    (64, 64, 32, 64, 2, 1),         # decoder.conv.3.0: 128 -> 64 @ 64x64
]


@pytest.mark.parametrize('case', SPLIT_CASES, ids=[f'{c[0]}+{c[1]}to{c[3]}_{2 * c[2]}' for c in SPLIT_CASES])
def test_split_skip_subpixel_block_full_width(case):
    """The decoder stage-entry convolutions exactly as the training step runs them (reference module/conv.py:270,331-349 +
    module/srvp.py:222-223): hoisted skip half conv_s(skip) once per sample + sub-pixel main half on the low-resolution
    tensor, forward, data-gradients (wrt the low-resolution main input and wrt each sample's skip tensor) and both weight
    gradients, against torch autograd on cat([upsample(h_t), skip]) with bf16-rounded operands."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Block
    c0r, c1r, Hs, cout, T, B = case
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(21)
    N = T * B
    f0 = make_feat(N, Hs, Hs, c0r, dev, g)
    f1 = make_feat(B + 3, 2 * Hs, 2 * Hs, c1r, dev, g)
    sel = torch.tensor([(3 * b + 1) % (B + 3) for b in range(B)], dtype=torch.int32, device=dev)
    smap = sel.repeat(T)
    spec = dict(kind='conv', key='w', bnkey='bn', cin=c0r + c1r, cout=cout, k=3, s=1, p=1, act='leaky_relu')
    blk = Block(spec, 'mfma', [f0, f1], True, N, dev, True, skip_map=smap, skip_sel=sel)
    assert blk.split and blk.subpix
    blk._fwd, blk._dg, blk._wg = blk.fwd_descs(), blk.dgrad_descs(), blk.wgrad_desc()
    w = (torch.randn(cout, c0r + c1r, 3, 3, generator=g) * 0.05).to(dev)
    st = L.stream()
    blk.pack(w, st)
    blk.stats.zero_()
    L.call('srvp_conv_mfma', C.byref(blk._fwd[0]), st)
    L.call('srvp_conv_mfma_multi', (L.ConvDesc * 4)(*blk._fwd[-4:]), 4, st)
    torch.cuda.synchronize()
    x0 = F.interpolate(feat_nchw(f0), scale_factor=2, mode='nearest')
    x1 = feat_nchw(f1)
    x0 = x0.clone().requires_grad_(True)
    x1 = x1.clone().requires_grad_(True)
    xin = torch.cat([x0, x1[smap.cpu().long()]], 1)
    wr = bf(w.cpu()).clone().requires_grad_(True)
    ref = F.conv2d(xin, wr, None, 1, 1)
    raw = blk.raw[..., :cout].permute(0, 3, 1, 2).float().cpu()
    assert rel_err(raw, ref) < 2 ** -7, rel_err(raw, ref)
    s1, s2 = ref.sum(dim=(0, 2, 3)).double(), (ref.double() ** 2).sum(dim=(0, 2, 3))
    assert rel_err(blk.stats[0, :cout], s1) < 4e-3 * max(1.0, (s2.sqrt().max() / (s1.abs().max() + 1e-9)).item())
    assert rel_err(blk.stats[1, :cout], s2) < 4e-3
    # ---- backward: draw (per frame) and its sum over time (what srvp_bn_bwd_apply writes beside it)
    OH = 2 * Hs
    dr = torch.randn(N, OH, OH, cout, generator=g) * 0.5
    blk.put_draw(dr.to(dev))
    drf = blk.get_draw()[..., :cout].float()
    blk.draw_sum.zero_()
    blk.draw_sum[:, 1:-1, 1:-1, :cout].copy_(drf.view(T, B, OH, OH, cout).sum(0))
    ref.backward(drf.permute(0, 3, 1, 2).cpu())
    gw = torch.zeros_like(w)
    blk.dw.zero_(); blk.dw_s.zero_()
    for d in blk._wg:
        L.call('srvp_wgrad_mfma', C.byref(d), st)
    for src, dst, pd in blk.unpack_jobs(gw):
        L.call('srvp_unpack_wgrad', L.ptr(src), dst, C.byref(pd), st)
    for d in blk._dg:
        L.call('srvp_conv_mfma', C.byref(d), st)
    blk.finish_dgrad(st)
    torch.cuda.synchronize()
    # main half: exact up to the fp32 summation order; skip half: the time-summed gradient is rounded to bf16 once more
    assert rel_err(gw[:, :c0r], wr.grad[:, :c0r]) < 2e-3, rel_err(gw[:, :c0r], wr.grad[:, :c0r])
    assert rel_err(gw[:, c0r:], wr.grad[:, c0r:]) < 6e-3, rel_err(gw[:, c0r:], wr.grad[:, c0r:])
    d0 = blk.dcat[..., :c0r].permute(0, 3, 1, 2).float().cpu()
    assert rel_err(d0, F.avg_pool2d(x0.grad, 2) * 4) < 2 ** -7
    dsel = blk.dsel[..., :c1r].permute(0, 3, 1, 2).float().cpu()
    assert rel_err(dsel, x1.grad[sel.cpu().long()]) < 2 ** -6


@pytest.mark.parametrize('mode', ['plain', 'ups', 'pool', 'skip'])
@pytest.mark.parametrize('C_', [64, 40])
def test_bn_act_fwd_bwd(mode, C_):
    """bn_finalize + bn_act (+pool) forward and bn_bwd_{reduce,finalize,apply} against torch batch_norm autograd."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import Feat, cpad, BN_EPS, BN_MOMENTUM
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    N, H = 4, 8
    Cp = cpad(C_)
    raw = torch.zeros(N, H, H, Cp, dtype=torch.bfloat16, device=dev)
    raw[..., :C_] = (torch.randn(N, H, H, C_, generator=g) * 1.5 + 0.3)
    rawf = raw[..., :C_].permute(0, 3, 1, 2).float().cpu().requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C_, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C_, generator=g)).to(dev)
    rm, rv = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    stats = torch.zeros(2, Cp, dtype=torch.float64, device=dev)
    stats[0, :C_] = raw[..., :C_].double().sum(dim=(0, 1, 2))
    stats[1, :C_] = (raw[..., :C_].double() ** 2).sum(dim=(0, 1, 2))
    coef = torch.zeros(4, Cp, device=dev)
    st = L.stream()
    cnt = float(N * H * H)
    L.call('srvp_bn_finalize', L.ptr(stats), cnt, L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
           L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]), Cp, C_, BN_EPS, BN_MOMENTUM, st)
    out = Feat(N, H, H, C_, dev)
    pool = Feat(N, H // 2, H // 2, C_, dev) if mode == 'pool' else None
    L.call('srvp_bn_act', L.ptr(raw), L.ptr(coef[0]), L.ptr(coef[1]), L.ACT_LRELU, N, H, H, Cp, L.ptr(out.t), 1,
           L.ptr(pool.t) if pool else None, 1, None, st)
    torch.cuda.synchronize()
    rm_ref, rv_ref = torch.zeros(C_), torch.ones(C_)
    y = F.batch_norm(rawf, rm_ref, rv_ref, gamma.cpu(), beta.cpu(), True, BN_MOMENTUM, BN_EPS)
    a = F.leaky_relu(y, 0.2)
    assert rel_err(feat_nchw(out), a) < 2 ** -7
    assert rel_err(rm, rm_ref) < 1e-4 and rel_err(rv, rv_ref) < 1e-4 and int(nbt) == 1
    assert out.t[:, 0].abs().max().item() == 0 and out.t[:, :, 0].abs().max().item() == 0      # border stays zero
    # ---- backward: build dA in the requested form
    d = L.BnBwdDesc()
    d.raw, d.act, d.act_border = L.ptr(raw), L.ptr(out.t), 1
    d.scale, d.shift, d.mean, d.invstd, d.act_kind = L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]), L.ACT_LRELU
    d.N, d.H, d.W, d.C = N, H, H, Cp
    d.da_border, d.da_is_f32, d.da2, d.da2_idx = 0, 0, None, None
    keep = []
    if mode in ('plain', 'skip'):
        da = torch.zeros(N, H, H, Cp + 32, dtype=torch.bfloat16, device=dev)         # channel slice of a wider tensor
        da[..., 32:32 + C_] = torch.randn(N, H, H, C_, generator=g)
        d.da, d.da_mode, d.da_cstride, d.da_coff = L.ptr(da), 0, Cp + 32, 32
        da_ref = da[..., 32:32 + C_].permute(0, 3, 1, 2).float().cpu()
        loss_in = a
        if mode == 'skip':
            da2 = torch.zeros(2, H, H, Cp, dtype=torch.bfloat16, device=dev)
            da2[..., :C_] = torch.randn(2, H, H, C_, generator=g)
            idx = torch.tensor([-1, 1, -1, 0], dtype=torch.int32, device=dev)
            d.da2, d.da2_idx = L.ptr(da2), L.ptr(idx)
            extra = torch.zeros_like(da_ref)
            extra[1] = da2[1, ..., :C_].permute(2, 0, 1).float().cpu()
            extra[3] = da2[0, ..., :C_].permute(2, 0, 1).float().cpu()
            da_ref = da_ref + extra
            keep += [da2, idx]
        (loss_in * da_ref).sum().backward()
    elif mode == 'ups':
        da = torch.zeros(N, 2 * H, 2 * H, Cp, dtype=torch.bfloat16, device=dev)
        da[..., :C_] = torch.randn(N, 2 * H, 2 * H, C_, generator=g)
        d.da, d.da_mode, d.da_cstride, d.da_coff = L.ptr(da), 1, Cp, 0
        up = F.interpolate(a, scale_factor=2, mode='nearest')
        (up * da[..., :C_].permute(0, 3, 1, 2).float().cpu()).sum().backward()
    else:
        da = torch.zeros(N, H // 2, H // 2, Cp, dtype=torch.bfloat16, device=dev)
        da[..., :C_] = torch.randn(N, H // 2, H // 2, C_, generator=g)
        d.da, d.da_mode, d.da_cstride, d.da_coff = L.ptr(da), 2, Cp, 0
        # pool on the bf16-rounded activations (what the kernel pooled), gradient routed by torch's own arg-max
        a_b = bf(a).detach()
        pooled = F.max_pool2d(a_b + (a - a.detach()), 2, 2)
        assert rel_err(feat_nchw(pool), pooled.detach()) < 2 ** -7      # the activations may differ by one bf16 ulp from torch.s own BN arithmetic
        (pooled * da[..., :C_].permute(0, 3, 1, 2).float().cpu()).sum().backward()
    red = torch.zeros(2, Cp, dtype=torch.float64, device=dev)
    bcoef = torch.zeros(3, Cp, device=dev)
    dgamma, dbeta = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    draw = torch.zeros(N, H + 2, H + 2, Cp, dtype=torch.bfloat16, device=dev)
    L.call('srvp_bn_bwd_reduce', C.byref(d), L.ptr(red), st)
    L.call('srvp_bn_bwd_finalize', L.ptr(red), cnt, L.ptr(coef[0]), L.ptr(coef[2]), L.ptr(coef[3]), L.ptr(dgamma), L.ptr(dbeta),
           L.ptr(bcoef), Cp, C_, 1, 1.0, st)
    L.call('srvp_bn_bwd_apply', C.byref(d), L.ptr(bcoef), L.ptr(draw), 1, st)
    torch.cuda.synchronize()
    got = draw[:, 1:-1, 1:-1, :C_].permute(0, 3, 1, 2).float().cpu()
    assert rel_err(got, rawf.grad) < 2 ** -6, rel_err(got, rawf.grad)
    assert draw[:, 0].abs().max().item() == 0
    # parameter gradients via a second autograd pass on gamma/beta
    gm, bt = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    y2 = F.batch_norm(rawf.detach(), None, None, gm, bt, True, BN_MOMENTUM, BN_EPS)
    a2 = F.leaky_relu(y2, 0.2)
    if mode in ('plain', 'skip'):
        (a2 * da_ref).sum().backward()
    elif mode == 'ups':
        (F.interpolate(a2, scale_factor=2, mode='nearest') * da[..., :C_].permute(0, 3, 1, 2).float().cpu()).sum().backward()
    else:
        (F.max_pool2d(bf(a2).detach() + (a2 - a2.detach()), 2, 2) * da[..., :C_].permute(0, 3, 1, 2).float().cpu()).sum().backward()
    assert rel_err(dgamma, gm.grad) < 5e-3 and rel_err(dbeta, bt.grad) < 5e-3
    # ---- the one-launch forms (srvp_bn_finalize_act, srvp_bn_bwd_finalize_apply: coefficients derived per workgroup in LDS) must
    # reproduce the separate finalize + act / finalize + apply launches BIT FOR BIT: outputs, coefficients, running statistics,
    # parameter gradients
    rm2, rv2 = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
    nbt2 = torch.zeros((), dtype=torch.int64, device=dev)
    coef2 = torch.zeros(4, Cp, device=dev)
    out2 = Feat(N, H, H, C_, dev)
    pool2 = Feat(N, H // 2, H // 2, C_, dev) if mode == 'pool' else None
    rawp = torch.zeros(N, H // 2, H // 2, Cp, dtype=torch.bfloat16, device=dev) if mode == 'pool' else None    # raw value behind every pooled activation
    L.call('srvp_bn_finalize_act', L.ptr(raw), L.ptr(stats), cnt, L.ptr(gamma), L.ptr(beta), L.ptr(rm2), L.ptr(rv2), L.ptr(nbt2),
           L.ptr(coef2[0]), L.ptr(coef2[1]), L.ptr(coef2[2]), L.ptr(coef2[3]), C_, BN_EPS, BN_MOMENTUM, L.ACT_LRELU, N, H, H, Cp,
           L.ptr(out2.t), 1, L.ptr(pool2.t) if pool2 else None, 1, None, None, L.ptr(rawp) if pool2 else None, 0, 0, st)
    draw2 = torch.zeros_like(draw)
    bcoef2 = torch.zeros(3, Cp, device=dev)
    dgamma2, dbeta2 = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    L.call('srvp_bn_bwd_finalize_apply', C.byref(d), L.ptr(red), cnt, L.ptr(dgamma2), L.ptr(dbeta2), L.ptr(bcoef2), C_, 1.0, L.ptr(draw2), 1, st)
    torch.cuda.synchronize()
    assert torch.equal(out2.t, out.t) and torch.equal(coef2, coef) and torch.equal(rm2, rm) and torch.equal(rv2, rv) and int(nbt2) == 1
    assert pool is None or torch.equal(pool2.t, pool.t)
    if pool2 is not None:
        # raw_pool: activating it again gives the pooled tensor, and it is one of the window's four raw values
        a4 = F.leaky_relu(rawp[..., :C_].float() * coef2[0, :C_] + coef2[1, :C_], 0.2).to(torch.bfloat16).float()
        diff = (a4 - pool2.interior().float()).abs()
        assert (diff > 0).float().mean().item() < 2e-3 and diff.max().item() <= 2 ** -7 * max(1.0, a4.abs().max().item())
        win = raw[..., :C_].view(N, H // 2, 2, H // 2, 2, C_).permute(0, 1, 3, 5, 2, 4).reshape(N, H // 2, H // 2, C_, 4)
        assert bool((win == rawp[..., :C_].unsqueeze(-1)).any(-1).all())
    assert torch.equal(draw2, draw) and torch.equal(bcoef2, bcoef) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta)
    del keep


def test_latent_glue_kernels():
    """srvp_latent_to_z / srvp_dz_split / srvp_rows_scatter_add_f32 (the decoder-input assembly [w | y_t] of srvp.py:216-221, its
    backward, and the backward of the row gathers) against the torch expressions they replaced, exactly."""
    from srvp_amd import _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(12)
    nt, B, nh, ny, ne = 5, 7, 24, 10, 2
    Cz = 64
    w = torch.randn(B, nh, generator=g).to(dev)
    y_all = torch.randn((nt - 1) * ne + 1, B, ny, generator=g).to(dev)
    st = L.stream()
    for dt_, f32 in ((torch.bfloat16, 0), (torch.float32, 1)):
        z = torch.full((nt * B, Cz), 7.0, dtype=dt_, device=dev)
        L.call('srvp_latent_to_z', L.ptr(w), L.ptr(y_all), ne * B * ny, L.ptr(z), nt, B, nh, ny, Cz, f32, st)
        ref = torch.zeros(nt * B, Cz, device=dev)
        ref[:, :nh + ny] = torch.cat([w.repeat(nt, 1), y_all[::ne].reshape(nt * B, ny)], 1)
        assert torch.equal(z.float(), ref.to(dt_).float())
        dz = torch.randn(nt * B, Cz, generator=g).to(dev).to(dt_)
        d_w_add, d_y_add = torch.randn(B, nh, generator=g).to(dev), torch.randn(nt, B, ny, generator=g).to(dev)
        d_w = torch.empty(B, nh, device=dev)
        d_y_all = torch.zeros_like(y_all)
        L.call('srvp_dz_split', L.ptr(dz), Cz, f32, nt, B, nh, ny, L.ptr(d_w_add), L.ptr(d_y_add), L.ptr(d_w), L.ptr(d_y_all), ne * B * ny, st)
        dzf = dz.float()
        sw = torch.zeros(B, nh, device=dev)
        for t in range(nt):                                   # the kernel's summation order (t ascending, then the addend)
            sw = sw + dzf[t * B:(t + 1) * B, :nh]
        assert torch.equal(d_w, sw + d_w_add)
        assert torch.equal(d_y_all[::ne], dzf[:, nh:nh + ny].reshape(nt, B, ny) + d_y_add)
        assert d_y_all[1::ne].abs().max().item() == 0
    dst = torch.randn(40, 16, generator=g).to(dev)
    ref = dst.clone()
    idx = torch.randperm(40, generator=g)[:9].to(dev)
    src = torch.randn(9, 16, generator=g).to(dev)
    L.call('srvp_rows_scatter_add_f32', L.ptr(dst), L.ptr(idx), 1, L.ptr(src), 9, 16, st)
    ref.index_add_(0, idx, src)
    assert torch.equal(dst, ref)
    dst2 = ref.clone()
    L.call('srvp_rows_scatter_add_f32', L.ptr(dst2), L.ptr(idx.to(torch.int32)), 0, L.ptr(src), 9, 16, st)
    assert torch.equal(dst2, ref.index_add(0, idx, src))


@pytest.mark.parametrize('kind,cin,cout,k,split', [('conv', 128, 64, 3, False), ('conv', 96, 40, 3, False), ('convT', 64, 128, 4, False),
                                                   ('conv', 64, 128, 4, False), ('conv', 128, 64, 3, True)])
def test_unpack_tiled_equals_item_path(kind, cin, cout, k, split, monkeypatch):
    """The LDS-tiled srvp_unpack_wgrad_multi path (SRVP_PACK_TILED bit 1, the default) adds exactly what the item-per-thread path adds
    to the fp32 gradient: every layout the networks use -- OIHW / IOHW, padded channel counts, two channel segments (skip concat)."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, json, torch, ctypes as C
sys.path.insert(0, {root!r})
from srvp_amd import _lib as L
from srvp_amd.convnet import Block, Feat, ConvNetBase
dev = torch.device('cuda')
g = torch.Generator().manual_seed(5)
N, Hs = 2, 8
f0 = Feat(N, Hs, Hs, {cin}, dev)
srcs = [f0]
skip_map = None
if {split}:
    f1 = Feat(2, Hs, Hs, {cin} // 2, dev); srcs.append(f1)
    skip_map = torch.zeros(N, dtype=torch.int32, device=dev)
c_tot = {cin} + ({cin} // 2 if {split} else 0)
spec = dict(kind={kind!r}, key='w', bnkey='bn', cin=c_tot, cout={cout}, k={k}, s=(2 if {k} == 4 else 1), p=1, act='leaky_relu')
blk = Block(spec, 'mfma', srcs, False, N, dev, True, skip_map=skip_map)
blk.dw.copy_(torch.randn(blk.dw.shape, generator=g))
wshape = ({cout}, c_tot, {k}, {k}) if {kind!r} == 'conv' else (c_tot, {cout}, {k}, {k})
gw = torch.randn(*wshape, generator=g).to(dev)
net = ConvNetBase(); net.dev = dev
jobs = blk.unpack_jobs(gw)
c = net._job_table(jobs, dev, dict(), True)
if c['tiles']:
    L.call('srvp_unpack_wgrad_tiles', L.ptr(c['tiles'][0]), c['tiles'][1], c['tiles'][2], L.stream())
if c['multi']:
    L.call('srvp_unpack_wgrad_multi', L.ptr(c['multi'][0]), c['multi'][1], c['multi'][2], L.stream())
torch.cuda.synchronize()
torch.save(gw.cpu(), sys.argv[1])
"""
    outs = []
    for tiles, mode in (('1', '2'), ('0', '2'), ('0', '0')):      # lean tile kernel / 8 x 64 LDS-tiled multi path / item-per-thread multi path
        path = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'unpack_{tiles}{mode}_{kind}_{cin}_{cout}_{k}_{int(split)}.pt')
        r = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, SRVP_PACK_TILES=tiles, SRVP_PACK_TILED=mode), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize('config', ['bair', 'smmnist'])
def test_big_tile_pack_unpack_equal_item_path(config):
    """The lean tile kernels (srvp_pack_weight_tiles / srvp_unpack_wgrad_tiles, the default for every eligible job) write exactly the
    bytes the item-per-thread multi paths write, for every job of a whole network at full width: VGG (3x3, folded sub-pixel tap sets,
    hoisted skip halves, fragment-major and tap-major layouts, space-to-depth phase jobs) and DCGAN (4x4 kernels, IOHW)."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ('1', '0'):
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'pack_time.py'), config], env=dict(os.environ, SRVP_PACK_TILES=mode, SRVP_PACK_TILED='0'),
                           capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['1']['dec_jobs']['tiles'] and not res['0']['dec_jobs']['tiles'], res
    for k in ('enc_pack_digest', 'dec_pack_digest', 'enc_unpack_digest', 'dec_unpack_digest'):
        assert res['1'][k] == res['0'][k], (k, res)
    rep = os.path.join(root, 'gpurun_out', 'parity_report.jsonl')
    os.makedirs(os.path.dirname(rep), exist_ok=True)
    with open(rep, 'a') as f:
        f.write(json.dumps(dict(test='pack_unpack_us', config=config, **{f'{k}_{m}': res[m][k] for m in res for k in res[m] if k.endswith('_us')})) + '\n')


@pytest.mark.parametrize('nc,k,s,p', [(3, 3, 1, 1), (1, 4, 2, 1), (3, 4, 2, 1)])
def test_image_side_layers(nc, k, s, p):
    """First encoder conv (fp32 frames in) and last decoder transposed conv + sigmoid (fp32 frames out)."""
    from srvp_amd import _lib as L
    from srvp_amd.convnet import cpad
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(11)
    N, cout_r = 3, 24
    Cp = cpad(cout_r)
    x = torch.rand(N, nc, 64, 64, generator=g)
    w = torch.randn(cout_r, nc, k, k, generator=g) * 0.2
    OH = (64 + 2 * p - k) // s + 1
    raw = torch.zeros(N, OH, OH, Cp, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(2, Cp, dtype=torch.float64, device=dev)
    st = L.stream()
    xd, wd = x.to(dev), w.to(dev)
    L.call('srvp_conv_in_fwd', L.ptr(xd), L.ptr(wd), L.ptr(raw), L.ptr(stats), N, nc, 64, 64, Cp, cout_r, k, s, p, st)
    torch.cuda.synchronize()
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, s, p)
    assert rel_err(raw[..., :cout_r].permute(0, 3, 1, 2).float(), ref) < 2 ** -7
    assert rel_err(stats[0, :cout_r], ref.sum(dim=(0, 2, 3))) < 2e-3
    assert rel_err(stats[1, :cout_r], (ref ** 2).sum(dim=(0, 2, 3))) < 1e-3
    draw = torch.zeros(N, OH + 2, OH + 2, Cp, dtype=torch.bfloat16, device=dev)
    draw[:, 1:-1, 1:-1, :cout_r] = torch.randn(N, OH, OH, cout_r, generator=g)
    dw = torch.zeros_like(wd)
    L.call('srvp_conv_in_wgrad', L.ptr(xd), L.ptr(draw), L.ptr(dw), N, nc, 64, 64, Cp, cout_r, k, s, p, st)
    torch.cuda.synchronize()
    ref.backward(draw[:, 1:-1, 1:-1, :cout_r].permute(0, 3, 1, 2).float().cpu())
    assert rel_err(dw, wr.grad) < 1e-3
    # ---- last layer: ConvTranspose2d(cin -> nc) + sigmoid on the MFMA kernel (Cout padded to 32, fp32 frame epilogue),
    # input = [main, skip] bf16 padded tensors; backward = srvp_out_dpre + the generic MFMA wgrad / dgrad
    from srvp_amd.convnet import Block
    Hin = 64 if s == 1 else 32
    c0r, c1r = 24, (24 if s == 2 else 0)
    f0 = make_feat(N, Hin, Hin, c0r, dev, g)
    srcs, mp = [f0], None
    if c1r:
        f1 = make_feat(2, Hin, Hin, c1r, dev, g)
        mp = torch.tensor([1, 0, 1], dtype=torch.int32, device=dev)
        srcs.append(f1)
    spec = dict(kind='convT', key='w', bnkey=None, cin=c0r + c1r, cout=nc, k=k, s=s, p=p, act='none')
    blk = Block(spec, 'out', srcs, False, N, dev, True, skip_map=mp)
    blk._fwd, blk._dg, blk._wg = blk.fwd_descs(), blk.dgrad_descs(), blk.wgrad_desc()
    wt = (torch.randn(c0r + c1r, nc, k, k, generator=g) * 0.2)
    wtd = wt.to(dev)
    blk.pack(wtd, st)
    for d in blk._fwd:
        L.call('srvp_conv_mfma', C.byref(d), st)
    xin = feat_nchw(f0)
    if c1r:
        xin = torch.cat([xin, feat_nchw(f1)[mp.cpu().long()]], 1)
    xin = xin.clone().requires_grad_(True)
    wtr = bf(wt).clone().requires_grad_(True)
    ref = torch.sigmoid(F.conv_transpose2d(xin, wtr, None, s, p))
    torch.cuda.synchronize()
    assert (blk.x_out.cpu() - ref).abs().max().item() < 1e-4
    dxo = torch.randn(N, nc, 64, 64, generator=g)
    ref.backward(dxo)
    dxd = dxo.to(dev)
    L.call('srvp_out_dpre', L.ptr(blk.x_out), L.ptr(dxd), L.ptr(blk.draw), None, N, nc, 64, 64, blk.cout, 1, st)
    grads = {'w.weight': torch.zeros_like(wtd)}
    blk.dw.zero_()
    L.call('srvp_wgrad_mfma', C.byref(blk._wg), st)
    L.call('srvp_unpack_wgrad', L.ptr(blk.dw), L.ptr(grads['w.weight']), C.byref(blk.pu), st)
    for d in blk._dg:
        L.call('srvp_conv_mfma', C.byref(d), st)
    blk.finish_dgrad(st)
    torch.cuda.synchronize()
    assert rel_err(grads['w.weight'], wtr.grad) < 1e-2            # dpre is rounded to bf16 before the reduction
    dact = blk.dcat
    assert rel_err(dact[..., :c0r].permute(0, 3, 1, 2).float(), xin.grad[:, :c0r]) < 2 ** -6
    if c1r:
        assert rel_err(dact[..., f0.C:f0.C + c1r].permute(0, 3, 1, 2).float(), xin.grad[:, c0r:]) < 2 ** -6


@pytest.mark.parametrize('M,C,sk,with_stats', [(2304, 128, 16, True), (37, 320, 5, False), (1, 8, 2, True), (288, 64, 64, True)])
def test_splitk_finish(M, C, sk, with_stats):
    """srvp_splitk_finish: fixed-order sum of the split-K slabs -> bf16 rows, + the per-column sum / sum of squares of the fp32 sums."""
    from srvp_amd import _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(M + C)
    parts = torch.randn(sk, M, C, generator=g).to(dev)
    dst = torch.full((M, C), 7.0, dtype=torch.bfloat16, device=dev)
    stat_mod = C // 2 if C % 16 == 0 else C                      # columns fold onto stat_mod statistics channels
    stats = torch.zeros(2, stat_mod, dtype=torch.float64, device=dev)
    L.call('srvp_splitk_finish', L.ptr(parts), sk, M * C, M, C, L.ptr(dst), L.ptr(stats) if with_stats else None, stat_mod, L.stream())
    torch.cuda.synchronize()
    ref = parts[0].clone()
    for z in range(1, sk):
        ref += parts[z]                                            # the kernel's summation order (z ascending, fp32)
    assert torch.equal(dst.view(torch.int16), ref.to(torch.bfloat16).view(torch.int16))
    if with_stats:
        r64 = ref.double().view(M, C // stat_mod, stat_mod)
        assert rel_err(stats[0], r64.sum(dim=(0, 1))) < 1e-6 and rel_err(stats[1], (r64 ** 2).sum(dim=(0, 1))) < 1e-6
    else:
        assert stats.abs().max().item() == 0
