"""Image-side output layer (64 -> 3, 3x3 transposed, sigmoid) at N frames: streaming kernel (srvp_conv_out_fwd) vs the MFMA tile kernel, time and agreement."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srvp_amd import _lib as L
from srvp_amd.convnet import Block, Feat
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
dev = torch.device('cuda')
g = torch.Generator().manual_seed(3)
f0 = Feat(N, 64, 64, 64, dev)
f0.t[:, 1:-1, 1:-1, :].copy_((torch.randn(N, 64, 64, 64, generator=g) * 0.5).to(torch.bfloat16))
spec = dict(kind='convT', key='w', bnkey=None, cin=64, cout=3, k=3, s=1, p=1, act='sigmoid')
blk = Block(spec, 'out', [f0], False, N, dev, False)
blk._fwd = blk.fwd_descs()
w = (torch.randn(64, 3, 3, 3, generator=g) * 0.1).to(dev)
st = L.stream()
blk.pack(w, st)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = torch.empty_like(blk.x_out)
def mfma():
    for d in blk._fwd:
        L.call('srvp_conv_mfma', C.byref(d), st)
mfma(); torch.cuda.synchronize(); ref.copy_(blk.x_out); blk.x_out.zero_()
def stream():
    L.call('srvp_conv_out_fwd', L.ptr(f0.t), L.ptr(blk.wt_o), L.ptr(blk.x_out), N, 3, 1, st)
stream(); torch.cuda.synchronize()
err = (blk.x_out - ref).abs().max().item()
print(f'N={N}: max |stream - mfma| = {err:.3e}   mfma {t(mfma):.3f} ms   stream {t(stream):.3f} ms')
bad = (blk.x_out - ref).abs() > 1e-3
if bad.any():
    idx = bad.nonzero()
    print('mismatches:', idx.shape[0], 'first', idx[:5].tolist(), 'rows', sorted(set(idx[:, 2].tolist()))[:20], 'cols', sorted(set(idx[:, 3].tolist()))[:20], 'imgs', sorted(set(idx[:, 0].tolist()))[:20])
