"""Times srvp_conv_in_fwd / srvp_conv_in_wgrad at the headline shape (N = 2304 frames, 3 -> 64 channels, 64x64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srvp_amd import _lib as L
N = 2304
dev = torch.device('cuda')
x = torch.rand(N, 3, 64, 64, device=dev)
w = torch.randn(64, 3, 3, 3, device=dev) * 0.1
raw = torch.empty(N, 64, 64, 64, dtype=torch.bfloat16, device=dev)
stats = torch.zeros(2, 64, dtype=torch.float64, device=dev)
draw = torch.randn(N, 66, 66, 64, device=dev).to(torch.bfloat16)
dw = torch.zeros(64, 27, device=dev)
st = L.stream()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = int(os.environ.get('N', N))
praw = torch.randn(N, 64, 64, 64, device=dev).to(torch.bfloat16)
coef = torch.stack([torch.rand(64) + 0.5, torch.randn(64) * 0.3, torch.randn(64) * 0.1, torch.rand(64) + 0.5]).to(dev).contiguous()
red = torch.zeros(2, 64, dtype=torch.float64, device=dev)
for on in (1, 0):
    L.call('srvp_conv_set_in_stream', on)
    print('in_stream=%d  conv_in_fwd  %.3f ms' % (on, t(lambda: L.call('srvp_conv_in_fwd', L.ptr(x), L.ptr(w), L.ptr(raw), L.ptr(stats), N, 3, 64, 64, 64, 64, 3, 1, 1, st))))
    print('in_stream=%d  conv_in_fwd_bnr  %.3f ms' % (on, t(lambda: L.call('srvp_conv_in_fwd_bnr', L.ptr(x), L.ptr(w), L.ptr(raw), N, 3, 64, 64, 64, 64, 3, 1, 1,
                                                                          L.ptr(praw), L.ptr(coef), L.ptr(red), st))))
L.call('srvp_conv_set_in_stream', 1)
print('conv_in_wgrad %.3f ms' % t(lambda: L.call('srvp_conv_in_wgrad', L.ptr(x), L.ptr(draw), L.ptr(dw), N, 3, 64, 64, 64, 64, 3, 1, 1, st)))
# accuracy of the weight gradient against a float64 reference (small N so that the CPU reference is quick)
N2 = 8
x2 = torch.rand(N2, 3, 64, 64, device=dev) * 2 - 0.7
d2 = torch.zeros(N2, 66, 66, 64, device=dev)
d2[:, 1:65, 1:65] = torch.randn(N2, 64, 64, 64, device=dev)
d2 = d2.to(torch.bfloat16)
dw2 = torch.zeros(64, 27, device=dev)
L.call('srvp_conv_in_wgrad', L.ptr(x2), L.ptr(d2), L.ptr(dw2), N2, 3, 64, 64, 64, 64, 3, 1, 1, st)
torch.cuda.synchronize()
xr = x2.double().cpu().requires_grad_(False)
g = d2[:, 1:65, 1:65].double().cpu().permute(0, 3, 1, 2)
ref = torch.nn.grad.conv2d_weight(xr, (64, 3, 3, 3), g, stride=1, padding=1).reshape(64, 27)
err = (dw2.double().cpu() - ref).abs().max() / ref.abs().max()
print('conv_in_wgrad max err / max |ref| vs float64: %.3e' % err.item())
