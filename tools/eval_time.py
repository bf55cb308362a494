"""Time the best-of-S evaluation rollout both ways: S inference forward passes (the reference's loop,
train.py:170-174) vs model.sample (one encoding, samples fanned into the batch).  BAIR VGG-64 shapes.
usage: python tools/eval_time.py [B] [S] [nt] [nt_cond]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd

_a = [int(a) for a in sys.argv[1:5]]
B, S, nt, ntc = _a + [16, 100, 30, 2][len(_a):]
dev = torch.device('cuda')
m = srvp_amd.StochasticLatentResidualVideoPredictor(64, 3, 64, 128, 20, 20, True, 2, 256, 3, 512, 4, 'vgg')
m.init(1.41)
m.to(dev).eval()
x = torch.rand(ntc, B, 3, 64, 64, device=dev)
LIM = int(os.environ.get('LIM', 2304))
chunk = max(1, min(S, LIM // (nt * B)))


def loop():
    for _ in range(S):
        m(x, nt, 0.5)


def fan():
    for s0 in range(0, S, chunk):
        m.sample(x, nt, min(chunk, S - s0), dt=0.5)


for name, fn in (('forward x S', loop), ('sample (chunk %d)' % chunk, fan)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-20s B=%d S=%d nt=%d: %.1f ms  (%.0f frames/s)  peak mem %.1f GB' % (name, B, S, nt, dt * 1e3, B * S * nt / dt,
          torch.cuda.max_memory_allocated() / 2**30))
