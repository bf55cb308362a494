"""Clock / power under the big conv layers vs a BN pass (evidence for the power-ceiling statement)."""
import ctypes as C, os, sys, subprocess, threading, time, re
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd, bench
from srvp_amd import _lib as L
from srvp_amd.train import train
cfg = bench.CONFIGS['bair']; B = 192; T = cfg['T']
dev = torch.device('cuda', 0); torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor']); model.init(res_gain=cfg['res_gain']); model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, 3, 64, 64).to(dev)
for _ in range(2): train(model, optim, None, x, dev, opt)
torch.cuda.synchronize()
pl = list(model._plans.values())[0]; st = L.stream()
samples = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r'sclk clock level.*?\((\d+)Mhz\)', o)
            pw = re.findall(r'Power \(W\):\s*([\d.]+)', o)
            samples.append((time.time(), sclk[0] if sclk else '?', pw[0] if pw else '?'))
        except Exception as e:
            samples.append((time.time(), 'err', str(e)[:40]))
th = threading.Thread(target=poll); th.start()
def phase(name, fn, secs=4.0):
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50 if 'training step' not in name else 5): fn()
        torch.cuda.synchronize(); n += 50 if 'training step' not in name else 5
    t1 = time.time()
    s = [(c, p) for (t, c, p) in samples if t0 + 1.0 <= t <= t1]
    print(f'{name}: {1e3*(t1-t0)/n:.3f} ms/launch  samples(sclk MHz, W): {s[:6]}', flush=True)
blk = pl['enc'].blocks[8]
d = blk._fwd[0]
phase('idle', lambda: None, 3.0)
phase('enc08 fwd conv (512->512, 8x8)', lambda: L.call('srvp_conv_mfma', C.byref(d), st))
blk1 = pl['enc'].blocks[1]
phase('enc01 fwd conv (64->64, 64x64)', lambda: L.call('srvp_conv_mfma', C.byref(blk1._fwd[0]), st))
a = torch.empty(600_000_000, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a)
phase('copy 1.2 GB (HBM bound)', lambda: b.copy_(a))
w1 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); w2 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
phase('hipBLASLt bf16 8192^3', lambda: torch.matmul(w1, w2))
z1 = torch.zeros(8192, 8192, device=dev, dtype=torch.bfloat16)
phase('hipBLASLt bf16 8192^3 on zeros', lambda: torch.matmul(z1, z1))
phase('whole training step (B=192)', lambda: train(model, optim, None, x, dev, opt), 6.0)
# round 6: do the weight gradients lose their rate in-step because the chip is at its power cap (clocks drop under sharing)
# or because of LDS / L2 contention?  (i) a data-gradient conv and a weight-gradient launch of the same 8x8x512 layer back to back on ONE
# stream, (ii) the same two on TWO streams (co-running), (iii) the step with the second stream off
dg, wg = blk._dg[0], (blk._wg if not isinstance(blk._wg, list) else blk._wg[0])
s2 = torch.cuda.Stream()
def serial():
    L.call('srvp_conv_mfma', C.byref(dg), st); L.call('srvp_wgrad_mfma', C.byref(wg), st)
def corun():
    L.call('srvp_conv_mfma', C.byref(dg), st)
    with torch.cuda.stream(s2):
        L.call('srvp_wgrad_mfma', C.byref(wg), L.stream())
phase('enc08 dgrad alone', lambda: L.call('srvp_conv_mfma', C.byref(dg), st))
phase('enc08 wgrad alone', lambda: L.call('srvp_wgrad_mfma', C.byref(wg), st))
phase('enc08 dgrad + wgrad, one stream (per pair)', serial)
phase('enc08 dgrad + wgrad, two streams co-running (per pair)', corun)
import srvp_amd.model as _m, srvp_amd.convnet as _cn
_m.OVERLAP_WGRAD, _m.OVERLAP_SKIP, _m.OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN = False, False, False, 0
for _ in range(2): train(model, optim, None, x, dev, opt)
phase('whole training step (B=192), second stream OFF', lambda: train(model, optim, None, x, dev, opt), 6.0)
stop = True; th.join()
