"""Debug aid (round 6): which part of the step leaves unjoined work in a stream capture.  PART = fwd | step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from srvp_amd.train import train, fused_step, elbo_terms_and_grads
import bench
cfg = bench.CONFIGS[os.environ.get('CFG', 'bair')]
B = int(os.environ.get('B', 8)); T = cfg['T']
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
model.init(res_gain=cfg['res_gain'])
model.to(dev).train()
optim = srvp_amd.FusedAdam(model, lr=3e-4)
opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
x = torch.rand(T, B, cfg['ctor'][1], 64, 64).to(dev)
for _ in range(3):
    train(model, optim, None, x, dev, opt)
nt_inf, ny, nz = cfg['ctor'][7], cfg['ctor'][4], cfg['ctor'][5]
tape = dict(t_w=torch.stack([torch.randperm(T)[:nt_inf] for _ in range(B)], 1), eps_y0=torch.randn(B, ny, device=dev), eps_z=torch.randn(T - 1, B, nz, device=dev))
if cfg['ctor'][6]:
    tape['t_skip'] = torch.randint(T, (B,))
part = os.environ.get('PART', 'step')
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
from srvp_amd import _lib as L
_orig_call = L.call
state = dict(bad=None, n=0)


def status():
    st_ = ctypes.c_int(-1)
    hip.hipStreamIsCapturing(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(st_))
    return st_.value


def traced(name, *a):
    _orig_call(name, *a)
    state['n'] += 1
    s_ = status()
    if s_ != 1 and state['bad'] is None:
        state['bad'] = (name, state['n'], s_)
        print('FIRST NON-ACTIVE STATUS after', name, 'call #', state['n'], 'status', s_, flush=True)
optim.zero_grad(); fused_step(model, x, opt, tape=tape); optim.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.graph(g, stream=st, capture_error_mode=os.environ.get('MODE', 'thread_local')):
        if os.environ.get('TRACE'):
            L.call = traced
        if part == 'fwd':
            outs = model._forward_impl(x, T, opt.n_euler_steps, tape, training=True)
        elif part == 'fwd_elbo':
            outs = model._forward_impl(x, T, opt.n_euler_steps, tape, training=True)
            acc, gr = elbo_terms_and_grads(model, x, outs, opt)
        else:
            optim.zero_grad(); fused_step(model, x, opt, tape=tape); optim.step()
        L.call = _orig_call
        for nm in os.environ.get('JOIN', '').split(','):
            s2 = getattr(model, nm, None) if nm else None
            if s2 is not None:
                torch.cuda.current_stream().wait_stream(s2)
    print('RESULT', part, 'OK')
except Exception as e:
    print('RESULT', part, 'FAIL', str(e).splitlines()[0])
