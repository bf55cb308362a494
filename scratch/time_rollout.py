import sys, ctypes, os, torch
sys.path.insert(0, '/root/repo')
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.latent import LatentNet
B = int(sys.argv[1]); fused = int(sys.argv[2])
ne, T = 2, 12
dims = (128, 50, 50, 256, 512, 3, 4, 2)
nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = dims
ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
torch.manual_seed(5)
model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor); model.init(1.2)
g = torch.Generator().manual_seed(9)
hx = torch.tanh(torch.randn(T, B, nhx, generator=g))
model = model.cuda(); model.flatten_parameters_(); grads = model._grads(); params = model._named_tensors()
st = L.stream()
lat = LatentNet(model._cfg(), T, B, T, ne, torch.device('cuda'), True)
hxg = hx.cuda()
t_w = torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1).cuda()
eps_y0 = torch.randn(B, ny, generator=g).cuda(); eps_z = torch.randn(T - 1, B, nz, generator=g).cuda()
lat.infer_w(hxg, params, t_w, st)
y0, _ = lat.infer_y(hxg[:nt_inf], params, eps_y0, st)
lat.posterior(hxg, params, st)
lat.generate(y0, T, params, eps_z, st)
if not fused:
    lat._rd.fused_ws = None
bd = L.RolloutBwdDesc()
bd.f = lat._rd
bd.d_y_all, bd.d_z, bd.d_pz, bd.d_res = L.ptr(lat.d_y_all), None, None, L.ptr(torch.randn_like(lat.res))
bd.d_y0, bd.d_qz, bd.dhid_dyn, bd.dhid_pz, bd.work = (L.ptr(lat.d_y0), L.ptr(lat.d_qz_samp), L.ptr(lat.dhid_dyn), L.ptr(lat.dhid_pz), L.ptr(lat.work))
bd.dinp_all = L.ptr(lat.dinp_all)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tf = timeit(lambda: L.call('srvp_rollout_fwd', ctypes.byref(lat._rd), st))
tb = timeit(lambda: L.call('srvp_rollout_bwd', ctypes.byref(bd), st))
print(f'B={B} fused={fused} dbg={os.environ.get("SRVP_RF_DEBUG","0")} fwd {tf*1e3:.0f} us  bwd {tb*1e3:.0f} us')
