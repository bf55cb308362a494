import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.latent import LatentNet
def run(ne, T, B, dims):
    nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = dims
    ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
    torch.manual_seed(5)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor); model.init(1.2)
    g = torch.Generator().manual_seed(9)
    hx = torch.tanh(torch.randn(T, B, nhx, generator=g))
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                eps_y0=torch.randn(B, ny, generator=g), eps_z=torch.randn(T - 1, B, nz, generator=g))
    model = model.cuda(); model.flatten_parameters_(); grads = model._grads(); params = model._named_tensors()
    st = L.stream()
    lat = LatentNet(model._cfg(), T, B, T, ne, torch.device('cuda'), True)
    hxg = hx.cuda(); tg = {k: v.cuda() for k, v in tape.items()}
    lat.infer_w(hxg, params, tg['t_w'], st)
    y0_g, _ = lat.infer_y(hxg[:nt_inf], params, tg['eps_y0'], st)
    lat.posterior(hxg, params, st)
    for rep in range(3):
        lat.generate(y0_g, T, params, tg['eps_z'], st)
        torch.cuda.synchronize()
        fused = bool(lat._rd.fused_ws)
        keep = dict(y=lat.y_all.clone(), res=lat.res.clone(), hid=lat.hid_dyn.clone())
        lat._rd.fused_ws = None
        L.call('srvp_rollout_fwd', ctypes.byref(lat._rd), st)
        torch.cuda.synchronize()
        ref = dict(y=lat.y_all, res=lat.res, hid=lat.hid_dyn)
        S = lat.S
        msg = []
        for l in range(nl_res - 1):
            for i in range(S):
                d = (keep['hid'][l, i] - ref['hid'][l, i]).abs()
                if d.max() > 1e-4 * ref['hid'][l, i].abs().max():
                    bad = (d > 1e-4).nonzero()
                    msg.append(f'hid l{l} step{i} max {d.max().item():.3e} nbad {len(bad)} rows {sorted(set(bad[:,0].tolist()))[:8]} cols {sorted(set((bad[:,1]//32).tolist()))[:8]}')
                    break
        for i in range(S):
            d = (keep['res'][i] - ref['res'][i]).abs()
            if d.max() > 1e-4 * ref['res'][i].abs().max():
                bad = (d > 1e-5).nonzero()
                msg.append(f'res step{i} max {d.max().item():.3e} rows {sorted(set(bad[:,0].tolist()))[:10]}')
                break
        print(ne, T, B, dims, 'fused', fused, 'rep', rep, 'OK' if not msg else msg)
for c in [(2, 5, 6, (128, 50, 50, 256, 512, 3, 4, 2)), (1, 6, 70, (32, 20, 20, 64, 512, 2, 4, 3)), (2, 4, 33, (16, 10, 6, 32, 64, 2, 2, 2)),
          (2, 4, 40, (16, 12, 9, 32, 96, 2, 3, 2)), (4, 3, 192, (32, 50, 50, 64, 512, 2, 4, 2))]:
    run(*c)
