"""Persistent latent kernels in isolation (csrc/rollout_fused.hip): fused rollout forward / backward (22 Euler steps x 4 layers at the headline
dimensions) and the persistent LSTM forward / backward (12 steps), HIP-event time per launch, with the XCD-local exchange on and off
(srvp_cluster_set_xcd_local), and a bit-equality check between the two.     usage: python tools/rollout_time.py [B ...]   (default 24 192)"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.latent import LatentNet


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(B):
    ne, T = 2, 12
    nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = 128, 50, 50, 256, 512, 3, 4, 2
    ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
    torch.manual_seed(5)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    g = torch.Generator().manual_seed(9)
    dev = torch.device('cuda')
    model = model.to(dev)
    model.flatten_parameters_()
    model._grads()
    params = model._named_tensors()
    st = L.stream()
    lat = LatentNet(model._cfg(), T, B, T, ne, dev, True)
    hx = torch.tanh(torch.randn(T, B, nhx, generator=g)).to(dev)
    t_w = torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1).to(dev)
    eps_y0, eps_z = torch.randn(B, ny, generator=g).to(dev), torch.randn(T - 1, B, nz, generator=g).to(dev)
    lat.infer_w(hx, params, t_w, st)
    y0, _ = lat.infer_y(hx[:nt_inf], params, eps_y0, st)
    lat.posterior(hx, params, st)
    lat.generate(y0, T, params, eps_z, st)
    assert lat._rd.fused_ws
    d_res = torch.randn(lat.S, B, ny, generator=g).to(dev)
    lat.d_y_all.copy_(torch.randn(lat.S + 1, B, ny, generator=g))
    lat.d_hz.copy_(torch.randn(T * B, nh_inf, generator=g) * 0.1)
    bd = L.RolloutBwdDesc()
    bd.f = lat._rd
    bd.d_y_all, bd.d_z, bd.d_pz, bd.d_res = L.ptr(lat.d_y_all), None, None, L.ptr(d_res)
    bd.d_y0, bd.d_qz, bd.dhid_dyn, bd.dhid_pz, bd.work = (L.ptr(lat.d_y0), L.ptr(lat.d_qz_samp), L.ptr(lat.dhid_dyn), L.ptr(lat.dhid_pz), L.ptr(lat.work))
    bd.dinp_all = L.ptr(lat.dinp_all)
    whh = params['inf_z.weight_hh_l0']
    need = int(L.load().srvp_lstm_fused_ws_bytes(T, B, nh_inf))
    ws = lat._lstm_ws
    fns = dict(
        rollout_fwd=lambda: L.call('srvp_rollout_fwd', ctypes.byref(lat._rd), st),
        rollout_bwd=lambda: L.call('srvp_rollout_bwd', ctypes.byref(bd), st),
        lstm_fwd=lambda: L.call('srvp_lstm_fwd_fused', L.ptr(lat.gates_x), L.ptr(whh), L.ptr(lat.hz), L.ptr(lat.cz), L.ptr(lat.gates_act), T, B, nh_inf, L.ptr(ws), need, st),
        lstm_bwd=lambda: L.call('srvp_lstm_bwd_fused', L.ptr(lat.d_hz), L.ptr(whh), L.ptr(lat.cz), L.ptr(lat.gates_act), L.ptr(lat.dgates), T, B, nh_inf, L.ptr(ws), need, st))
    outs = dict(rollout_fwd=lambda: (lat.y_all, lat.res, lat.hid_dyn), rollout_bwd=lambda: (lat.dinp_all, lat.d_y0, lat.dhid_dyn[:nl_res - 1]),
                lstm_fwd=lambda: (lat.hz, lat.cz, lat.gates_act), lstm_bwd=lambda: (lat.dgates,))
    host = torch.zeros(2, dtype=torch.int32).pin_memory()
    row = dict(B=B, steps=lat.S)
    keep = {}
    for on in (0, 1):
        L.call('srvp_cluster_set_xcd_local', on)
        L.call('srvp_cluster_stats_read', host.data_ptr(), st)
        torch.cuda.synchronize()
        before = int(host[1])
        for name, fn in fns.items():
            row[f'{name}_us_xcd{on}'] = round(timeit(fn), 1)
            fn()
            torch.cuda.synchronize()
            got = [t.clone() for t in outs[name]()]
            if on == 0:
                keep[name] = got
            else:
                row[f'{name}_bit_equal'] = all(torch.equal(a, b) for a, b in zip(got, keep[name]))
        L.call('srvp_cluster_stats_read', host.data_ptr(), st)
        torch.cuda.synchronize()
        row[f'clusters_xcd_local_{on}'] = int(host[1]) - before
        row['cluster_timeouts'] = int(host[0])
    L.call('srvp_cluster_set_xcd_local', 1)
    print(json.dumps(row), flush=True)


if __name__ == '__main__':
    for b in ([int(v) for v in sys.argv[1:]] or [24, 192]):
        run(b)
