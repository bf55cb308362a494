"""
Worker of tests/test_two_rank_equality.py::test_two_communicators_two_streams_stress, one process per
rank under torch.distributed.run.

srvp_amd.distributed drives TWO native RCCL communicators from TWO streams: SyncBatchNorm statistics in line on the compute stream, the
decoder's gradient slice from the weight-gradient side stream (distributed.py: Sync.allreduce_stats / grads_ready('decoder')).  RCCL only
guarantees progress for concurrent communicators if the kernels of both can be co-resident on every rank, and the product's persistent
cluster kernels (csrc/rollout_fused.hip: <= 256 co-resident workgroups spinning on counters) are exactly the neighbours that make that
assumption worth a test.  Per iteration, on every rank:

    main stream : statistics all-reduce ([2][512] fp64)  ->  persistent fused rollout forward + backward  ->  statistics all-reduce
    side stream : gradient-slice all-reduce (8 M fp32 = 32 MB)

with the two streams issued in OPPOSITE host order on even and odd ranks (and the order swapped every iteration), which is the order
inversion a deadlock would need.  Inputs are small integers times (rank + 1), so every sum is exact and is checked on the device every
iteration (a mismatch count accumulates on the device; ONE host read at the end).  Must finish; sums exact; zero cluster-barrier timeouts;
rollout results bit-identical to the first iteration's.   With WORLD_SIZE=1 and SRVP_FORCE_COLLECTIVES=1 the same loop runs on one rank
(1-rank communicators: the single-GPU boxes can at least exercise the call pattern).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2000)
    ap.add_argument('--out', required=True)
    ap.add_argument('--grad-elems', type=int, default=8 << 20)
    a = ap.parse_args()
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
    dev = torch.device('cuda')
    import srvp_amd
    from srvp_amd import _lib as L
    from srvp_amd import distributed as sdist
    from srvp_amd.latent import LatentNet
    sync = sdist.init_process_group('nccl')
    assert sync.native_stats is not None and sync.native_grads is not None, f'native RCCL path not taken: {sync.transport}'
    info = (sync.native_stats.info(), sync.native_grads.info())
    assert info[0]['ranks'] == world and info[1]['ranks'] == world, info

    # ---- a persistent rollout launch of the headline latent dimensions at 24 sequences (config 4's per-GPU share)
    ne, T, B = 2, 12, 24
    nhx, ny, nz, nh_inf, nh_res, nl_inf, nl_res, nt_inf = 128, 50, 50, 256, 512, 3, 4, 2
    ctor = (64, 1, 4, nhx, ny, nz, False, nt_inf, nh_inf, nl_inf, nh_res, nl_res, 'dcgan')
    torch.manual_seed(5)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    model = model.to(dev)
    model.flatten_parameters_()
    params = model._named_tensors()
    g = torch.Generator().manual_seed(9)
    st = L.stream()
    lat = LatentNet(model._cfg(), T, B, T, ne, dev, True)
    hx = torch.tanh(torch.randn(T, B, nhx, generator=g)).to(dev)
    t_w = torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1).to(dev)
    eps_y0, eps_z = torch.randn(B, ny, generator=g).to(dev), torch.randn(T - 1, B, nz, generator=g).to(dev)
    lat.infer_w(hx, params, t_w, st)
    y0, _ = lat.infer_y(hx[:nt_inf], params, eps_y0, st)
    lat.posterior(hx, params, st)
    d_res = torch.randn(lat.S, B, ny, generator=g).to(dev)
    lat.d_y_all.copy_(torch.randn(lat.S + 1, B, ny, generator=g))

    def rollout():
        lat.generate(y0, T, params, eps_z, st)
        assert lat._rd.fused_ws, 'the persistent fused rollout kernel is the neighbour under test'
        bd = L.RolloutBwdDesc()
        bd.f = lat._rd
        bd.d_y_all, bd.d_z, bd.d_pz, bd.d_res = L.ptr(lat.d_y_all), None, None, L.ptr(d_res)
        bd.d_y0, bd.d_qz, bd.dhid_dyn, bd.dhid_pz, bd.work = (L.ptr(lat.d_y0), L.ptr(lat.d_qz_samp), L.ptr(lat.dhid_dyn),
                                                               L.ptr(lat.dhid_pz), L.ptr(lat.work))
        bd.dinp_all = L.ptr(lat.dinp_all)
        L.call('srvp_rollout_bwd', ctypes.byref(bd), st)
    rollout()
    torch.cuda.synchronize()
    ref = [t.clone() for t in (lat.y_all, lat.res, lat.dinp_all, lat.d_y0)]

    side = torch.cuda.Stream()
    s1 = torch.zeros(2, 512, dtype=torch.float64, device=dev)
    s2 = torch.zeros(2, 512, dtype=torch.float64, device=dev)
    gr = torch.zeros(a.grad_elems, dtype=torch.float32, device=dev)
    bad = torch.zeros(3, dtype=torch.int64, device=dev)           # mismatches: statistics, gradients, rollout
    tri = world * (world + 1) // 2
    done_side = None
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(a.iters):
        k1, k2 = float(it % 7 + 1), float(it % 5 + 1)

        def main_part():
            s1.fill_(k1 * (rank + 1))
            sync.native_stats.allreduce(s1)
            rollout()
            s2.fill_(k1 * (rank + 1) + 1.0)
            sync.native_stats.allreduce(s2)
            bad[0] += (s1 != k1 * tri).sum() + (s2 != k1 * tri + world).sum()
            if it % 50 == 0:
                bad[2] += sum((x != r).sum() for x, r in zip((lat.y_all, lat.res, lat.dinp_all, lat.d_y0), ref))

        def side_part():
            nonlocal done_side
            with torch.cuda.stream(side):
                gr.fill_(k2 * (rank + 1))
                sync.native_grads.allreduce(gr)                  # (L.stream() = the current stream: the side stream here)
                bad[1] += (gr != k2 * tri).sum()
                done_side = torch.cuda.Event()
                done_side.record()
        if (rank + it) % 2 == 0:
            main_part(); side_part()
        else:
            side_part(); main_part()
        torch.cuda.current_stream().wait_event(done_side)        # (bad[] is shared: the side stream's update lands before the next main one)
        if it % 250 == 249:
            torch.cuda.synchronize()                             # bounds the host's run-ahead; progress line for the test log
            print(f'rank {rank}: {it + 1} iterations, {time.time() - t0:.1f} s', file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    secs = time.time() - t0
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    L.call('srvp_cluster_timeouts_read', host.data_ptr(), st)
    torch.cuda.synchronize()
    res = dict(rank=rank, world=world, iters=a.iters, seconds=secs, mismatches=bad.cpu().tolist(), cluster_timeouts=int(host[0]),
               rccl=info[0], transport=sync.transport)
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if rank == 0:
        with open(a.out, 'w') as f:
            json.dump(allres, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
