"""
Latency of the SyncBatchNorm statistics exchange (reference train.py:278-283): one all-reduce of a [2][C] fp64 tensor per
BatchNorm layer and direction, 84 per VGG training step, each feeding the very next kernel.  Measures, in stream order between
two tiny dependent kernels (the situation inside the step), the native in-stream RCCL path (csrc/comm.hip) and the same
collective through torch.distributed -- so DESIGN §5's cost estimate becomes a number the moment >= 2 GPUs are available.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/allreduce_latency.py
    SRVP_FORCE_COLLECTIVES=1 python tools/allreduce_latency.py          # one rank: call-path overhead only (no peer)
    SRVP_COMM=peer ... (either form)                                     # adds the peer-read prototype (srvp_peer_allreduce_f64)
    SRVP_COMM=peer SRVP_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ...   # two ranks sharing ONE GPU: the
                                                                         # protocol's own cost (publish, flag, wait, read), no xGMI

Rank 0 prints one JSON line: microseconds per (kernel, all-reduce) pair for C in {64, 512}, both transports, and the pure
kernel-pair baseline.
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    # RCCL prints a version banner through C stdio at communicator creation: keep stdout for the JSON line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from srvp_amd import distributed as sdist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('SRVP_FORCE_COLLECTIVES', '1')
    backend = os.environ.get('SRVP_DIST_BACKEND', 'nccl')       # gloo: ranks sharing one GPU (peer path only; RCCL refuses that)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0)
    sync = sdist.init_process_group(backend)
    out = dict(world=world, transport=sync.transport, iters=200)
    for C in (64, 512):
        t = torch.zeros(2, C, dtype=torch.float64, device='cuda')

        def loop(kind, n=200):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                t.add_(1.0)                       # stands for the producer of the sums (dependent tiny kernel)
                if kind == 'native':
                    sync.native_stats.allreduce(t)
                elif kind == 'torch':
                    dist.all_reduce(t, group=sync.stat_group)
                elif kind == 'peer':
                    sync.peer.allreduce(t)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        for kind in ('none', 'native', 'peer', 'torch'):
            if (kind == 'native' and sync.native_stats is None) or (kind == 'peer' and getattr(sync, 'peer', None) is None) or \
                    (kind == 'torch' and backend != 'nccl'):
                out[f'C{C}_{kind}_us'] = None
                continue
            loop(kind, 20)
            out[f'C{C}_{kind}_us'] = round(loop(kind), 2)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
