"""
Worker of the multi-rank equality tests (tests/test_two_rank_equality.py), started once per rank by torch.distributed.run,
or once with WORLD_SIZE unset for the single-process reference run on the same GLOBAL batch.

    --mode hip    : the product (srvp_amd on cuda:0, collectives on gloo so that two ranks can share one GPU): rank r trains on
                    samples [r*B/world, (r+1)*B/world) of the global batch with the matching slice of the noise tape
    --mode oracle : the CPU oracle with its SyncBatchNorm hook -- the same recipe in the reference's arithmetic

Rank 0 writes {loss (global batch average), flat gradient (after the all-reduce), BN running statistics} to --out.
What the reference does here: train.py:205-219 (DDP: gradients averaged over ranks), 278-283 (SyncBatchNorm), 106 (loss / local batch).
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CTOR = (64, 3, 8, 16, 5, 7, True, 2, 24, 3, 32, 4, 'vgg')
T, B_GLOBAL, NE = 4, 6, 2
HP = dict(obs_scale=0.5, beta_y=1.0, beta_z=1.0, l2_res=1.0)


def problem():
    import srvp_amd
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*CTOR)
    model.init(1.2)
    g = torch.Generator().manual_seed(123)
    x = torch.rand(T, B_GLOBAL, 3, 64, 64, generator=g)
    tape = dict(t_skip=torch.randint(T, (B_GLOBAL,), generator=g),
                t_w=torch.stack([torch.randperm(T, generator=g)[:2] for _ in range(B_GLOBAL)], 1),
                eps_y0=torch.randn(B_GLOBAL, 5, generator=g), eps_z=torch.randn(T - 1, B_GLOBAL, 7, generator=g))
    return model, x, tape


def shard(x, tape, rank, world):
    n = B_GLOBAL // world
    sl = slice(rank * n, (rank + 1) * n)
    return x[:, sl].contiguous(), dict(t_skip=tape['t_skip'][sl].contiguous(), t_w=tape['t_w'][:, sl].contiguous(),
                                       eps_y0=tape['eps_y0'][sl].contiguous(), eps_z=tape['eps_z'][:, sl].contiguous())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=['hip', 'oracle'], required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--backend', choices=['gloo', 'nccl'], default='gloo',
                    help='nccl: one rank per GPU over RCCL (needs >= world GPUs); srvp_amd.distributed then takes its native in-stream '
                         'transport (csrc/comm.hip) unless SRVP_COMM=torch')
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
        dist.init_process_group(a.backend)
    model, x, tape = problem()
    xs, ts = shard(x, tape, rank, world)
    n_local = xs.shape[1]
    if a.mode == 'hip':
        import srvp_amd
        from srvp_amd import distributed as sdist
        from srvp_amd.train import fused_step
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', rank)) if (a.backend == 'nccl' and world > 1) else 0)
        model = model.to(dev).train()
        if world > 1:
            sync = sdist.Sync(stat_group=dist.new_group())
            sdist.DataParallel(model, sync)
            transport = sync.transport
        optim = srvp_amd.FusedAdam(model, lr=1e-3)
        optim.zero_grad()
        opt = srvp_amd.DotDict(dict(n_euler_steps=NE, **HP))
        acc = fused_step(model, xs.to(dev), opt, tape=ts)
        torch.cuda.synchronize()
        nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
        loss = torch.tensor([(nll + kl_y0 + kl_z + l2) / n_local], dtype=torch.float64, device=dev if a.backend == 'nccl' else 'cpu')
        flat_g = model._flat[1][:sum(p.numel() for p in model.parameters())].detach().cpu().double()
        bufs = {k: v.detach().cpu().clone() for k, v in model.named_buffers()}
    else:
        from oracle import srvp_oracle as O
        import torch.distributed.nn.functional as dfn
        if world > 1:
            def bn_sync(s1, s2, n):
                return dfn.all_reduce(s1), dfn.all_reduce(s2), n * world
            O.BN_SYNC = bn_sync
        torch.set_num_threads(4)
        # float64: an exact statement of the recipe (in fp32 the autograd of this small-batch BatchNorm network is itself only
        # good to ~5e-4 between two summation orders)
        f64 = lambda v: v.double() if v.is_floating_point() else v.clone()
        sd = {k: f64(v.detach()) for k, v in model.state_dict().items()}
        scal, _, grads = O.train_step(sd, O.make_cfg(*CTOR), xs.double(), NE, {k: f64(v) for k, v in ts.items()}, HP)
        loss = torch.tensor([scal['loss']], dtype=torch.float64)
        flat_g = torch.cat([g.flatten().double() for g in grads.values()])
        if world > 1:
            dist.all_reduce(flat_g)
            flat_g /= world                                   # DDP averages gradients over ranks
        bufs = {k: v.clone() for k, v in sd.items() if k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
    if world > 1:
        dist.all_reduce(loss)
        loss /= world                                         # mean of the per-rank batch averages = global batch average
    if rank == 0:
        torch.save(dict(loss=loss.item(), grad=flat_g, bufs=bufs, transport=locals().get('transport')), a.out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
