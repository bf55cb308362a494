"""
Data parallelism over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI.  Replaces DistributedDataParallel + SyncBatchNorm of the reference (train.py:205-219, 278-283, 309-314):

  * gradients live in ONE flat fp32 buffer (model.flatten_parameters_), so the exchange is a few large all-reduces
    of contiguous slices -- the decoder's slice is launched as soon as the decoder backward has been queued, so it
    overlaps the latent + encoder backward; averaged over ranks like DDP;
  * BatchNorm statistics (sum, sum of squares; and the two backward sums) are all-reduced per layer in fp64, which
    makes N-GPU results equal the 1-GPU results on the same global batch up to summation order (SyncBatchNorm).
"""
import os

import torch
import torch.distributed as dist


class Sync:
    def __init__(self, group=None, stat_group=None):
        self.group = group
        # BatchNorm statistics travel on their OWN communicator: RCCL executes the collectives of one communicator in issue
        # order, and the decoder's gradient slice (issued from the second stream, behind ~5 ms of weight-gradient kernels)
        # would otherwise hold up every statistics all-reduce of the encoder backward issued after it
        self.stat_group = stat_group if stat_group is not None else group
        self.world = dist.get_world_size(group)
        self.handles = []
        self.sync_bn = True
        # SRVP_FORCE_COLLECTIVES=1: issue every collective even on a single rank (exercises the RCCL call path on a
        # 1-GPU box: tests/test_gpu_model.py::test_single_rank_collectives)
        self.force = os.environ.get('SRVP_FORCE_COLLECTIVES', '0') == '1'

    def allreduce_stats(self, t, count):
        """In-place sum of a small fp64 statistics tensor over ranks; returns the global element count."""
        if self.sync_bn and (self.world > 1 or self.force):
            dist.all_reduce(t, group=self.stat_group)
            return count * self.world
        return count

    def _slices(self, model):
        """[decoder slice, rest] of the flat gradient buffer (parameters are registered encoder, decoder, latent)."""
        off, enc_end, dec_end = 0, None, None
        for name, p in model.named_parameters():
            if name.startswith('decoder.') and enc_end is None:
                enc_end = off
            if not name.startswith(('encoder.', 'decoder.')) and dec_end is None:
                dec_end = off
            off += p.numel()
        return enc_end, dec_end, off

    def grads_ready(self, what, model):
        if self.world == 1 and not self.force:
            return
        flat_g = model._flat[1]
        enc_end, dec_end, total = self._slices(model)
        if what == 'decoder':
            self.handles.append(dist.all_reduce(flat_g[enc_end:dec_end], group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(flat_g[:enc_end], group=self.group, async_op=True))
            self.handles.append(dist.all_reduce(flat_g[dec_end:], group=self.group, async_op=True))
            for h in self.handles:
                h.wait()
            self.handles = []
            flat_g.mul_(1.0 / self.world)     # DDP averages gradients over ranks


def init_process_group(backend=None):
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend=backend)
    return Sync(stat_group=dist.new_group() if dist.get_world_size() > 1 else None)


class DataParallel(torch.nn.Module):
    """Thin wrapper with the DistributedDataParallel calling convention (`.module`, forward passthrough)."""

    def __init__(self, module, sync):
        super().__init__()
        self.module = module
        module.sync = sync
        # same initial parameters / buffers on every rank (DDP broadcasts from rank 0)
        if sync.world > 1 or sync.force:
            module.flatten_parameters_()
            dist.broadcast(module._flat[0], 0, group=sync.group)
            for b in module.buffers():
                dist.broadcast(b, 0, group=sync.group)

    def forward(self, *a, **k):
        return self.module(*a, **k)
