"""
World-size-2 test of the data-parallel layer on CPU (gloo): parameter broadcast at wrap time, the two-phase
flat-buffer gradient all-reduce (decoder slice first, then encoder + latent; averaged like DDP, reference
train.py:309-314) and the fp64 BatchNorm statistics all-reduce that stands for SyncBatchNorm (train.py:278-283).
The collectives are the same torch.distributed calls the GPU path issues over RCCL.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import srvp_amd
    from srvp_amd import distributed as sdist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                    # different initial weights per rank on purpose
        model = srvp_amd.StochasticLatentResidualVideoPredictor(64, 1, 4, 8, 3, 3, True, 2, 8, 3, 16, 4, 'dcgan')
        model.init()
        sync = sdist.Sync()
        wrapped = sdist.DataParallel(model, sync)
        assert wrapped.module is model and model.sync is sync
        # 1. broadcast: every rank now holds rank 0's parameters and buffers
        flat = model._flat[0].clone()
        ref = flat.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(flat, ref)
        # 2. gradient exchange: rank r contributes (r + 1) everywhere -> mean = 1.5
        grads = model._grads()
        model._flat[1].fill_(float(rank + 1))
        sync.grads_ready('decoder', model)
        sync.grads_ready('all', model)
        n = sum(p.numel() for p in model.parameters())
        assert torch.allclose(model._flat[1][:n], torch.full((n,), 1.5))
        for k, g in grads.items():
            assert g.data_ptr() == dict(model.named_parameters())[k].grad.data_ptr()
        # slices cover encoder | decoder | latent in registration order
        e, d, t = sync._slices(model)
        names = [k for k, _ in model.named_parameters()]
        assert names[0].startswith('encoder.') and 0 < e < d < t == n
        # 3. SyncBN statistics: fp64 sums add up, count scales with the world size
        st = torch.tensor([[1.0 + rank, 2.0], [3.0, 4.0 * (rank + 1)]], dtype=torch.float64)
        cnt = sync.allreduce_stats(st, 10.0)
        assert cnt == 20.0 and torch.equal(st, torch.tensor([[3.0, 4.0], [6.0, 12.0]], dtype=torch.float64))
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def _watchdog_worker(rank, world, port, q, meet):
    """Rank 0 spends 1.5 s between steps (validation / checkpoint writes) with a 0.5 s step limit armed on every rank."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    from srvp_amd import distributed as sdist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sync = sdist.Sync(native=False)
        fired = []
        wd = sdist.StepWatchdog(timeout_s=0.5, first_s=0.5, rank=rank, on_timeout=lambda w: fired.append(w.step))
        for it in range(2):
            wd.begin()                                    # srvp_amd.train.train on entry
            t = torch.ones(4, dtype=torch.float64)
            sync.allreduce_stats(t, 1.0)                  # the step's first collective: blocks until every rank is in the step
            wd.beat()
            if rank == 0:
                time.sleep(1.5)                           # rank 0 alone: evaluate() + checkpoint writes
            if meet:
                sync.after_rank0_phase()                  # what srvp_amd.train.main calls on every rank at such an iteration
        wd.begin()
        sync.allreduce_stats(torch.ones(4, dtype=torch.float64), 1.0)
        wd.beat()
        wd.stop()
        q.put((rank, bool(fired)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _run_watchdog(meet):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_watchdog_worker, args=(r, 2, port, q, meet)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def test_watchdog_does_not_count_rank0_phase():
    """ranks >= 1 used to start the next step's clock while rank 0 was still validating and were killed in the step's first
    all-reduce.  With the host meeting point of after_rank0_phase no clock fires; without it (control) rank 1's does."""
    assert _run_watchdog(meet=True) == {0: False, 1: False}
    res = _run_watchdog(meet=False)
    assert res[1] is True and res[0] is False, res


def test_watchdog_cancel_and_stop():
    """a step that raised leaves no armed clock behind; stop() ends the thread."""
    import time
    from srvp_amd import distributed as sdist
    fired = []
    wd = sdist.StepWatchdog(timeout_s=0.3, first_s=0.3, on_timeout=lambda w: fired.append(1))
    wd.begin()
    wd.cancel()                                           # (train()'s except path)
    time.sleep(0.8)
    assert not fired and wd.step == -1
    wd.begin()
    wd.stop()
    time.sleep(0.6)
    assert not fired and not wd._thread.is_alive()
