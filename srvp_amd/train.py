"""
Training driver mirroring the reference's train.py (same function names, arguments, CLI flags and outputs):
`train(forward_fn, optimizer, scaler, batch, device, opt)` (reference train.py:49-129), `evaluate(...)`
(train.py:132-189) and `main(opt)` (train.py:192-384).

The optimisation step is the MI355X-native one: forward, ELBO terms with their gradients, backward and Adam are all
HIP kernels (libsrvp_hip.so); the four logged scalars come back in ONE device->host copy per step instead of the
reference's four `.item()` syncs (train.py:124-127).
"""
import os
import random
import sys

import numpy as np
import torch

from . import _lib as L
from . import metrics as _metrics
from .model import StochasticLatentResidualVideoPredictor
from .optim import FusedAdam


def _unwrap(forward_fn):
    m = getattr(forward_fn, 'module', forward_fn)
    return m if isinstance(m, StochasticLatentResidualVideoPredictor) else None


def elbo_terms_and_grads(model, x, outs, opt, want_grads=True):
    """
    train.py:90-106 on the device: returns a float64 tensor [nll, kl_y0, kl_z, l2_res] (sums, not yet /batch) and
    the gradients of loss = (nll + beta_y kl_y0 + beta_z kl_z + l2_res * sum||res||) / batch wrt the forward outputs.
    """
    x_, y, z, w, q_y0, qz, pz, res = outs
    B = x.shape[1]
    dev = x.device
    st = L.stream()
    acc = torch.zeros(4, dtype=torch.float64, device=dev)
    inv = 1.0 / B
    bufs = model._last_plan.setdefault('elbo_bufs', {})

    def buf(name, like):
        if like is None or not want_grads:
            return None
        t = bufs.get(name)
        if t is None or t.shape != like.shape:
            t = torch.empty_like(like, dtype=torch.float32)
            bufs[name] = t
        return t
    d_x, d_qy0, d_qz, d_pz, d_res = buf('d_x', x_), buf('d_qy0', q_y0), buf('d_qz', qz), buf('d_pz', pz), buf('d_res', res)
    xc = x.contiguous().float()
    L.call('srvp_nll', L.ptr(x_), L.ptr(xc), L.ptr(d_x), x_.numel(), float(opt.obs_scale), inv, L.ptr(acc[0:1]), st)
    L.call('srvp_kl', L.ptr(q_y0), None, L.ptr(d_qy0), None, q_y0.shape[0], q_y0.shape[1] // 2, float(opt.beta_y) * inv,
           L.ptr(acc[1:2]), st)
    if qz is not None:
        rows = qz.shape[0] * qz.shape[1]
        L.call('srvp_kl', L.ptr(qz), L.ptr(pz), L.ptr(d_qz), L.ptr(d_pz), rows, qz.shape[2] // 2, float(opt.beta_z) * inv,
               L.ptr(acc[2:3]), st)
    l2 = float(opt.l2_res) if opt.l2_res is not None else 0.0
    if l2 > 0:
        L.call('srvp_l2rows', L.ptr(res), L.ptr(d_res), res.shape[0] * res.shape[1], res.shape[2], l2 * inv, L.ptr(acc[3:4]), st)
    else:
        d_res = None
    return acc, (d_x, d_qy0, d_qz, d_pz, d_res)


def fused_step(model, x, opt, tape=None):
    """Forward + ELBO + backward entirely on the HIP path (no autograd graph).  Returns the float64 device tensor
    [nll, kl_y0, kl_z, l2_res]."""
    T = x.shape[0]
    outs = model._forward_impl(x, T, opt.n_euler_steps, tape, training=True)
    acc, (d_x, d_qy0, d_qz, d_pz, d_res) = elbo_terms_and_grads(model, x, outs, opt)
    # the ELBO terms are final here: copy them to pinned host memory now and mark the point with an event, so that the caller's
    # read-back (train(): the step's single host sync) waits for the FORWARD only and the host can queue the next step while the
    # backward still runs (a stream-wide sync at the end of the step left the GPU idle for ~0.9 ms of host prologue per step)
    host = model.__dict__.get('_elbo_host')
    if host is None:
        host = model.__dict__['_elbo_host'] = torch.empty(4, dtype=torch.float64).pin_memory()
    host.copy_(acc, non_blocking=True)
    model.__dict__['_elbo_event'] = L.record()
    model._backward_impl(d_x, None, None, d_qy0, d_qz, d_pz, d_res)
    return acc


def _poll_cluster_timeouts(model, every=32):
    """Every `every` steps (and on the first): the library's count of cluster failures in the persistent latent kernels (model._poll_cluster)."""
    model._poll_cluster(every)


def train(forward_fn, optimizer, scaler, batch, device, opt):
    """
    One optimisation step (reference train.py:49-129): returns (loss, nll, kl_y_0, kl_z) as python floats, all
    batch-averaged like the reference's logs.  `scaler` (torch.cuda.amp.GradScaler in the reference) is accepted for
    signature compatibility and ignored: the bf16 kernels keep fp32 accumulation and need no loss scaling.
    """
    model = _unwrap(forward_fn)
    if model is None:
        raise TypeError('srvp_amd.train.train needs the srvp_amd StochasticLatentResidualVideoPredictor (or a wrapper '
                        'exposing it as .module)')
    wd = model.__dict__.get('_watchdog')
    if wd is not None:
        wd.begin()
    try:
        with L.step_scope():                                # (torch's current stream looked up once for the whole step)
            optimizer.zero_grad()
            if batch.dtype == torch.uint8:                 # stacked uint8 videos (data.collate_u8): finish the collate on the GPU
                from .data import frames_from_u8
                x = frames_from_u8(batch, device)
            else:
                x = batch.to(device, non_blocking=True)
            n = x.shape[1]
            fused_step(model, x, opt)
            optimizer.step()
            _poll_cluster_timeouts(model)
        model._elbo_event.synchronize()                     # the step's single host sync: waits for the forward + ELBO only
    except BaseException:
        #  a step that raised -- SrvpHipError of the cluster poll, an OOM, KeyboardInterrupt inside the sync -- is not a step in
        # flight any more: an armed clock would os._exit() the process up to SRVP_WATCHDOG_S later, e.g. in the middle of main()'s final
        # checkpoint writes, or kill a pytest session after a test that expected the error
        if wd is not None:
            wd.cancel()
        raise
    if wd is not None:
        wd.beat()                                       # (distributed.StepWatchdog: a step that never gets here ends the job with a report)
    nll, kl_y_0, kl_z, l2 = model._elbo_host.tolist()
    loss = nll + opt.beta_y * kl_y_0 + opt.beta_z * kl_z
    if opt.l2_res is not None and opt.l2_res > 0:
        loss += opt.l2_res * l2
    return loss / n, nll / n, kl_y_0 / n, kl_z / n


def evaluate(forward_fn, val_loader, device, opt):
    """Validation PSNR, best of n_samples_test predictions per video (reference train.py:132-189); returns -PSNR."""
    inf_len = opt.nt_cond
    assert val_loader is not None and opt.n_iter_test <= len(val_loader)
    n, global_psnr = 0, 0.0
    with torch.no_grad():
        for j, batch in enumerate(val_loader):
            if j >= opt.n_iter_test:
                break
            x = batch.to(device)
            x_inf = x[:inf_len]
            nt, n_b = x.shape[0], x.shape[1]
            n += n_b
            best_psnr, best_x = None, None
            model = _unwrap(forward_fn)
            if model is not None and opt.n_samples_test > 1 and hasattr(model, 'sample') and not model.training:
                # one encoding of the conditioning frames, samples fanned out inside the latent path / decoder,
                # in chunks of <= SRVP_EVAL_FRAMES decoded frames (9216 frames of VGG-64 keep ~31 GB of the 288 GB resident)
                lim = int(os.environ.get('SRVP_EVAL_FRAMES', 9216))
                chunk = max(1, min(opt.n_samples_test, lim // max(1, nt * n_b)))
                # ONE call: the encoder and the latent path (posterior + prior rollout: latency-bound, batch-size independent) run once
                # for all samples, the decoder works through them `chunk` at a time
                xs = model.sample(x_inf, nt, opt.n_samples_test, dt=1 / opt.n_euler_steps, chunk=chunk)
                samples = (xs[:, i] for i in range(opt.n_samples_test))
            else:
                samples = (forward_fn(x_inf, nt, dt=1 / opt.n_euler_steps)[0] for _ in range(opt.n_samples_test))
            for x_s in samples:
                mse = _metrics.mse(x_s, x)                                # (nt, B, C), one pass on the device (§8f-4)
                psnr = torch.mean(10 * torch.log10(1 / mse), dim=[0, 2])  # (B,)
                if best_psnr is None:
                    best_psnr, best_x = psnr, x_s.clone()
                else:
                    better = psnr > best_psnr
                    best_psnr = torch.where(better, psnr, best_psnr)
                    best_x[:, better] = x_s[:, better]
            psnr = _metrics.psnr(best_x, x)
            global_psnr += psnr[inf_len:].mean().item() * n_b
    model = _unwrap(forward_fn)
    if model is not None:
        model.drop_sample_plans()          # tens of GB of S > 1 activation buffers must not stay resident during training
    return -global_psnr / n


def write_config(opt, path):
    """The run's options as JSON (the file reference test.py:177 loads to rebuild the model)."""
    import json
    cfg = {k: v for k, v in dict(opt).items() if isinstance(v, (int, float, str, bool, list, tuple, type(None)))}
    with open(path, 'w') as f:
        json.dump(cfg, f, indent=1, sort_keys=True)


def _rng_state():
    return dict(python=random.getstate(), numpy=np.random.get_state(), torch=torch.get_rng_state(),
                cuda=torch.cuda.get_rng_state() if torch.cuda.is_available() else None)


def rank_rng_path(path, rank):
    """RNG streams differ per rank (numpy is seeded seed + local_rank, train.py:226-228): each rank keeps its own file."""
    return f'{path}.rng{rank}'


def save_rank_rng(path, rank, itr=None):
    """itr: the iteration the streams belong to -- load_train_state ignores a file whose iteration differs from the model's
    (ranks write independently of rank 0's train_state.pt: a crash between the two writes, or a directory left by an earlier run,
    would otherwise pair RNG state of iteration N with a model of iteration M unnoticed)."""
    tmp = rank_rng_path(path, rank) + '.tmp'
    torch.save(dict(_rng_state(), itr=itr), tmp)
    os.replace(tmp, rank_rng_path(path, rank))


def save_train_state(path, model, optimizer, lr_scheduler, itr, best_val_metric):
    """Everything needed to continue a run bit-for-bit on the same hardware: weights + BN buffers, Adam moments and step,
    LR schedule, iteration counter, best validation metric, RNG states (python / numpy / torch CPU and device) of the
    writing rank (rank 0); the other ranks of a data-parallel run save theirs with save_rank_rng."""
    state = dict(model=model.state_dict(), optimizer=optimizer.state_dict(), lr_scheduler=lr_scheduler.state_dict(), itr=itr,
                 best_val_metric=best_val_metric, rng=_rng_state())
    tmp = path + '.tmp'
    torch.save(state, tmp)
    os.replace(tmp, path)


def load_train_state(path, model, optimizer, lr_scheduler, device, rank=0, seed=None):
    """rank > 0 restores ITS OWN RNG streams (train_state.pt.rng<rank>, written by save_rank_rng); if that file is missing the
    numpy stream is re-derived from (seed, rank, iteration) so that the ranks never share one data stream after a resume."""
    state = torch.load(path, map_location=device, weights_only=False)
    model.load_state_dict(state['model'])
    optimizer.load_state_dict(state['optimizer'])
    lr_scheduler.load_state_dict(state['lr_scheduler'])
    rng = state.get('rng') or {}
    if rank > 0:
        rp = rank_rng_path(path, rank)
        own = torch.load(rp, map_location='cpu', weights_only=False) if os.path.exists(rp) else None
        if own is not None and own.get('itr') == int(state['itr']):
            rng = own
        else:
            if own is not None:
                print(f'srvp_amd.train: {rp} belongs to iteration {own.get("itr")}, the model to {state["itr"]}: RNG file ignored '
                      '(seed-derived stream instead)')
            rng = dict(rng, numpy=None)
            np.random.seed(((seed or 0) + rank + 7919 * int(state['itr'])) % (2 ** 32))
    if rng.get('python') is not None:
        random.setstate(rng['python'])
    if rng.get('numpy') is not None:
        np.random.set_state(rng['numpy'])
    if rng.get('torch') is not None:
        torch.set_rng_state(rng['torch'].cpu())
    if rng.get('cuda') is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(rng['cuda'].cpu())
    return state['itr'], state['best_val_metric']


def main(opt):
    """Trains SRVP and saves the resulting model (reference train.py:192-384); same flags (srvp_amd/args.py)."""
    from . import data as sdata
    from . import distributed as sdist
    if opt.device is None:
        raise SystemExit('srvp_amd trains on MI355X GPUs only: pass --device')
    opt.n_gpu = len(opt.device)
    local_rank = int(os.environ.get('LOCAL_RANK', opt.local_rank or 0))
    opt.local_rank = local_rank
    # train.py:210-212 overwrites CUDA_VISIBLE_DEVICES per process; here --device indexes the devices this process can
    # already see (an exported HIP_VISIBLE_DEVICES list stays valid, and every rank lands on its own GPU)
    device = torch.device('cuda', int(opt.device[local_rank]))
    torch.cuda.set_device(device)
    sync = None
    if opt.n_gpu > 1 or local_rank > 0:
        sync = sdist.init_process_group()
        assert opt.seed is not None
        assert opt.batch_size % opt.n_gpu == 0
        opt.global_batch_size = opt.batch_size          # what the user passed (and what config.json records)
        opt.batch_size = opt.batch_size // opt.n_gpu
    if opt.seed is None:
        opt.seed = random.randint(1, 10000)
    print(f'Learning on {opt.n_gpu} GPU(s) (seed: {opt.seed})')
    random.seed(opt.seed)
    np.random.seed(opt.seed + local_rank)
    torch.manual_seed(opt.seed)

    print('Loading data...')
    train_loader, val_loader, sampler = sdata.make_loaders(opt, local_rank)

    print('Building model...')
    model = StochasticLatentResidualVideoPredictor(opt.nx, opt.nc, opt.nf, opt.nhx, opt.ny, opt.nz, opt.skipco, opt.nt_inf,
                                                   opt.nh_inf, opt.nlayers_inf, opt.nh_res, opt.nlayers_res, opt.archi)
    model.init(res_gain=opt.res_gain)
    model.to(device)
    forward_fn = model
    if sync is not None:
        forward_fn = sdist.DataParallel(model, sync)

    optimizer = FusedAdam(model, lr=opt.lr)
    opt.n_iter = opt.lr_scheduling_burnin + opt.lr_scheduling_n_iter
    n_sch = opt.lr_scheduling_n_iter
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda i: max(0, (n_sch - i) / n_sch))

    assert opt.n_iter > 0
    os.makedirs(opt.save_path, exist_ok=True)
    itr, finished, status_code = 0, False, 0
    val_metric = best_val_metric = None
    # what the reference lacks.  config.json (test.py:177 expects it next to the weights) and a resumable
    # training state (optimizer moments, LR schedule, iteration, best validation metric, RNG) beside the reference-compatible
    # model*.pt state dicts.  Resume: opt.resume / SRVP_RESUME = directory holding train_state.pt (no new CLI flag).
    if local_rank == 0:
        cfg_out = dict(opt)
        cfg_out['batch_size'] = getattr(opt, 'global_batch_size', None) or opt.batch_size
        write_config(cfg_out, os.path.join(opt.save_path, 'config.json'))
    resume_dir = getattr(opt, 'resume', None) or os.environ.get('SRVP_RESUME')
    if resume_dir:
        itr, best_val_metric = load_train_state(os.path.join(resume_dir, 'train_state.pt'), model, optimizer, lr_scheduler, device,
                                                rank=local_rank, seed=opt.seed)
        print(f'Resumed from {resume_dir} at iteration {itr}')
        gen = getattr(train_loader, 'gen', None)
        if gen is not None and hasattr(gen, 'counter'):
            gen.counter = itr          # device Moving-MNIST generator: batch k is a function of (seed, k) -- continue the same data stream
        vgen = getattr(val_loader, 'gen', None) if val_loader is not None else None
        if vgen is not None and hasattr(vgen, 'counter'):
            vgen.counter = (itr // opt.val_interval) * opt.n_iter_test      # ... and the validation stream where the validations so far left it
    # the plans / descriptor tables built by the first steps are long-lived: keep them out of the cyclic collector's generations
    # (a full collection over them is a ~90 ms host stall, several steps' worth at the small configurations)
    import gc
    gc.collect()
    gc.freeze()
    import time as _time
    t_loop, t_val, itr0 = _time.perf_counter(), 0.0, itr
    # ONE prefetcher (and one copy stream) for the run, re-iterated per epoch (a fresh one -- a new stream -- per epoch)
    prefetch = sdata.Prefetcher(train_loader, device)
    try:
        while not finished:
            if sampler is not None:
                sampler.set_epoch(opt.seed + itr)
            # batch i + 1 is copied to the device (and, for uint8 videos, collated there) on a copy stream under step i (data.Prefetcher)
            for batch in prefetch:
                if itr >= opt.n_iter:
                    finished = True
                    break
                itr += 1
                model.train()
                loss, nll, kl_y_0, kl_z = train(forward_fn, optimizer, None, batch, device, opt)
                if itr >= opt.lr_scheduling_burnin:
                    lr_scheduler.step()
                if local_rank > 0 and opt.chkpt_interval is not None and itr % opt.chkpt_interval == 0:
                    save_rank_rng(os.path.join(opt.save_path, 'train_state.pt'), local_rank, itr)
                if local_rank == 0:
                    if itr % opt.val_interval == 0 and val_loader is not None:
                        model.eval()
                        _tv = _time.perf_counter()
                        val_metric = evaluate(model, val_loader, device, opt)
                        t_val += _time.perf_counter() - _tv
                        if best_val_metric is None or best_val_metric > val_metric:
                            best_val_metric = val_metric
                            torch.save(model.state_dict(), os.path.join(opt.save_path, 'model_best.pt'))
                    if opt.chkpt_interval is not None and itr % opt.chkpt_interval == 0:
                        torch.save(model.state_dict(), os.path.join(opt.save_path, f'model_{itr}.pt'))
                        save_train_state(os.path.join(opt.save_path, 'train_state.pt'), model, optimizer, lr_scheduler, itr,
                                         best_val_metric)
                if sync is not None and (itr % opt.val_interval == 0
                                         or (opt.chkpt_interval is not None and itr % opt.chkpt_interval == 0)):
                    sync.after_rank0_phase()       # (no-op unless the statistics exchange waits on the device with a deadline)
                if local_rank == 0:
                    if itr % 50 == 0 or itr == 1:
                        print(f'itr {itr}: loss {loss:.3f} nll {nll:.3f} kl_y_0 {kl_y_0:.4f} kl_z {kl_z:.4f} '
                              f'val {val_metric} best {best_val_metric}', flush=True)
    except KeyboardInterrupt:
        status_code = 130
    wd = model.__dict__.get('_watchdog')
    if wd is not None:
        wd.stop()                                  # nothing may os._exit() during the final checkpoint writes below
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if itr > itr0 + 1:
        dt_loop = _time.perf_counter() - t_loop
        print(f'{itr - itr0} iterations in {dt_loop:.1f} s: {1e3 * (dt_loop - t_val) / (itr - itr0):.2f} ms / iteration (data + step + logging), '
              f'validation {t_val:.1f} s', flush=True)
    print('Saving...')
    if local_rank > 0:
        save_rank_rng(os.path.join(opt.save_path, 'train_state.pt'), local_rank, itr)
    if local_rank == 0:
        torch.save(model.state_dict(), os.path.join(opt.save_path, 'model.pt'))
        save_train_state(os.path.join(opt.save_path, 'train_state.pt'), model, optimizer, lr_scheduler, itr, best_val_metric)
    print('Done')
    return status_code


if __name__ == '__main__':
    from . import args as sargs
    from .helper import DotDict
    p = sargs.create_args()
    opt = DotDict(vars(p.parse_args()))
    if int(os.environ.get('LOCAL_RANK', opt.local_rank or 0)) != 0:
        sys.stdout = open(os.devnull, 'w')
    sys.exit(main(opt))
