"""
Headline benchmark of the SRVP training hot path on MI355X (BASELINE.json metric):

    training frames/sec (B x T) per node, BAIR VGG-64 (vgg + skipco, nc=3, seq_len=12, n_euler=2, batch 192 per GPU)

One "step" = one full optimisation step of reference train.py:49-129 -- forward, ELBO, backward, Adam -- through the
HIP kernels, on a synthetic (T, B, C, 64, 64) batch already resident in HBM.  `python bench.py --gpus N --steps K
--warmup W` (for N > 1 launched by torch.distributed.run, one rank per GPU over RCCL); rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     : the dominant kernel class (srvp_conv_mfma: forward + data-gradient implicit GEMMs), its algorithmic
                 FLOPs per step (2*MACs of the reference layer definitions) divided by its summed launch
                 durations measured with HIP events on the launch stream during the timed steps; peak = dense bf16 MFMA.
  cpu_baseline : the CPU oracle (port of the reference path, fixture-verified) timed on the host cores on a bounded
                 sample of the same workload (reduced batch: frames/s is batch-independent on CPU; B=192 needs > 60 GB).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
CONFIGS = {
    # name: (ctor args in the reference's positional order, T, hyper-parameters)  -- README.md:111-128 recipes
    'bair': dict(ctor=(64, 3, 64, 128, 50, 50, True, 2, 256, 3, 512, 4, 'vgg'), T=12, n_euler=2, obs_scale=0.71, beta_z=1.0,
                 res_gain=1.41, batch=192, label='BAIR 64x64x3 vgg+skipco seq_len=12 n_euler=2'),
    'kth': dict(ctor=(64, 1, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg'), T=20, n_euler=2, obs_scale=0.2, beta_z=1.0,
                res_gain=1.2, batch=100, label='KTH 64x64x1 vgg+skipco seq_len=20 n_euler=2'),
    # Human3.6M (BASELINE.json config 5): training T=16, B=100; test.py protocol: 8 conditioning frames -> 53 frames, 100 samples
    'human': dict(ctor=(64, 3, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg'), T=16, n_euler=2, obs_scale=0.2, beta_z=1.0,
                  res_gain=1.2, batch=100, label='Human3.6M 64x64x3 vgg+skipco seq_len=16 n_euler=2',
                  nt_cond=8, nt_test=53, batch_test=8, n_samples=100),
    'smmnist': dict(ctor=(64, 1, 64, 128, 20, 20, False, 5, 256, 3, 512, 4, 'dcgan'), T=15, n_euler=1, obs_scale=1.0, beta_z=2.0,
                    res_gain=1.41, batch=128, label='SM-MNIST 64x64x1 dcgan seq_len=15'),
}


def conv_flops(model, N_enc, N_dec):
    """Algorithmic FLOPs (2*MACs of the reference layer definitions) per step, split by kernel class."""
    out = dict(fwd_mfma=0.0, dgrad_mfma=0.0, wgrad_mfma=0.0, fwd_all=0.0, bytes_mfma=0.0)
    for blocks, N, enc in ((model._enc_blocks, N_enc, True), (model._dec_blocks, N_dec, False)):
        h = 64 if enc else 1
        for i, b in enumerate(blocks):
            if enc and b.get('pre') == 'pool':
                h //= 2
            hin = h
            if b['kind'] == 'conv':
                ho = (hin + 2 * b['p'] - b['k']) // b['s'] + 1
                macs = N * ho * ho * b['cout'] * b['cin'] * b['k'] ** 2
            else:
                ho = (hin - 1) * b['s'] - 2 * b['p'] + b['k']
                macs = N * hin * hin * b['cout'] * b['cin'] * b['k'] ** 2
            h = ho * (2 if (not enc and b.get('post_up')) else 1)
            f = 2.0 * macs
            out['fwd_all'] += f
            image_side = (enc and i == 0) or ((not enc) and i == len(blocks) - 1)
            if not image_side:
                out['fwd_mfma'] += f
                out['dgrad_mfma'] += f
                out['wgrad_mfma'] += f
                # ideal bf16 traffic of the forward + the data-gradient launch of the layer: input read once, output written
                # once (each way), weights once per launch
                hi = hin * (2 if (not enc and i > 0 and blocks[i - 1].get('post_up')) else 1)
                out['bytes_mfma'] += 2 * 2.0 * (N * (hi * hi * b['cin'] + ho * ho * b['cout']) + b['cout'] * b['cin'] * b['k'] ** 2)
    return out


def cpu_baseline(cfg, seconds=20.0):
    """Oracle train step (forward + ELBO + backward + Adam) on the host cores, reduced batch."""
    from oracle import srvp_oracle as O
    import srvp_amd
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)          # the CPU convolutions of a 48-frame batch stop scaling (and oversubscribe) beyond that
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
    m.init(cfg['res_gain'])
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ocfg = O.make_cfg(*cfg['ctor'])
    T, B = cfg['T'], 4
    g = torch.Generator().manual_seed(123)
    x = torch.rand(T, B, cfg['ctor'][1], 64, 64, generator=g)
    hp = dict(obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0)
    nt_inf, ny, nz = cfg['ctor'][7], cfg['ctor'][4], cfg['ctor'][5]

    def tape():
        t = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:nt_inf] for _ in range(B)], 1),
                 eps_y0=torch.randn(B, ny, generator=g), eps_z=torch.randn(T - 1, B, nz, generator=g))
        if cfg['ctor'][6]:
            t['t_skip'] = torch.randint(T, (B,), generator=g)
        return t
    adam = {}
    O.train_step(sd, ocfg, x, cfg['n_euler'], tape(), hp, adam, 3e-4)      # warm-up
    t0, n = time.time(), 0
    while time.time() - t0 < seconds or n < 2:
        O.train_step(sd, ocfg, x, cfg['n_euler'], tape(), hp, adam, 3e-4)
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=B * T / dt, unit='frames/s', cores=cores, kind='port',
                sample=f'{n} oracle train steps (fwd+ELBO+bwd+Adam, fp32, torch {torch.__version__} CPU) of the same '
                       f'architecture at B={B}, T={T} ({B * T} frames/step, {dt:.2f} s/step)')


def rollout_cpu_baseline(cfg, nt_cond, nt, seconds=15.0):
    """The reference's protocol on the host cores through the oracle: one inference forward (encode + posterior on the conditioning
    frames + prior rollout + decode of all nt frames) per sampled future, B = 1 video."""
    from oracle import srvp_oracle as O
    import srvp_amd
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    m = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
    m.init(cfg['res_gain'])
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ocfg = O.make_cfg(*cfg['ctor'])
    g = torch.Generator().manual_seed(5)
    x = torch.rand(nt_cond, 1, cfg['ctor'][1], 64, 64, generator=g)
    ny, nz = cfg['ctor'][4], cfg['ctor'][5]
    t0, n = time.time(), 0
    with torch.no_grad():
        while time.time() - t0 < seconds or n < 2:
            tape = dict(eps_y0=torch.randn(1, ny, generator=g), eps_z=torch.randn(nt - 1, 1, nz, generator=g))
            O.forward(sd, ocfg, x, nt, cfg['n_euler'], tape, training=False)
            n += 1
    dt = (time.time() - t0) / n
    return dict(value=nt / dt, unit='frames/s', cores=cores, kind='port',
                sample=f'{n} oracle inference forwards (fp32, torch {torch.__version__} CPU): 1 video, {nt_cond} conditioning frames -> '
                       f'{nt} decoded frames per sampled future ({dt:.2f} s each)')


def rollout_bench(args, cfg, dev, world, rank, real_stdout):
    """--mode rollout: reference test.py:219-246 / train.py:170-174 (best-of-S prediction) as model.sample launches."""
    import srvp_amd
    from srvp_amd import _lib as L
    nt_cond, nt = cfg.get('nt_cond', 2), cfg.get('nt_test', 30)
    B = args.batch if args.batch is not None else cfg.get('batch_test', 16)
    S = args.samples if args.samples is not None else cfg.get('n_samples', 100)
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
    model.init(res_gain=cfg['res_gain'])
    model.to(dev).eval()
    g = torch.Generator().manual_seed(123 + rank)
    x = torch.rand(nt_cond, B, cfg['ctor'][1], 64, 64, generator=g).to(dev)
    lim = int(os.environ.get('SRVP_EVAL_FRAMES', 9216))
    chunk = max(1, min(S, lim // (nt * B)))
    nchunks = -(-S // chunk)
    dt_e = 1.0 / cfg['n_euler']

    def step():
        # one encoding + one latent pass for all S futures, decoded `chunk` futures at a time (as train.evaluate does)
        model.sample(x, nt, S, dt=dt_e, chunk=chunk)
    for _ in range(max(1, args.warmup)):
        step()
    prof = None
    if not args.no_kernel_timing:
        L.PROFILE, L.PROFILE_ONLY = {}, {'srvp_conv_mfma', 'srvp_conv_mfma_multi'}
        step()                                       # first timing event of the process: outside the timed region
        L.PROFILE, L.PROFILE_ONLY = None, None
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = tmax.item()
    if not args.no_kernel_timing:
        prof = L.PROFILE = {}
        L.PROFILE_ONLY = None
        step()
        torch.cuda.synchronize()
        L.PROFILE = None
    if rank != 0:
        torch.distributed.destroy_process_group()
        return
    frames_step = S * B * nt
    fl = conv_flops(model, nt_cond * B, nt * B * chunk)          # per decoder pass (the encoder runs once per step: counted nchunks times, <1 %)
    line = {
        'metric': f'rollout decoded frames/sec ({cfg["label"].split(" seq_len")[0]}, {nt_cond} conditioning frames -> {nt}-frame horizon, {S} futures per video)',
        'value': frames_step * world * args.steps / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': model.precision,
        'data': 'synthetic (uniform random conditioning frames, random-init weights)',
        'config': {'workload': f'{cfg["label"].split(" seq_len")[0]} rollout: nt_cond={nt_cond}, seq_len_test={nt}, n_samples={S}, test batch {B}',
                   'per_gpu_batch': B, 'samples_per_call': chunk, 'calls_per_step': nchunks, 'parallelism': f'dp{world} (replicas)',
                   'step': 'model.sample: one encoding of the conditioning frames, posterior on them, prior rollout and decoding of every '
                           'frame of every future (reference test.py:237-246 runs one inference forward + generate + decode per future)'},
        'model_flops_frac_of_bf16_peak': ((fl['fwd_all']) * nchunks * world * args.steps / dt) / (PEAK_BF16_TFLOPS * 1e12 * world),
    }
    if prof:
        full = {name: sum(a.elapsed_time(b) for a, b in evs) for name, evs in prof.items()}
        conv_ms = full.get('srvp_conv_mfma', 0.0) + full.get('srvp_conv_mfma_multi', 0.0)
        nl = len(prof.get('srvp_conv_mfma', [])) + len(prof.get('srvp_conv_mfma_multi', []))
        ach = fl['fwd_mfma'] * nchunks / (conv_ms * 1e-3) / 1e12
        line['roofline'] = {'bound': 'mfma', 'kernel': 'conv_halo_kernel / conv_mfma_kernel (srvp_conv_mfma: decoder + encoder forward implicit GEMMs)',
                            'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS, 'traffic': None,
                            'algorithmic_flops_per_launch': fl['fwd_mfma'] * nchunks / max(1, nl), 'launches_per_step': nl,
                            'ms_per_step': conv_ms, 'avg_launch_us': conv_ms / max(1, nl) * 1e3,
                            'note': 'algorithmic = 2*MACs of the reference layer definitions for every decoded frame; the hoisted skip '
                                    'half and the sub-pixel convolutions execute fewer'}
        line['kernel_ms_per_step'] = {k: round(v, 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1])}
        line['kernel_ms_total'] = round(sum(full.values()), 3)
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = rollout_cpu_baseline(cfg, nt_cond, nt)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(line) + '\n').encode())


def self_launch(n):
    """Re-exec this command line under torch.distributed.run with n ranks; returns its exit status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    # the launcher's own chatter (warnings about OMP_NUM_THREADS etc.) goes to stderr already; the children inherit fd 1
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f'bench.py: the {n}-rank launch exited with status {rc}', file=sys.stderr)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='bair', choices=list(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (weak scaling) unless --global-batch is given; default: the '
                    "config's recipe batch (bair 192, kth 100, smmnist 128: BASELINE.json configs 4, 3, 2)")
    ap.add_argument('--global-batch', type=int, default=None, help='fixed global batch split over the ranks (strong scaling)')
    ap.add_argument('--no-strong', action='store_true', help='N > 1: skip the extra strong-scaling measurement (global batch 192 split over the ranks)')
    ap.add_argument('--h2d', choices=['none', 'u8'], default='none',
                    help="u8: every step starts from a pinned uint8 host batch (H2D copy + device-side /255 collate inside the timed "
                         "region, as reference train.py:84 pays it); default: batch resident in HBM (the contract's `value`)")
    ap.add_argument('--mode', choices=['train', 'rollout'], default='train',
                    help="rollout: the long-horizon prediction protocol of reference test.py:219-246 (config 5: --config human -> 8 conditioning "
                         "frames, 53-frame horizon, 100 sampled futures per video, test batch 8) through model.sample; one step = the "
                         "S futures of one test batch, value = decoded frames/s")
    ap.add_argument('--samples', type=int, default=None, help='rollout mode: futures per video (default: the recipe, 100)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-unshared', action='store_true', help='skip the extra untimed steps with the second stream off (a rocprofv3 --stats run of this '
                    'command then tabulates default-schedule steps only)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks ourselves (one process per GPU under torch.distributed.run,
        # rendezvous on 127.0.0.1 at a free port -- the reference's launcher, README.md:99-103 / train.py:205-219); rank 0 of the
        # children prints the ONE JSON line on the stdout they inherit from this process.
        return self_launch(args.gpus)
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner through C stdio at
    # communicator creation): everything that goes to file descriptor 1 during the run is sent to stderr, and the JSON line is
    # written to the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import srvp_amd
    from srvp_amd import _lib as L
    from srvp_amd import distributed as sdist
    from srvp_amd.train import train

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'
    # SRVP_DIST_BACKEND=gloo: diagnostic only -- lets the N>1 control flow be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices, which RCCL refuses); the measured configuration is always RCCL, one rank per GPU
    backend = os.environ.get('SRVP_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sync = None
    if world > 1 or os.environ.get('SRVP_FORCE_COLLECTIVES') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        sync = sdist.init_process_group(backend)

    cfg = CONFIGS[args.config]
    if args.mode == 'rollout':
        return rollout_bench(args, cfg, dev, world, rank, real_stdout)
    T = cfg['T']
    # N > 1: the headline is BASELINE.json's config as the reference runs it -- ONE global batch (bair: 192) split over the ranks
    # (train.py:218-219), i.e. strong scaling; the weak line (the recipe batch on every GPU) is reported beside it as `weak_scaling`.
    # An explicit --batch / --global-batch / --no-strong keeps the old meaning (per-GPU batch fixed: weak).
    headline_strong = (world > 1 and args.batch is None and args.global_batch is None and not args.no_strong
                       and cfg['batch'] % world == 0)
    if headline_strong:
        args.global_batch = cfg['batch']
    if args.batch is None:
        args.batch = cfg['batch']
    B = args.batch if args.global_batch is None else args.global_batch // world
    scaling = 'weak' if args.global_batch is None else 'strong'
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*cfg['ctor'])
    model.init(res_gain=cfg['res_gain'])
    model.to(dev).train()
    fwd = sdist.DataParallel(model, sync) if sync is not None else model
    optim = srvp_amd.FusedAdam(model, lr=3e-4)
    opt = srvp_amd.DotDict(dict(n_euler_steps=cfg['n_euler'], obs_scale=cfg['obs_scale'], beta_y=1.0, beta_z=cfg['beta_z'], l2_res=1.0))
    g = torch.Generator().manual_seed(123 + rank)
    x = torch.rand(T, B, cfg['ctor'][1], 64, 64, generator=g).to(dev)          # synthetic batch, resident in HBM
    if args.h2d == 'u8':
        # the reference's loader hands over uint8 videos (data/base.py:71-84): stacked [B][T][H][W][C] in pinned host memory;
        # srvp_amd.train.train() copies them and finishes the collate (transpose + /255) on the device
        x = (torch.rand(B, T, 64, 64, cfg['ctor'][1], generator=g) * 255).to(torch.uint8).pin_memory()
    # --h2d u8: the step's input comes through srvp_amd.data.Prefetcher, as in srvp_amd.train.main: the pinned batch of step i + 1 is copied
    # and collated on a copy stream under step i (SRVP_BENCH_H2D_INLINE=1: the round-3 form, copy + collate in line at the top of the step)
    feed = None
    if args.h2d == 'u8' and os.environ.get('SRVP_BENCH_H2D_INLINE') != '1':
        import itertools
        from srvp_amd.data import Prefetcher
        feed = iter(Prefetcher(itertools.repeat(x), dev))
    batch = (lambda: next(feed)) if feed is not None else (lambda: x)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = train(fwd, optim, None, batch(), dev, opt)
    # HIP events around the launches of the two dominant kernel classes only (~110 per step).  Round 6: the
    # instrumented steps are NOT part of the timed region any more (each event record is a marker packet in the stream: ~1 ms per
    # instrumented step) -- they are N_EVENT_STEPS extra untimed steps of the same schedule directly after the timed loop (same process,
    # same box, same clocks), so `roofline` is still measured in-step while the headline stops paying for its own instrumentation.
    N_EVENT_STEPS = int(os.environ.get('SRVP_BENCH_EVENT_STEPS', 3))
    timing = not args.no_kernel_timing
    prof, n_prof_steps = ({} if timing else None), 0
    if timing:
        # one instrumented step before the timed region: the first timing event of a process costs ~75-100 ms once (the runtime
        # switches the queue to profiling mode); taken here so that neither the timed steps nor the roofline steps see the switch
        L.PROFILE, L.PROFILE_ONLY = {}, {'srvp_conv_mfma', 'srvp_conv_mfma_multi', 'srvp_wgrad_mfma'}
        train(fwd, optim, None, batch(), dev, opt)
        L.PROFILE, L.PROFILE_ONLY = None, None
    # Same garbage-collector treatment as a real run (srvp_amd.train.main: gc.collect(); gc.freeze() after set-up, collector left
    # ON): the long-lived object graph (model, plans, descriptors) moves to the permanent generation, so the collections the
    # ~100 event objects of an instrumented step trigger stay young-generation and cheap -- without it one of them walks the whole
    # graph, a 90 ms host stall = +7 ms per step of a 10-step run of the short-step configs.
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = train(fwd, optim, None, batch(), dev, opt)
        if os.environ.get('SRVP_BENCH_TRACE'):
            print(f'step {i} host t={1e3 * (time.perf_counter() - t0):.2f} ms', file=sys.stderr)
    barrier()
    dt = time.perf_counter() - t0
    if timing:
        # the roofline's steps: same schedule, events around the two dominant classes only
        for _ in range(N_EVENT_STEPS):
            L.PROFILE, L.PROFILE_ONLY = prof, {'srvp_conv_mfma', 'srvp_conv_mfma_multi', 'srvp_wgrad_mfma'}
            n_prof_steps += 1
            train(fwd, optim, None, batch(), dev, opt)
            L.PROFILE, L.PROFILE_ONLY = None, None
        torch.cuda.synchronize()
    table = None
    if prof is not None and rank == 0:
        L.PROFILE = {}                      # untimed extra pass: every launch
        for _ in range(2):
            train(fwd, optim, None, batch(), dev, opt)
        torch.cuda.synchronize()
        table, L.PROFILE = L.PROFILE, None
    elif prof is not None:
        for _ in range(2):
            train(fwd, optim, None, batch(), dev, opt)
    # context for `roofline.achieved`: the same launches with NOTHING running beside them.  Since round 3 every weight gradient runs
    # on the second stream, concurrently with the data-gradient / BatchNorm kernels of the main stream (a faster step, and launch
    # durations that include the sharing); two more untimed steps with the second stream switched off give the kernels' own rate.
    unshared = None
    if prof is not None and not args.no_unshared:
        import srvp_amd.model as _m
        import srvp_amd.convnet as _cn
        saved = (_m.OVERLAP_WGRAD, _m.OVERLAP_SKIP, _m.OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN)
        _m.OVERLAP_WGRAD, _m.OVERLAP_SKIP, _m.OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN = False, False, False, 0
        try:
            train(fwd, optim, None, batch(), dev, opt)
            L.PROFILE, L.PROFILE_ONLY = {}, {'srvp_conv_mfma', 'srvp_conv_mfma_multi', 'srvp_wgrad_mfma'}
            for _ in range(2):
                train(fwd, optim, None, batch(), dev, opt)
            torch.cuda.synchronize()
            unshared, L.PROFILE, L.PROFILE_ONLY = L.PROFILE, None, None
        finally:
            _m.OVERLAP_WGRAD, _m.OVERLAP_SKIP, _m.OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN = saved
            L.PROFILE, L.PROFILE_ONLY = None, None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = tmax.item()
    # ---- N > 1: the other scaling mode beside the headline.  Headline strong (default): the recipe batch on EVERY GPU (weak scaling,
    # global batch N x 192 -- a batch the reference never runs, kept as the per-GPU-work-fixed line).  Headline weak (--batch given):
    # the reference's DDP split of ONE global batch of 192 (strong).
    def side_run(Bs):
        xs = torch.rand(T, Bs, cfg['ctor'][1], 64, 64, generator=g).to(dev)
        for _ in range(max(2, args.warmup)):
            train(fwd, optim, None, xs, dev, opt)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            train(fwd, optim, None, xs, dev, opt)
        barrier()
        ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(ts, op=torch.distributed.ReduceOp.MAX)
        return ts.item()
    strong = weak = None
    if world > 1 and headline_strong:
        tw = side_run(cfg['batch'])
        weak = dict(scaling='weak', global_batch=cfg['batch'] * world, per_gpu_batch=cfg['batch'], ms_per_step=tw / args.steps * 1e3,
                    value=cfg['batch'] * world * T * args.steps / tw, unit='frames/s')
    elif world > 1 and not args.no_strong and args.global_batch is None and cfg['batch'] % world == 0:
        Bs = cfg['batch'] // world
        tsx = side_run(Bs)
        strong = dict(scaling='strong', global_batch=cfg['batch'], per_gpu_batch=Bs, ms_per_step=tsx / args.steps * 1e3,
                      value=cfg['batch'] * T * args.steps / tsx, unit='frames/s')
    # ---- N > 1 (or SRVP_FORCE_COLLECTIVES=1): what the exchange itself costs on the transports the step used (every rank takes part): the
    # transport per exchange, RCCL's own rank count per communicator, the in-stream latency of one statistics all-reduce, the bandwidth of one
    # 95 MB gradient all-reduce -- so that a scaling curve explains itself (srvp_amd.distributed.Sync.diagnostics)
    comm_diag = None
    if sync is not None:
        try:
            comm_diag = sync.diagnostics()
        except Exception as exc:           # a diagnostic must never take the bench line down
            comm_diag = {'error': str(exc)}
        wd = model.__dict__.get('_watchdog')
        if wd is not None:
            comm_diag['step_watchdog_s'] = wd.timeout_s
    if rank != 0:
        torch.distributed.destroy_process_group()
        return
    frames = B * T * world
    ms_step = dt / args.steps * 1e3
    fl = conv_flops(model, T * B, T * B)
    line = {
        'metric': 'training frames/sec (BxT) per node, BAIR VGG-64 seq_len=12' if args.config == 'bair' else f'training frames/sec ({cfg["label"]})',
        'value': frames * args.steps / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': model.precision,
        'data': 'synthetic (uniform random frames, random-init weights)',
        'config': {'workload': cfg['label'], 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': T,
                   'parallelism': f'dp{world}', 'step': 'forward + ELBO + backward + Adam (reference train.py:49-129)',
                   'input': ('uint8 pinned host batch per step; H2D + device collate ' + ('on a copy stream under the previous step (srvp_amd.data.Prefetcher)' if feed is not None else 'in line at the top of the step'))
                            if args.h2d == 'u8' else 'float32 batch resident in HBM',
                   'collectives': sync.transport if sync is not None else None},
        'loss': loss[0] if loss else None,
        'model_flops_frac_of_bf16_peak': (3 * fl['fwd_all'] * world * args.steps / dt) / (PEAK_BF16_TFLOPS * 1e12 * world),
    }
    if strong is not None:
        line['strong_scaling'] = strong
    if weak is not None:
        line['weak_scaling'] = weak
    if comm_diag is not None:
        line['comm'] = comm_diag
    if prof:
        per = {}
        for name, evs in prof.items():
            per[name] = sum(a.elapsed_time(b) for a, b in evs) / n_prof_steps    # ms per (instrumented) step
        dom = 'srvp_conv_mfma'
        # (srvp_conv_mfma_multi = the same kernels, four sub-pixel phase launches issued as one grid)
        per[dom] = per.get(dom, 0.0) + per.pop('srvp_conv_mfma_multi', 0.0)
        # kernel launches per step as rocprofv3 counts them (the four sub-pixel phases of a multi call are ONE grid)
        nlaunch = (len(prof.get(dom, [])) + len(prof.get('srvp_conv_mfma_multi', []))) // n_prof_steps
        ach = (fl['fwd_mfma'] + fl['dgrad_mfma']) / (per[dom] * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel class: PMC counters cannot be collected from inside this process, so the
        # number comes from the committed summary of a separate `rocprofv3 --pmc` pass of this very command
        # (tools/hbm_traffic.py -> profiles/r01_hbm_traffic.json, corrections of MI355X_MICROARCH.md's HBM section)
        traffic, tpath = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_hbm_traffic.json')))
        if args.config == 'bair' and B == 192 and cands:
            tpath = cands[-1]                                   # the latest round's PMC pass
            traffic = json.load(open(tpath))['srvp_conv_mfma']['bytes_per_launch']
        # what the matrix cores actually execute: MACs of the launch descriptors (padded channels, hoisted skip half evaluated
        # once per sample, sub-pixel upsample convolutions) -- less than the reference's layer definitions ask for
        pl = model._last_plan
        ex_macs = 0
        for net in (pl['enc'], pl['dec']):
            for blk in net.blocks:
                if blk.role == 'in' or (blk.role == 'out' and net._f32_out() if hasattr(net, '_f32_out') else False):
                    continue
                for d in list(getattr(blk, '_fwd', [])) + list(getattr(blk, '_dg', [])):
                    ex_macs += d.N * d.OH * d.OW * d.Cout * d.ntaps * (d.C0 + d.C1)
                if blk.role == 'out':                           # (its data-gradient runs on the fp32 image-side kernel instead)
                    for d in getattr(blk, '_dg', []):
                        ex_macs -= d.N * d.OH * d.OW * d.Cout * d.ntaps * (d.C0 + d.C1)
        ex = 2.0 * ex_macs / (per[dom] * 1e-3) / 1e12
        line['roofline'] = {'bound': 'mfma', 'kernel': 'conv_halo_kernel / conv_mfma_kernel (srvp_conv_mfma: forward + data-gradient implicit GEMMs)',
                            'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                            'executed_flops_per_step': 2.0 * ex_macs, 'executed': ex, 'executed_frac': ex / PEAK_BF16_TFLOPS,
                            'traffic': traffic, 'traffic_unit': f'HBM bytes per launch (rocprofv3 PMC pass, {os.path.relpath(tpath, ROOT) if tpath else None})',
                            'algorithmic_flops_per_launch': (fl['fwd_mfma'] + fl['dgrad_mfma']) / max(1, nlaunch),
                            'algorithmic_bytes_per_launch': fl['bytes_mfma'] / max(1, nlaunch),
                            'launches_per_step': nlaunch, 'ms_per_step': per[dom], 'avg_launch_us': per[dom] / max(1, nlaunch) * 1e3,
                            'timed_with_events': f'0 of the {args.steps} timed steps: {n_prof_steps} extra untimed steps of the same schedule directly after the timed loop (same process)'}
        if unshared and rank == 0:
            um = sum(a.elapsed_time(b) for k in ('srvp_conv_mfma', 'srvp_conv_mfma_multi') for a, b in unshared.get(k, [])) / 2
            uw = sum(a.elapsed_time(b) for a, b in unshared.get('srvp_wgrad_mfma', [])) / 2
            line['roofline']['unshared'] = {
                'what': 'the same launches in 2 extra untimed steps with the second stream switched off (nothing runs beside them): the '
                        "kernels' own rate; `achieved` above is measured inside the timed steps, where the weight gradients of the second "
                        'stream share the chip with these launches',
                'ms_per_step': um, 'achieved': (fl['fwd_mfma'] + fl['dgrad_mfma']) / (um * 1e-3) / 1e12,
                'frac': (fl['fwd_mfma'] + fl['dgrad_mfma']) / (um * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                'wgrad_ms_per_step': uw, 'wgrad_achieved': fl['wgrad_mfma'] / (uw * 1e-3) / 1e12 if uw else None}
        wg = fl['wgrad_mfma'] / (per['srvp_wgrad_mfma'] * 1e-3) / 1e12
        line['roofline_wgrad'] = {'bound': 'mfma', 'kernel': 'wgrad_halo_kernel / wgrad_mfma_kernel', 'achieved': wg, 'peak': PEAK_BF16_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': wg / PEAK_BF16_TFLOPS, 'ms_per_step': per['srvp_wgrad_mfma']}
        if table:
            full = {name: sum(a.elapsed_time(b) for a, b in evs) / 2 for name, evs in table.items()}
            line['kernel_ms_per_step'] = {k: round(v, 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1])}
            line['kernel_ms_total'] = round(sum(full.values()), 3)
    if prof:
        # context for `frac`: what the vendor GEMM library (hipBLASLt through torch.matmul) sustains on this very device for
        # a large bf16 GEMM -- clocks and power limit included.  A measuring stick only; nothing in the product calls it.
        try:
            ga = torch.randn(16384, 4608, device=dev, dtype=torch.bfloat16)
            gb = torch.randn(4608, 8192, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                ga @ gb
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ga @ gb
            e1.record()
            torch.cuda.synchronize()
            line['roofline']['vendor_gemm_tflops'] = 10 * 2 * 16384 * 8192 * 4608 / (e0.elapsed_time(e1) * 1e-3) / 1e12
            line['roofline']['vendor_gemm'] = 'hipBLASLt bf16 16384x8192x4608 via torch.matmul, same device, same run'
        except Exception as exc:   # the measuring stick must never take the bench line down
            line['roofline']['vendor_gemm_tflops'] = None
            line['roofline']['vendor_gemm'] = f'unavailable: {exc}'
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(cfg)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(line) + '\n').encode())


if __name__ == '__main__':
    main()
