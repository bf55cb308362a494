"""
StochasticLatentResidualVideoPredictor -- MI355X-native drop-in for the reference class of the same name
(reference module/srvp.py:29-470): same constructor signature, public attributes, method names / shapes and
state-dict keys, with all arithmetic in hand-written HIP kernels (libsrvp_hip.so).

The nn.Conv2d / nn.BatchNorm2d / nn.Linear / nn.LSTM objects built here are *parameter containers only*: they
reproduce the reference's module hierarchy (hence its state-dict keys, its default-initialisation RNG order and
compatibility with `SyncBatchNorm.convert_sync_batchnorm`), but their forward methods are never called.

Differences from the reference, all documented in DESIGN.md:
  * random draws can be supplied explicitly (`tape=`) so that results are reproducible against the CPU oracle;
    by default they are drawn in the reference's order (CPU generator for indices, device generator for normals);
  * `forward` is differentiable through one fused autograd node; the granular methods (`encode`, `infer_*`,
    `generate`, `decode`) run the same kernels without recording a graph (inference / test.py call pattern);
  * the Euler schedule is the integer one (identical to the reference for n_euler in {1, 2, 4, ...}).
"""
import math

import os

import torch
import torch.nn as nn

from . import _lib as L
from . import arch
from .convnet import EncoderNet, DecoderNet, cpad
from .latent import LatentNet


def _set_module(root, path, module):
    """Places `module` at dotted `path` under `root`, creating nn.Sequential / nn.ModuleList containers like the
    reference hierarchy (integer path elements index containers)."""
    parts = path.split('.')
    cur = root
    for i, p in enumerate(parts[:-1]):
        nxt = getattr(cur, p, None) if not p.isdigit() else (cur._modules.get(p))
        if nxt is None:
            nxt = nn.Module()
            cur.add_module(p, nxt)
        cur = nxt
    cur.add_module(parts[-1], module)


def _to_dev(v, dev):
    if v.device == dev:
        return v
    if v.device.type == 'cpu' and dev.type == 'cuda':
        return v.pin_memory().to(dev, non_blocking=True)
    return v.to(dev)


OVERLAP_WGRAD = os.environ.get('SRVP_OVERLAP_WGRAD', '1') != '0'
PZ_AUX = os.environ.get('SRVP_PZ_AUX', '1') != '0'                       # 0: the batched prior MLP of a training forward in line
SKIP_REDUCE_AUX = os.environ.get('SRVP_SKIP_REDUCE_AUX', '1') != '0'      # 0: the pooled stages' skip-gradient reductions in line on the main stream
LATENT_AUX = os.environ.get('SRVP_LATENT_AUX', '1') != '0'        # independent chains of the latent path (posterior / w / y_0; their backward) on two streams
PZ_BWD_AUX = os.environ.get('SRVP_PZ_BWD_AUX', '1') != '0'          # the prior MLP's backward on the auxiliary stream under the decoder backward
DEC_ALLREDUCE_STREAM = os.environ.get('SRVP_DEC_ALLREDUCE_STREAM', '1') != '0'   # N > 1: the gradient slices' all-reduces on a stream of their own
GRAD_SLICES = os.environ.get('SRVP_GRAD_SLICES', '1') != '0'      # N > 1: gradient exchange slice by slice under the backward (0: decoder slice + the rest at the end)
ENC_WGRAD_STREAM2 = os.environ.get('SRVP_ENC_WGRAD_STREAM2', '0') != '0'   # the encoder's weight gradients on a third stream (not behind the decoder's)
LATENT_WGRAD_STREAM = os.environ.get('SRVP_LATENT_WGRAD_STREAM', '1') != '0'    # the latent networks' weight gradients on a stream of their own
OVERLAP_SKIP = os.environ.get('SRVP_OVERLAP_SKIP', '1') == '1'
# training: hoisted skip convs stage by stage under the encoder forward (overrides SKIP_LATE).  Measured (round 5, same box, ms per step on / off):
# 192 sequences 37.57 / 37.67, 24 sequences 7.34 / 7.25, KTH 32.79 / 32.60, Human3.6M 27.13 / 26.98: off
SKIP_EARLY = os.environ.get('SRVP_SKIP_EARLY', '0') == '1'
SKIP_LATE = os.environ.get('SRVP_SKIP_LATE', '1') == '1'        # hoisted skip convs under the rollout kernel (1) / under the inference chain (0)
OVERLAP_PACK = os.environ.get('SRVP_OVERLAP_PACK', '1') == '1'    # decoder weight packing on the second stream, under the encoder


SIDE_PRIORITY = os.environ.get('SRVP_SIDE_PRIORITY', 'default')      # 'low' / 'default' / 'high' (A/B switch)


# deterministic parity mode: process-wide state shared by the models that are in it (StochasticLatentResidualVideoPredictor.set_deterministic)
_DET = dict(count=0, saved=None, ws=None)


def _make_side_stream():
    """The second stream (weight gradients, weight packing, hoisted skip convolutions: everything off the critical path).
    Measured at three stream priorities (same box, ms per step at 192 sequences / 24 sequences): lowest priority of the device
    (srvp_stream_create_low_priority) 40.76 / 8.78 -- the deprioritised weight gradients pile up behind the main stream's last kernel --
    default 40.60 / 8.76: default kept."""
    if SIDE_PRIORITY == 'low':
        import ctypes as C
        h, pr = C.c_void_p(), C.c_int32()
        L.check(L.load().srvp_stream_create_low_priority(C.byref(h), C.byref(pr)), 'srvp_stream_create_low_priority')
        return torch.cuda.ExternalStream(h.value)
    if SIDE_PRIORITY == 'high':
        return torch.cuda.Stream(priority=-1)
    return torch.cuda.Stream()


class _Holder(nn.Module):
    """Generic container (stands for nn.Sequential / nn.ModuleList levels of the reference)."""


def _build_convnet(root, blocks):
    for b in blocks:
        if b['kind'] == 'conv':
            m = nn.Conv2d(b['cin'], b['cout'], b['k'], b['s'], b['p'], bias=False)
        else:
            m = nn.ConvTranspose2d(b['cin'], b['cout'], b['k'], b['s'], b['p'], bias=False)
        _set_module(root, b['key'], m)
        if b['bnkey'] is not None:
            _set_module(root, b['bnkey'], nn.BatchNorm2d(b['cout']))


def _mlp(n_inp, n_hid, n_out, n_layers):
    """Same parameter layout as reference module/mlp.py:47-90: module.0.0, module.1.1, module.2.1, ..."""
    root = _Holder()
    seq = _Holder()
    root.add_module('module', seq)
    for il in range(n_layers):
        blk = _Holder()
        lin = nn.Linear(n_inp if il == 0 else n_hid, n_out if il == n_layers - 1 else n_hid)
        blk.add_module('0' if il == 0 else '1', lin)
        seq.add_module(str(il), blk)
    return root


class _SrvpForward(torch.autograd.Function):
    """One autograd node for the whole training forward (srvp.py:415-470): backward runs the HIP backward chain and
    writes parameter gradients straight into the model's flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, x, nt, n_euler, tape, *params):
        outs = model._forward_impl(x, nt, n_euler, tape, training=True)
        ctx.model = model
        ctx.gen = model._fwd_gen
        # fresh tensors, like the reference's outputs: the plan-owned buffers are overwritten by the next forward
        outs = tuple(o.clone() if o is not None else None for o in outs)
        ctx.mark_non_differentiable(outs[2])      # z is a sample: gradient flows through q_z / p_z params
        return outs

    @staticmethod
    def backward(ctx, d_x, d_y, d_z, d_w, d_qy0, d_qz, d_pz, d_res):
        if ctx.gen != ctx.model._fwd_gen:
            raise RuntimeError('srvp_amd: backward() of a forward pass whose saved activations were overwritten by a later '
                               'training-mode forward of the same model (activation buffers are preallocated per problem '
                               'size and shared between calls): run backward before the next forward')
        ctx.model._backward_impl(d_x, d_y, d_w, d_qy0, d_qz, d_pz, d_res)
        return (None,) * (5 + len(list(ctx.model.parameters())))


class StochasticLatentResidualVideoPredictor(nn.Module):
    def __init__(self, nx, nc, nf, nhx, ny, nz, skipco, nt_inf, nh_inf, nlayers_inf, nh_res, nlayers_res, archi):
        super().__init__()
        assert nx == 64, 'the encoder / decoder families are 64x64 architectures (reference conv.py:45-48)'
        self.nx, self.nc, self.ny, self.nz, self.skipco = nx, nc, ny, nz, skipco
        self.nt_inf, self.nh_inf, self.nlayers_inf = nt_inf, nh_inf, nlayers_inf
        self.nh_res, self.nlayers_res, self.nhx = nh_res, nlayers_res, nhx
        self.nf, self.archi = nf, archi
        self._enc_blocks = arch.encoder_blocks(archi, nc, nhx, nf)
        self._dec_blocks = arch.decoder_blocks(archi, nc, nh_inf + ny, nf, skipco)
        # module construction order = reference order (srvp.py:124-137), so that default initialisation consumes the
        # global RNG identically
        self.encoder = _Holder()
        self.decoder = _Holder()
        _build_convnet(self, self._enc_blocks)
        _build_convnet(self, self._dec_blocks)
        self.w_proj = _Holder()
        self.w_proj.add_module('0', nn.Linear(nhx, nh_inf))
        self.w_inf = _Holder()
        self.w_inf.add_module('0', nn.Linear(nh_inf, nh_inf))
        self.q_y = _mlp(nhx * nt_inf, nh_inf, ny * 2, nlayers_inf)
        self.inf_z = nn.LSTM(nhx, nh_inf, 1)
        self.q_z = nn.Linear(nh_inf, nz * 2)
        self.p_z = _mlp(ny, nh_res, nz * 2, nlayers_res)
        self.dynamics = _mlp(ny + nz, nh_res, ny, nlayers_res)
        self._plans = {}
        self._flat = None
        self._pack_version = None
        self.sync = None            # set by srvp_amd.distributed for multi-GPU (SyncBN statistics + gradient all-reduce)
        self.last_tape = None
        self._fwd_gen = 0           # training forwards so far (guards backward against overwritten activations)
        # 'bf16' (production: bf16 MFMA operands / activation storage, fp32 accumulation) or 'fp32' (parity mode: every tensor
        # fp32, contractions in exact fp32 on the matrix cores) -- set_precision()
        self.precision = os.environ.get('SRVP_PRECISION', 'bf16')

    # ------------------------------------------------------------------------------------------------ init
    def init(self, res_gain=1.41):
        """srvp.py:139-154 / utils.py:51-85: N(0, 0.02) conv weights, N(1, 0.02) BN weights (zero bias) for the encoder and
        decoder; orthogonal(res_gain) weights and zero bias for `dynamics`."""
        for root in (self.encoder, self.decoder):
            for m in root.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                    nn.init.normal_(m.weight.data, 0.0, 0.02)
                    if getattr(m, 'bias', None) is not None:
                        nn.init.constant_(m.bias.data, 0.0)
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.normal_(m.weight.data, 1.0, 0.02)
                    nn.init.constant_(m.bias.data, 0.0)
        for m in self.dynamics.modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight.data, gain=res_gain)
                nn.init.constant_(m.bias.data, 0.0)

    def set_precision(self, precision):
        """'bf16': the production path.  'fp32': parity mode -- the conv stacks run on fp32 activations / gradients / packed
        weights with exact-fp32 MFMA contractions (csrc/conv_f32.hip), so that the whole pipeline can be held against the
        reference's fp32 arithmetic at 1e-5 (tests/test_gpu_fp32_mode.py); ~16x lower matrix throughput by construction."""
        assert precision in ('bf16', 'fp32'), precision
        if precision != self.precision:
            self.precision = precision
            self._plans = {}
            self._pack_version = None
        return self

    def set_deterministic(self, on=True):
        """Bit-reproducible runs of the fp32 parity mode (SRVP_DETERMINISTIC=1 switches it on at construction): every
        cross-workgroup sum that is normally formed with atomics in arrival order -- BatchNorm statistics, BatchNorm-backward sums,
        split-K weight gradients, the image-side weight gradient, the latent weight gradients, the ELBO accumulators -- is formed in a
        fixed order instead (csrc/common.h), and everything is issued on ONE stream.  Process-wide (the library holds the switch and
        the 8 MiB workspace of the two-launch reductions); the production bf16 path keeps its atomics and is refused here.
        The process-wide part (overlap switches, library switch, workspace) is held at MODULE level with a count of the models that are in
        the mode: it is entered by the first and left by the last (per-instance saves restored the wrong values with two models)."""
        from . import convnet as _cn
        global OVERLAP_WGRAD, OVERLAP_SKIP, OVERLAP_PACK
        on = bool(on)
        if on and self.precision != 'fp32':
            raise ValueError("deterministic mode covers precision = 'fp32' only: call set_precision('fp32') first")
        was = bool(getattr(self, 'deterministic', False))
        if on and not was:
            if _DET['count'] == 0:
                _DET['saved'] = (OVERLAP_WGRAD, OVERLAP_SKIP, OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN)
                OVERLAP_WGRAD, OVERLAP_SKIP, OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN = False, False, False, 0
                _cn.DETERMINISTIC = True
            _DET['count'] += 1
        if on:
            ws = _DET['ws']
            if ws is None or ws.device != self._device():
                ws = _DET['ws'] = torch.zeros(8 << 20, dtype=torch.uint8, device=self._device())
            L.call('srvp_set_deterministic', 1, L.ptr(ws), ws.numel())
        elif was:
            _DET['count'] -= 1
            if _DET['count'] == 0:
                L.call('srvp_set_deterministic', 0, None, 0)
                OVERLAP_WGRAD, OVERLAP_SKIP, OVERLAP_PACK, _cn.ENC_WGRAD_SIDE_MAXN = _DET['saved']
                _DET['saved'], _DET['ws'] = None, None
                _cn.DETERMINISTIC = False
        if not on:
            self._det_env_done = True            # an explicit set_deterministic(False) is not undone by SRVP_DETERMINISTIC=1 in _plan
        self.deterministic = on
        self._plans = {}
        self._pack_version = None
        return self

    # ------------------------------------------------------------------------------------------------ plumbing
    def _cfg(self):
        return dict(nx=self.nx, nc=self.nc, nf=self.nf, nhx=self.nhx, ny=self.ny, nz=self.nz, skipco=self.skipco,
                    nt_inf=self.nt_inf, nh_inf=self.nh_inf, nlayers_inf=self.nlayers_inf, nh_res=self.nh_res,
                    nlayers_res=self.nlayers_res, archi=self.archi)

    def _device(self):
        return self._enumerate()['plist'][0].device

    def _require_gpu(self):
        dev = self._device()
        if dev.type != 'cuda':
            raise L.SrvpHipError('srvp_amd runs on an MI355X only: move the model to the GPU (model.to("cuda")); '
                                 'there is no CPU compute path')
        L.load()
        return dev

    # ---- cached enumeration of the module tree.  nn.Module.parameters() / named_parameters() / named_buffers() walk the whole hierarchy
    # (~100 modules) on every call; the step asked for them ~10 times (1.5 ms of the 4 ms of host work per step at 24 sequences,
    # tools/host_profile.py).  The Parameter OBJECTS survive everything the training loop does to the model (.to() / .cuda() re-home
    # .data in place, load_state_dict copies in place, SyncBatchNorm conversion re-uses them); buffers are replaced by _apply(), so
    # the cache is dropped there.  Adding / removing submodules after the first step is not supported.
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__.pop('_enum', None)
        self._pack_version = None
        return r

    def _enumerate(self):
        e = self.__dict__.get('_enum')
        if e is None:
            named_p = list(self.named_parameters())
            d = dict(named_p)
            d.update(dict(self.named_buffers()))
            e = self.__dict__['_enum'] = dict(plist=[p for _, p in named_p], pnames=[k for k, _ in named_p], named=d)
        return e

    def flatten_parameters_(self):
        """Re-homes every parameter (and its gradient) into one flat fp32 buffer: one fused Adam launch and one
        contiguous all-reduce payload.  Parameter objects (and hence optimizers / state dicts) are unchanged."""
        ps = self._enumerate()['plist']
        dev = ps[0].device
        if self._flat is not None and self._flat[0].device == dev and all(
                p.data_ptr() == v.data_ptr() for p, v in zip(ps, self._flat[2])):
            return
        n = sum(p.numel() for p in ps)
        n_al = (n + 3) // 4 * 4
        flat_p = torch.zeros(n_al, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n_al, dtype=torch.float32, device=dev)
        views, gviews, off = [], [], 0
        for p in ps:
            v = flat_p[off:off + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            g = flat_g[off:off + p.numel()].view_as(p)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g
            views.append(v)
            gviews.append(g)
            off += p.numel()
        self._flat = (flat_p, flat_g, views, gviews)
        self._pack_version = None

    def _named_tensors(self):
        """name -> Parameter / buffer of the whole model (cached: _enumerate)."""
        return self._enumerate()['named']

    def _grads(self):
        """name -> gradient view inside the flat gradient buffer (re-attached if an optimizer set .grad to None)."""
        self.flatten_parameters_()
        _, flat_g, _, gviews = self._flat
        e = self._enumerate()
        fresh = False
        for p, g in zip(e['plist'], gviews):
            if p.grad is not g and (p.grad is None or p.grad.data_ptr() != g.data_ptr()):
                fresh = True
                break
        if fresh:
            flat_g.zero_()
            for p, g in zip(e['plist'], gviews):
                p.grad = g
        gd = e.get('gdict')
        if gd is None or gd[0] is not gviews:
            gd = e['gdict'] = (gviews, dict(zip(e['pnames'], gviews)))
        return gd[1]

    def _plan(self, T, B, nt, n_euler, training, S=1, S_lat=None):
        """S > 1 (inference only): the conditioning frames are encoded once for B videos, the latent path and the decoder run
        on B*S (video, sample) rows -- row s*B + b -- sharing the B skip tensors / hoisted skip halves through the image maps.
        S_lat > S: the latent path carries S_lat samples per video at once (its chains are latency-bound: 800 rows cost what 160
        do) while the decoder, whose activations set the memory footprint, works through them S at a time."""
        if (os.environ.get('SRVP_DETERMINISTIC') == '1' and not getattr(self, 'deterministic', False) and self.precision == 'fp32'
                and not getattr(self, '_det_env_done', False)):
            self._det_env_done = True                     # (the environment switch applies ONCE, when the model first sits on its device)
            self.set_deterministic(True)
        f32 = self.precision == 'fp32'
        S_lat = S if S_lat is None else S_lat
        key = (T, B, nt, n_euler, training, str(self._device()) + ('/fp32' if f32 else '')) + ((S,) if S > 1 or S_lat > 1 else ()) + ((S_lat,) if S_lat != S else ())
        pl = self._plans.get(key)
        if pl is None:
            assert S == 1 or not training
            dev = self._device()
            enc = EncoderNet(self._enc_blocks, T * B, dev, training, f32=f32, use_skips=bool(self.skipco)) if T > 0 else None
            # the step's small index tensors live in ONE int32 buffer, filled by a single host-to-device copy per forward:
            # [keep (T*B) | skip_idx (T*B) | skip_sel (B) | skip_map (nt*B*S)]  (a dozen tiny device kernels otherwise)
            nk = max(T * B, 1)
            ibuf = torch.zeros(2 * nk + B + nt * B * S, dtype=torch.int32, device=dev)
            skip_sel = ibuf[2 * nk:2 * nk + B] if self.skipco else None
            skip_map = ibuf[2 * nk + B:] if self.skipco else None
            dec = DecoderNet(self._dec_blocks, nt * B * S, dev, training, enc.skips if (self.skipco and enc) else None, skip_map,
                             skip_sel, f32=f32)
            lat = LatentNet(self._cfg(), T, B * S_lat, nt, n_euler, dev, training)
            pl = dict(enc=enc, dec=dec, lat=lat, skip_map=skip_map, skip_sel_t=skip_sel, ibuf=ibuf, keep=ibuf[:nk], skip_idx=ibuf[nk:2 * nk])
            # keep at most two training plans alive (they own all activation memory)
            if training:
                for k in [k for k in self._plans if isinstance(k[0], int) and k[4]]:
                    del self._plans[k]
            self._plans[key] = pl
            self._pack_version = None
        return pl

    def _pack(self, pl, params, st, overlap=False):
        """bf16 MFMA-layout copies of the conv weights, redone when a parameter changed.  overlap: on the second stream -- the
        encoder's share under the image-side layer (which reads the fp32 weights), the decoder's (60 % of the bytes; nothing reads
        it before the decoder forward) under the rest of the encoder; returns the two events to wait for (None: nothing pending)."""
        ver = (id(pl), tuple(p._version for p in self._enumerate()['plist']))
        if ver == self._pack_version:
            return None, None
        self._pack_version = ver
        if not (overlap and pl['enc'] is not None):
            if pl['enc'] is not None:
                pl['enc'].pack_weights(params, st)
            pl['dec'].pack_weights(params, st)
            return None, None
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = _make_side_stream()
        ev = L.record()
        with L.on_stream(self._side_stream):
            self._side_stream.wait_event(ev)
            pl['enc'].pack_weights(params, L.stream())
            enc_done = L.record()
            pl['dec'].pack_weights(params, L.stream())
            dec_done = L.record()
        return enc_done, dec_done

    def _draw_tape(self, T, B, nt, training, dev, t_skip=None, have=None):
        """Random draws in the reference's order: CPU generator for the frame indices, device
        generator for the normals.  have (optional): a partial tape -- only what it lacks is drawn (e.g. host-drawn frame indices given, normals
        drawn here: what a stream capture of the step needs, tools/graph_probe.py)."""
        tape = dict(have) if have else {}
        if training:
            if self.skipco and 't_skip' not in tape:
                tape['t_skip'] = t_skip if t_skip is not None else torch.randint(T, size=(B,))
            if 't_w' not in tape:
                tape['t_w'] = torch.stack([torch.randperm(T)[:self.nt_inf] for _ in range(B)], 1)
        if 'eps_y0' not in tape:
            tape['eps_y0'] = torch.randn(B, self.ny, device=dev)
        if 'eps_z' not in tape:
            tape['eps_z'] = torch.randn(max(nt - 1, 1), B, self.nz, device=dev)
        return tape

    @staticmethod
    def _fill_index_host(pl, T, B, nt, t_skip, training):
        """The step's small index tensors [keep | skip_idx | skip_sel | skip_map] formed on the host (B integers) in the plan's ONE pinned
        staging buffer, refilled in place (a fresh .pin_memory() per step is a pinned allocation on the host's critical path right at the
        step boundary); the caller (or a captured graph's copy node) moves it to pl['ibuf'] in one copy.  t_skip: CPU tensor (B,)."""
        ts = t_skip.cpu().to(torch.int32) if training else torch.full((B,), T - 1, dtype=torch.int32)
        sel_h = ts * B + torch.arange(B, dtype=torch.int32)
        host = pl.get('ibuf_host')
        if host is None:
            host = pl['ibuf_host'] = torch.zeros(pl['ibuf'].numel(), dtype=torch.int32).pin_memory()
        host[:T * B] = 0
        host[sel_h.long()] = 1                                              # keep
        host[T * B:2 * T * B] = -1
        host[T * B + sel_h.long()] = torch.arange(B, dtype=torch.int32)    # skip_idx: frame -> sample (or -1)
        host[2 * T * B:2 * T * B + B] = sel_h                               # skip_sel
        host[2 * T * B + B:] = sel_h.repeat(nt)                             # skip_map
        return host

    def _poll_cluster(self, every=32):
        """Every `every` calls (and on the first): the library's count of cluster failures of the persistent latent kernels -- barrier timeouts
        (workgroups of a cluster not co-resident) or a generation launch that found its cluster spread over several XCCs -- as it stood when it
        was last copied: no extra host sync, the word is read back in stream order into pinned memory and looked at one poll later.  Non-zero =
        some launch left garbage behind: stop, do not train on / return it."""
        n = self.__dict__.get('_ct_step', 0)
        self.__dict__['_ct_step'] = n + 1
        if n % every:
            return
        host = self.__dict__.get('_ct_host')
        if host is None:
            host = self.__dict__['_ct_host'] = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.__dict__['_ct_event'] = None
        ev = self.__dict__['_ct_event']
        if ev is not None and ev.query() and int(host[0]) != 0:
            raise L.SrvpHipError(f'{int(host[0])} cluster failure(s) in the persistent latent kernels (workgroups of a cluster were not co-resident, or not '
                                 'on one XCD for the generation chain): results since then are invalid; set SRVP_ROLLOUT_FUSED=0 SRVP_LSTM_FUSED=0 '
                                 'SRVP_LSTM_BWD_FUSED=0')
        L.call('srvp_cluster_timeouts_read', host.data_ptr(), L.stream())
        self.__dict__['_ct_event'] = L.record()

    def _check_cluster_now(self):
        """INFERENCE entry points fail closed: a generation launch whose cluster was not on one XCD (or a chain whose barrier
        timed out) writes nothing and only counts a failure, so frames decoded after it would come from stale latent buffers.  The deferred
        poll of training (one poll late, every n-th call) is not good enough for a caller that returns results: the failure word is read back
        in stream order behind EVERYTHING the call queued and waited for before the call returns.  Cost: the pipeline bubble between two
        inference calls (the callers -- evaluate, test.py's protocol -- read results on the host between calls anyway)."""
        host = self.__dict__.get('_ct_host_now')
        if host is None:
            host = self.__dict__['_ct_host_now'] = torch.zeros(1, dtype=torch.int32).pin_memory()
        L.call('srvp_cluster_timeouts_read', host.data_ptr(), L.stream())
        ev = L.record()
        ev.synchronize()
        if int(host[0]) != 0:
            raise L.SrvpHipError(f'{int(host[0])} cluster failure(s) in the persistent latent kernels (workgroups of a cluster were not co-resident, or not '
                                 'on one XCD for the generation chain): the results of this call are invalid and were not returned; set '
                                 'SRVP_ROLLOUT_GEN_FUSED=0 (launch-per-layer generation chain)')

    # ------------------------------------------------------------------------------------------------ core
    def _forward_impl(self, x, nt, n_euler, tape, training):
        self._require_gpu()                       # (fails loudly on a CPU model before anything touches the device runtime)
        with L.step_scope():
            return self._forward_body(x, nt, n_euler, tape, training)

    def _forward_body(self, x, nt, n_euler, tape, training):
        dev = self._require_gpu()
        T, B = x.shape[0], x.shape[1]
        if training:
            assert nt == T, 'training mode requires observations for every generated frame (srvp.py:391)'
            self.flatten_parameters_()
            self._fwd_gen += 1
        st = L.stream()
        pl = self._plan(T, B, nt, n_euler, training)
        params = self._named_tensors()
        enc_packed, pack_done = self._pack(pl, params, st, overlap=training and OVERLAP_PACK)
        enc, dec, lat = pl['enc'], pl['dec'], pl['lat']
        x = x.contiguous().float()
        # frames whose encoder activations feed the skip connections (srvp.py:185-190): drawn first (the reference draws t_skip
        # before the t_w permutations too), because the pooled encoder blocks keep their full-resolution output for these only
        t_skip = None
        if self.skipco and training:
            t_skip = tape['t_skip'] if tape is not None else torch.randint(T, size=(B,))
        keep = pl['keep']
        sel = None
        if self.skipco:
            # frame t_skip[b] * B + b of every sample: the index tensors are formed on the host (B integers) and travel in one copy.
            # (Safe to overwrite the staging buffer: the host has waited for the previous step's ELBO event, which lies behind the previous
            # copy out of it; a caller that runs two forwards without a sync in between is held on the copy's event)
            if pl.get('ibuf_copied') is not None and not torch.cuda.is_current_stream_capturing():
                pl['ibuf_copied'].synchronize()
            host = self._fill_index_host(pl, T, B, nt, t_skip, training)
            pl['ibuf'].copy_(host, non_blocking=True)
            pl['ibuf_copied'] = L.record()
            sel = pl['skip_sel_t']
        lat_early = training and LATENT_AUX and OVERLAP_WGRAD
        if lat_early:
            # (what the posterior chain needs besides hx -- the summed LSTM bias -- is formed on the auxiliary stream under the encoder)
            if getattr(self, '_lat_stream', None) is None:
                self._lat_stream = torch.cuda.Stream()
            ev_p = L.record()                         # behind the previous step's optimizer
            with L.on_stream(self._lat_stream):
                self._lat_stream.wait_event(ev_p)
                lat.lstm_bias_prep(params, L.stream())
        # (round 5, A/B switch, off) the decoder's hoisted skip convolutions stage by stage on the second stream, each as soon as the encoder
        # block it reads is done, instead of under the rollout kernel (where a traced step at 192 sequences shows the main stream waiting
        # 0.46 ms for them now that the rollout takes 0.3 ms; untraced the difference is 0.1 ms, and the other configs lose)
        skip_early = training and self.skipco and OVERLAP_SKIP and SKIP_EARLY and any(b.split for b in dec.blocks)
        on_skip = None
        if skip_early:
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = _make_side_stream()
            cats = {b.spec['cat'] for b in dec.blocks if b.split}

            def on_skip(cat):
                if cat not in cats:
                    return
                ev = L.record()
                with L.on_stream(self._side_stream):
                    self._side_stream.wait_event(ev)
                    dec.precompute_skips(L.stream(), cat=cat)
        hx = enc.forward(x.view(T * B, *x.shape[2:]), params, st, self.sync if training else None, keep=keep, packed=enc_packed, on_skip=on_skip)
        hx = hx.contiguous().view(T, B, self.nhx)
        ev_hx = None
        if lat_early:
            ev_hx = L.record()                        # hx exists (recorded BEFORE the draws below: the posterior chain does not wait for them)
        # the draws come AFTER the encoder launches (same order within the CPU and the device generator as the reference, which
        # draws them inside encode / infer_w / infer_y / generate): the per-sample randperm calls cost ~0.3 ms of host time that
        # the GPU now spends in the encoder instead of idling; the small index tensors travel through pinned memory so that the
        # copy does not block the host behind the work queued so far
        if tape is None or any(k not in tape for k in ('eps_y0', 'eps_z') + (('t_w',) if training else ())):
            tape = self._draw_tape(T, B, nt, training, dev, t_skip=t_skip, have=tape)
        t_w_host = tape.get('t_w') if (training and torch.is_tensor(tape.get('t_w')) and not tape['t_w'].is_cuda) else None
        # (the host-drawn frame indices t_skip / t_w are consumed on the host -- they stay there; the normals go to the device)
        tape = {k: (_to_dev(v, dev) if (torch.is_tensor(v) and not (k in ('t_skip', 't_w') and not v.is_cuda)) else v) for k, v in tape.items()}
        self.last_tape = tape
        if self.skipco:
            pl['skip_sel'] = sel
        def skips_on_side():
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = _make_side_stream()
            ev = L.record()
            with L.on_stream(self._side_stream):
                self._side_stream.wait_event(ev)
                dec.precompute_skips(L.stream())
                done = L.record()
            return done
        s_done = None
        overlap_skips = self.skipco and OVERLAP_SKIP and any(b.split for b in dec.blocks)
        if skip_early:
            dec._skips_done = True                                  # (every stage's launch is on the second stream by now)
            s_done = torch.cuda.Event()
            s_done.record(self._side_stream)
            overlap_skips = False
        if overlap_skips and not SKIP_LATE:
            s_done = skips_on_side()
        t_w_arg = (t_w_host if t_w_host is not None else tape.get('t_w')) if training else None
        w_done = None
        if training and LATENT_AUX and OVERLAP_WGRAD:
            # the three inference chains are independent: the posterior (LSTM + q_z) and the content variable w on the auxiliary stream, y_0 on
            # the main one; the rollout needs y_0 and q_z, the decoder w (round 5: ~25 dependent micro-kernels became the longest of three chains)
            if getattr(self, '_lat_stream', None) is None:
                self._lat_stream = torch.cuda.Stream()
            aux = self._lat_stream
            main_stream = L.cur_stream()
            ev_tape = L.record()                      # the index / noise tensors of the tape are on the device
            with L.on_stream(aux):
                aux.wait_event(ev_hx)
                lat.posterior(hx, params, L.stream(), bias_ready=True)
                post_done = L.record()
                aux.wait_event(ev_tape)
                w = lat.infer_w(hx, params, t_w_arg, L.stream())
                if lat.w_rows is not lat.__dict__.get('_w_rows_dev'):
                    lat.w_rows.record_stream(main_stream)    # (allocated on the auxiliary stream, read by the backward's scatter on the main one)
                w_done = L.record()
            y0, q_y0 = lat.infer_y(hx[:self.nt_inf], params, tape['eps_y0'], st)
            L.wait(post_done)
        else:
            w = lat.infer_w(hx, params, t_w_arg, st)
            y0, q_y0 = lat.infer_y(hx[:self.nt_inf], params, tape['eps_y0'], st)
            lat.posterior(hx, params, st)
        if overlap_skips and SKIP_LATE:
            # the hoisted skip halves of the decoder need the encoder only: second stream, under the ROLLOUT kernel (one persistent
            # launch on nh/32 x batch-tiles CUs that leaves the rest of the chip idle for 0.5 ms).  Issued earlier -- under the
            # inference MLPs and the LSTM chain -- their big workgroups starved those small dependent kernels (15 -> 60-80 us each)
            s_done = skips_on_side()
        # (training: the prior MLP over the stored states feeds the KL term only -- on the auxiliary stream, under the decoder, instead of
        # 0.12 ms of small dependent launches between the rollout and the decoder; SRVP_PZ_AUX=0: in line)
        pz_stream = None
        if training and PZ_AUX and OVERLAP_WGRAD:
            if getattr(self, '_lat_stream', None) is None:
                self._lat_stream = torch.cuda.Stream()
            pz_stream = self._lat_stream
        y, z, qz, pz, res = lat.generate(y0, T, params, tape['eps_z'], st, pz_stream=pz_stream)
        if s_done is not None:
            L.wait(s_done)
        if pack_done is not None:
            L.wait(pack_done)
        if w_done is not None:
            L.wait(w_done)
        # decoder input rows [w[b] | y[t][b]] (srvp.py:216-221) assembled by the library from w and the stored states y_all[t * n_euler]
        x_flat = dec.forward(None, params, st, self.sync if training else None,
                             latent=(w, lat.y_all, lat.ne * B * self.ny, nt, B, self.nh_inf, self.ny))
        if getattr(lat, 'pz_done', None) is not None:
            L.wait(lat.pz_done)          # p_z is read (KL term, callers) from here on
        x_ = x_flat.view(nt, B, *x_flat.shape[1:])
        pl['hx'], pl['x'] = hx, x
        self._last_plan = pl
        if not training:
            self._check_cluster_now()             # (training: srvp_amd.train.train polls once per step)
        return x_, y, z, w, q_y0, qz, pz, res

    @torch.no_grad()
    def sample(self, x, nt, n_samples, dt=1.0, tape=None, chunk=None):
        """n_samples stochastic futures of every video from ONE encoding of the conditioning frames x (T, B, C,
        64, 64) -- what train.evaluate (train.py:170-174) / test.py:237-246 obtain from n_samples forward passes, each of
        which re-encodes the same frames.  Inference mode only.  Returns x_ (nt, n_samples, B, C, 64, 64).
        tape (optional): eps_y0 (n_samples*B, ny), eps_z (nt-1, n_samples*B, nz), row s*B + b.
        chunk (optional): decode `chunk` samples per video at a time (bounds the activation memory: nt * B * chunk frames resident);
        the encoder and the whole latent path -- posterior on the conditioning frames, prior rollout: a latency-bound chain whose
        cost does not depend on the number of rows -- still run ONCE for all n_samples."""
        assert not self.training, 'sample() is an inference entry point (model.eval())'
        dev = self._require_gpu()
        n_euler = int(round(1 / dt))
        T, B, S = x.shape[0], x.shape[1], int(n_samples)
        Sd = S if (chunk is None or chunk >= S) else max(1, int(chunk))
        st = L.stream()
        pl = self._plan(T, B, nt, n_euler, False, S=Sd, S_lat=S)
        params = self._named_tensors()
        self._pack(pl, params, st)
        if tape is None:
            tape = self._draw_tape(T, B * S, nt, False, dev)
        tape = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tape.items()}
        enc, dec, lat = pl['enc'], pl['dec'], pl['lat']
        self._last_sample_lat = lat
        x = x.contiguous().float()
        hx = enc.forward(x.view(T * B, *x.shape[2:]), params, st, None).contiguous().view(T, B, self.nhx)
        if self.skipco:
            sel = (T - 1) * B + torch.arange(B, device=dev, dtype=torch.int32)
            pl['skip_map'].copy_(sel.repeat(nt * Sd))
            pl['skip_sel_t'].copy_(sel)
        hx_s = hx.repeat(1, S, 1).contiguous()                          # (T, S*B, nhx): row s*B + b
        w = lat.infer_w(hx_s, params, None, st)
        y0, _ = lat.infer_y(hx_s[:self.nt_inf], params, tape['eps_y0'], st)
        lat.posterior(hx_s, params, st)
        lat.generate(y0, T, params, tape['eps_z'], st)
        if Sd == S:
            x_flat = dec.forward(None, params, st, None, latent=(w, lat.y_all, lat.ne * B * S * self.ny, nt, B * S, self.nh_inf, self.ny))
            out = x_flat.view(nt, S, B, *x_flat.shape[1:]).clone()
            self._check_cluster_now()            # fail closed: behind everything queued, before anything is returned
            return out
        out = torch.empty(nt, S, B, *dec.x_out.shape[1:], dtype=torch.float32, device=dev)
        for s0 in range(0, S, Sd):
            # the last chunk is decoded at the common size (one decoder plan per evaluation shape), re-using the final samples
            s0 = min(s0, S - Sd)
            x_flat = dec.forward(None, params, st, None, latent=(w[s0 * B:], lat.y_all[:, s0 * B:], lat.ne * B * S * self.ny, nt, B * Sd,
                                                                  self.nh_inf, self.ny), coeffs_current=s0 > 0)
            out[:, s0:s0 + Sd] = x_flat.view(nt, Sd, B, *x_flat.shape[1:])
        self._check_cluster_now()                # fail closed: behind everything queued, before anything is returned
        return out

    def _grad_offsets(self):
        """name -> (offset, numel) of every parameter's gradient inside the flat gradient buffer (registration order: encoder, decoder,
        latent networks); cached per flat buffer."""
        c = self.__dict__.get('_goff')
        if c is None or c[0] is not self._flat[1]:
            off, d = 0, {}
            for name, p in zip(self._enumerate()['pnames'], self._enumerate()['plist']):
                d[name] = (off, p.numel())
                off += p.numel()
            enc_end = min(o for k, (o, _) in d.items() if k.startswith('decoder.'))
            dec_end = min(o for k, (o, _) in d.items() if not k.startswith(('encoder.', 'decoder.')))
            c = self.__dict__['_goff'] = (self._flat[1], d, (enc_end, dec_end, off))
        return c[1], c[2]

    def _block_range(self, net, lo, hi):
        """[start, end) flat gradient elements of the parameters (conv weight, BatchNorm weight / bias) of net.blocks[lo:hi]."""
        offs, _ = self._grad_offsets()
        keys = []
        for blk in net.blocks[lo:hi]:
            keys.append(blk.spec['key'] + '.weight')
            if blk.spec['bnkey'] is not None:
                keys += [blk.spec['bnkey'] + '.weight', blk.spec['bnkey'] + '.bias']
        a = min(offs[k][0] for k in keys)
        b = max(offs[k][0] + offs[k][1] for k in keys)
        assert b - a == sum(offs[k][1] for k in keys), 'the parameters of consecutive blocks are contiguous in the flat buffer'
        return a, b

    def _exchange(self, lo, hi, after=None):
        """Data-parallel exchange of flat gradient elements [lo, hi) (complete behind event `after`, or behind the current stream's work so
        far) on the communication stream: every gradient collective of a step goes through this ONE stream and ONE communicator in program
        order -- the same order on every rank."""
        if hi <= lo:
            return
        if after is None:
            after = L.record()
        if getattr(self, '_comm_stream', None) is None:
            self._comm_stream = torch.cuda.Stream()
        with L.on_stream(self._comm_stream):
            self._comm_stream.wait_event(after)
            self.sync.reduce_slice(self, lo, hi)
        self._exchanged.append((lo, hi))

    def _backward_impl(self, d_x, d_y, d_w, d_qy0, d_qz, d_pz, d_res):
        with L.step_scope():
            return self._backward_body(d_x, d_y, d_w, d_qy0, d_qz, d_pz, d_res)

    def _backward_body(self, d_x, d_y, d_w, d_qy0, d_qz, d_pz, d_res):
        pl = self._last_plan
        enc, dec, lat = pl['enc'], pl['dec'], pl['lat']
        st = L.stream()
        params = self._named_tensors()
        grads = self._grads()
        tape = self.last_tape
        T, B = lat.T, lat.B
        nt = lat.nt
        cz = lambda t: None if t is None else t.contiguous().float()
        if d_x is None:
            d_x = torch.zeros_like(dec.x_out)
        # The decoder's weight gradients feed nothing but the optimizer: they run on a second stream, concurrently with the
        # latent backward (a ~3 ms chain of tiny dependent kernels that leaves the GPU almost idle) and the encoder backward
        overlap = OVERLAP_WGRAD
        if overlap and getattr(self, '_side_stream', None) is None:
            self._side_stream = _make_side_stream()
        # the prior MLP's backward needs the KL gradient only: on the auxiliary stream, under the decoder backward (SRVP_PZ_BWD_AUX=0: in
        # line between the decoder's and the rollout's backward, as before round 5)
        pz_pre = None
        d_pz_c = cz(d_pz)
        if overlap and LATENT_AUX and PZ_BWD_AUX and getattr(lat, 'pz_ext', False) and lat.S > 0:
            if getattr(self, '_lat_stream', None) is None:
                self._lat_stream = torch.cuda.Stream()
            ev_in = L.record()                        # d_pz exists
            with L.on_stream(self._lat_stream):
                self._lat_stream.wait_event(ev_in)
                lat.pz_backward_chain(params, d_pz_c, L.stream())
                pz_pre = L.record()
        # multi-GPU: the flat gradient buffer is exchanged slice by slice, each as soon as it is final (SURVEY §5 / train.py:309-314: DDP's
        # bucketed all-reduce under backward): decoder tail, decoder head, encoder deep stages, latent networks; the encoder's first stages
        # (< 1 MB) at the step's end.  SRVP_GRAD_SLICES=0: decoder slice + everything else at the end (the round-5 form).
        exch = self.sync is not None and (self.sync.world > 1 or self.sync.force)
        sliced = exch and overlap and GRAD_SLICES and DEC_ALLREDUCE_STREAM
        self._exchanged = []
        (_, (enc_end, dec_end, g_total)) = self._grad_offsets() if exch else (None, (0, 0, 0))

        def dec_part(lo, hi):
            self._exchange(*self._block_range(dec, lo, hi))            # (called with the weight-gradient stream current, behind the unpack)
        dz = dec.backward(cz(d_x).view(nt * B, *dec.x_out.shape[1:]), params, grads, st, self.sync, defer_wgrad=overlap,
                          side=self._side_stream if overlap else None, on_part=dec_part if sliced else None)
        ev_dec = None
        if overlap:
            ev_dec = L.record()                       # the decoder's output gradients exist: its weight gradients may start (second stream, below)
        # backward of the time-expansion of w and of the concatenation [w | y_t] (srvp.py:216-221): d_w = sum over time, d_y_t straight
        # into the rollout's state-gradient buffer (frame t = Euler step t * n_euler)
        if not hasattr(lat, 'd_w_tot'):
            lat.d_w_tot = torch.empty(B, self.nh_inf, dtype=torch.float32, device=dz.device)
        lat.d_y_all.zero_()
        L.call('srvp_dz_split', L.ptr(dz), dz.shape[1], 1 if dz.dtype == torch.float32 else 0, nt, B, self.nh_inf, self.ny,
               L.ptr(cz(d_w)), L.ptr(cz(d_y)), L.ptr(lat.d_w_tot), L.ptr(lat.d_y_all), lat.ne * B * self.ny, st)
        if exch and not overlap:
            self._exchange(enc_end, dec_end)
        deferred = [] if overlap else None
        lat_aux = None
        if overlap and LATENT_AUX:
            if getattr(self, '_lat_stream', None) is None:
                self._lat_stream = torch.cuda.Stream()
            lat_aux = self._lat_stream
        d_hx = lat.backward(pl['hx'], params, grads, tape['eps_y0'], tape['eps_z'], 'in_place', lat.d_w_tot,
                            cz(d_qy0), cz(d_qz), d_pz_c, cz(d_res), None, st, defer=deferred, aux=lat_aux, pz_pre=pz_pre)
        if overlap:
            # (host order: the main-stream launches of the latent backward -- the critical path -- go out BEFORE the ~25 second-stream
            # launches of the decoder's weight gradients, which only wait for ev_dec on the device: a host that is just ahead of the device
            # -- small batches, a profiler attached -- then does not leave the main queue empty for the 0.25 ms the enqueueing takes)
            with L.on_stream(self._side_stream):
                self._side_stream.wait_event(ev_dec)
                b_lo, b_hi = dec.deferred_wgrads(grads, L.stream())
                if exch and not DEC_ALLREDUCE_STREAM:
                    self.sync.reduce_slice(self, enc_end, dec_end)          # (A/B: in line on the weight-gradient stream, the round-4 form)
                    self._exchanged.append((enc_end, dec_end))
                elif exch:
                    # the decoder's (remaining) slice on the communication stream: enqueued on the second stream it sat in front of the
                    # ENCODER's weight gradients there and held them back for as long as the exchange takes over the links
                    # (blocks [b_lo, b_hi) were unpacked just now; the blocks behind them went out early: dec_part)
                    self._exchange(enc_end, dec_end if b_hi == len(dec.blocks) else self._block_range(dec, b_hi, len(dec.blocks))[0])
        ev_lat = None
        if deferred:
            ev_lat = L.record()                       # every delta the latent weight gradients read exists
        skip_grads = None
        if self.skipco:
            skip_grads = {}
            idx = pl['skip_idx']                                   # frame -> sample whose skip connection it feeds, or -1 (forward)
            for stage, dsel in dec.skip_grads(nt, B, st).items():
                skip_grads[stage] = (dsel, idx, pl['skip_sel_t'])     # gradient rows, frame -> row (or -1), row -> frame
        nhp = cpad(self.nhx)
        if nhp != self.nhx:
            d_hx_p = torch.zeros(T * B, nhp, dtype=torch.float32, device=d_hx.device)
            d_hx_p[:, :self.nhx] = d_hx
        else:
            d_hx_p = d_hx
        aux = None
        if overlap and LATENT_WGRAD_STREAM and SKIP_REDUCE_AUX:
            if getattr(self, '_lat_stream', None) is None:
                self._lat_stream = torch.cuda.Stream()
            aux = self._lat_stream
        enc_side = self._side_stream if overlap else None
        if overlap and ENC_WGRAD_STREAM2:
            # the encoder's weight gradients (and their unpacking) on a stream of their own: behind the decoder's on ONE stream the first of
            # them waited for the last of those, and the step ended on that queue after the main queue had run dry
            if getattr(self, '_side2_stream', None) is None:
                self._side2_stream = _make_side_stream()
            enc_side = self._side2_stream
        def enc_part(lo, hi):
            self._exchange(*self._block_range(enc, lo, hi))               # (weight-gradient stream current, behind the deep blocks' unpack)
        enc.backward(pl['x'].view(T * B, *pl['x'].shape[2:]), d_hx_p, skip_grads, params, grads, st, self.sync,
                     side=enc_side, aux=aux, on_part=enc_part if sliced else None)
        lat_stream = None
        if deferred:
            # the latent networks' weight gradients feed nothing but the optimizer.  ~25 small launches (a few workgroups each): on the second
            # stream they sat BEHIND the encoder's weight gradients and ran as the step's tail, 0.3 ms after the main stream's last kernel;
            # on a stream of their own (SRVP_LATENT_WGRAD_STREAM=0: the second stream) they run as soon as the latent backward is done,
            # beside everything else.  Enqueued last so that the encoder backward's main-stream launches are not held up on the host.
            lat_stream = self._side_stream
            if LATENT_WGRAD_STREAM:
                if getattr(self, '_lat_stream', None) is None:
                    self._lat_stream = torch.cuda.Stream()
                lat_stream = self._lat_stream
            with L.on_stream(lat_stream):
                lat_stream.wait_event(ev_lat)
                s2 = L.stream()
                for fn in deferred:
                    fn(s2)
                if sliced:
                    self._exchange(dec_end, g_total)                      # every gradient of the latent networks is final behind this point
        if overlap:
            # (the side stream holds the decoder's weight gradients + unpack and, behind them, the encoder's unpack)
            side_done = torch.cuda.Event()
            side_done.record(self._side_stream)
            L.wait(side_done)
            if enc_side is not self._side_stream:
                side2_done = torch.cuda.Event()
                side2_done.record(enc_side)
                L.wait(side2_done)
            if lat_stream is not None and lat_stream is not self._side_stream:
                lat_done = torch.cuda.Event()
                lat_done.record(lat_stream)
                L.wait(lat_done)
        if exch:
            # what has not been exchanged yet (sliced: the encoder's first stages; otherwise encoder + latent slice, or everything)
            done = sorted(self._exchanged)
            cur = 0
            for lo, hi in done + [(g_total, g_total)]:
                if lo > cur:
                    self._exchange(cur, lo)
                cur = max(cur, hi)
            if getattr(self, '_comm_stream', None) is not None:
                comm_done = torch.cuda.Event()
                comm_done.record(self._comm_stream)
                L.wait(comm_done)     # the optimizer reads the averaged gradients
            self.sync.grads_finish(self)

    # ------------------------------------------------------------------------------------------------ reference API
    def forward(self, x, nt, dt, remove_intermediate=True, tape=None):
        """srvp.py:415-470.  Returns (x_, y, z, w, q_y_0_params, q_z_params, p_z_params, res).
        remove_intermediate=False (srvp.py:402: every Euler sub-step is kept and decoded -- generation at 1/dt times the frame rate)
        is an inference path: the composition of the granular entry points, as srvp.py:459-469 writes it."""
        n_euler = int(round(1 / dt))
        assert abs(n_euler * dt - 1) < 1e-6, 'dt must be the inverse of an integer (srvp.py:371)'
        if not remove_intermediate:
            assert not self.training, 'remove_intermediate=False: inference only (the ELBO of train.py compares x_ with nt data frames)'
            tape = tape or {}
            with torch.no_grad():
                hx, skipco = self.encode(x)
                w = self.infer_w(hx)
                y_0, q_y_0 = self.infer_y(hx[:self.nt_inf], eps=tape.get('eps_y0'))
                y, z, q_z, p_z, res = self.generate(y_0, hx, nt, dt, remove_intermediate=False, eps_z=tape.get('eps_z'))
                return self.decode(w, y, skipco), y, z, w, q_y_0, q_z, p_z, res
        if self.training and torch.is_grad_enabled():
            return _SrvpForward.apply(self, x, nt, n_euler, tape, *self.parameters())
        with torch.no_grad():
            outs = self._forward_impl(x, nt, n_euler, tape, training=self.training)
            # fresh tensors, like the reference's outputs (the plan-owned buffers are overwritten by the next call)
            return tuple(o.clone() if o is not None else None for o in outs)

    def drop_sample_plans(self):
        """Frees the S > 1 inference plans of sample() (activation buffers for S futures per video)."""
        for k in [k for k in self._plans if isinstance(k[0], int) and len(k) > 6]:      # (keys with a sample count)
            del self._plans[k]

    def _infer_plan(self, T, B, nt, n_euler=1):
        return self._plan(T, B, nt, n_euler, False)

    @torch.no_grad()
    def encode(self, x, tape=None):
        """srvp.py:156-193 -> (hx (T,B,nhx), skips: list of (B,C,H,W) deepest first, or None)."""
        dev = self._require_gpu()
        T, B = x.shape[0], x.shape[1]
        training = self.training
        pl = self._plan(T, B, T, 1, training)
        st = L.stream()
        params = self._named_tensors()
        self._pack(pl, params, st)
        enc = pl['enc']
        hx = enc.forward(x.contiguous().float().view(T * B, *x.shape[2:]), params, st, self.sync if training else None)
        hx = hx.contiguous().view(T, B, self.nhx).clone()
        skips = None
        if self.skipco:
            if training:
                t = tape['t_skip'] if tape is not None else torch.randint(T, size=(B,))
                sel = t.to(dev).long() * B + torch.arange(B, device=dev)
            else:
                sel = (T - 1) * B + torch.arange(B, device=dev)
            skips = []
            for stage in sorted(enc.skips):
                f = enc.skips[stage]
                skips.append(f.interior()[sel].permute(0, 3, 1, 2).float().contiguous())
        return hx, skips

    @torch.no_grad()
    def decode(self, w, y, skip):
        """srvp.py:195-227: w (B, nh_inf), y (nt, B, ny), skip list (B, C, H, W) or None -> x_ (nt, B, C, 64, 64)."""
        dev = self._require_gpu()
        nt, B = y.shape[0], y.shape[1]
        assert (skip is None) == (not self.skipco)
        f32 = self.precision == 'fp32'
        key = ('decode', nt, B, self.training, str(dev) + ('/fp32' if f32 else ''))
        pl = self._plans.get(key)
        if pl is None:
            from .convnet import Feat
            feats = None
            skip_map = None
            if self.skipco:
                feats = {i: Feat(B, s.shape[2], s.shape[3], s.shape[1], dev, dtype=torch.float32 if f32 else torch.bfloat16)
                         for i, s in enumerate(skip)}
                skip_map = torch.arange(B, dtype=torch.int32, device=dev).repeat(nt)
            dec = DecoderNet(self._dec_blocks, nt * B, dev, False, feats, skip_map,
                             torch.arange(B, dtype=torch.int32, device=dev) if self.skipco else None, f32=f32)
            pl = dict(dec=dec, feats=feats)
            self._plans[key] = pl
        st = L.stream()
        params = self._named_tensors()
        if self.training:
            raise NotImplementedError('decode() in training mode: use forward() (fused autograd node)')
        pl['dec'].pack_weights(params, st)
        if self.skipco:
            for i, s in enumerate(skip):
                pl['feats'][i].load_nchw(s.to(dev))
        z_in = torch.cat([w.repeat(nt, 1), y.reshape(nt * B, self.ny)], 1).float().contiguous()
        x_flat = pl['dec'].forward(z_in, params, st, None)
        return x_flat.view(nt, B, *x_flat.shape[1:]).clone()

    @torch.no_grad()
    def infer_w(self, hx, tape=None):
        """srvp.py:229-256."""
        self._require_gpu()
        T, B = hx.shape[0], hx.shape[1]
        lat = self._infer_plan(T, B, max(T, 2))['lat']
        t_w = None
        if self.training:
            t_w = tape['t_w'] if tape is not None else torch.stack([torch.randperm(T)[:self.nt_inf] for _ in range(B)], 1)
            t_w = t_w.to(hx.device)
        return lat.infer_w(hx.contiguous().float(), self._named_tensors(), t_w, L.stream()).clone()

    @torch.no_grad()
    def infer_y(self, hx, eps=None):
        """srvp.py:258-278 -> (y_0, q_y_0_params)."""
        self._require_gpu()
        B = hx.shape[1]
        lat = self._infer_plan(hx.shape[0], B, max(hx.shape[0], 2))['lat']
        if eps is None:
            eps = torch.randn(B, self.ny, device=hx.device)
        y0, q = lat.infer_y(hx.contiguous().float(), self._named_tensors(), eps.to(hx.device), L.stream())
        return y0.clone(), q.clone()

    @torch.no_grad()
    def infer_z(self, hx, eps=None):
        """srvp.py:280-298: hx here is the LSTM output for one frame (B, nh_inf) -> (z, q_z_params)."""
        self._require_gpu()
        from .latent import linear_fwd
        st = L.stream()
        B = hx.shape[0]
        q = torch.empty(B, 2 * self.nz, dtype=torch.float32, device=hx.device)
        linear_fwd(st, hx.contiguous().float(), self.q_z.weight, self.q_z.bias, q)
        if eps is None:
            eps = torch.randn(B, self.nz, device=hx.device)
        z = torch.empty(B, self.nz, dtype=torch.float32, device=hx.device)
        L.call('srvp_rsample_fwd', L.ptr(q), L.ptr(eps.contiguous()), L.ptr(z), B, self.nz, st)
        return z, q

    @torch.no_grad()
    def _residual_step(self, y_t, z_tp1, dt):
        """srvp.py:300-323."""
        self._require_gpu()
        from .latent import linear_fwd, mlp_keys
        st = L.stream()
        cur = torch.cat([y_t, z_tp1], 1).float().contiguous()
        params = self._named_tensors()
        keys = mlp_keys('dynamics', self.nlayers_res)
        for i, k in enumerate(keys):
            last = i == len(keys) - 1
            out = torch.empty(cur.shape[0], params[k + '.weight'].shape[0], dtype=torch.float32, device=cur.device)
            linear_fwd(st, cur, params[k + '.weight'], params[k + '.bias'], out, L.ACT_NONE if last else L.ACT_RELU)
            cur = out
        res = dt * cur
        return y_t + res, res

    @torch.no_grad()
    def generate(self, y_0, hx, nt, dt, remove_intermediate=True, eps_z=None):
        """srvp.py:325-413 -> (y, z, q_z_params, p_z_params, res); hx may be [] (pure prior rollout, test.py:244).
        remove_intermediate=False: y holds the state after EVERY Euler sub-step, (nt - 1) / dt + 1 rows (srvp.py:402)."""
        self._require_gpu()
        n_euler = int(round(1 / dt))
        B = y_0.shape[0]
        T = len(hx)
        assert not (self.training and nt > T), 'prior sampling is an inference-only path (srvp.py:391)'
        key = ('gen', T, B, nt, n_euler, str(y_0.device))
        lat = self._plans.get(key)
        if lat is None:
            lat = LatentNet(self._cfg(), T, B, nt, n_euler, y_0.device, False)
            self._plans[key] = lat
        st = L.stream()
        params = self._named_tensors()
        if eps_z is None:
            eps_z = torch.randn(max(nt - 1, 1), B, self.nz, device=y_0.device)
        if T > 0:
            lat.posterior(hx.contiguous().float(), params, st)
        y, z, qz, pz, res = lat.generate(y_0.contiguous().float(), T, params, eps_z.contiguous().float(), st)
        if not remove_intermediate:
            y = lat.y_all[:lat.S + 1]                     # the rollout stores every sub-step anyway (BPTT / weight gradients)
        cl = lambda t: None if t is None else t.clone()
        outs = cl(y), (cl(z) if nt > 1 else None), cl(qz), (cl(pz) if nt > 1 else None), cl(res)
        self._check_cluster_now()                         # fail closed (granular inference entry point: test.py:244)
        return outs
