"""
Data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.  Replaces DistributedDataParallel +
SyncBatchNorm of the reference (train.py:205-219, 278-283, 309-314):

  * gradients live in ONE flat fp32 buffer (model.flatten_parameters_), so the exchange is a few large all-reduces of
    contiguous slices -- the decoder's slice is launched as soon as the decoder backward has been queued, so it overlaps the
    latent + encoder backward; averaged over ranks like DDP;
  * BatchNorm statistics (sum, sum of squares; and the two backward sums) are all-reduced per layer in fp64, which makes N-GPU
    results equal the 1-GPU results on the same global batch up to summation order (SyncBatchNorm) -- checked by
    tests/test_two_rank_equality.py.  Each of these 42 small all-reduces per step feeds the very next kernel (the layer's
    normalisation / its gradient), so they cannot be coalesced across layers; what can be removed is their host and stream
    overhead:
  * transport: RCCL driven from the C ABI (csrc/comm.hip: ncclAllReduce enqueued on the compute stream -- one kernel in stream
    order, no hop to a communicator stream, no Python dispatch per collective) on two communicators (statistics / gradients,
    so that a gradient slice queued behind the weight-gradient kernels on the side stream cannot hold up the statistics of
    the layers behind it).  `SRVP_COMM=torch`, a non-RCCL backend (gloo: CPU tests, several ranks on one GPU) or a failed
    self-test fall back to the same collectives through torch.distributed.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L


class NativeComm:
    """One RCCL communicator owned by libsrvp_hip.so (srvp_comm_*), bootstrapped through torch.distributed."""

    def __init__(self, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        lib = L.load()
        # Every rank reaches the broadcast whatever happens on rank 0 (librccl missing, ncclGetUniqueId failing): rank 0 sends the
        # error text instead of the id and ALL ranks raise, so the caller's fallback decision is the same everywhere -- a rank
        # that raised before the broadcast would leave the others blocked inside it (mismatched collectives).
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            if lib.srvp_comm_unique_id(buf) == 0:
                ident = [bytes(buf.raw)]
            else:
                ident = ['error: ' + lib.srvp_last_error().decode()]
        dist.broadcast_object_list(ident, src=0, group=group)
        if not isinstance(ident[0], bytes):
            raise L.SrvpHipError(f'srvp_comm_unique_id failed on rank 0 ({ident[0]})')
        self.handle = C.c_void_p()
        self.rank, self.world = rank, world
        # communicator creation is itself collective inside RCCL; a rank whose srvp_comm_init fails reports it through the
        # MIN all-reduce of Sync.__init__ (init_ok), it does not raise past its peers
        self.init_error = None
        if lib.srvp_comm_init(ident[0], rank, world, C.byref(self.handle)) != 0:
            self.init_error = lib.srvp_last_error().decode()
            self.handle = C.c_void_p()

    def allreduce(self, t):
        assert t.is_cuda and t.is_contiguous()
        name = {torch.float64: 'srvp_allreduce_f64', torch.float32: 'srvp_allreduce_f32'}[t.dtype]
        L.call(name, self.handle, L.ptr(t), t.numel(), L.stream())

    def broadcast(self, t, root=0):
        assert t.is_cuda and t.is_contiguous()
        L.call('srvp_bcast_bytes', self.handle, L.ptr(t), t.numel() * t.element_size(), root, L.stream())

    def self_test(self):
        """Every rank contributes rank + 1: the sum must be world (world + 1) / 2 in both dtypes."""
        want = self.world * (self.world + 1) / 2
        for dt in (torch.float64, torch.float32):
            t = torch.full((257,), float(self.rank + 1), dtype=dt, device='cuda')
            self.allreduce(t)
            if not bool((t == want).all().item()):
                return False
        return True

    def info(self):
        """What RCCL itself reports about this communicator: dict(ranks=ncclCommCount, rank=ncclCommUserRank, version=ncclGetVersion)."""
        a = (C.c_int32 * 3)()
        L.call('srvp_comm_info', self.handle, a)
        return dict(ranks=int(a[0]), rank=int(a[1]), version=int(a[2]))

    def close(self):
        if self.handle:
            L.load().srvp_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class StepWatchdog:
    """a training step that does not complete within `timeout_s` (a collective that never meets its peers, a persistent
    cluster kernel spinning on workgroups that are not resident) must END the job with the evidence on stderr, on every rank, instead of
    hanging until a scheduler kills it.  A daemon thread watches the time a step has been in flight (`begin()` on entry of
    srvp_amd.train.train, `beat()` after the step's host sync); when it is exceeded the thread prints rank, step, the library's last error text, the
    cluster-timeout word of the persistent latent kernels (read on a stream of its own with a bounded wait: the copy engine still works when
    a compute queue is stuck), the transports in use -- and leaves with os._exit(EXIT_CODE).  Every rank runs its own: when one rank's
    collective blocks, its peers block in the same collective and report too.  SRVP_WATCHDOG_S sets the limit (default 30; 0: off); the
    first step gets `first_s` (plan building, kernel loading, RCCL's lazy channel set-up)."""
    EXIT_CODE = 17

    def __init__(self, timeout_s=None, first_s=180.0, rank=0, describe=None, on_timeout=None):
        import threading
        import time
        if timeout_s is None:
            timeout_s = float(os.environ.get('SRVP_WATCHDOG_S', '30'))
        self.timeout_s, self.first_s, self.rank, self.describe = float(timeout_s), float(first_s), rank, describe
        self.on_timeout = on_timeout                          # (tests: called instead of os._exit)
        self._time = time
        self.step, self.t_begin, self.fired = -1, None, False
        self._stop = threading.Event()
        self._thread = None
        if self.timeout_s > 0:
            self._thread = threading.Thread(target=self._run, name='srvp-step-watchdog', daemon=True)
            self._thread.start()

    def begin(self):
        """A step starts (srvp_amd.train.train on entry): the clock runs until beat().  Time BETWEEN steps (data loading, validation,
        checkpoint writes on rank 0) is not a step's time and is not counted."""
        self.t_begin = self._time.monotonic()

    def beat(self):
        """The step's host sync returned."""
        self.step += 1
        self.t_begin = None

    def cancel(self):
        """The step was abandoned (an exception left srvp_amd.train.train): the clock stops, the step is not counted."""
        self.t_begin = None

    def stop(self):
        """Ends the watchdog thread (before the final checkpoint writes: nothing may os._exit() in the middle of them)."""
        self.t_begin = None
        self._stop.set()
        if self._thread is not None and self._thread.is_alive():
            self._thread.join(timeout=2.0)

    def _limit(self):
        return self.first_s if self.step < 0 else self.timeout_s

    def _run(self):
        while not self._stop.wait(0.25):
            t0 = self.t_begin
            if t0 is None:
                continue
            waited = self._time.monotonic() - t0
            if waited > self._limit():
                self.fired = True
                self.report(waited)
                if self.on_timeout is not None:
                    self.on_timeout(self)
                    return
                os._exit(self.EXIT_CODE)

    def cluster_timeout_word(self, wait_s=2.0):
        """The library's count of cluster-barrier timeouts, read on a stream of its own; None if the copy does not come back in time."""
        try:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            host[0] = -1
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                L.call('srvp_cluster_timeouts_read', host.data_ptr(), st.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(st)
            t0 = self._time.monotonic()
            while not ev.query():
                if self._time.monotonic() - t0 > wait_s:
                    return None
                self._time.sleep(0.01)
            return int(host[0])
        except Exception as e:                               # noqa: BLE001 -- a diagnostic must not raise inside the watchdog
            return f'unreadable ({e})'

    def report(self, waited):
        import sys
        try:
            err = L.load().srvp_last_error().decode()
        except Exception as e:                               # noqa: BLE001
            err = f'unreadable ({e})'
        word = self.cluster_timeout_word() if torch.cuda.is_available() else None
        what = self.describe() if callable(self.describe) else (self.describe or '')
        sys.stderr.write(f'[srvp_amd watchdog] rank {self.rank}: step {self.step + 1} has not completed after {waited:.1f} s (limit {self._limit():g} s).  '
                         f'srvp_last_error: {err!r}; cluster-barrier timeouts: {word}; {what}\n')
        sys.stderr.flush()


class PeerStats:
    """PROTOTYPE (SRVP_COMM=peer): SyncBatchNorm statistics through one-sided peer reads of hipIpc-shared slabs instead of all-reduce
    collectives (csrc/comm.hip: srvp_peer_*).  One slot (two parities) per call site, assigned in order of first use -- the launch
    sequence is the same on every rank; seq counts the uses of the slot.  Ranks of ONE node only (hipIpc), world <= 8."""

    SLOTS, NMAX = 128, 2 * 2048          # call sites (21 BatchNorm layers x 2 directions = 42 for VGG), doubles per collective

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise L.SrvpHipError('peer statistics exchange: at most 8 ranks (one node)')
        lib = L.load()
        self.stride = self.NMAX * 8 + 128                    # data + a 128-byte line for the flag
        nbytes = self.SLOTS * 2 * self.stride
        self.own, handle = C.c_void_p(), C.create_string_buffer(64)
        err = None
        if lib.srvp_peer_slab_create(nbytes, C.byref(self.own), handle) != 0:
            err = lib.srvp_last_error().decode()
        # every rank reaches the gather whatever happened locally; a failure anywhere fails the construction everywhere
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (err, bytes(handle.raw)), group=group)
        self.ptrs = (C.c_void_p * self.world)()
        bad = [f'rank {r}: {e}' for r, (e, _) in enumerate(gathered) if e]
        if bad:
            if err is None:
                lib.srvp_peer_slab_close(self.own, 1)
            raise L.SrvpHipError('; '.join(bad))
        handles = [h for _, h in gathered]
        for r, h in enumerate(handles):
            if r == self.rank:
                self.ptrs[r] = self.own
            else:
                p = C.c_void_p()
                L.check(lib.srvp_peer_slab_open(h, C.byref(p)), 'srvp_peer_slab_open')
                self.ptrs[r] = p
        self.err = torch.zeros(1, dtype=torch.int32, device='cuda')
        self.slot_of, self.uses = {}, {}

    def allreduce(self, t, site=None):
        """`site`: the call site's identity, the same on every rank whatever its allocation history -- (BatchNorm layer key,
        direction), fixed by the model definition; training plans of different batch sizes share the slot of a layer (they run in
        the same order on every rank).  Without one the tensor's address stands in (self-test only)."""
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64 and t.numel() <= self.NMAX
        key = site if site is not None else t.data_ptr()
        slot = self.slot_of.setdefault(key, len(self.slot_of))
        if slot >= self.SLOTS:
            raise L.SrvpHipError('peer statistics exchange: more call sites than slots')
        n = self.uses.get(slot, 0) + 1
        self.uses[slot] = n
        off = (slot * 2 + (n & 1)) * self.stride
        L.call('srvp_peer_allreduce_f64', L.ptr(t), t.numel(), self.rank, self.world, self.ptrs, off, off + self.NMAX * 8, n, L.ptr(self.err), L.stream())

    def check(self):
        """(synchronises) raises if a collective gave up waiting for a peer."""
        if int(self.err.item()) != 0:
            raise L.SrvpHipError('peer statistics exchange: a rank gave up waiting for a peer that never published its sums')

    def self_test(self):
        want = self.world * (self.world + 1) / 2
        t = torch.full((2, 64), float(self.rank + 1), dtype=torch.float64, device='cuda')
        for _ in range(3):                                   # three uses: both parities and the reuse of the first
            u = t.clone()
            self.slot_of[u.data_ptr()] = self.SLOTS - 1      # (the last slot is the test's)
            self.allreduce(u)
            del self.slot_of[u.data_ptr()]
            if not bool((u == want).all().item()):
                return False
        return int(self.err.item()) == 0

    def close(self):
        lib = L.load()
        for r in range(self.world):
            if self.ptrs[r]:
                lib.srvp_peer_slab_close(self.ptrs[r], 1 if r == self.rank else 0)
                self.ptrs[r] = None


class Sync:
    def __init__(self, group=None, stat_group=None, native=None):
        self.group = group
        # BatchNorm statistics travel on their OWN communicator: RCCL executes the collectives of one communicator in issue
        # order, and the decoder's gradient slice (issued from the second stream, behind ~5 ms of weight-gradient kernels)
        # would otherwise hold up every statistics all-reduce of the encoder backward issued after it
        self.stat_group = stat_group if stat_group is not None else group
        self.world = dist.get_world_size(group)
        self.handles = []
        self.sync_bn = True
        # host meeting point of after_rank0_phase: a gloo group with a day-long timeout.  Not the training backend's barrier: rank 0's
        # validation may take longer than ProcessGroupNCCL's collective timeout (10 minutes by default), which would abort a healthy job
        self.host_group = group
        if self.world > 1 and dist.get_backend(group) != 'gloo':
            import datetime
            if os.environ.get('MASTER_ADDR') in ('127.0.0.1', 'localhost'):
                os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')     # one node: the loopback interface (the container's hostname may not resolve)
            try:
                self.host_group = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group), backend='gloo',
                                                 timeout=datetime.timedelta(hours=24))
            except Exception as e:                       # noqa: BLE001 -- (every rank of the node fails alike: same host, same interfaces)
                print(f'srvp_amd.distributed: no gloo host group ({e}); rank-0 phases meet on the training backend\'s barrier')
        # SRVP_FORCE_COLLECTIVES=1: issue every collective even on a single rank (exercises the RCCL call path on a
        # 1-GPU box: tests/test_gpu_model.py::test_single_rank_collectives)
        self.force = os.environ.get('SRVP_FORCE_COLLECTIVES', '0') == '1'
        self.native_grads = self.native_stats = None
        if native is None:
            native = (dist.get_backend(group) == 'nccl' and os.environ.get('SRVP_COMM', 'rccl') != 'torch'
                      and torch.cuda.is_available() and (self.world > 1 or self.force))
        if native:
            made, why = [], None

            def agree(ok):
                flag = torch.tensor([1 if ok else 0], device='cuda')
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)      # every rank takes the same path
                return int(flag.item()) == 1
            try:
                for _ in range(2):                       # gradients, statistics
                    made.append(NativeComm(group))
                ok = all(c.init_error is None for c in made)
                why = next((c.init_error for c in made if c.init_error), None)
            except L.SrvpHipError as e:                  # raised on EVERY rank (NativeComm.__init__), so all ranks are here together
                ok, why = False, str(e)
            # two rounds of agreement: no rank enters the self-test's collectives unless every rank holds both communicators
            ok = agree(ok) and agree(all(c.self_test() for c in made))
            if ok:
                self.native_grads, self.native_stats = made
            else:
                for c in made:                           # nothing created so far is leaked
                    c.close()
                print(f'srvp_amd.distributed: native RCCL path unavailable ({why or "failed on another rank / self-test"}); using torch.distributed')
        self.transport = 'rccl (C ABI, in-stream)' if self.native_stats is not None else f'torch.distributed ({dist.get_backend(group)})'
        # SRVP_COMM=peer (prototype): statistics through peer reads of hipIpc-shared slabs; gradients keep the transport chosen above
        self.peer = None
        if os.environ.get('SRVP_COMM') == 'peer' and torch.cuda.is_available() and (self.world > 1 or self.force):
            made = None
            try:
                made = PeerStats(group)
                ok = True
            except L.SrvpHipError as e:
                print(f'srvp_amd.distributed: peer statistics exchange unavailable ({e})')
                ok = False
            agree = lambda v: self._agree(v, group)
            ok = agree(ok) and agree(made.self_test())
            if ok:
                self.peer = made
                self.transport += ' + peer-read statistics (hipIpc slabs)'
            elif made is not None:
                made.close()

    def describe(self):
        """One line for logs / the watchdog: world size and the transport of each exchange."""
        return (f'world {self.world}, statistics: {self.stats_transport()}, gradients: {self.grads_transport()}')

    def stats_transport(self):
        if self.peer is not None:
            return 'peer reads of hipIpc slabs (prototype)'
        return 'rccl, C ABI, in-stream' if self.native_stats is not None else f'torch.distributed ({dist.get_backend(self.stat_group)})'

    def grads_transport(self):
        return 'rccl, C ABI, in-stream' if self.native_grads is not None else f'torch.distributed ({dist.get_backend(self.group)})'

    def diagnostics(self, stats_channels=512, grad_mb=95, samples=200):
        """numbers that let a scaling curve explain itself, measured on the transports the step really uses -- called by
        every rank (collective).  Returns dict(transport per exchange, what RCCL reports about each native communicator, the in-stream
        latency of ONE SyncBatchNorm statistics all-reduce ([2][C] fp64, `samples` of them back to back behind a dependent kernel each:
        median / p90 in us), and the algorithm bandwidth of ONE gradient all-reduce of `grad_mb` MB (the VGG model's 23.85 M fp32 gradients
        are 95 MB): GB/s = bytes / time, bus bandwidth = 2 (n - 1) / n of it)."""
        d = dict(world=self.world, statistics=self.stats_transport(), gradients=self.grads_transport())
        if self.native_stats is not None:
            d['rccl_statistics_comm'] = self.native_stats.info()
            d['rccl_gradients_comm'] = self.native_grads.info()
        if not torch.cuda.is_available() or not (self.world > 1 or self.force):
            return d
        st = torch.zeros(2, stats_channels, dtype=torch.float64, device='cuda')
        evs = []
        for i in range(samples + 20):
            st.add_(1.0)                                     # the dependent kernel in front (a BatchNorm layer's sums are written just before)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.allreduce_stats(st, 1.0, site=('diagnostics', 'f'))
            e1.record()
            st.mul_(1.0 / self.world)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[20:])
        d['statistics_allreduce_us'] = dict(bytes=st.numel() * 8, samples=len(us), median=us[len(us) // 2], p90=us[int(len(us) * 0.9)], min=us[0])
        n = grad_mb * 1000 * 1000 // 4
        g = torch.ones(n, dtype=torch.float32, device='cuda')
        ts = []
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if self.native_grads is not None:
                self.native_grads.allreduce(g)
            else:
                dist.all_reduce(g, group=self.group)
            e1.record()
            torch.cuda.synchronize()
            g.fill_(1.0)
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts[1:])[len(ts[1:]) // 2]
        gbs = n * 4 / (ms * 1e-3) / 1e9
        d['gradient_allreduce'] = dict(bytes=n * 4, ms=ms, algbw_GBps=gbs, busbw_GBps=gbs * 2 * (self.world - 1) / max(self.world, 1))
        return d

    @staticmethod
    def _agree(ok, group):
        """True iff `ok` on every rank (the decision must be the same everywhere).  On the backend's own device type."""
        dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
        flag = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    def allreduce_stats(self, t, count, site=None):
        """In-place sum of a small fp64 statistics tensor over ranks; returns the global element count.  `site` names the call site
        (BatchNorm key, direction) for the transports that keep per-call-site state."""
        if self.sync_bn and (self.world > 1 or self.force):
            if self.peer is not None:
                self.peer.allreduce(t, site)
            elif self.native_stats is not None:
                self.native_stats.allreduce(t)
            else:
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

    def _poll_peer_error(self):
        """Once per step, without a device sync: the error word of the peer exchange as it stood one step ago."""
        if self.peer is None:
            return
        host = self.__dict__.get('_peer_err_host')
        if host is None:
            host = self._peer_err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._peer_err_event = None
        if self._peer_err_event is not None and self._peer_err_event.query() and int(host[0]) != 0:
            raise L.SrvpHipError('peer statistics exchange: a rank gave up waiting for a peer that never published its sums '
                                 '(the sums of that collective were replaced by NaN); set SRVP_COMM=rccl')
        host.copy_(self.peer.err, non_blocking=True)
        self._peer_err_event = torch.cuda.Event()
        self._peer_err_event.record()

    # ---- gradient exchange (reference train.py:309-314: DistributedDataParallel's bucketed all-reduce overlapped with backward, averaged over
    # ranks).  The flat fp32 gradient buffer is exchanged in SLICES, each as soon as the backward has completed it (model._backward_impl:
    # decoder tail, decoder head, encoder deep stages, latent networks; only the first encoder stages -- < 1 MB -- are left for the step's end),
    # all on ONE stream and ONE communicator in the same order on every rank.  The average rides the collective (ncclAvg) on the native
    # transport; torch.distributed transports sum and grads_finish() scales.  SRVP_GRAD_BF16=1: bf16 payload (half the bytes over xGMI; the
    # sum is then formed in bf16 by RCCL: relative error <= 2^-8 per addend, tests/test_two_rank_equality.py states the tolerance).
    def reduce_slice(self, model, lo, hi):
        """All-reduce (average over ranks) of flat gradient elements [lo, hi), enqueued on the CURRENT stream."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        flat_g = model._flat[1]
        sl = flat_g[lo:hi]
        bf16 = os.environ.get('SRVP_GRAD_BF16', '0') == '1'
        if bf16:
            buf = self.__dict__.get('_bf16_buf')
            if buf is None or buf.numel() != flat_g.numel() or buf.device != flat_g.device:
                buf = self._bf16_buf = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
            pay = buf[lo:hi]
        if self.native_grads is not None:
            st = L.stream()
            if bf16:
                L.call('srvp_cast_f32_bf16', L.ptr(sl), L.ptr(pay), 1, hi - lo, hi - lo, st)
                L.call('srvp_allreduce', self.native_grads.handle, L.ptr(pay), hi - lo, 2, 1, st)
                L.call('srvp_cast_bf16_f32', L.ptr(pay), L.ptr(sl), hi - lo, 1.0, st)
            else:
                L.call('srvp_allreduce', self.native_grads.handle, L.ptr(sl), hi - lo, 0, 1, st)
            return
        if bf16:
            pay.copy_(sl)
            self.handles.append((dist.all_reduce(pay, group=self.group, async_op=True), pay, sl))
        else:
            self.handles.append((dist.all_reduce(sl, group=self.group, async_op=True), None, None))
        self._pending_scale = True

    def grads_finish(self, model):
        """After the last reduce_slice of a step (same stream): torch.distributed transports wait for their handles and apply the 1/world
        average; the native transport has nothing left to do (ncclAvg).  Also the once-per-step poll of the peer exchange's error word."""
        if self.world == 1 and not self.force:
            return
        self._poll_peer_error()
        if self.handles:
            for h, pay, sl in self.handles:
                h.wait()
                if pay is not None:
                    sl.copy_(pay)
            self.handles = []
        if self.__dict__.get('_pending_scale'):
            model._flat[1].mul_(1.0 / self.world)     # DDP averages gradients over ranks
            self._pending_scale = False

    def grads_ready(self, what, model):
        """Two-phase form kept for callers of the older interface (tests): 'decoder' = the decoder slice, 'all' = the rest + finish."""
        if self.world == 1 and not self.force:
            return
        enc_end, dec_end, total = self._slices(model)
        if what == 'decoder':
            self.reduce_slice(model, enc_end, dec_end)
        else:
            self.reduce_slice(model, 0, enc_end)
            self.reduce_slice(model, dec_end, total)
            self.grads_finish(model)

    def after_rank0_phase(self):
        """Called by EVERY rank at an iteration where rank 0 alone did something long (validation, checkpoint writes; reference
        train.py:355-366 has no barrier there, an RCCL all-reduce simply blocks until rank 0 arrives).  The ranks meet on the HOST here,
        on every transport: without it ranks >= 1 enter the next step, start their StepWatchdog clock and sit in the step's
        first statistics all-reduce for as long as rank 0 validates / writes -- a healthy job killed after SRVP_WATCHDOG_S; and the peer
        exchange waits on the device with a deadline.  A monitored barrier is not needed: a rank-0 phase has no step clock running."""
        if self.world > 1:
            dist.barrier(group=self.host_group)

    def broadcast(self, t):
        if self.native_grads is not None:
            self.native_grads.broadcast(t, 0)
        else:
            dist.broadcast(t, 0, group=self.group)


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
        # a step that does not complete ends the job with a report instead of hanging it (StepWatchdog; SRVP_WATCHDOG_S=0: off)
        if (sync.world > 1 or sync.force) and float(os.environ.get('SRVP_WATCHDOG_S', '30')) > 0:
            module.__dict__['_watchdog'] = StepWatchdog(rank=dist.get_rank(sync.group) if dist.is_initialized() else 0, describe=sync.describe)
        # same initial parameters / buffers on every rank (DDP broadcasts from rank 0)
        if sync.world > 1 or sync.force:
            module.flatten_parameters_()
            sync.broadcast(module._flat[0])
            for b in module.buffers():
                if b.is_cuda and b.numel() > 0:
                    sync.broadcast(b)
                else:
                    dist.broadcast(b, 0, group=sync.group)

    def forward(self, *a, **k):
        return self.module(*a, **k)
