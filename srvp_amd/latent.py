"""
Latent path of SRVP on the HIP library: content variable w, initial state y_0, posterior LSTM + q_z, prior p_z and the
residual Euler rollout, forward and backward (reference module/srvp.py:229-413, module/mlp.py, module/utils.py).
fp32 throughout; every GEMM / recurrence / sampling step is a libsrvp_hip.so kernel, torch only allocates, views and
gathers rows.
"""
import ctypes as C

import torch

from . import _lib as L


def _gemm(st, A, a_rs, a_cs, B, b_rs, b_cs, bias, Cm, c_rs, M, N, K, act=L.ACT_NONE, acc=0):
    L.call('srvp_gemm_f32', L.ptr(A), a_rs, a_cs, L.ptr(B), b_rs, b_cs, L.ptr(bias), L.ptr(Cm), c_rs, M, N, K, act, acc, st)


def linear_fwd(st, x, w, b, out, act=L.ACT_NONE):
    """out[M][N] = act(x[M][K] w[N][K]^T + b)"""
    M, K = x.shape
    N = w.shape[0]
    _gemm(st, x, K, 1, w, 1, K, b, out, N, M, N, K, act)
    return out


def linear_bwd(st, x, w, dy, gw, gb, dx=None, dx_acc=0, defer=None):
    """dy: [M][N] (contiguous).  gw += dy^T x ; gb += colsum(dy) ; dx (+)= dy w
    defer (list or None): the weight / bias gradient launch is appended as a closure fn(stream) instead of being issued (it feeds
    nothing but the optimizer: the caller runs it on a second stream)."""
    M, K = x.shape
    N = w.shape[0]
    wg = lambda s_: L.call('srvp_linear_wgrad_f32', L.ptr(dy), N, L.ptr(x), K, L.ptr(gw), K, L.ptr(gb), N, K, M, s_)      # weight + bias gradient
    if defer is not None:
        defer.append(wg)
    else:
        wg(st)
    if dx is not None:
        _gemm(st, dy, N, 1, w, K, 1, None, dx, K, M, K, N, acc=dx_acc)


def axpby(st, out, a, x, b=0.0, y=None):
    L.call('srvp_axpby_f32', L.ptr(out), float(a), L.ptr(x), float(b), L.ptr(y), out.numel(), st)


import os
PZ_BATCHED = os.environ.get('SRVP_PZ_BATCHED', '1') != '0'
LSTM_BWD_FUSED = os.environ.get('SRVP_LSTM_BWD_FUSED', '1') != '0'
ROLLOUT_GEN_FUSED = True      # inference chain (p_z inside the loop) as persistent launches; False: per-layer launch sequence (A/B switch of the tests)


def mlp_keys(prefix, n):
    return [f'{prefix}.module.{i}.{0 if i == 0 else 1}' for i in range(n)]


class LatentNet:
    """Buffers + launch sequence for one (T, B, nt, n_euler) problem size."""

    def __init__(self, cfg, T, B, nt, n_euler, device, training):
        self.cfg, self.T, self.B, self.nt, self.ne, self.dev, self.training = cfg, T, B, nt, n_euler, device, training
        f32 = dict(dtype=torch.float32, device=device)
        nhx, nh, ny, nz, nhr = cfg['nhx'], cfg['nh_inf'], cfg['ny'], cfg['nz'], cfg['nh_res']
        self.nl_inf, self.nl_res, self.nt_inf = cfg['nlayers_inf'], cfg['nlayers_res'], cfg['nt_inf']
        S = n_euler * (nt - 1)
        F = nt - 1
        self.S, self.F = S, F
        ti = self.nt_inf
        z = lambda *s: torch.zeros(*s, **f32)
        # content variable
        self.h_sel = z(ti * B, nhx)
        self.proj = z(ti * B, nh)
        self.hsum = z(B, nh)
        self.w = z(B, nh)
        # y_0
        self.qy_in = z(B, ti * nhx)
        self.qy_hid = [z(B, nh) for _ in range(self.nl_inf - 1)]
        self.q_y0 = z(B, 2 * ny)
        self.y0 = z(B, ny)
        # LSTM / q_z
        if T > 0:
            self.lstm_bias = z(4 * nh)
            self.gates_x = z(T * B, 4 * nh)
            self.gates_act = z(T * B, 4 * nh)
            self.hz = z(T * B, nh)
            self.cz = z(T * B, nh)
        self.q_z = z(max(F, 1), B, 2 * nz)
        # rollout
        self.y_all = z(S + 1, B, ny)
        self.z = z(max(F, 1), B, nz)
        self.p_z = z(max(F, 1), B, 2 * nz)
        self.res = z(max(S, 1), B, ny)
        self.inp_all = z(max(S, 1), B, ny + nz)
        self.scratch_out = z(B, max(ny, 2 * nz))
        nlr = self.nl_res
        if training:
            self.hid_dyn = z(nlr - 1, max(S, 1), B, nhr)
            self.hid_pz = z(nlr - 1, max(F, 1), B, nhr)
            self._pz_in = z(max(F, 1) * B, ny)
            self._pz_dx = z(max(F, 1) * B, ny)
            self.scratch_hid = None
            dwd, dwp = max(nhr, ny), max(nhr, 2 * nz)
            self.dwd, self.dwp = dwd, dwp
            self.dhid_dyn = z(nlr, max(S, 1), B, dwd)
            self.dhid_pz = z(nlr, max(F, 1), B, dwp)
            self.work = z(3 * B * ny + B * (ny + nz) + B * nz)
            self.dinp_all = z(max(S, 1), B, ny + nz)
            self.d_y_all = z(S + 1, B, ny)
            self.d_y0 = z(B, ny)
            self.d_qz_samp = z(max(F, 1), B, 2 * nz)
            self.d_qy0_tot = z(B, 2 * ny)
            self.d_qz_tot = z(max(F, 1), B, 2 * nz)
            self.d_hz = z(max(T, 1) * B, nh)
            self.dgates = z(max(T, 1) * B, 4 * nh)
            self.lstm_scratch = z(2 * B * nh)
            self.d_hx = z(max(T, 1) * B, nhx)
            self.d_qy_hid = [z(B, nh) for _ in range(self.nl_inf - 1)]
            self.d_qy_in = z(B, ti * nhx)
            self.d_wpre = z(B, nh)
            self.d_hsum = z(B, nh)
            self.d_proj = z(ti * B, nh)
            self.d_hsel = z(ti * B, nhx)
        else:
            self.hid_dyn = self.hid_pz = None
            self.scratch_hid = z(nlr - 1, B, nhr)

    # ------------------------------------------------------------------------------------------------
    def _rollout_desc(self, params, n_data, eps_z, y0, nsteps=None):
        cfg = self.cfg
        d = L.RolloutDesc()
        d.B, d.ny, d.nz, d.nh, d.nl = self.B, cfg['ny'], cfg['nz'], cfg['nh_res'], self.nl_res
        d.nsteps, d.n_euler, d.n_data_frames, d.dt = self.S, self.ne, n_data, 1.0 / self.ne
        for i, k in enumerate(mlp_keys('dynamics', self.nl_res)):
            d.dyn_w[i], d.dyn_b[i] = L.ptr(params[k + '.weight']), L.ptr(params[k + '.bias'])
        for i, k in enumerate(mlp_keys('p_z', self.nl_res)):
            d.pz_w[i], d.pz_b[i] = L.ptr(params[k + '.weight']), L.ptr(params[k + '.bias'])
        d.y0, d.q_z_params, d.eps_z = L.ptr(y0), L.ptr(self.q_z), L.ptr(eps_z)
        d.y_all, d.z, d.p_z_params, d.res = L.ptr(self.y_all), L.ptr(self.z), L.ptr(self.p_z), L.ptr(self.res)
        d.inp_all, d.hid_dyn, d.hid_pz = L.ptr(self.inp_all), L.ptr(self.hid_dyn), L.ptr(self.hid_pz)
        d.scratch_hid, d.scratch_out = L.ptr(self.scratch_hid), L.ptr(self.scratch_out)
        return d

    def infer_w(self, hx, params, t_w, st):
        """srvp.py:229-256.  hx: (T, B, nhx) fp32 contiguous; t_w: (nt_inf, B) long (train) or None (last nt_inf frames)."""
        B, ti = self.B, self.nt_inf
        T = hx.shape[0]
        if t_w is not None and not t_w.is_cuda:
            # frame indices still on the host (the training forward draws them there): the row numbers are formed on the host too and travel
            # in ONE copy instead of five small device kernels on the latent path's serial chain -- through ONE pinned staging buffer and ONE
            # device buffer per plan (round 6: no pinned allocation per step; fixed addresses, so a captured step can be replayed after the
            # host has refilled the staging buffer, fill_w_rows_host)
            if self.__dict__.get('_w_rows_copied') is not None and not torch.cuda.is_current_stream_capturing():
                self._w_rows_copied.synchronize()          # (a caller that runs two forwards without a sync in between)
            host = self.fill_w_rows_host(t_w)
            rows = self.__dict__.get('_w_rows_dev')
            if rows is None:
                rows = self._w_rows_dev = torch.zeros(ti * B, dtype=torch.long, device=hx.device)
            rows.copy_(host, non_blocking=True)
            self._w_rows_copied = L.record()
        elif t_w is not None:
            rows = (t_w.reshape(-1) * B + torch.arange(B, device=hx.device).repeat(ti)).to(torch.long)
        else:
            rows = torch.arange((T - ti) * B, T * B, device=hx.device)
        self.w_rows = rows
        torch.index_select(hx.view(T * B, -1), 0, rows, out=self.h_sel)
        linear_fwd(st, self.h_sel, params['w_proj.0.weight'], params['w_proj.0.bias'], self.proj, L.ACT_RELU)
        pv = self.proj.view(ti, B, -1)
        axpby(st, self.hsum, 1.0, pv[0], 1.0 if ti > 1 else 0.0, pv[1] if ti > 1 else None)
        for i in range(2, ti):
            axpby(st, self.hsum, 1.0, self.hsum, 1.0, pv[i])
        linear_fwd(st, self.hsum, params['w_inf.0.weight'], params['w_inf.0.bias'], self.w, L.ACT_TANH)
        return self.w

    def fill_w_rows_host(self, t_w):
        """t_w (nt_inf, B) CPU frame indices -> the rows t_w[i][b] * B + b of hx they select, in the plan's pinned staging buffer."""
        B, ti = self.B, self.nt_inf
        host = self.__dict__.get('_w_rows_host')
        if host is None:
            host = self._w_rows_host = torch.zeros(ti * B, dtype=torch.long).pin_memory()
        torch.add(t_w.reshape(-1).to(torch.long) * B, torch.arange(B).repeat(ti), out=host)
        return host

    def infer_y(self, hx_first, params, eps_y0, st):
        """srvp.py:258-278.  hx_first: (nt_inf, B, nhx)."""
        B = self.B
        self.qy_in.view(B, self.nt_inf, -1).copy_(hx_first.permute(1, 0, 2))
        keys = mlp_keys('q_y', self.nl_inf)
        cur = self.qy_in
        for i, k in enumerate(keys):
            last = i == len(keys) - 1
            dst = self.q_y0 if last else self.qy_hid[i]
            linear_fwd(st, cur, params[k + '.weight'], params[k + '.bias'], dst, L.ACT_NONE if last else L.ACT_RELU)
            cur = dst
        L.call('srvp_rsample_fwd', L.ptr(self.q_y0), L.ptr(eps_y0), L.ptr(self.y0), B, self.cfg['ny'], st)
        return self.y0, self.q_y0

    def lstm_bias_prep(self, params, st):
        """b_ih + b_hh of the posterior LSTM (it depends on the parameters only: the caller may form it ahead of the encoder's output)."""
        axpby(st, self.lstm_bias, 1.0, params['inf_z.bias_ih_l0'], 1.0, params['inf_z.bias_hh_l0'])

    def posterior(self, hx, params, st, bias_ready=False):
        """LSTM over the frame encodings + q_z (srvp.py:366,387,296): fills q_z[f] for frames f+1 < T."""
        T, B = hx.shape[0], self.B
        nh, nz = self.cfg['nh_inf'], self.cfg['nz']
        if not bias_ready:
            self.lstm_bias_prep(params, st)
        linear_fwd(st, hx.view(T * B, -1), params['inf_z.weight_ih_l0'], self.lstm_bias, self.gates_x[:T * B])
        need = int(L.load().srvp_lstm_fused_ws_bytes(T, B, nh))          # 0: shape not eligible for the persistent kernel
        if need > 0:
            ws = self.__dict__.get('_lstm_ws')
            if ws is None or ws.numel() < need:
                ws = self._lstm_ws = torch.zeros(need, dtype=torch.uint8, device=self.dev)
            L.call('srvp_lstm_fwd_fused', L.ptr(self.gates_x), L.ptr(params['inf_z.weight_hh_l0']), L.ptr(self.hz), L.ptr(self.cz),
                   L.ptr(self.gates_act), T, B, nh, L.ptr(ws), need, st)
        else:
            L.call('srvp_lstm_fwd', L.ptr(self.gates_x), L.ptr(params['inf_z.weight_hh_l0']), L.ptr(self.hz), L.ptr(self.cz),
                   L.ptr(self.gates_act), T, B, nh, st)
        if T > 1:
            nq = min(T - 1, self.F)
            linear_fwd(st, self.hz[B:(nq + 1) * B], params['q_z.weight'], params['q_z.bias'], self.q_z.view(-1, 2 * nz)[:nq * B])

    def generate(self, y0, n_data, params, eps_z, st, pz_stream=None):
        """srvp.py:325-413 (remove_intermediate=True); the LSTM/q_z part must have run if n_data > 1.
        pz_stream (torch stream, optional; training): the batched prior MLP p_z(y_t) -- it feeds the KL term only, not the decoder -- runs
        there behind the rollout; self.pz_done is the event its readers wait for (None when it ran in line)."""
        if self.S > 0:
            self._rd = self._rollout_desc(params, n_data, eps_z, y0)
            # training (every frame has data): z always comes from the posterior, so the prior MLP p_z(y_t) only feeds the KL
            # term -- it leaves the serial chain and runs once, batched over all frames, on the stored states
            self.pz_ext = bool(self.training and n_data >= self.nt and PZ_BATCHED and self.dwp == self.cfg['nh_res'])
            self._rd.pz_external = 1 if self.pz_ext else 0
            if self.pz_ext:
                # persistent fused chain (csrc/rollout_fused.hip): the library says how much workspace it wants, 0 = not eligible
                need = int(L.load().srvp_rollout_fused_ws_bytes(C.byref(self._rd)))
                if need > 0:
                    ws = self.__dict__.get('_fused_ws')
                    if ws is None or ws.numel() < need:
                        ws = self._fused_ws = torch.zeros(need, dtype=torch.uint8, device=self.dev)
                    self._rd.fused_ws, self._rd.fused_ws_bytes = L.ptr(ws), need
            elif not self.training and ROLLOUT_GEN_FUSED:
                # inference chain (p_z inside the loop): persistent launches where the library takes it (0 = per-layer launch sequence)
                need = int(L.load().srvp_rollout_gen_ws_bytes(C.byref(self._rd)))
                if need > 0:
                    ws = self.__dict__.get('_gen_ws')
                    if ws is None or ws.numel() < need:
                        ws = self._gen_ws = torch.zeros(need, dtype=torch.uint8, device=self.dev)
                    self._rd.fused_ws, self._rd.fused_ws_bytes = L.ptr(ws), need
            L.call('srvp_rollout_fwd', C.byref(self._rd), st)
            self.pz_done = None
            if self.pz_ext:
                B, F, nlr, ny, nz, nhr = self.B, self.F, self.nl_res, self.cfg['ny'], self.cfg['nz'], self.cfg['nh_res']

                def prior(s_):
                    cur = self._pz_in
                    cur.view(F, B, ny).copy_(self.y_all[0:self.S:self.ne])      # states at the frame starts as contiguous rows
                    for l, k in enumerate(mlp_keys('p_z', nlr)):
                        last = l == nlr - 1
                        out = self.p_z.view(F * B, 2 * nz) if last else self.hid_pz[l].view(F * B, nhr)
                        linear_fwd(s_, cur, params[k + '.weight'], params[k + '.bias'], out, L.ACT_NONE if last else L.ACT_RELU)
                        cur = out
                if pz_stream is None:
                    prior(st)
                else:
                    ev = L.record()
                    with L.on_stream(pz_stream):
                        pz_stream.wait_event(ev)
                        prior(L.stream())
                        self.pz_done = L.record()
        else:
            self.y_all[0].copy_(y0)
        y = self.y_all[::self.ne]
        nq = max(min(n_data, self.nt) - 1, 0)
        return y, self.z[:self.F], (self.q_z[:nq] if nq > 0 else None), self.p_z[:self.F], self.res[:self.S]

    # ------------------------------------------------------------------------------------------------
    def pz_backward_chain(self, params, d_pz, st):
        """Backward of the batched prior MLP p_z(y_t) of a training forward (generate, pz_ext) down to its input gradient self._pz_dx: fills
        the deltas self.dhid_pz the weight gradients read.  It needs d_pz (the KL term's gradient) only, which exists before the decoder
        backward starts: the caller may run it on another stream under the decoder backward (round 5: ~10 dependent micro-kernels, 90 us at
        24 sequences, sat between the decoder's and the rollout's backward on the step's serial path)."""
        B, F, nlr = self.B, self.F, self.nl_res
        ny, nz, nhr = self.cfg['ny'], self.cfg['nz'], self.cfg['nh_res']
        keys = mlp_keys('p_z', nlr)
        dwp = self.dwp
        top = self.dhid_pz[nlr - 1].view(F * B, dwp)
        if d_pz is not None:
            top[:, :2 * nz].copy_(d_pz.reshape(F * B, 2 * nz))
        else:
            top.zero_()
        for l in range(nlr - 1, 0, -1):
            w = params[keys[l] + '.weight']                        # [cout][nhr]
            cout = 2 * nz if l == nlr - 1 else nhr
            dst = self.dhid_pz[l - 1].view(F * B, dwp)             # dwp == nhr (checked when pz_ext was chosen)
            _gemm(st, self.dhid_pz[l].view(F * B, dwp), dwp, 1, w, nhr, 1, None, dst, dwp, F * B, nhr, cout)
            L.call('srvp_act_bwd_f32', L.ptr(self.hid_pz[l - 1]), L.ptr(dst), L.ptr(dst), F * B * nhr, L.ACT_RELU, 1, st)
        _gemm(st, self.dhid_pz[0].view(F * B, dwp), dwp, 1, params[keys[0] + '.weight'], ny, 1, None, self._pz_dx, ny, F * B, ny, nhr)

    def backward(self, hx, params, grads, eps_y0, eps_z, d_y, d_w, d_qy0, d_qz, d_pz, d_res, d_z, st, defer=None, aux=None, pz_pre=None):
        """
        Gradients wrt the latent-path outputs -> parameter gradients (accumulated into `grads`) and d_hx (T*B, nhx).
        Any d_* may be None.  Training forward (n_data = T = nt) must have run.
        d_y == 'in_place': the caller has already written the state gradients into self.d_y_all (zero elsewhere).
        defer (list or None): weight-gradient launches (off the critical path) are collected as closures fn(stream).
        aux (torch stream or None; round 5): the backward of the two chains that do not pass through the LSTM -- the content variable w (needs
        d_w only: runs from the start, under the rollout backward) and y_0 (needs the rollout's d_y0: runs beside the q_z / LSTM chain) -- is
        issued there up to their input gradients; the main stream joins it and adds both into d_hx (the row scatter-adds and the LSTM's input
        GEMM all accumulate into d_hx, so those stay in stream order).  A dozen dependent micro-kernels less on the step's serial path.
        """
        cfg, B, T, ne, S, F = self.cfg, self.B, self.T, self.ne, self.S, self.F
        ny, nz, nh, nhr, nhx = cfg['ny'], cfg['nz'], cfg['nh_inf'], cfg['nh_res'], cfg['nhx']
        nlr = self.nl_res
        ti = self.nt_inf
        self.d_hx.zero_()

        def y0_chain(s_):
            L.call('srvp_rsample_bwd', L.ptr(self.q_y0), L.ptr(eps_y0), L.ptr(self.d_y0), L.ptr(self.d_qy0_tot), B, ny, 0, s_)
            if d_qy0 is not None:
                axpby(s_, self.d_qy0_tot, 1.0, self.d_qy0_tot, 1.0, d_qy0)
            keys = mlp_keys('q_y', self.nl_inf)
            dcur = self.d_qy0_tot
            for i in range(len(keys) - 1, -1, -1):
                x_in = self.qy_in if i == 0 else self.qy_hid[i - 1]
                dx = self.d_qy_in if i == 0 else self.d_qy_hid[i - 1]
                linear_bwd(s_, x_in, params[keys[i] + '.weight'], dcur, grads[keys[i] + '.weight'], grads[keys[i] + '.bias'], dx=dx, defer=defer)
                if i > 0:
                    L.call('srvp_act_bwd_f32', L.ptr(self.qy_hid[i - 1]), L.ptr(dx), L.ptr(dx), dx.numel(), L.ACT_RELU, 1, s_)
                dcur = dx

        def w_chain(s_):
            L.call('srvp_act_bwd_f32', L.ptr(self.w), L.ptr(d_w), L.ptr(self.d_wpre), d_w.numel(), L.ACT_TANH, 1, s_)
            linear_bwd(s_, self.hsum, params['w_inf.0.weight'], self.d_wpre, grads['w_inf.0.weight'], grads['w_inf.0.bias'],
                       dx=self.d_hsum, defer=defer)
            dp = self.d_proj.view(ti, B, nh)
            for i in range(ti):
                L.call('srvp_act_bwd_f32', L.ptr(self.proj.view(ti, B, nh)[i]), L.ptr(self.d_hsum), L.ptr(dp[i]), B * nh,
                       L.ACT_RELU, 1, s_)
            linear_bwd(s_, self.h_sel, params['w_proj.0.weight'], self.d_proj, grads['w_proj.0.weight'], grads['w_proj.0.bias'],
                       dx=self.d_hsel, defer=defer)
        if aux is not None and d_w is not None:
            ev0 = L.record()                                    # d_w (and everything the caller prepared) exists
            with L.on_stream(aux):
                aux.wait_event(ev0)
                w_chain(L.stream())
        # ---- rollout
        if not (isinstance(d_y, str) and d_y == 'in_place'):
            self.d_y_all.zero_()
            if d_y is not None:
                self.d_y_all[::ne].copy_(d_y)
        bd = L.RolloutBwdDesc()
        bd.f = self._rd
        bd.d_y_all, bd.d_z, bd.d_pz, bd.d_res = L.ptr(self.d_y_all), L.ptr(d_z), L.ptr(d_pz), L.ptr(d_res)
        bd.d_y0, bd.d_qz, bd.dhid_dyn, bd.dhid_pz, bd.work = (L.ptr(self.d_y0), L.ptr(self.d_qz_samp), L.ptr(self.dhid_dyn),
                                                               L.ptr(self.dhid_pz), L.ptr(self.work))
        bd.dinp_all = L.ptr(self.dinp_all)
        self.d_qz_samp.zero_()
        if getattr(self, 'pz_ext', False) and S > 0:
            # batched p_z backward (see generate): deltas of every frame at once (pz_backward_chain: in line here, or issued by the caller on
            # another stream as soon as d_pz existed -- `pz_pre` is the event behind it), input gradient into d_y_all[f * ne]
            if pz_pre is None:
                self.pz_backward_chain(params, d_pz, st)
            else:
                L.wait(pz_pre)
            L.call('srvp_add_blocks_f32', L.ptr(self.d_y_all), ne * B * ny, L.ptr(self._pz_dx), F, B * ny, st)
        L.call('srvp_rollout_bwd', C.byref(bd), st)
        if aux is not None:
            ev1 = L.record()                                    # d_y0 exists
            with L.on_stream(aux):
                aux.wait_event(ev1)
                y0_chain(L.stream())
                aux_done = L.record()
        # weight gradients of dynamics / p_z: one GEMM per layer over all (step, sample) rows
        for name, nrow, width, dh, hid, inp, nin, nout in (
                ('dynamics', S * B, self.dwd, self.dhid_dyn, self.hid_dyn, self.inp_all.view(S * B, -1), ny + nz, ny),
                ('p_z', F * B, self.dwp, self.dhid_pz, self.hid_pz, None, ny, 2 * nz)):
            keys = mlp_keys(name, nlr)
            for l, k in enumerate(keys):
                cout = nout if l == nlr - 1 else nhr
                delta = dh[l].view(nrow, width)
                if l == 0:
                    a_prev = inp if inp is not None else None
                    cin = nin
                else:
                    a_prev, cin = hid[l - 1].view(nrow, nhr), nhr
                if a_prev is None:
                    # p_z input = state at the start of every frame: y_all[f*ne]
                    # (contiguous rows: with B == 1 the strided slice reshapes to a VIEW whose row stride is ne * ny, not ny)
                    a_prev = self.y_all[0:S:ne].reshape(nrow, ny).contiguous()
                wg = (lambda s_, delta=delta, width=width, a_prev=a_prev, cin=cin, k=k, cout=cout, nrow=nrow:
                      L.call('srvp_linear_wgrad_f32', L.ptr(delta), width, L.ptr(a_prev), cin, L.ptr(grads[k + '.weight']), cin,
                             L.ptr(grads[k + '.bias']), cout, cin, nrow, s_))
                if defer is not None:
                    defer.append(wg)
                else:
                    wg(st)
        # ---- q_z + LSTM
        nq = T - 1
        if nq > 0:
            axpby(st, self.d_qz_tot[:nq], 1.0, self.d_qz_samp[:nq], 1.0 if d_qz is not None else 0.0, d_qz)
            dq = self.d_qz_tot[:nq].view(nq * B, 2 * nz)
            self.d_hz[:B].zero_()
            linear_bwd(st, self.hz[B:T * B], params['q_z.weight'], dq, grads['q_z.weight'], grads['q_z.bias'],
                       dx=self.d_hz[B:T * B], defer=defer)
            ws = self.__dict__.get('_lstm_ws')
            if LSTM_BWD_FUSED and ws is not None and int(L.load().srvp_lstm_fused_ws_bytes(T, B, nh)) > 0:
                # the recurrence as one persistent launch (csrc/rollout_fused.hip), like the forward that allocated the workspace
                L.call('srvp_lstm_bwd_fused', L.ptr(self.d_hz), L.ptr(params['inf_z.weight_hh_l0']), L.ptr(self.cz), L.ptr(self.gates_act),
                       L.ptr(self.dgates), T, B, nh, L.ptr(ws), ws.numel(), st)
            else:
                L.call('srvp_lstm_bwd', L.ptr(self.d_hz), L.ptr(params['inf_z.weight_hh_l0']), L.ptr(self.cz), L.ptr(self.gates_act),
                       L.ptr(self.dgates), L.ptr(self.lstm_scratch), T, B, nh, st)
            dg = self.dgates[:T * B]
            # W_hh: sum_t dgates[t]^T h_{t-1}
            whh = lambda s_: _gemm(s_, dg[B:], 1, 4 * nh, self.hz[:(T - 1) * B], nh, 1, None, grads['inf_z.weight_hh_l0'], nh, 4 * nh, nh,
                                   (T - 1) * B, acc=1)
            bhh = lambda s_: L.call('srvp_colsum_f32', L.ptr(dg), 4 * nh, L.ptr(grads['inf_z.bias_hh_l0']), T * B, 4 * nh, 1, s_)
            if defer is not None:
                defer.extend([whh, bhh])
            else:
                whh(st)
            linear_bwd(st, hx.view(T * B, nhx), params['inf_z.weight_ih_l0'], dg, grads['inf_z.weight_ih_l0'],
                       grads['inf_z.bias_ih_l0'], dx=self.d_hx, dx_acc=1, defer=defer)
            if defer is None:
                bhh(st)
        # ---- y_0 and w: see y0_chain / w_chain above; their input gradients are added into d_hx here, in stream order behind the LSTM's
        if aux is None:
            y0_chain(st)
            if d_w is not None:
                w_chain(st)
        else:
            L.wait(aux_done)
        rows = self.__dict__.get('_qy_rows')
        if rows is None:
            rows = self._qy_rows = (torch.arange(ti, dtype=torch.int32).view(1, ti) * B + torch.arange(B, dtype=torch.int32).view(B, 1)).reshape(-1).to(self.dev)
        # d_hx[t][b] += d_qy_in[b][t] for the first nt_inf frames (backward of the (B, nt_inf * nhx) flattening, srvp.py:268)
        L.call('srvp_rows_scatter_add_f32', L.ptr(self.d_hx), L.ptr(rows), 0, L.ptr(self.d_qy_in), B * ti, nhx, st)
        if d_w is not None:
            L.call('srvp_rows_scatter_add_f32', L.ptr(self.d_hx), L.ptr(self.w_rows), 1, L.ptr(self.d_hsel), self.w_rows.numel(), nhx, st)
        return self.d_hx
