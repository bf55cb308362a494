"""
ctypes binding of libsrvp_hip.so (C ABI declared in include/srvp_hip.h).

The library is the product's only compute path: there is no Python / CPU fallback.  If it is missing, loading
fails loudly with instructions to build it (`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SRVP_LIB') or os.path.join(_HERE, 'libsrvp_hip.so')      # SRVP_LIB: A/B a second build in one run
MAX_TAPS = 16

ACT_NONE, ACT_LRELU, ACT_TANH, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3, 4

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p
TAPS = c_i32 * MAX_TAPS
P8 = c_vp * 8


class ConvDesc(C.Structure):
    _fields_ = [('src0', c_vp), ('src1', c_vp), ('map1', c_vp),
                ('C0', c_i32), ('C1', c_i32),
                ('H0p', c_i32), ('W0p', c_i32), ('H1p', c_i32), ('W1p', c_i32),
                ('ups0', c_i32), ('ups1', c_i32), ('si', c_i32),
                ('ntaps', c_i32), ('dy', TAPS), ('dx', TAPS),
                ('wt', c_vp), ('Cout', c_i32),
                ('N', c_i32), ('OH', c_i32), ('OW', c_i32),
                ('dst', c_vp), ('DHp', c_i32), ('DWp', c_i32), ('so', c_i32), ('ooy', c_i32), ('oox', c_i32),
                ('Cdst', c_i32), ('cdst_off', c_i32),
                ('stats', c_vp), ('stat_mod', c_i32),
                ('out_f32', c_vp), ('out_nc', c_i32), ('out_sigmoid', c_i32),
                ('map0', c_vp), ('dst_is_f32', c_i32), ('add_f32', c_vp), ('add_mod', c_i32), ('wt_fragmajor', c_i32),
                ('tap_phase_chunks', c_i32), ('elem_f32', c_i32), ('splitk', c_i32), ('f32_quad', c_i32),
                ('bnr_raw', c_vp), ('bnr_coef', c_vp), ('bnr_red', c_vp), ('ep_coef', c_vp), ('ep_act', c_i32), ('ep_border', c_i32)]


class WgradDesc(C.Structure):
    _fields_ = [('src0', c_vp), ('src1', c_vp), ('map1', c_vp),
                ('C0', c_i32), ('C1', c_i32),
                ('H0p', c_i32), ('W0p', c_i32), ('H1p', c_i32), ('W1p', c_i32),
                ('ups0', c_i32), ('ups1', c_i32), ('si', c_i32),
                ('ntaps', c_i32), ('dy', TAPS), ('dx', TAPS),
                ('dout', c_vp), ('DHp', c_i32), ('DWp', c_i32), ('so', c_i32), ('ooy', TAPS), ('oox', TAPS),
                ('Cout', c_i32), ('N', c_i32), ('OH', c_i32), ('OW', c_i32),
                ('dw', c_vp), ('splitk', c_i32), ('map0', c_vp), ('elem_f32', c_i32), ('dout_cstride', c_i32), ('dout_coff', c_i32), ('dout_phase_taps', c_i32)]


class BnBwdDesc(C.Structure):
    _fields_ = [('raw', c_vp), ('act', c_vp), ('act_border', c_i32),
                ('scale', c_vp), ('shift', c_vp), ('mean', c_vp), ('invstd', c_vp), ('act_kind', c_i32),
                ('da', c_vp), ('da_mode', c_i32), ('da_cstride', c_i32), ('da_coff', c_i32), ('da_border', c_i32),
                ('da_is_f32', c_i32),
                ('da2', c_vp), ('da2_idx', c_vp),
                ('N', c_i32), ('H', c_i32), ('W', c_i32), ('C', c_i32), ('tsum', c_vp), ('tsum_T', c_i32), ('elem_f32', c_i32), ('draw_s2d', c_i32)]


class PackDesc(C.Structure):
    _fields_ = [('ntaps', c_i32), ('tap_off', TAPS), ('J', c_i32), ('K', c_i32),
                ('J0', c_i32), ('J0r', c_i32), ('J1r', c_i32), ('K0', c_i32), ('K0r', c_i32), ('K1r', c_i32),
                ('sj', c_i64), ('sk', c_i64), ('tap_set', TAPS), ('layout', c_i32), ('dst_f32', c_i32), ('kc_total', c_i32), ('kc_off', c_i32)]


class PackJob(C.Structure):
    _fields_ = [('src', c_vp), ('dst', c_vp), ('d', PackDesc)]


class RolloutDesc(C.Structure):
    _fields_ = [('B', c_i32), ('ny', c_i32), ('nz', c_i32), ('nh', c_i32), ('nl', c_i32),
                ('nsteps', c_i32), ('n_euler', c_i32), ('n_data_frames', c_i32), ('dt', c_f32),
                ('dyn_w', P8), ('dyn_b', P8), ('pz_w', P8), ('pz_b', P8),
                ('y0', c_vp), ('q_z_params', c_vp), ('eps_z', c_vp),
                ('y_all', c_vp), ('z', c_vp), ('p_z_params', c_vp), ('res', c_vp),
                ('inp_all', c_vp), ('hid_dyn', c_vp), ('hid_pz', c_vp), ('scratch_hid', c_vp), ('scratch_out', c_vp),
                ('pz_external', c_i32), ('fused_ws', c_vp), ('fused_ws_bytes', c_i64)]


class RolloutBwdDesc(C.Structure):
    _fields_ = [('f', RolloutDesc),
                ('d_y_all', c_vp), ('d_z', c_vp), ('d_pz', c_vp), ('d_res', c_vp),
                ('d_y0', c_vp), ('d_qz', c_vp), ('dhid_dyn', c_vp), ('dhid_pz', c_vp), ('work', c_vp), ('dinp_all', c_vp)]


_SIGS = {
    'srvp_version': ([], c_i32),
    'srvp_stream_create_low_priority': ([C.POINTER(c_vp), C.POINTER(c_i32)], c_i32),
    'srvp_conv_mfma': ([C.POINTER(ConvDesc), c_vp], c_i32),
    'srvp_conv_set_halo': ([c_i32], c_i32),
    'srvp_conv_wants_fragmajor': ([C.POINTER(ConvDesc)], c_i32),
    'srvp_conv_runs_on_halo': ([C.POINTER(ConvDesc)], c_i32),
    'srvp_conv_mfma_multi': ([C.POINTER(ConvDesc), c_i32, c_vp], c_i32),
    'srvp_wgrad_mfma': ([C.POINTER(WgradDesc), c_vp], c_i32),
    'srvp_wgrad_set_tr': ([c_i32], c_i32),
    'srvp_wgrad_set_halo': ([c_i32], c_i32),
    'srvp_bn_finalize': ([c_vp, c_f64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp], c_i32),
    'srvp_bn_eval_coeffs': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp], c_i32),
    'srvp_bn_act': ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp], c_i32),
    'srvp_bn_act_keep': ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp], c_i32),
    'srvp_bn_act_keep_f32': ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp], c_i32),
    'srvp_bn_finalize_act': ([c_vp, c_vp, c_f64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32, c_i32, c_i32, c_i32, c_i32,
                             c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp], c_i32),
    'srvp_bn_act_s2d': ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp], c_i32),
    'srvp_bn_bwd_finalize_apply': ([C.POINTER(BnBwdDesc), c_vp, c_f64, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_i32, c_vp], c_i32),
    'srvp_bn_bwd_reduce': ([C.POINTER(BnBwdDesc), c_vp, c_vp], c_i32),
    'srvp_conv_in_wgrad_bn_ok': ([C.POINTER(BnBwdDesc), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32], c_i32),
    'srvp_conv_in_wgrad_bn': ([c_vp, C.POINTER(BnBwdDesc), c_vp, c_f64, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_bn_bwd_finalize': ([c_vp, c_f64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp], c_i32),
    'srvp_bn_bwd_apply': ([C.POINTER(BnBwdDesc), c_vp, c_vp, c_i32, c_vp], c_i32),
    'srvp_conv_in_fwd': ([c_vp, c_vp, c_vp, c_vp] + [c_i32] * 9 + [c_vp], c_i32),
    'srvp_conv_in_fwd_bnr': ([c_vp, c_vp, c_vp] + [c_i32] * 9 + [c_vp, c_vp, c_vp, c_vp], c_i32),
    'srvp_conv_in_fwd_bnr_ok': ([c_i32] * 7, c_i32),
    'srvp_conv_in_wgrad': ([c_vp, c_vp, c_vp] + [c_i32] * 9 + [c_vp], c_i32),
    'srvp_conv_in_fwd_f32': ([c_vp, c_vp, c_vp, c_vp] + [c_i32] * 9 + [c_vp], c_i32),
    'srvp_conv_in_wgrad_f32': ([c_vp, c_vp, c_vp] + [c_i32] * 9 + [c_vp], c_i32),
    'srvp_out_dpre_f32': ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_out_dpre': ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_pack_weight': ([c_vp, c_vp, C.POINTER(PackDesc), c_vp], c_i32),
    'srvp_pack_weight_multi': ([c_vp, c_i32, c_i64, c_vp], c_i32),
    'srvp_unpack_wgrad_multi': ([c_vp, c_i32, c_i64, c_vp], c_i32),
    'srvp_pack_job_wgs': ([c_i64], c_i32),
    'srvp_cluster_timeouts_read': ([c_vp, c_vp], c_i32),
    'srvp_cluster_set_xcd_local': ([c_i32], c_i32),
    'srvp_cluster_stats_read': ([c_vp, c_vp], c_i32),
    'srvp_set_deterministic': ([c_i32, c_vp, c_i64], c_i32),
    'srvp_get_deterministic': ([], c_i32),
    'srvp_bn_stats_f32_det': ([c_vp, c_i64, c_i32, c_vp, c_vp], c_i32),
    'srvp_pack_job_tiles': ([C.POINTER(PackDesc), c_i32], c_i32),
    'srvp_conv_set_stream64': ([c_i32], c_i32),
    'srvp_conv_stream_count': ([c_i32], c_i64),
    'srvp_conv_set_in_stream': ([c_i32], c_i32),
    'srvp_conv_out_eligible': ([c_i32] * 7, c_i32),
    'srvp_conv_out_fwd': ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_conv_up_out_eligible': ([c_i32] * 7, c_i32),
    'srvp_conv_up_out_fwd': ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_pack_weight_tiles': ([c_vp, c_i32, c_i64, c_vp], c_i32),
    'srvp_unpack_wgrad_tiles': ([c_vp, c_i32, c_i64, c_vp], c_i32),
    'srvp_unpack_wgrad': ([c_vp, c_vp, C.POINTER(PackDesc), c_vp], c_i32),
    'srvp_gemm_f32': ([c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_linear_wgrad_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_add_blocks_f32': ([c_vp, c_i64, c_vp, c_i32, c_i64, c_vp], c_i32),
    'srvp_latent_to_z': ([c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_dz_split': ([c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp], c_i32),
    'srvp_rows_scatter_add_f32': ([c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_vp], c_i32),
    'srvp_axpby_f32': ([c_vp, c_f32, c_vp, c_f32, c_vp, c_i64, c_vp], c_i32),
    'srvp_colsum_f32': ([c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_act_bwd_f32': ([c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp], c_i32),
    'srvp_lstm_fwd': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_lstm_fused_ws_bytes': ([c_i32, c_i32, c_i32], c_i64),
    'srvp_lstm_fwd_fused': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp], c_i32),
    'srvp_lstm_bwd_fused': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp], c_i32),
    'srvp_lstm_bwd': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_rollout_fwd': ([C.POINTER(RolloutDesc), c_vp], c_i32),
    'srvp_rollout_fused_ws_bytes': ([C.POINTER(RolloutDesc)], c_i64),
    'srvp_rollout_gen_ws_bytes': ([C.POINTER(RolloutDesc)], c_i64),
    'srvp_rollout_bwd': ([C.POINTER(RolloutBwdDesc), c_vp], c_i32),
    'srvp_nll': ([c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp], c_i32),
    'srvp_kl': ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp], c_i32),
    'srvp_l2rows': ([c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp], c_i32),
    'srvp_rsample_fwd': ([c_vp, c_vp, c_vp, c_i64, c_i32, c_vp], c_i32),
    'srvp_rsample_bwd': ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp], c_i32),
    'srvp_adam': ([c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_vp], c_i32),
    'srvp_comm_unique_id': ([c_vp], c_i32),
    'srvp_comm_init': ([c_vp, c_i32, c_i32, C.POINTER(c_vp)], c_i32),
    'srvp_comm_destroy': ([c_vp], c_i32),
    'srvp_comm_info': ([c_vp, C.POINTER(c_i32)], c_i32),
    'srvp_allreduce_f64': ([c_vp, c_vp, c_i64, c_vp], c_i32),
    'srvp_allreduce_f32': ([c_vp, c_vp, c_i64, c_vp], c_i32),
    'srvp_allreduce': ([c_vp, c_vp, c_i64, c_i32, c_i32, c_vp], c_i32),
    'srvp_bcast_bytes': ([c_vp, c_vp, c_i64, c_i32, c_vp], c_i32),
    'srvp_peer_slab_create': ([c_i64, C.POINTER(c_vp), c_vp], c_i32),
    'srvp_peer_slab_open': ([c_vp, C.POINTER(c_vp)], c_i32),
    'srvp_peer_slab_close': ([c_vp, c_i32], c_i32),
    'srvp_peer_allreduce_f64': ([c_vp, c_i32, c_i32, c_i32, C.POINTER(c_vp), c_i64, c_i64, C.c_uint64, c_vp, c_vp], c_i32),
    'srvp_fill_f64': ([c_vp, c_i64, c_f64, c_vp], c_i32),
    'srvp_frames_u8_to_f32': ([c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    'srvp_mmnist_render': ([c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp], c_i32),
    'srvp_mmnist_trajectories': ([C.c_uint64, C.c_uint64] + [c_i32] * 9 + [c_vp, c_vp, c_vp, c_vp], c_i32),
    'srvp_cast_f32_bf16': ([c_vp, c_vp, c_i64, c_i32, c_i32, c_vp], c_i32),
    'srvp_cast_bf16_f32': ([c_vp, c_vp, c_i64, c_f32, c_vp], c_i32),
    'srvp_splitk_finish': ([c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp], c_i32),
    'srvp_pad_f32': ([c_vp, c_vp, c_i64, c_i32, c_i32, c_vp], c_i32),
    'srvp_skip_grad_reduce_f32': ([c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp], c_i32),
    'srvp_frame_metrics': ([c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp], c_i32),
    'srvp_skip_grad_reduce': ([c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp], c_i32),
}

_lib = None


class SrvpHipError(RuntimeError):
    pass


def load():
    """Loads libsrvp_hip.so (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrvpHipError(
            f'{LIB_PATH} not found: the HIP kernel library is the only compute path of srvp_amd. Build it with '
            '`make -C srvp_amd/csrc` (or `python -c "import __graft_entry__ as g; g.build()"`).')
    lib = C.CDLL(LIB_PATH)
    lib.srvp_last_error.restype = C.c_char_p
    lib.srvp_last_error.argtypes = []
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def exported_symbols():
    return ['srvp_last_error'] + list(_SIGS)


def check(rc, what=''):
    if rc != 0:
        msg = load().srvp_last_error().decode()
        raise SrvpHipError(f'{what} failed (code {rc}): {msg}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_get_dev = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """Raw handle of torch's CURRENT stream on the current device (what `with torch.cuda.stream(s)` sets).  Through torch's raw getter:
    torch.cuda.current_stream().cuda_stream builds a Stream object behind four Python-level device lookups, 9 us a call, ~120 calls per
    step (round 6, tools/host_profile.py)."""
    if _raw_stream is not None and _get_dev is not None:
        return _raw_stream(_get_dev())
    return torch.cuda.current_stream().cuda_stream


PROFILE = None      # set to a dict by bench.py: name -> list of (start_event, end_event) on the launch stream
PROFILE_ONLY = None  # optional set of entry-point names to time (None = all); keeps the event overhead out of a timed run


# ---- torch's CURRENT stream, tracked on the step path.  torch.cuda.current_stream() costs ~10 us of Python per call (four device-index
# helpers and a Stream object), and `with torch.cuda.stream(s)`, `event.record()`, `event.wait()` each call it: ~120 calls = 1 ms of the
# host work of a step (round 6, tools/host_profile.py).  The step therefore sets / restores torch's current stream through the helpers
# below, which remember the Stream OBJECT they made current: step_scope() looks the caller's stream up once, on_stream(s) replaces
# `with torch.cuda.stream(s)`, record() / wait(e) replace the argument-less Event.record() / current_stream().wait_event(e).  torch's own
# current stream is always the one these helpers believe it is (they call torch.cuda.set_stream), so torch ops inside keep working.
import threading

_tls = threading.local()


def _stack():
    st = getattr(_tls, 'stack', None)
    if st is None:
        st = _tls.stack = []
    return st


def cur_stream():
    """torch's current Stream object (tracked inside a step_scope / on_stream; looked up otherwise)."""
    st = _stack()
    return st[-1] if st else torch.cuda.current_stream()


class step_scope:
    """Looks the caller's current stream up ONCE for everything below (model._forward_impl / _backward_impl, optimizer step)."""
    __slots__ = ('pushed',)

    def __enter__(self):
        st = _stack()
        self.pushed = not st
        if self.pushed:
            st.append(torch.cuda.current_stream())
        return st[-1]

    def __exit__(self, *exc):
        if self.pushed:
            _stack().pop()
        return False


class on_stream:
    """`with torch.cuda.stream(s)` on the same device, without the look-ups."""
    __slots__ = ('s', 'prev')

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        st = _stack()
        self.prev = st[-1] if st else torch.cuda.current_stream()
        torch.cuda.set_stream(self.s)
        st.append(self.s)
        return self.s

    def __exit__(self, *exc):
        _stack().pop()
        torch.cuda.set_stream(self.prev)
        return False


def record(event=None):
    """A (new) event recorded on the current stream."""
    e = torch.cuda.Event() if event is None else event
    e.record(cur_stream())
    return e


def wait(event):
    """The current stream waits for `event`."""
    event.wait(cur_stream())


_FN = {}


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    if PROFILE is None or (PROFILE_ONLY is not None and name not in PROFILE_ONLY):
        rc = fn(*args)
        if rc != 0:
            check(rc, name)
        return
    # torch.cuda.Event records on torch's current stream, which is the stream every launch is issued on (stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), name)
    e1.record()
    PROFILE.setdefault(name, []).append((e0, e1))


def taps(vals):
    a = TAPS()
    for i, v in enumerate(vals):
        a[i] = int(v)
    return a
