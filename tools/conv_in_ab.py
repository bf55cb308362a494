"""where does the bf16 ELBO error of the full-width KTH fixture (tests/golden/full_c3_kth_vgg.npz: 40 frames, one
channel) come from now that the image-side layer runs on csrc/conv_in_stream.hip?  The fixture's training forward is run in production
precision with the streaming kernel ON and OFF; reported separately for the first block: its raw (pre-BatchNorm) bf16 output against the
float64 convolution of the same operands, its BatchNorm statistics against float64 sums of the exact outputs, and the resulting ELBO
against the reference-made fixture value.  One JSON line per setting.    usage: python tools/conv_in_ab.py [fixture]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import srvp_amd
from srvp_amd import _lib as L
from srvp_amd.train import elbo_terms_and_grads
from test_full_width_golden import Full


def main(name):
    fx = Full(name)
    hp, ne = fx.meta['hp'], fx.meta['n_euler']
    opt = srvp_amd.DotDict(dict(n_euler_steps=ne, **hp))
    ref = float(fx.z['train.scalars'][0])
    rows = []
    keep = {}
    for on in (1, 0):
        L.load()
        L.call('srvp_conv_set_in_stream', on)
        try:
            m, x = fx.model_and_input()
            m = m.cuda().train().set_precision('bf16')
            xg = x.cuda()
            T, B = x.shape[:2]
            with torch.no_grad():
                outs = m._forward_impl(xg, T, ne, fx.tape(), training=True)
                acc, _ = elbo_terms_and_grads(m, xg, outs, opt, want_grads=False)
            nll, kl_y0, kl_z, l2 = acc.cpu().tolist()
            loss = (nll + hp['beta_y'] * kl_y0 + hp['beta_z'] * kl_z + hp['l2_res'] * l2) / B
            blk = m._last_plan['enc'].blocks[0]
            raw = blk.raw.float().cpu().view(T * B, 64, 64, -1)[..., :blk.cout]
            stats = blk.stats.double().cpu().view(2, -1)[:, :blk.cout]
            w = dict(m.named_parameters())[blk.spec['key'] + '.weight'].detach().double().cpu()
            exact = F.conv2d(x.view(T * B, *x.shape[2:]).double(), w, None, 1, 1).permute(0, 2, 3, 1)        # float64, NHWC
            want = exact.to(torch.bfloat16).double()
            off = (raw.double() != want).double().mean().item()
            s_ref = torch.stack([exact.sum((0, 1, 2)), (exact ** 2).sum((0, 1, 2))])
            n = exact.numel() / exact.shape[-1]
            mean, var = s_ref[0] / n, s_ref[1] / n - (s_ref[0] / n) ** 2
            mean_k, var_k = stats[0] / n, stats[1] / n - (stats[0] / n) ** 2
            row = dict(fixture=name, conv_in_stream=on, frames=T * B, loss=loss, loss_ref=ref, e_loss=abs(loss - ref) / abs(ref),
                       raw_frac_off_bf16_of_exact=off, raw_max_abs_err=(raw.double() - exact).abs().max().item(),
                       stats_rel_err=((stats - s_ref).abs().max(1).values / s_ref.abs().max(1).values).tolist(),
                       mean_abs_err_over_std=((mean_k - mean).abs() / var.sqrt()).max().item(),
                       var_rel_err=((var_k - var).abs() / var).max().item(), mean_over_std_max=(mean.abs() / var.sqrt()).max().item())
            keep[on] = (raw, stats)
            rows.append(row)
            print(json.dumps(row), flush=True)
        finally:
            L.call('srvp_conv_set_in_stream', 1)
    d = (keep[1][0] != keep[0][0]).float().mean().item()
    print(json.dumps(dict(fixture=name, raw_frac_differing_between_kernels=d,
                          stats_rel_diff_between_kernels=((keep[1][1] - keep[0][1]).abs().max() / keep[0][1].abs().max()).item())), flush=True)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'full_c3_kth_vgg')
