"""VERDICT r4 item 3c, measured on the numerics model (CPU, no GPU needed): would keeping the END of the encoder in higher precision bring the
production-precision ELBO of config 3 / 5 at their INITIAL weights (3.4e-3 / 2.4e-4 off the fp32 reference at 400 frames) under 1e-4?

oracle/srvp_oracle.py's PRECISION = 'bf16' places the product's rounding points into the reference algorithm (bf16 MFMA operands, bf16 storage of
raw conv outputs and activations, fp32 statistics / latent path); the 400-frame gates hold the HIP path to this model (bf16_vs_model <= 1.5e-3).
Three variants of the model on the gate's own batch / tape / seeds (tests/test_gpu_parity_gate.py::test_elbo_gate_undiluted_recipes_400_frames):
    bf16        every conv layer as the product stores it
    raw32       encoder stage 4 (the 8x8 layers) + last_conv keep their RAW outputs unrounded (fp32 `raw`): one of their two roundings goes
    enc4_fp32   encoder stage 4 + last_conv entirely in fp32 arithmetic (operands, raw, activations): the UPPER BOUND of any mixed-storage scheme
                there (the product would pay the 16x slower fp32-exact MFMA path on 13 % of the encoder+decoder FLOPs for it)
    enc_fp32    the WHOLE encoder in fp32 arithmetic (decoder bf16): where the error would have to come from for the idea to work at all
One JSON line per (recipe, variant): relative ELBO / NLL error against the fp32 oracle.     usage: python tools/mixed_storage_model.py [kth|human]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from make_golden import synth_video
from oracle import srvp_oracle as O

RECIPES = {'kth': dict(nc=1, T=20, B=20), 'human': dict(nc=3, T=16, B=26)}


def run(name):
    import srvp_amd
    r = RECIPES[name]
    nc, T, B, ne = r['nc'], r['T'], r['B'], 2
    ctor = (64, nc, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg')
    hp = dict(obs_scale=0.2, beta_y=1.0, beta_z=1.0, l2_res=1.0)
    torch.manual_seed(1)
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)      # (host-side constructor only: same-seed initial weights as the reference)
    model.init(1.2)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(321)
    x = torch.from_numpy(synth_video(T, B, nc, seed=77))
    tape = dict(t_w=torch.stack([torch.randperm(T, generator=g)[:3] for _ in range(B)], 1), eps_y0=torch.randn(B, 50, generator=g),
                eps_z=torch.randn(T - 1, B, 50, generator=g), t_skip=torch.randint(T, (B,), generator=g))
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    cfg = O.make_cfg(*ctor)
    enc_keys = [s['key'] for s in O.encoder_spec(cfg['archi'], cfg['nc'], cfg['nhx'], cfg['nf'])]
    tail = [k for k in enc_keys if k.startswith('encoder.conv.3.') or k.startswith('encoder.last_conv')]
    orig = O._conv_block_bf16

    def elbo(prec, fp32_keys=(), raw32_keys=()):
        def patched(h, sd_, spec, training):
            if spec['key'] in fp32_keys:
                O.PRECISION = 'fp32'
                try:
                    return O.conv_block(h, sd_, spec, training)
                finally:
                    O.PRECISION = 'bf16'
            if spec['key'] in raw32_keys:
                real = O._bf
                calls = [0]

                def bf_skip_raw(t):
                    # _conv_block_bf16 rounds: operands (2 calls), then the raw output (3rd call), then the activation (4th)
                    calls[0] += 1
                    return t if calls[0] == 3 else real(t)
                O._bf = bf_skip_raw
                try:
                    return orig(h, sd_, spec, training)
                finally:
                    O._bf = real
            return orig(h, sd_, spec, training)
        O.PRECISION, O._conv_block_bf16 = prec, patched
        try:
            with torch.no_grad():
                outs = O.forward({k: v.clone() for k, v in sd.items()}, cfg, x, T, ne, tape, training=True)
                res = O.elbo(x, outs, hp['obs_scale'], hp['beta_y'], hp['beta_z'], hp['l2_res'])
        finally:
            O.PRECISION, O._conv_block_bf16 = 'fp32', orig
        return float(res['loss']), float(res['nll']) / B
    ref = elbo('fp32')
    rel = lambda a, b: abs(a - b) / abs(b)
    for variant, kw in (('bf16', {}), ('raw32', dict(raw32_keys=tail)), ('enc4_fp32', dict(fp32_keys=tail)), ('enc_fp32', dict(fp32_keys=enc_keys))):
        got = elbo('bf16', **kw)
        print(json.dumps(dict(recipe=name, frames=T * B, variant=variant, layers=len(kw.get('fp32_keys', kw.get('raw32_keys', ()))),
                              loss_fp32=ref[0], e_loss=rel(got[0], ref[0]), e_nll=rel(got[1], ref[1]))), flush=True)


if __name__ == '__main__':
    for n in (sys.argv[1:] or ['kth', 'human']):
        run(n)
