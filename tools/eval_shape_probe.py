"""Diagnostic: eval-mode forward vs the CPU oracle over (nc, nt, B) -- looks for shape-dependent decoder errors."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srvp_amd
from oracle import srvp_oracle as O

torch.set_num_threads(8)
for nc, ntc, nt, B in [(1, 3, 33, 2), (1, 3, 53, 1)]:
    torch.manual_seed(1)
    ctor = (64, nc, 64, 128, 50, 50, True, 3, 256, 3, 512, 4, 'vgg')
    model = srvp_amd.StochasticLatentResidualVideoPredictor(*ctor)
    model.init(1.2)
    model.cuda().train()
    g = torch.Generator().manual_seed(321)
    xw = torch.rand(ntc, 6, nc, 64, 64, generator=g)
    with torch.no_grad():
        for _ in range(12):
            model(xw.cuda(), ntc, 0.5)
    model.eval()
    x = xw[:, :B].contiguous()
    tape = dict(eps_y0=torch.randn(B, 50, generator=g), eps_z=torch.randn(nt - 1, B, 50, generator=g))
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = O.forward(sd, O.make_cfg(*ctor), x, nt, 2, tape, training=False)
        out = model(x.cuda(), nt, 0.5, tape=tape)
    d = (out[0].cpu() - ref[0]).abs()
    pf = d.amax(dim=(2, 3, 4)).reshape(-1)          # per frame n = t*B + b
    bad = [i for i, v in enumerate(pf.tolist()) if v > 0.03]
    O.PRECISION = 'bf16'
    with torch.no_grad():
        refm = O.forward(sd, O.make_cfg(*ctor), x, nt, 2, tape, training=False)
    O.PRECISION = 'fp32'
    dm = (out[0].cpu() - refm[0]).abs().amax(dim=(2, 3, 4)).reshape(-1)
    print('  |y_t| max per t:', [round(v, 1) for v in ref[1].abs().amax(dim=(1, 2)).tolist()])
    print('  err vs fp32 oracle per frame:', [round(v, 3) for v in pf.tolist()])
    print('  err vs bf16-model oracle per frame:', [round(v, 3) for v in dm.tolist()])
    print('  fp32 oracle vs bf16-model oracle:', [round(v, 3) for v in (ref[0] - refm[0]).abs().amax(dim=(2, 3, 4)).reshape(-1).tolist()])
    print(f'nc={nc} nt={nt} B={B} N={nt*B}: x max err {d.max().item():.4f}; bad frames n:', bad[:40], flush=True)
