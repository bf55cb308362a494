import sys, ctypes as C, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo')
from srvp_amd import _lib as L
from srvp_amd.convnet import Feat
dev=torch.device('cuda'); g=torch.Generator().manual_seed(3)
N,H,Cp=2,8,32
raw=torch.zeros(N,H,H,Cp,dtype=torch.bfloat16,device=dev); raw[:]=torch.randn(N,H,H,Cp,generator=g)
coef=torch.zeros(4,Cp,device=dev); coef[0]=1.0
out=Feat(N,H,H,Cp,dev); pool=Feat(N,H//2,H//2,Cp,dev)
st=L.stream()
L.call('srvp_bn_act', L.ptr(raw), L.ptr(coef[0]), L.ptr(coef[1]), L.ACT_LRELU, N,H,H,Cp, L.ptr(out.t),1, L.ptr(pool.t),1, None, st)
da=torch.zeros(N,H//2,H//2,Cp,dtype=torch.bfloat16,device=dev); da[:]=torch.randn(N,H//2,H//2,Cp,generator=g)
d=L.BnBwdDesc(); d.raw,d.act,d.act_border=L.ptr(raw),L.ptr(out.t),1
d.scale,d.shift,d.mean,d.invstd,d.act_kind=L.ptr(coef[0]),L.ptr(coef[1]),L.ptr(coef[2]),L.ptr(coef[3]),L.ACT_LRELU
d.N,d.H,d.W,d.C=N,H,H,Cp; d.da_border,d.da_is_f32,d.da2,d.da2_idx=0,0,None,None
d.da,d.da_mode,d.da_cstride,d.da_coff=L.ptr(da),2,Cp,0
bcoef=torch.zeros(3,Cp,device=dev)
L.call('srvp_bn_bwd_finalize', None,1.0,None,None,None,None,None,L.ptr(bcoef),Cp,Cp,0,st)
draw=torch.zeros(N,H+2,H+2,Cp,dtype=torch.bfloat16,device=dev)
L.call('srvp_bn_bwd_apply', C.byref(d), L.ptr(bcoef), L.ptr(draw),1,st)
red=torch.zeros(2,Cp,dtype=torch.float64,device=dev)
L.call('srvp_bn_bwd_reduce', C.byref(d), L.ptr(red), st)
torch.cuda.synchronize()
a=out.interior().permute(0,3,1,2).float().cpu()
_,idx=F.max_pool2d(a,2,2,return_indices=True)
dA=torch.zeros_like(a).flatten(2); dA.scatter_(2, idx.flatten(2), da.permute(0,3,1,2).float().cpu().flatten(2)); dA=dA.view_as(a)
rawc=raw.permute(0,3,1,2).float().cpu()
gref=dA*torch.where(rawc>0,torch.tensor(1.0),torch.tensor(0.2))
got=draw[:,1:-1,1:-1].permute(0,3,1,2).float().cpu()
print('nonzero got', (got!=0).float().mean().item(), 'ref', (gref!=0).float().mean().item())
print('max err', (got-gref).abs().max().item(), 'sum got', got.sum().item(), 'sum ref', gref.sum().item(), 'red0', red[0].sum().item())
print('window sample got\n', got[0,0,:4,:4], '\nref\n', gref[0,0,:4,:4], '\nact\n', a[0,0,:4,:4])
