import sys, math; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
import test_gpu_ops as t
from ppmstereo_amd import _lib as L
from ppmstereo_amd.weights import hash_normal
lib = L
T,H,W,segs,cout,k3 = 1,9,24,[32],128,(1,1,15)
P=T*H*W
xs=[hash_normal((P,c),100+i) for i,c in enumerate(segs)]
cin=sum(segs)
wt=hash_normal((cout,cin,*k3),200)/math.sqrt(cin*15)
bs=hash_normal((cout,),201)*0.1
ref=t._ref_conv(xs,wt,bs,k3,T,H,W)
def split(v):
    hi=v.to(torch.bfloat16).float(); lo=(v-hi).to(torch.bfloat16).float(); return hi,lo
xh,xl=split(xs[0]); wh,wl=split(wt.reshape(cout,cin,15))
xim=lambda v: v.reshape(H,W,cin)
found=False
for trial in range(40):
    v = 5008 if trial%2==0 else 5007
    got=t._run_conv(lib,xs,wt,bs,k3,T,H,W,version=v,seg_pad=[32])
    err=(got-ref)
    bad=err.abs()>1e-4
    if not bad.any(): continue
    print('trial',trial,'v',v,'frac bad',bad.float().mean().item())
    idx=bad.float().sum(1).nonzero().flatten().tolist()
    print(' bad pixels', [divmod(p,W) for p in idx])
    for pi in idx[:3]:
        y,x=divmod(pi,W)
        bc=bad[pi].nonzero().flatten().tolist()
        print(' pixel',(y,x),'nbad',len(bc),'couts',bc[:40])
        c=bc[0]
        # candidate terms per tap / chunk / type
        best=[]
        for tap in range(15):
            xx=x+tap-7
            if not (0<=xx<W): continue
            for ch in range(2):
                ks=slice(ch*16,ch*16+16)
                for nm,(a,b) in {'hh':(wh,xh),'lh':(wl,xh),'hl':(wh,xl)}.items():
                    term=(a[c,ks,tap]*xim(b)[y,xx,ks]).sum().item()
                    best.append((abs(err[pi,c].item()+term),tap,ch,nm,term))
        best.sort()
        print('  err',err[pi,c].item(),'closest missing-term candidates',best[:3])
    found=True
    break
print('found',found)
