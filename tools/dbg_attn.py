import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from ppmstereo_amd import _lib as L
from ppmstereo_amd.weights import hash_normal
dev='cuda:0'
T,n,ks=1,int(sys.argv[1]) if len(sys.argv)>1 else 256,1
torch.set_printoptions(linewidth=220, precision=3, sci_mode=False)
qb=(hash_normal((T,n,128),1)*0.5).to(torch.bfloat16).to(dev)
kb=(hash_normal((T,ks,n,128),2)*0.5).to(torch.bfloat16).to(dev)
sel=torch.zeros(T,5,dtype=torch.int32,device=dev)
lib=L.load()
scale=128**-0.5
for mode in ("subtile","pos128"):
    V=torch.zeros(T,n,128)
    key=torch.arange(n)
    if mode=="subtile": V[0,key,(key//32)%128]=1.0
    else: V[0,key,key%128]=1.0
    vt=V.transpose(1,2).contiguous().to(torch.bfloat16).to(dev)
    X=L.SPTensor(T*n,256,dev)
    beta=torch.tensor([1.0],device=dev)
    raw=torch.zeros(T,n,128,dtype=torch.bfloat16,device=dev)
    ws=torch.empty(int(lib.ppms_mem_attn_workspace_bytes(T,ks,n)),dtype=torch.uint8,device=dev)
    L.check(lib.ppms_mem_attn(qb.data_ptr(),kb.data_ptr(),vt.data_ptr(),sel.data_ptr(),ks,scale,beta.data_ptr(),X.view(0,128),X.view(128,128),raw.data_ptr(),T,n,L.ptr(ws),L.stream_ptr()))
    torch.cuda.synchronize()
    S=(qb[0].float()@kb[0,0].float().T)*scale
    P=torch.softmax(S,dim=1)
    ref=(P@V[0].to(dev)).cpu()
    got=raw[0].float().cpu()
    nz = 16 if mode=="subtile" else 128
    err=(got-ref)[:, :nz]
    print(mode,'max err',err.abs().max().item())
    bad=(err.abs()>2e-3)
    print(' bad count', bad.sum().item(), 'bad channels', bad.any(0).nonzero().flatten().tolist(), 'bad queries', bad.any(1).nonzero().flatten().tolist()[:40])
    if bad.any():
        qi=bad.any(1).nonzero().flatten()[0].item()
        print(' query',qi,'got',got[qi,:nz][bad[qi]], 'ref', ref[qi,:nz][bad[qi]])
