import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from oracle import ppm_oracle as O
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.synth import T40_CASES, synth_scale_inputs
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.corr import CorrBlock1D
W=Wm.hot_path_weights(); DEV='cuda:0'
model=PPMStereoHotPath().load_hot_path_weights(W).to(DEV).eval()
g=lambda x: None if x is None else x.to(DEV)
for name,tag,ai,T,h,w,iters,isc,mh in (("fub04_T40","update_block04",2,40,8,32,3,1,True),("fub16_T40","update_block16",0,40,8,32,2,4,False)):
    d=synth_scale_inputs(T,h,w,with_mhs=mh,**T40_CASES[name])
    preds,uncs,rp,ru,trace=[],[],[],[],[]
    fo,net,mhs=model.forward_update_block(None,getattr(model,tag),CorrBlock1D(g(d["fmap1"]),g(d["fmap2"])),g(d["flow"]),g(d["net"]),g(d["inp"]),g(d["mhs"]),model.att[ai],preds,uncs,iters,isc,T)
    rfo,rnet,rmhs=O.forward_update_block(W[tag],W[f"att.{ai}"],O.corr_pyramid(d["fmap1"],d["fmap2"]),d["flow"],d["net"],d["inp"],d["mhs"],iters,isc,T,tag=="update_block16",rp,ru,trace)
    eng=getattr(model,tag).engine(T,h,w,DEV)
    sel=eng.SEL.cpu()
    ref_sel=trace[-1]['sel'][0]
    flips=[i for i in range(T) if sel[i].tolist()!=torch.nonzero(ref_sel[i]).flatten().tolist()]
    err=(fo.cpu()-rfo).abs()
    pf=err.amax(dim=(1,2,3))
    print(name,'pick flips in last iteration:',flips,'max err %.2e mean %.2e'%(err.max(),err.mean()),'|fo|max %.1f'%rfo.abs().max())
    print('  per-frame max err:',['%.1e'%x for x in pf.tolist()])
    print('  net err %.2e mhs err %.2e |net| %.2f'%((net.cpu()-rnet).abs().max(),(mhs.cpu()-rmhs).abs().max(),rnet.abs().max()))
