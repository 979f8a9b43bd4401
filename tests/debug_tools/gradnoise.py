import sys, copy; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, sifsr
from oracle import sif_oracle as O
from tests.conftest import rel_err
MEAN, STD = 307.2378, 5.5698
for kind,alpha,gamma,ws,bs in (("sr2",0.5,-0.25,31,41),("sr1",0.99,-0.5,32,42)):
    sd = O.synthetic_state(ws); lst,lst_up,ndvi = O.synthetic_batch(bs,2)
    _,_,g_o = O.forward_backward(copy.deepcopy(sd), lst,lst_up,ndvi,MEAN,STD,alpha,gamma,kind)
    sd64 = {k:(v.double() if v.dtype==torch.float32 else v.clone()) for k,v in sd.items()}
    _,_,g64 = O.forward_backward(sd64, lst.double(),lst_up.double(),ndvi.double(),MEAN,STD,alpha,gamma,kind)
    m = sifsr.ModelB_2(2); m.load_state_dict(sd); m=m.cuda().train()
    sr = m(torch.cat((lst_up,ndvi),1).cuda())
    _,_,loss = sifsr.sif_loss(kind, sr, lst.cuda(), ndvi.cuda(), MEAN,STD,alpha,gamma); loss.backward()
    eh=[];ec=[]
    for n,p in m.named_parameters():
        a=rel_err(p.grad,g64[n]); b=rel_err(g_o[n],g64[n]); eh.append(a); ec.append(b)
        print(f"{kind} {n:45s} hip {a:.2e} cpu {b:.2e} ratio {a/max(b,1e-12):.2f}")
    eh=torch.tensor(eh); ec=torch.tensor(ec)
    print(kind, "max hip %.2e cpu %.2e | rms hip %.2e cpu %.2e | median hip %.2e cpu %.2e"%(eh.max(),ec.max(),eh.pow(2).mean().sqrt(),ec.pow(2).mean().sqrt(),eh.median(),ec.median()))
