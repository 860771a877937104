import torch,sys
sys.path.insert(0,".")
from semivl_amd import ops
dev=torch.device("cuda:0")
for N in (21,81,150):
    B,S=16,512
    lg=torch.randn(B,N,S,S,device=dev); tg=torch.randint(0,N,(B,S,S),device=dev); cf=torch.rand(B,S,S,device=dev)
    ig=torch.zeros(B,S,S,dtype=torch.int64,device=dev); mc=torch.randint(0,N,(B,S,S),device=dev); dl=torch.empty_like(lg)
    gs=torch.ones(2,device=dev); sums=torch.zeros(4,dtype=torch.float64,device=dev)
    for _ in range(3): ops.ce_fused(lg,tg,False,conf=cf,ign=ig,conf_thresh=0.5,mc=mc,dlogits=dl,gscale=gs,sums_out=sums)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.ce_fused(lg,tg,False,conf=cf,ign=ig,conf_thresh=0.5,mc=mc,dlogits=dl,gscale=gs,sums_out=sums)
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    print("ce N=%d %.4f ms  real %.2f TB/s  algo %.2f TB/s"%(N,ms,B*S*S*(8*N+28)/ms/1e9,B*S*S*(12*N+40)/ms/1e9))
    del lg,dl
