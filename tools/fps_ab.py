import torch,sys,os
sys.path.insert(0,".")
from pointrcnn_amd import ops, rpn
dev=torch.device("cuda:0")
xyz=rpn.synthetic_clouds(32,16384,seed0=100,device=dev)
out=[os.path.basename(os.environ.get("PRCNN_POINTOPS_LIB","product"))]
for n,m in ((16384,4096),(4096,1024),(1024,256),(256,64)):
    x=xyz[:,:n].contiguous()
    for _ in range(3): ops.furthest_point_sample(x,m)
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(9):
        s.record(); ops.furthest_point_sample(x,m); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)*1e3)
    out.append("%d->%d %.1f us"%(n,m,sorted(ts)[4]))
print(" | ".join(out))
