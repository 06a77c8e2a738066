import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from yolo_v1_amd import ops
DEV="cuda:0"
def run(N,H,cin,cout=128):
    gen=torch.Generator(device=DEV).manual_seed(1)
    x=ops.Act((torch.randn(N,H,H,cin,generator=gen,device=DEV)*1.3+0.4).to(torch.bfloat16))
    bn=torch.nn.BatchNorm2d(cin).to(DEV)
    st=ops.bn_finalize(ops.bn_stats(x), x.npix, bn)
    param=torch.nn.Parameter((torch.randn(cout,cin,1,1,generator=gen,device=DEV)*0.1).contiguous(memory_format=torch.channels_last))
    w=ops.ConvWeights(param,1,1,0); w.refresh()
    dy=ops.Act((torch.randn(N,H,H,cout,generator=gen,device=DEV)*0.05).to(torch.bfloat16))
    G=ops.Act(torch.zeros(N,H,H,cin,dtype=torch.bfloat16,device=DEV))
    K=torch.zeros((2,cin),dtype=torch.float32,device=DEV)
    out=[]
    for pend in (None, K):
        for _ in range(3): ops.conv_dgrad_bn_deferred(dy,w,G,x,st,True,pending=pend)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv_dgrad_bn_deferred(dy,w,G,x,st,True,pending=pend)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1)/20*1e3)
    print("N=%d H=%d cin=%d: no pending %.1f us, pending %.1f us"%(N,H,cin,out[0],out[1]))
for s in [(64,112,160),(64,112,256),(64,56,320),(64,56,512),(64,28,640),(64,28,1024),(64,14,800)]:
    run(*s)
