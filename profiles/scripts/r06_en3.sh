cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python - > gpurun_out/r6_en3.log 2>&1 <<'P'
import torch
import hesic_amd
from hesic_amd import functional as Fn, _lib as L
hesic_amd.set_compute_dtype(torch.float16)
torch.manual_seed(0)
w1,w2=(torch.randn(32,32,3,3,device='cuda')*0.06 for _ in range(2)); b1,b2=(torch.randn(32,device='cuda')*0.1 for _ in range(2))
for (B,H,W) in ((8,512,512),(1,512,512),(2,70,100),(1,64,64)):
    x=(torch.randn(B,32,H,W,device='cuda')*0.5).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y=Fn.resblock_c32(x,w1,b1,w2,b2,act=L.ACT_LEAKY)
        t=Fn.conv3x3_c32(x,w1,b1,act=L.ACT_LEAKY)
        r=Fn.conv3x3_c32(t,w2,b2,act=L.ACT_LEAKY,res1=x)
    d=(y.float()-r.float()).abs().amax(1)   # B,H,W
    bad=(d>0.01)
    tiles=((W+29)//30)*((H+13)//14)*B
    print((B,H,W),'tiles',tiles,'bad frac',float(bad.float().mean()))
    if bad.any():
        idx=bad.nonzero()
        print(' first bad', idx[:5].tolist(), ' last bad', idx[-3:].tolist())
        # per tile stats
        ty=idx[:,1]//14; tx=idx[:,2]//30; bb=idx[:,0]
        tid=(bb*((H+13)//14)+ty)*((W+29)//30)+tx
        u=torch.unique(tid)
        print(' bad tiles', u.numel(), 'min', int(u.min()), 'max', int(u.max()), 'first few', u[:12].tolist())
        rows=torch.unique(idx[:,1]%14); cols=torch.unique(idx[:,2]%30)
        print(' rows in tile', rows.tolist(), 'cols in tile', cols.tolist()[:40])
P
timeout 600 python - >> gpurun_out/r6_en3.log 2>&1 <<'P'
import torch
import hesic_amd
from hesic_amd import functional as Fn, _lib as L
hesic_amd.set_compute_dtype(torch.float16)
w1,w2=(torch.randn(32,32,3,3,device='cuda')*0.06 for _ in range(2)); b1,b2=(torch.randn(32,device='cuda')*0.1 for _ in range(2))
x=(torch.randn(8,32,512,512,device='cuda')*0.5).half().contiguous(memory_format=torch.channels_last)
sk=(torch.randn(8,32,512,512,device='cuda')*0.5).half().contiguous(memory_format=torch.channels_last)
def tm(fn,n=20):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
with torch.no_grad():
    print('resblock us', tm(lambda: Fn.resblock_c32(x,w1,b1,w2,b2,act=L.ACT_LEAKY)), 'with res2', tm(lambda: Fn.resblock_c32(x,w1,b1,w2,b2,act=L.ACT_LEAKY,res2=sk)))
    print('single conv us', tm(lambda: Fn.conv3x3_c32(x,w1,b1,act=L.ACT_LEAKY)))
P
