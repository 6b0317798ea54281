set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_path_a.py -x -q 2>&1 | tail -40 > gpurun_out/r6_patha_tests.log
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "rd_sums or reductions" 2>&1 | tail -5 >> gpurun_out/r6_patha_tests.log
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -15 > gpurun_out/r6_multirank.log
timeout 600 python - > gpurun_out/r6_patha_time.log 2>&1 <<'P'
import torch, time, hesic_amd
from hesic_amd import models, path_a, synthetic, functional as Fn, handover
hesic_amd.set_compute_dtype(torch.float16)
net = models.HSIC(); synthetic.fill_state_dict_(net.state_dict()); net = net.to("cuda").eval()
a,b,h = (t.to("cuda") for t in synthetic.stereo_batch(0,8,512,512))
def loop(fn,n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
with torch.no_grad():
    print("path A handover ms", loop(lambda: path_a.hsic_forward(net,a,b,h)))
    models.OVERLAP_STREAMS=False
    print("twin single stream ms", loop(lambda: net(a,b,h)))
    models.OVERLAP_STREAMS=True
    print("twin overlap ms", loop(lambda: net(a,b,h)))
    handover.ENABLED=False
    print("path A literal ms", loop(lambda: path_a.hsic_forward(net,a,b,h)))
    handover.ENABLED=True
    # host time of issue only
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): path_a.hsic_forward(net,a,b,h)
    print("path A host issue ms (no sync)", (time.perf_counter()-t)/10*1e3)
    torch.cuda.synchronize()
P
