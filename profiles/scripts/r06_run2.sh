cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python profiles/scripts/aten_ops_in_forward.py > gpurun_out/r6_aten_ops.log 2>&1
timeout 600 python - > gpurun_out/r6_en.log 2>&1 <<'P'
import torch, json, sys
sys.argv=['bench.py']
import bench, hesic_amd
hesic_amd.set_compute_dtype(torch.float16)
print(json.dumps(bench.secondary_hesic_en(torch.device('cuda:0')), indent=1))
P
