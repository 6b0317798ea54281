cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -k "c32 or independent_en or stage2 or resblock or img6" 2>&1 | tail -5 > gpurun_out/r6_en_tests.log
timeout 600 python - > gpurun_out/r6_en.log 2>&1 <<'P'
import torch, json, sys
sys.argv=['bench.py']
import bench, hesic_amd
hesic_amd.set_compute_dtype(torch.float16)
d=bench.secondary_hesic_en(torch.device('cuda:0'))
print(json.dumps({k:d[k] for k in ('value','ms_per_step','independent_en_alone','roofline','conv3x3_c32_avg_us') if k in d} if 'error' not in d else d, indent=1))
P
