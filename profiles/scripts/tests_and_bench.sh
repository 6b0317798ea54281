mkdir -p gpurun_out/r03g
timeout 600 python -m pytest tests/test_gpu_phase_fusion.py tests/test_gpu_wgrad_batched.py -x -q -m gpu > gpurun_out/r03g/pytest_new.txt 2>&1; grep -v "^NCCL\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r03g/pytest_new.txt | tail -25
L=profiles/scripts/conv_layer_time.py
for lay in deconv3 deconv_plain; do
  HESIC_IGEMM_PHASE4=0 python $L --layer $lay --size 128 --graph --dump gpurun_out/r03g/${lay}_0.pt 2>&1 | grep us
  HESIC_IGEMM_PHASE4=1 python $L --layer $lay --size 128 --graph --dump gpurun_out/r03g/${lay}_1.pt 2>&1 | grep us
  python profiles/scripts/cmp_dump.py gpurun_out/r03g/${lay}_0.pt gpurun_out/r03g/${lay}_1.pt
done
rm -f gpurun_out/r03g/*.pt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03g/pytest_gpu.txt 2>&1; grep "passed\|failed" gpurun_out/r03g/pytest_gpu.txt | tail -3
