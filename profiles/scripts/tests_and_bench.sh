mkdir -p gpurun_out/r03g
timeout 600 python -m pytest tests/test_gpu_phase_fusion.py tests/test_gpu_wgrad_batched.py -x -q -m gpu > gpurun_out/r03g/pytest_new.txt 2>&1; tail -15 gpurun_out/r03g/pytest_new.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03g/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r03g/pytest_gpu.txt
python bench.py --no-cpu-baseline > gpurun_out/r03g/bench_quick.json 2> gpurun_out/r03g/bench.err
python bench.py --mode train > gpurun_out/r03g/bench_train.json 2>/dev/null
HESIC_WGRAD_FINISH_BATCH=0 python bench.py --mode train > gpurun_out/r03g/bench_train_nobatch.json 2>/dev/null
for f in bench_quick bench_train bench_train_nobatch; do python -c "
import json
d=json.loads(open('gpurun_out/r03g/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], list(d['roofline'].get('other_conv_kernels',{}).keys())[:14])"; done
