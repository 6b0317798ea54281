mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1; grep "passed\|failed\|^FAILED" gpurun_out/final/pytest_gpu.txt | tail -8
python bench.py --mode train > gpurun_out/final/bench_train.json 2>/dev/null
HESIC_WGRAD_BIAS_COLSUM=1 python bench.py --mode train > gpurun_out/final/bench_train_colsum.json 2>/dev/null
python bench.py --mode train > gpurun_out/final/bench_train2.json 2>/dev/null
for f in bench_train bench_train_colsum bench_train2; do python -c "
import json
d=json.loads(open('gpurun_out/final/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['losses_last_step'])"; done
