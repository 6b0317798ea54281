mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/ab/pytest.txt 2>&1; grep "passed\|failed\|^FAILED" gpurun_out/ab/pytest.txt | tail -5
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export HESIC_GROUPED_NOSPLIT=1; else unset HESIC_GROUPED_NOSPLIT; fi
  python bench.py --no-cpu-baseline > gpurun_out/ab/b_$v.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/ab/b_$v.json').read().strip().splitlines()[-1]); print('NOSPLIT=$v', d['value'], d['ms_per_step'], d['parity']['met'], d['parity'].get('abs_dbpp'))"
done
