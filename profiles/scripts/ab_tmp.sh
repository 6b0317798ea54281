mkdir -p gpurun_out/ab
for v in "0 384" "500 384" "500 500" "500 450" "500 384" "500 500"; do
  set -- $v
  HESIC_WGRAD_BLOCKS_BIG=$1 HESIC_WGRAD_BLOCKS=$2 timeout 300 python bench.py --mode train > gpurun_out/ab/t.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/ab/t.json').read().strip().splitlines()[-1]); print('BIG=$1 BLOCKS=$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
