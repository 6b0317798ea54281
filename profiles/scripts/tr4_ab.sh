set -x
mkdir -p gpurun_out/tr4
L=profiles/scripts/conv_layer_time.py
for lay in deconv3 deconv_plain; do
  HESIC_IGEMM_PHASE4=0 python $L --layer $lay --size 128 --graph --dump gpurun_out/tr4/${lay}_0.pt
  HESIC_IGEMM_PHASE4=1 python $L --layer $lay --size 128 --graph --dump gpurun_out/tr4/${lay}_1.pt
  python profiles/scripts/cmp_dump.py gpurun_out/tr4/${lay}_0.pt gpurun_out/tr4/${lay}_1.pt
  HESIC_IGEMM_PHASE4=0 python $L --layer $lay --size 128 --graph
  HESIC_IGEMM_PHASE4=1 python $L --layer $lay --size 128 --graph
done
rm -f gpurun_out/tr4/*.pt
HESIC_IGEMM_PHASE4_MIN=1 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5
