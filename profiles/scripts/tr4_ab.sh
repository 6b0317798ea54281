set -x
L=profiles/scripts/conv_layer_time.py
for cfg in "64 8" "128 4" "128 2" "64 16" "96 8"; do
  set -- $cfg
  for lay in deconv3 deconv_plain; do
    HESIC_IGEMM_PHASE4=0 python $L --layer $lay --size $1 --batch $2 --graph 2>&1 | grep us
    HESIC_IGEMM_PHASE4_MIN=1 python $L --layer $lay --size $1 --batch $2 --graph 2>&1 | grep us
  done
done
