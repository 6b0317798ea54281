cd $GRAFT_REPO_ROOT
for v in 0 3 4; do
  HESIC_IGEMM_BM256=$v python profiles/scripts/conv_layer_time.py --layer plain --graph --dump /tmp/d$v.pt 2>&1 | tail -1
  HESIC_IGEMM_BM256=$v python profiles/scripts/conv_layer_time.py --layer plain --stride 1 --size 128 --graph 2>&1 | tail -1
done
python -c "
import torch
a=torch.load('/tmp/d0.pt'); b=torch.load('/tmp/d3.pt'); c=torch.load('/tmp/d4.pt')
print('equal', torch.equal(a,b), torch.equal(a,c), float((a-b).abs().max()))"
