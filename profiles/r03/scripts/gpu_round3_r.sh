set -x
O=gpurun_out/r3r; mkdir -p $O
python - > $O/check.txt 2>&1 <<'PY'
import numpy as np, torch, sys
sys.path.insert(0,'.')
from lwsnet_amd import ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.weights import default_args, make_state_dict
from oracle import c_oracle as C
dev=torch.device('cuda:0'); sd=make_state_dict(7)
m=LWSNet(default_args(),device=dev).set_state_dict(sd).eval()
for stage,shape in [(1,(1,9,8,16)),(1,(2,9,30,70)),(2,(1,9,64,128)),(2,(1,5,7,33))]:
    c=(np.random.default_rng(stage+1).random(shape)*12).astype(np.float32)
    want=C.conv3d_stack(c,sd,stage)
    m.set_option('mid8_form',2)
    got=ops.conv3d_stack(m._h,stage,torch.from_numpy(c).to(dev)).cpu().numpy()
    print(stage,shape,'bit-exact' if np.array_equal(got,want) else 'DIFFERS')
PY
cat $O/check.txt
python tools/sbench.py --batch 1 > $O/sbench.txt 2>&1
python tools/sbench.py --batch 2 >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 >> $O/sbench.txt 2>&1
grep stage $O/sbench.txt
