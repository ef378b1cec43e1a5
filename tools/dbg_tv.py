import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from conftest import load_golden
from oracle import artspeech_oracle as O
from artspeech_amd.tract_variables import tract_variables_batched
g = load_golden("tract_variables"); arts=[str(a) for a in g["articulators"]]
v,p1,p2,idx = tract_variables_batched(torch.from_numpy(g["frames"]).cuda(), arts)
v=v.cpu().numpy(); idx=idx.cpu().numpy()
bad=0
for f in range(g["frames"].shape[0]):
    ov,_,_,oi = O.tract_variables(g["frames"][f], arts, dtype=np.float32)
    ov=ov.astype(np.float32)
    if not np.array_equal(v[f],ov) or not np.array_equal(idx[f],oi):
        bad+=1
        if bad<4: print(f, v[f].view(np.uint32)-ov.view(np.uint32), idx[f].tolist(), oi.tolist())
print("bad frames", bad)
