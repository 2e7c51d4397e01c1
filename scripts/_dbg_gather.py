import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from cusrl_amd import ops
DEV="cuda:0"
def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)
T,N=2,8
leaves=[torch.randn(T,N,k%5+1,device=DEV) for k in range(30)]
idx=torch.randperm(T*N,device=DEV)[:9]
outs=ops.gather_rows(leaves,idx,T,N); sync("30 leaves")
rng=np.random.default_rng(1)
T,N,B=24,64,384
narrow={f"f{i}":torch.randn(T,N,1,device=DEV) for i in range(6)}
narrow.update({f"b{i}":torch.rand(T,N,1,device=DEV)<0.4 for i in range(3)})
wide=[torch.randn(T,N,48,device=DEV),torch.randn(T,N,12,device=DEV)]
pack=ops.RecordPack(narrow); print(pack.record_bytes, pack.used_bytes, pack.offsets)
pack.build(); sync("build")
idx=torch.from_numpy(rng.permutation(T*N)[:B].astype(np.int64)).to(DEV)
names=list(narrow)[::-1]
o,p=ops.gather_rows_packed(wide,pack,names,idx,T,N); sync("gather packed")
for n_,out in zip(names,p):
    assert torch.equal(out, narrow[n_].flatten(0,1)[idx]), n_
print("values ok")
env=torch.from_numpy(rng.permutation(N)[:N//2].astype(np.int64)).to(DEV)
o,p=ops.gather_rows_packed([],pack,names[:3],env,T,N,temporal=True); sync("temporal")
for n_,out in zip(names[:3],p):
    assert torch.equal(out, narrow[n_][:,env]), n_
print("temporal ok")
