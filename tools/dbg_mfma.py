import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from coast_amd.engine import Engine, XmrConfig
eng=Engine()
rng=np.random.default_rng(1)
for batch in (1,3):
  for rep in (3,2,1):
    f=rng.integers(0,2**32,size=(batch,256,256),dtype=np.uint32); s=rng.integers(0,2**32,size=(batch,256,256),dtype=np.uint32)
    want=np.einsum('bik,bkj->bij',f.astype(np.uint64)&0xffffffff,s.astype(np.uint64)) .astype(np.uint32) if False else None
    want=np.stack([(f[b].astype(np.uint64)[:,:,None]*0).sum(-1) for b in range(batch)]) if False else None
    w=np.zeros((batch,256,256),dtype=np.uint32)
    for b in range(batch):
        acc=np.zeros((256,256),dtype=np.uint64)
        for k in range(256):
            acc+= (f[b,:,k].astype(np.uint64)[:,None]*s[b,k,:].astype(np.uint64)[None,:]) & 0xffffffff
            acc&=0xffffffff
        w[b]=acc.astype(np.uint32)
    r=eng.mm_batch(torch.from_numpy(f.view(np.int32)).cuda(),torch.from_numpy(s.view(np.int32)).cuda(),cfg=XmrConfig(rep))
    got=r[0].cpu().numpy().view(np.uint32) if isinstance(r,tuple) else r.cpu().numpy().view(np.uint32)
    bad=(got!=w)
    print(batch,rep,bad.sum(), [ (b, np.unique(np.nonzero(bad[b])[0]//32).tolist(), np.unique(np.nonzero(bad[b])[1]//8).tolist()) for b in range(batch) if bad[b].any()])
