import os, sys, numpy as np
def remap(b, nb):
    q, r, x, i = nb >> 3, nb & 7, b & 7, b >> 3
    return x * (q + 1) + i if x < r else r * (q + 1) + (x - r) * q + i
s=np.load(sys.argv[1]); n=s.shape[0]
inv=np.zeros(n,dtype=int)
for blk in range(n): inv[remap(blk,n)]=blk
rel=(s-s[:, [0,8]][s[:, [0,8]]>0].min())*10.0/1e3
first=inv<256
names={0:"start",6:"first drive: blocks stored",3:"(a_k, h) partial done",4:"pair constants made",5:"Q seen, pair products",1:"wave sums done",2:"scalar entries stored",7:"wave done"}
for off,lab in ((0,"first compute wave"),(8,"last compute wave")):
    print(lab)
    prev=None
    for k in [0,6,3,4,5,1,2,7]:
        a,bb=rel[first,off+k],rel[~first,off+k]
        d="" if prev is None else f"   (+{np.median(rel[first,off+k]-rel[first,off+prev]):.2f} / +{np.median(rel[~first,off+k]-rel[~first,off+prev]):.2f})"
        print(f"  {names[k]:28s} first on its CU {np.median(a):5.2f} (max {a.max():5.2f})   second {np.median(bb):5.2f} (max {bb.max():5.2f}){d}")
        prev=k
