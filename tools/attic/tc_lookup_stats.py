#!/usr/bin/env python3
"""Where the look-ups of the forward triangle count are (numpy, no GPU): symmetrized R-MAT scale S, vertices relabelled by
(degree, id) rank, DAG = edges to higher ranks.  A look-up is (u -> v, w in N+(u) behind v: is w in N+(v)?).  Prints
  * the share of look-ups in walks of at least L elements, and by the rank of the middle vertex v;
  * for cores of K ranks: look-ups covered, (u, v) pairs, bytes of the bit-matrix rows they read, work items by |C(u)|;
  * the rows that stay below a core of 16384 ranks by their look-up counts.
These are the numbers behind DESIGN 4.7 / profiles/r04_tc_core.txt.      python3 tools/tc_lookup_stats.py 22"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import graphio
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g=graphio.rmat_graph(scale,16,seed=1)
rp=np.asarray(g.rowptr); ci=np.asarray(g.colidx)
m=g.m
src=np.repeat(np.arange(m,dtype=np.int64),np.diff(rp.astype(np.int64)))
dst=ci.astype(np.int64)
keep=src!=dst
a=np.concatenate([src[keep],dst[keep]]); b=np.concatenate([dst[keep],src[keep]])
key=np.unique(a*m+b)
a=key//m; b=key%m
deg=np.bincount(a,minlength=m)
# rank by (degree,id)
order=np.lexsort((np.arange(m),deg)); rank=np.empty(m,np.int64); rank[order]=np.arange(m)
ra=rank[a]; rb=rank[b]
fw=ra<rb
dplus=np.bincount(ra[fw],minlength=m)  # out-degree in DAG indexed by rank
print("m",m,"dag edges",fw.sum(),"max d+",dplus.max())
d=dplus.astype(np.float64)
P=(d*(d-1)/2).sum()
print("probes forward: %.3g  per dag edge %.1f"%(P,P/fw.sum()))
for L in (16,32,48,64,128):
    dd=np.maximum(d,L)
    # sum_{t=L}^{d-1} t = (d-1)d/2 - (L-1)L/2 for d>L
    long_=np.where(d>L,(d*(d-1)/2-(L*(L-1)/2)),0).sum()
    ntails_long=np.where(d>L,d-L,0).sum()
    print("tails >= %d: share of probes %.3f ; such tails %.3g of %d (%.3f)"%(L,long_/P,ntails_long,fw.sum(),ntails_long/fw.sum()))
# in-degree side: set sizes N+(v) and number of in-neighbours
indeg=np.bincount(rb[fw],minlength=m)
print("rows v with set: ",(dplus>0).sum(),"; in-neighbour count mean",indeg[dplus>0].mean())
# probes by rank decile of v
pe=np.zeros(m)
# tail length for each edge: position of v in sorted N+(u)
o=np.lexsort((rb[fw],ra[fw])); ua=ra[fw][o]; vb=rb[fw][o]
start=np.concatenate([[0],np.cumsum(dplus)])[:-1]
pos=np.arange(len(ua))-start[ua]
tail=dplus[ua]-1-pos
np.add.at(pe,vb,tail)
cs=np.cumsum(pe)/pe.sum()
for q in (0.5,0.8,0.9,0.95,0.99,0.999):
    print("rank < %.3f m holds %.3f of probes"%(q,cs[int(q*m)-1]))
# colidx bytes by rank decile of u (list storage)
cb=np.cumsum(dplus)/dplus.sum()
for q in (0.5,0.8,0.9,0.95,0.99):
    print("lists of rank < %.2f m hold %.3f of the DAG edges"%(q,cb[int(q*m)-1]))
print("---- core analysis")
for K in (4096,8192,16384,32768,65536):
    base=m-K
    inc=vb>=base               # pairs (u,v) with v in the core
    share=tail[inc].sum()/tail.sum()
    # C(u) size per u
    cu=np.bincount(ua[inc],minlength=m)
    for T in (0,16,32,64):
        sel=inc&(cu[ua]>=T)
        pr=tail[sel].sum()/tail.sum()
        rowbytes=((m-vb[sel]).astype(np.float64)/8).sum()   # triangular rows
        print("K %6d T %3d: probes covered %.3f (core total %.3f), pairs %.3g, tri-row bytes %.3g (full rows %.3g)  bytes per covered probe %.2f"%(K,T,pr,share,sel.sum(),rowbytes,sel.sum()*K/8.0,rowbytes/max(tail[sel].sum(),1)))
print("---- items")
for K in (8192,16384):
    base=m-K
    inc=vb>=base
    cu=np.bincount(ua[inc],minlength=m)
    print("K",K,"items (cu>=2):",(cu>=2).sum()," cu>=8:",(cu>=8).sum()," cu>=32:",(cu>=32).sum()," cu>=64:",(cu>=64).sum(),"cu>=256:",(cu>=256).sum())
    for lo,hi in ((2,8),(8,32),(32,64),(64,256),(256,100000)):
        sel=(cu>=lo)&(cu<hi)
        print("   cu in [%d,%d): items %d, pairs(u,v) %d, pair-tests %.3g"%(lo,hi,sel.sum(),cu[sel].sum(),(cu[sel]*(cu[sel]-1)/2).sum()))
print("---- rows below the core")
K=16384
base=m-K
below=vb<base
rows=np.unique(vb[below])
lk=np.zeros(m); np.add.at(lk,vb[below],tail[below])
indeg_b=np.bincount(vb[below],minlength=m)
q=lk[rows]
print("rows below the core that are walked:",len(rows),"look-ups",q.sum(),"per row mean %.0f median %.0f"%(q.mean(),np.median(q)))
for lo,hi in ((0,64),(64,256),(256,1024),(1024,4096),(4096,1e12)):
    sel=(q>=lo)&(q<hi)
    print("  rows with look-ups in [%g,%g): %d rows (%.2f), %.3g look-ups (%.3f)"%(lo,hi,sel.sum(),sel.mean(),q[sel].sum(),q[sel].sum()/q.sum()))
print("in-neighbours per walked row: mean %.1f; set size d+(v) mean %.1f"%(indeg_b[rows].mean(), dplus[rows].mean()))
