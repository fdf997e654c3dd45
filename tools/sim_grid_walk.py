#!/usr/bin/env python3
"""CPU simulation behind LABNOTES.md section 9.1 (no GPU, checker-side only): cells crossed, records met, distinct leaves
and leaf boxes passed per depth-1 ray of the C5 scene for a uniform grid at 1, 2 and 4 cells per primitive, and the longest
of 64 walks (what a wave's loop runs).  Rays come from the oracle's stage functions.  usage: tools/sim_grid_walk.py [rays]"""
import os, sys, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
from oracle import binding as ob
w,h=1920,1080
text=scenes.stress_scene_text((22,22,21),res=(w,h),depth=8)
path=scenes.write_scene(text, os.path.join(tempfile.mkdtemp(),'s.txt'))
sc=capi.Scene(path,res=(w,h)); ob.load_scene(path,res=(w,h))
B=sc.bvh(); n=len(B)
bmin=np.array([[b.bmin[0],b.bmin[1],b.bmin[2]] for b in B],np.float64); bmax=np.array([[b.bmax[0],b.bmax[1],b.bmax[2]] for b in B],np.float64)
geom=np.array([b.geomIndex for b in B])
leaves=np.where(geom>=0)[0]
print('nodes',n,'leaves',len(leaves))
rng=np.random.default_rng(1)
NR=int(sys.argv[1]) if len(sys.argv)>1 else 640
# contiguous pixels in blocks of 64 to imitate wave groups
starts=rng.choice(w*h//64, NR//64, replace=False)*64
pix=np.concatenate([np.arange(s,s+64) for s in starts]).astype(np.int64)
O=np.zeros((3,NR),np.float32); D=np.zeros((3,NR),np.float32)
for i,p in enumerate(pix):
    o,d=ob.generate(int(p),1); O[:,i]=o[:,0]; D[:,i]=d[:,0]
hit0=ob.intersect(O,D)
it=np.ones(NR,np.int32); col=np.ones((3,NR),np.float32); rem=np.full(NR,8,np.int32)
O1,D1,c1,rem1=ob.shade(0,it,pix.astype(np.int32),hit0,O,D,col,rem)
alive=rem1>0
O1=O1[:,alive]; D1=D1[:,alive]
hit1=ob.intersect(O1,D1)
T1=np.where(hit1['t']>0,hit1['t'],np.inf)
print('depth1 rays',alive.sum(),'hit frac',(hit1['t']>0).mean())
gmin=bmin[0]-1e-3; gmax=bmax[0]+1e-3; ext=gmax-gmin
N=len(leaves)
for k in (1.0,2.0,4.0):
    dens=(k*N/np.prod(ext))**(1/3)
    res=np.maximum(1,np.round(ext*dens)).astype(int)
    cs=ext/res
    cells={}
    nrefs=0
    for li in leaves:
        lo=np.clip(np.floor((bmin[li]-1e-3-gmin)/cs).astype(int),0,res-1)
        hi=np.clip(np.floor((bmax[li]+1e-3-gmin)/cs).astype(int),0,res-1)
        for x in range(lo[0],hi[0]+1):
            for y in range(lo[1],hi[1]+1):
                for z in range(lo[2],hi[2]+1):
                    cells.setdefault((x,y,z),[]).append(li); nrefs+=1
    cnts=np.array([len(v) for v in cells.values()])
    print(f'k={k} res={res} cells={np.prod(res)} nonempty={len(cells)} refs={nrefs} mean items/nonempty={cnts.mean():.2f} max={cnts.max()} >3: {(cnts>3).mean():.3f}')
    margin=0.01
    ncell=[];nitem=[];nuniq=[];nbox=[]
    for r in range(O1.shape[1]):
        o=O1[:,r].astype(np.float64); d=D1[:,r].astype(np.float64)
        inv=1.0/d
        t0=(gmin-o)*inv; t1=(gmax-o)*inv
        tn=max(0.0,np.minimum(t0,t1).max()); tf=np.maximum(t0,t1).min()
        if tf<=tn: ncell.append(0);nitem.append(0);nuniq.append(0);nbox.append(0);continue
        p=o+d*tn
        ijk=np.clip(np.floor((p-gmin)/cs).astype(int),0,res-1)
        step=np.where(d>0,1,-1)
        tmax=((ijk+(d>0))*cs+gmin-o)*inv
        tdel=cs*np.abs(inv)
        te=tn; nc=0; ni=0; seen=set(); nb=0
        best=T1[r]   # ideal cull: best known as soon as hit cell reached (optimistic) -> emulate lag: known only after cell where hit found is processed
        found=np.inf
        while True:
            if te>found+margin: break
            nc+=1
            for li in cells.get(tuple(ijk),()):
                ni+=1
                if li not in seen:
                    seen.add(li)
                    # exact box test
                    a=(bmin[li]-o)*inv; b=(bmax[li]-o)*inv
                    if np.maximum(a,b).min()>max(0,np.minimum(a,b).max()):
                        nb+=1
                        if hit1['t'][r]>0 and False: pass
            # hit known if the true hit t lies within this cell's span
            ax=int(np.argmin(tmax))
            tx=tmax[ax]
            if best<=tx: found=best
            ijk[ax]+=step[ax]
            if ijk[ax]<0 or ijk[ax]>=res[ax]: break
            te=tx; tmax[ax]+=tdel[ax]
        ncell.append(nc);nitem.append(ni);nuniq.append(len(seen));nbox.append(nb)
    ncell=np.array(ncell);nitem=np.array(nitem);nuniq=np.array(nuniq);nbox=np.array(nbox)
    g=len(ncell)//64*64
    mx=ncell[:g].reshape(-1,64).max(1)
    print(f'   cells/ray mean {ncell.mean():.1f} p90 {np.percentile(ncell,90):.0f} max {ncell.max()} ; wave max-of-64 mean {mx.mean():.1f}; items/ray {nitem.mean():.1f} uniq {nuniq.mean():.1f} box-pass {nbox.mean():.1f}')
