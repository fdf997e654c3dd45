#!/usr/bin/env python3
"""CPU simulation behind LABNOTES.md section 9 (no GPU, checker-side only): node visits and leaf-box passes of depth-1
rays on the C5 scene under different scan orders / cull assumptions, and the loop-step counts of a 64-lane wave with
and without (ideal) work stealing.  Rays come from the oracle's stage functions; the scan itself is re-implemented
here in Python on the BVH table of the product's host builder.  usage: tools/sim_scan_orders.py"""
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
left=[b.left for b in B]; right=[b.right for b in B]; geom=[b.geomIndex for b in B]
rng=np.random.default_rng(1)
NR=1200
pix=np.sort(rng.choice(w*h, NR, replace=False)).astype(np.int64)
O=np.zeros((3,NR),np.float32); D=np.zeros((3,NR),np.float32)
for i,p in enumerate(pix):
    o,d=ob.generate(int(p),1); O[:,i]=o[:,0]; D[:,i]=d[:,0]
hit0=ob.intersect(O,D)
# depth-1 rays
it=np.ones(NR,np.int32); col=np.ones((3,NR),np.float32); rem=np.full(NR,8,np.int32)
O1,D1,c1,rem1=ob.shade(0,it,pix.astype(np.int32),hit0,O,D,col,rem)
alive=rem1>0
O1=O1[:,alive]; D1=D1[:,alive]
hit1=ob.intersect(O1,D1)
print('primary hits',(hit0['t']>0).mean(),'depth1 rays',alive.sum(),'hit frac',(hit1['t']>0).mean())
# subtree cut like build_top: split largest until 32
size={}
def sz(i):
    if left[i]<0: size[i]=1
    else: size[i]=1+sz(left[i])+sz(right[i])
    return size[i]
sys.setrecursionlimit(10000); sz(0)
cut=[0]
while len(cut)<32:
    best=max((c for c in cut if left[c]>=0), key=lambda c:size[c], default=None)
    if best is None: break
    cut[cut.index(best)]=right[best]; cut.append(left[best])
def slab(o,inv,i):
    t0=(bmin[i]-o)*inv; t1=(bmax[i]-o)*inv
    lo=np.minimum(t0,t1); hi=np.maximum(t0,t1)
    tmin=max(0.0,lo.max()); tmax=hi.min()
    return tmax>tmin, tmin
def near_first(i,d):
    L,R=left[i],right[i]
    cl=bmin[L]+bmax[L]; cr=bmin[R]+bmax[R]
    a=int(np.argmax(np.abs(cl-cr)))
    l_lower=cl[a]<=cr[a]
    return (L,R) if (l_lower != (d[a]<0)) else (R,L)
def walk(o,d,tbest,geom_best,mode,lag=0):
    # returns visits; mode: 'ref' (right first), 'oct'; cull after true leaf reached (lag = extra visits before known)
    inv=1.0/d.astype(np.float64); o=o.astype(np.float64)
    visits=0; found_at=None; cands=0
    ents=[]
    for c in cut:
        ok,tm=slab(o,inv,c); 
        if ok: ents.append((tm,c))
    if mode.endswith('sorted'): ents.sort()
    for tm,c in ents:
        stack=[c]; first=True
        while stack:
            i=stack.pop()
            if not first:
                visits+=1
            ok,tmn=slab(o,inv,i) if not first else (True,tm)
            first=False
            known = found_at is not None and visits>=found_at+lag
            if not ok or (known and tmn>tbest+0.02): continue
            if left[i]<0:
                cands+=1
                if geom[i]==geom_best and found_at is None: found_at=visits
            else:
                if mode.startswith('oct'):
                    a,b=near_first(i,d)
                else:
                    a,b=right[i],left[i]
                stack.append(b); stack.append(a)
    return visits,cands
def run(Os,Ds,hit,label):
    res={}
    for mode,cull,lag in [('ref',False,0),('ref',True,0),('oct',True,0),('oct_sorted',True,0),('oct_sorted',True,10),('oct_sorted',True,30),('ref_sorted',True,0)]:
        V=[];Cn=[]
        for k in range(Os.shape[1]):
            tb=float(hit['t'][k]); gb=int(hit['geom'][k])
            if not cull or tb<=0: tb=1e30; 
            if tb>=1e30: gb=-2
            v,c=walk(Os[:,k],Ds[:,k],tb,gb,mode,lag); V.append(v);Cn.append(c)
        print(label,mode,'cull' if cull else 'nocull','lag',lag,'visits mean %.1f max %d cands %.2f'%(np.mean(V),np.max(V),np.mean(Cn)))
run(O[:,:400],D[:,:400],{k:v[...,:400] for k,v in hit0.items()},'primary')
run(O1[:,:400],D1[:,:400],{k:v[...,:400] for k,v in hit1.items()},'depth1 ')

def pairs(o,d,tbest,geom_best):
    inv=1.0/d.astype(np.float64); o=o.astype(np.float64)
    out=[]; found=False
    for c in cut:
        ok,tm=slab(o,inv,c)
        if not ok: continue
        if left[c]<0: continue
        v=0; stack=[right[c],left[c]][::-1]
        stack=[left[c],right[c]]
        while stack:
            i=stack.pop(); v+=1
            ok,tmn=slab(o,inv,i)
            if not ok or (found and tmn>tbest+0.02): continue
            if left[i]<0:
                if geom[i]==geom_best: found=True
            else:
                stack.append(left[i]); stack.append(right[i])
        out.append(v)
    return out
P=[]
for k in range(O1.shape[1]):
    tb=float(hit1['t'][k]); gb=int(hit1['geom'][k])
    if tb<=0: tb=1e30; gb=-2
    P.append(pairs(O1[:,k],D1[:,k],tb,gb))
tot=np.array([sum(p) for p in P]); npair=np.array([len(p) for p in P]); mx=np.array([max(p) if p else 0 for p in P])
print('rays',len(P),'mean total visits %.1f'%tot.mean(),'mean #subtrees %.2f'%npair.mean(),'mean max-pair %.1f'%mx.mean(),'overall max pair',mx.max())
rng=np.random.default_rng(0)
its_now=[];its_bal=[];its_lb=[]
for g in range(200):
    idx=rng.choice(len(P),64,replace=False)
    its_now.append(tot[idx].max())
    allp=sorted([v for i in idx for v in P[i]],reverse=True)
    # greedy list scheduling on 64 lanes (upper bound of a good steal)
    lanes=np.zeros(64)
    for v in allp: lanes[lanes.argmin()]+=v
    its_bal.append(lanes.max()); its_lb.append(max(tot[idx].sum()/64, allp[0] if allp else 0))
print('iterations per group: now (max over lanes) %.0f | ideal stealing %.0f | lower bound %.0f'%(np.mean(its_now),np.mean(its_bal),np.mean(its_lb)))
