import sys, torch, numpy as np
sys.path.insert(0,'.')
from devo_amd import synth
from oracle import pops
cfg=synth.workload("cfg2")
n,M,H,W=cfg["n"],cfg["M"],cfg["H"],cfg["W"]
poses=synth.make_poses(n,1234); patches,_=synth.make_patches(n,M,H,W,seed=1234); intr=synth.make_intrinsics(n,H,W)
ii,jj,kk=synth.full_graph(n,M)
from oracle import fastba
co=fastba.reproject(poses,patches,intr,ii,jj,kk)  # try
print(type(co), getattr(co,'shape',None))
c=co[0].numpy().reshape(-1,2,9)
for lvl,sc in ((0,1.0),(1,4.0)):
    x=np.floor(c[:,0]/sc); y=np.floor(c[:,1]/sc)
    key=(x*100000+y)
    nd=np.array([len(np.unique(k)) for k in key])
    bw=(x.max(1)-x.min(1)+8); bh=(y.max(1)-y.min(1)+8)
    print("level",lvl,"distinct offsets hist",np.bincount(nd,minlength=10)[1:], "mean",nd.mean())
    print("  box w mean",bw.mean(),"h",bh.mean(),"area mean",(bw*bh).mean(), "pct>128",(bw*bh>128).mean(), "pct<=64",(bw*bh<=64).mean(), "<=81", (bw*bh<=81).mean(), "<=100",(bw*bh<=100).mean())
    # spacing
    sx=np.abs(c[:,0,1]-c[:,0,0])/sc
    print("  spacing pct", np.percentile(sx,[5,25,50,75,95]))
x=np.floor(c[:,0]); y=np.floor(c[:,1])
bw=(x.max(1)-x.min(1)+8); bh=(y.max(1)-y.min(1)+8); ar=bw*bh
for t in (100,110,121,128,144,160,192,256): print(t, (ar>t).mean())
def pitch(w): s=3*w; return s+((8-s)&15)
for (tp,ts) in ((128,422),(160,480),(160,528),(160,560),(192,600)):
    ok=(ar<=tp)&(np.array([bh[i]*pitch(int(bw[i])) for i in range(len(bw))])<=ts)
    print(tp,ts,"heavy frac",1-ok.mean())
