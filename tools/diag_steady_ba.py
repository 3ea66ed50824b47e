import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from devo_amd import synth
from devo_amd.backends import cuda_ba
from oracle import fastba as F
N_KF,M,H,W=40,96,120,160
poses=synth.make_poses(48,11); patches,_=synth.make_patches(48,M,H,W,seed=11); intr=synth.make_intrinsics(48,H,W)
ii,jj,kk=synth.sliding_window_graph(N_KF,M)
d=lambda t:t.cuda()
P_,Q_=d(poses).clone(),d(patches).clone()
coords=cuda_ba.transform(P_,Q_,d(intr),d(ii),d(jj),d(kk),layout="2pp")
delta,weight=synth.make_update_outputs(len(ii),11,sigma=0.5)
target=coords[:,:,:,1,1]+d(delta)
lm=torch.tensor([1e-4]).cuda()
iters=int(os.environ.get("ITERS","2"))
cuda_ba.forward(P_,Q_,d(intr),target,d(weight),lm,d(ii),d(jj),d(kk),30,40,iters)
print(cuda_ba.last_path())
p64,q64=F.ba(poses.double(),patches.double(),intr.double(),target.cpu().double(),weight.double(),torch.tensor([1e-4]),ii,jj,kk,30,40,iters,dtype=torch.float64)
e=(Q_.cpu()[0,:,2,1,1].double()-q64[0,:,2,1,1]).abs()
print("max depth err", float(e.max()), "pose err", float((P_.cpu().double()-p64).abs().max()))
bad=(e>1e-4).nonzero().squeeze(1)
print("bad patches", len(bad), "frames", sorted(set((bad//M).tolist())))
for b in bad[:8].tolist():
    m=(kk==b); print(b, b//M, "edges", int(m.sum()), "targets", sorted(jj[m].tolist()), "err", float(e[b]), "hip", float(Q_[0,b,2,1,1]), "ref", float(q64[0,b,2,1,1]), "old", float(patches[0,b,2,1,1]))
