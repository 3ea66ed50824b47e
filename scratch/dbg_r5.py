import sys, torch, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_altcorr import _case, _run
from oracle import altcorr as A
R = int(sys.argv[1])
c = _case(R=R, seed=5)
ref = A.corr_forward(*c)
got = _run(*c, layout="cl").cpu().double()
D = 2*R+2
err = (got - ref).abs()[0]          # [E, c, a, i0, j0]
e_err = err.flatten(1).max(-1).values
coords = c[2][0]
ox = coords[:, 0].flatten(1).floor(); oy = coords[:, 1].flatten(1).floor()
bw = (ox.max(1).values - ox.min(1).values + D); bh = (oy.max(1).values - oy.min(1).values + D)
bad = (e_err > 1e-3).nonzero()[:, 0]
print("R", R, "n bad", len(bad), "of", len(e_err))
for e in bad[:4].tolist():
    print("edge", e, float(e_err[e]), int(bw[e]), int(bh[e]))
    be = err[e] > 1e-3     # [c,a,i0,j0]
    idx = be.nonzero()
    # map to bbox position of the 4 taps
    xmin = ox[e].min(); ymin = oy[e].min()
    poss = set()
    for (cc, aa, i0, j0) in idx.tolist()[:400]:
        p = i0*3+j0
        gy = oy[e, p] - R + aa; gx = ox[e, p] - R + cc
        pos = int((gy - (ymin - R)) * bw[e] + (gx - (xmin - R)))
        poss.add(pos)
    print(" bad outputs", len(idx), "min/max pos(top-left tap)", min(poss), max(poss))
e = bad[0].item()
be = (err[e] > 1e-3).nonzero()
for (cc, aa, i0, j0) in be.tolist()[:5]:
    print("c,a,i0,j0", cc, aa, i0, j0, "got", float(got[0, e, cc, aa, i0, j0]), "ref", float(ref[0, e, cc, aa, i0, j0]))
print("ox", ox[e].tolist(), "oy", oy[e].tolist())
raw = A.corr_raw(*c)[0, e]   # [a,c,i0,j0]
p_i0, p_j0 = be[0][2].item(), be[0][3].item()
aa, cc = be[0][1].item(), be[0][0].item()
print("raw taps", [float(raw[aa+da, cc+dc, p_i0, p_j0]) for da in (0,1) for dc in (0,1)])
x = coords[e,0,p_i0,p_j0]; y = coords[e,1,p_i0,p_j0]
print("dx,dy", float(x-x.floor()), float(y-y.floor()))
