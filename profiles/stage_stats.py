import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xugrid_amd import meshgen
n=20000
sxy,sf=meshgen.triangle_mesh(n,0); txy,tf=meshgen.triangle_mesh(n,1,30.,.7)
def ccw(xy,f):
    p=xy[f]; a=(p[:,1,0]-p[:,0,0])*(p[:,2,1]-p[:,0,1])-(p[:,1,1]-p[:,0,1])*(p[:,2,0]-p[:,0,0])
    f=f.copy(); f[a<0]=f[a<0][:,::-1]; return xy[f]
S=ccw(sxy,sf); T=ccw(txy,tf)
sb=np.concatenate([S.min(1),S.max(1)],1); tb=np.concatenate([T.min(1),T.max(1)],1)
pairs=[]
for i0 in range(0,len(T),500):
    t=tb[i0:i0+500]
    m=(t[:,None,0]<sb[None,:,2])&(sb[None,:,0]<t[:,None,2])&(t[:,None,1]<sb[None,:,3])&(sb[None,:,1]<t[:,None,3])
    ti,si=np.nonzero(m); pairs.append(np.stack([ti+i0,si],1))
pairs=np.concatenate(pairs); print("cand",len(pairs), "per target", len(pairs)/len(T))
tv=T[pairs[:,0]]; sv=S[pairs[:,1]]
# inside flags of original target verts vs the 3 clipper edges (r->s): edges (2->0),(0->1),(1->2)
def flags(tv,sv,k):
    r=sv[:,(k+2)%3]; s=sv[:,k]; U=s-r
    return (U[:,None,0]*(tv[:,:,1]-r[:,None,1]) > U[:,None,1]*(tv[:,:,0]-r[:,None,0]))
F=[flags(tv,sv,k) for k in range(3)]
dead_clip_edge=np.zeros(len(pairs),bool)
for k in range(3): dead_clip_edge|=~F[k].any(1)
# subject-edge separation: all clipper verts outside (right of) a CCW target edge 
G=[flags(sv,tv,k) for k in range(3)]
dead_subj_edge=np.zeros(len(pairs),bool)
for k in range(3): dead_subj_edge|=~G[k].any(1)
print("dead by clipper edge",dead_clip_edge.mean(),"dead by subject edge",dead_subj_edge.mean(),"either",(dead_clip_edge|dead_subj_edge).mean())
print("stage1 dead",(~F[0].any(1)).mean(), "stage1 all-inside",F[0].all(1).mean())
# fully inside: target inside source (all F all true)
allin=F[0].all(1)&F[1].all(1)&F[2].all(1); print("target fully inside source",allin.mean())
allin2=G[0].all(1)&G[1].all(1)&G[2].all(1); print("source fully inside target",allin2.mean())

# emulate S-H stage by stage (vectorised over pairs; polygons up to 7 verts)
N=len(pairs)
poly=np.zeros((N,8,2)); poly[:,:3]=tv; n=np.full(N,3); alive=np.ones(N,bool)
for k in range(3):
    r=sv[:,(k+2)%3]; s_=sv[:,k]; U=s_-r; Nn=np.stack([-U[:,1],U[:,0]],1)
    newpoly=np.zeros_like(poly); nn=np.zeros(N,int)
    idx=np.arange(N)
    a=poly[idx,n-1]
    ains=U[:,0]*(a[:,1]-r[:,1])>U[:,1]*(a[:,0]-r[:,0])
    ncross=np.zeros(N,int)
    for j in range(7):
        act=alive&(j<n)
        b=poly[:,j]
        bins=U[:,0]*(b[:,1]-r[:,1])>U[:,1]*(b[:,0]-r[:,0])
        cross=act&(bins!=ains)
        V=b-a
        nw=Nn[:,0]*(r[:,0]-a[:,0])+Nn[:,1]*(r[:,1]-a[:,1]); nv=Nn[:,0]*V[:,0]+Nn[:,1]*V[:,1]
        with np.errstate(all='ignore'):
            t=nw/nv
        pt=a+t[:,None]*V
        ii=np.nonzero(cross)[0]; newpoly[ii,nn[ii]]=pt[ii]; nn[ii]+=1; ncross[ii]+=1
        ii=np.nonzero(act&bins)[0]; newpoly[ii,nn[ii]]=b[ii]; nn[ii]+=1
        a=np.where(act[:,None],b,a); ains=np.where(act,bins,ains)
    was=alive.copy()
    alive=alive&(nn>=3)
    print(f"stage {k+1}: alive before {was.mean():.3f}  crossing lanes {(was&(ncross>0)).mean():.3f}  unchanged {(was&(ncross==0)&(nn>=3)).mean():.3f} died {(was&~alive).mean():.3f}  mean n after {nn[alive].mean():.2f}")
    poly=newpoly; n=np.where(alive,nn,3)
print("survivors",alive.mean())
