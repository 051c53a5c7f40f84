import numpy as np
d = np.load("/tmp/an/pairs.npz")  # written by analysis_gen_pairs.py (CPU oracle; offline analysis only)
sxy, sf, txy, tf = d["sxy"], d["sf"], d["txy"], d["tf"]
def boxes(xy, f):
    p = xy[f]; return p[:,:,0].min(1), p[:,:,0].max(1), p[:,:,1].min(1), p[:,:,1].max(1)
sx0,sx1,sy0,sy1 = boxes(sxy,sf); tx0,tx1,ty0,ty1 = boxes(txy,tf)
ext = np.maximum(sx1-sx0, sy1-sy0)
mean_ext = 0.5*((sx1-sx0).mean() + (sy1-sy0).mean())
X0, Y0 = sx0.min(), sy0.min()
T = tf.shape[0]; W=T//64
for fac in (1.0, 1.25, 1.5, 1.75, 2.0, 2.5):
    h0 = fac*mean_ext
    lvl = np.maximum(0, np.ceil(np.log2(ext*1.001/h0))).astype(int)
    tests = np.zeros(T,int); steps=[]; nl=0
    for l in range(3):
        m = lvl==l
        if m.sum() < 2000: continue
        nl+=1
        h = h0*2**l
        cx = np.floor((sx0[m]-X0)/h).astype(int); cy = np.floor((sy0[m]-Y0)/h).astype(int)
        nx = int(np.ceil((sx1.max()-X0)/h))+2; ny = int(np.ceil((sy1.max()-Y0)/h))+2
        cnt = np.zeros((ny, nx+1), int); np.add.at(cnt, (cy, cx), 1)
        cum = np.concatenate([np.zeros((ny,1),int), np.cumsum(cnt,1)],1)
        qx0 = np.maximum(np.floor((tx0-X0)/h).astype(int)-1,0); qx1 = np.floor((tx1-X0)/h).astype(int)
        qy0 = np.maximum(np.floor((ty0-Y0)/h).astype(int)-1,0); qy1 = np.floor((ty1-Y0)/h).astype(int)
        nrows = qy1-qy0+1
        st = np.zeros((T,4),int)
        for k in range(4):
            row = np.minimum(qy0+k, ny-1)
            ln = np.where(k < nrows, cum[row, np.minimum(qx1+1,nx)] - cum[row, np.minimum(qx0,nx)], 0)
            tests += ln; st[:,k] = (ln+3)//4
        steps.append(st)
    s_all = np.concatenate(steps,1)[:W*64].reshape(W,64,-1)
    cur = sum(s[:W*64].reshape(W,64,4).max(1).sum(1) for s in steps).mean()
    print("h0 factor %.2f levels(walked) %d share L0 %.3f tests/face %.1f | wave steps: current %.1f flattened %.1f perfect %.1f" % (fac, nl, (lvl==0).mean(), tests.mean(), cur, s_all.sum(2).max(1).mean(), np.ceil(tests[:W*64].reshape(W,64)/4).max(1).mean()))
