import numpy as np
d = np.load("/tmp/an/pairs.npz")  # written by analysis_gen_pairs.py (CPU oracle; offline analysis only)
sxy, sf, txy, tf = d["sxy"], d["sf"], d["txy"], d["tf"]
def boxes(xy, f):
    p = xy[f]; return p[:,:,0].min(1), p[:,:,0].max(1), p[:,:,1].min(1), p[:,:,1].max(1)
sx0,sx1,sy0,sy1 = boxes(sxy,sf); tx0,tx1,ty0,ty1 = boxes(txy,tf)
ext = np.maximum(sx1-sx0, sy1-sy0)
mean_ext = 0.5*((sx1-sx0).mean() + (sy1-sy0).mean())
h0 = 1.25*mean_ext
X0, Y0 = sx0.min(), sy0.min()
lvl = np.maximum(0, np.ceil(np.log2(ext*1.001/h0))).astype(int)
print("h0", h0, "levels", np.bincount(lvl))
T = tf.shape[0]
tot_tests = np.zeros(T, int); tot_steps = np.zeros(T, int); tot_rows=np.zeros(T,int)
wave_cost = np.zeros(T//64+1)
steps_rows = []   # per level list of (T, maxrows) arrays of steps
for l in range(2):
    h = h0*2**l
    m = lvl==l
    cx = np.floor((sx0[m]-X0)/h).astype(int); cy = np.floor((sy0[m]-Y0)/h).astype(int)
    nx = int(np.ceil((sx1.max()-X0)/h))+2; ny = int(np.ceil((sy1.max()-Y0)/h))+2
    cnt = np.zeros((ny, nx+1), int); np.add.at(cnt, (cy, cx), 1)
    cum = np.concatenate([np.zeros((ny,1),int), np.cumsum(cnt,1)],1)  # cum[row, c] = records in cells < c
    qx0 = np.maximum(np.floor((tx0-X0)/h).astype(int)-1,0); qx1 = np.floor((tx1-X0)/h).astype(int)
    qy0 = np.maximum(np.floor((ty0-Y0)/h).astype(int)-1,0); qy1 = np.floor((ty1-Y0)/h).astype(int)
    nrows = qy1-qy0+1
    tot_rows += nrows
    print("level",l,"rows per face hist", np.bincount(nrows)[:8])
    st = np.zeros((T, 4), int)
    for k in range(4):
        row = np.minimum(qy0+k, ny-1)
        ln = np.where(k < nrows, cum[row, np.minimum(qx1+1,nx)] - cum[row, np.minimum(qx0,nx)], 0)
        tot_tests += ln; st[:,k] = (ln+3)//4
    tot_steps += st.sum(1)
    steps_rows.append(st)
print("tests per face mean", tot_tests.mean(), "steps per face mean", tot_steps.mean(), "rows", tot_rows.mean())
# wave cost: for each wave, per level per k: max over lanes of steps
W = T//64
cost=0; 
for st in steps_rows:
    s = st[:W*64].reshape(W,64,4)
    cost += s.max(1).sum(1)
print("steps per WAVE (max over lanes, summed over rows/levels): mean", cost.mean(), " vs mean per-lane steps", tot_steps.mean(), " ratio", cost.mean()/tot_steps.mean())
print("useful records per wave-step-slot: ", tot_tests.sum()/ (cost.sum()*64*4))
s_all = np.concatenate(steps_rows,1)[:W*64].reshape(W,64,8)
print("flattened runs: max over lanes of sum of steps: mean", s_all.sum(2).max(1).mean())
tt = tot_tests[:W*64].reshape(W,64)
print("perfect flattening ceil(sum len/4): mean", np.ceil(tt/4).max(1).mean(), " lane mean", np.ceil(tt/4).mean())
for wl in (2,6,8):
    print("loads per step", wl, ": flattened runs", (np.ceil(np.concatenate([ (lambda st: st)(s) for s in steps_rows],1)*4/wl)).shape)
print("---- groupings (h0 1.25)")
S = s_all  # (W,64,8): [L0 k0..3, L1 k0..3]
def cost(groups): return sum(S[:,:,g].sum(2).max(1) for g in groups).mean()
print("singles", cost([[i] for i in range(8)]))
print("pairs across levels (L0k,L1k)", cost([[k,4+k] for k in range(4)]))
print("pairs within level", cost([[0,1],[2,3],[4,5],[6,7]]))
print("quads per level", cost([[0,1,2,3],[4,5,6,7]]))
print("rows 0-1 of both levels + rows 2-3 of both", cost([[0,1,4,5],[2,3,6,7]]))
print("octet", cost([list(range(8))]))
