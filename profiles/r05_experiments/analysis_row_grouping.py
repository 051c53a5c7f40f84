import numpy as np
d = np.load("/tmp/an/pairs.npz")  # written by analysis_gen_pairs.py (CPU oracle; offline analysis only)
tgt, src = d["tgt"], d["src"]; sxy, sf, txy, tf = d["sxy"], d["sf"], d["txy"], d["tf"]
T, S = tf.shape[0], sf.shape[0]
order = np.lexsort((src, tgt)); tgt, src = tgt[order], src[order]
indptr = np.zeros(T+1, np.int64); np.add.at(indptr, tgt+1, 1); indptr = np.cumsum(indptr)
tc = txy[tf].mean(1)
def morton(ix, iy):
    def part(v):
        v = v.astype(np.uint64) & 0xffff
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return part(ix) | (part(iy) << 1)
def block_stats(row_perm, label, rows_per_block=256):
    nb = T//rows_per_block
    U=[];L64=[];L128=[];P=[]
    rng = np.random.default_rng(0)
    for b in rng.choice(nb, min(nb, 300 if rows_per_block <= 1024 else 24), replace=False):
        rows = row_perm[b*rows_per_block:(b+1)*rows_per_block]
        cols = np.concatenate([src[indptr[r]:indptr[r+1]] for r in rows])
        u = np.unique(cols)
        U.append(u.size); L64.append(np.unique(u>>3).size); L128.append(np.unique(u>>4).size); P.append(np.unique(u>>1).size)
    print("%-44s rows/blk %d: cols %.0f pairs16B %.0f lines64 %.0f lines128 %.0f   per row: cols %.2f lines128 %.3f" % (label, rows_per_block, np.mean(U), np.mean(P), np.mean(L64), np.mean(L128), np.mean(U)/rows_per_block, np.mean(L128)/rows_per_block))
def tiled(run, cellsz_faces):
    nrun = (T + run - 1)//run
    mid = np.minimum(np.arange(nrun)*run + run//2, T-1)
    h = 0.7*np.sqrt(cellsz_faces/ T); lo = tc.min(0)
    key = morton(((tc[mid,0]-lo[0])/h).astype(np.int64), ((tc[mid,1]-lo[1])/h).astype(np.int64))
    o = np.argsort(key, kind="stable")
    perm = (o[:,None]*run + np.arange(run)[None,:]).ravel()
    return perm[perm < T]
def by_column(run):
    nrun = T//run
    mid = np.arange(nrun)*run + run//2
    key = src[indptr[mid]]           # first (lowest) column of the middle row
    o = np.argsort(key, kind="stable")
    return (o[:,None]*run + np.arange(run)[None,:]).ravel()
def by_column_med(run):
    nrun = T//run
    key = np.array([np.median(src[indptr[r*run]:indptr[(r+1)*run]]) for r in range(nrun)])
    o = np.argsort(key, kind="stable")
    return (o[:,None]*run + np.arange(run)[None,:]).ravel()
if 0: pass
if 0: pass
if 0: pass
for run in ():
    block_stats(by_column(run), "sorted by first column of mid row, run %d" % run)
    block_stats(by_column_med(run), "sorted by median column of run, run %d" % run)
    block_stats(by_column_med(run), "sorted by median column of run, run %d" % run, 128)
pass
print("---- super tiles")
perm = tiled(16,144)
for n in (256, 1024, 4096, 16384, 65536):
    block_stats(perm, "morton tiles run16", n)
