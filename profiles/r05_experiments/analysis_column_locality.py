import numpy as np
d = np.load("/tmp/an/pairs.npz")  # written by analysis_gen_pairs.py (CPU oracle; offline analysis only)
tgt, src = d["tgt"], d["src"]; sxy, sf, txy, tf = d["sxy"], d["sf"], d["txy"], d["tf"]
T, S = tf.shape[0], sf.shape[0]
tc = txy[tf].mean(1); sc = sxy[sf].mean(1)
# numbering coherence
for name, c in (("src", sc), ("tgt", tc)):
    dd = np.hypot(*(c[1:] - c[:-1]).T)
    ext = np.sqrt(1.0 / c.shape[0]) if name=="src" else 0.7*np.sqrt(1.0/c.shape[0])
    print(name, "consecutive-id distance / mean extent: median %.2f mean %.2f p90 %.2f p99 %.2f" % tuple(np.array([np.median(dd), dd.mean(), np.percentile(dd,90), np.percentile(dd,99)])/ext))
def morton(ix, iy):
    def part(v):
        v = v.astype(np.uint64) & 0xffff
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return part(ix) | (part(iy) << 1)
order = np.lexsort((src, tgt)); tgt, src = tgt[order], src[order]
indptr = np.zeros(T+1, np.int64); np.add.at(indptr, tgt+1, 1); indptr = np.cumsum(indptr)
def block_stats(row_perm, label, colmap=None):
    # row_perm: stored row r -> caller row
    nb = (T + 255)//256
    U=[];L64=[];L128=[];runs=[]
    rng = np.random.default_rng(0)
    for b in rng.choice(nb, 400, replace=False):
        rows = row_perm[b*256:(b+1)*256]
        cols = np.concatenate([src[indptr[r]:indptr[r+1]] for r in rows])
        if colmap is not None: cols = colmap[cols]
        u = np.unique(cols)
        U.append(u.size); L64.append(np.unique(u>>3).size); L128.append(np.unique(u>>4).size)
        runs.append(1 + np.count_nonzero(np.diff(u) != 1))
    print(label, "entries/block %.0f distinct cols %.0f lines64 %.0f lines128 %.0f runs %.0f" % (4052965/nb, np.mean(U), np.mean(L64), np.mean(L128), np.mean(runs)))
# caller order rows
block_stats(np.arange(T), "rows caller order      ")
# engine tiling: runs of 64 ids keyed by Morton of the middle face's coarse cell
def tiled(run, cellsz_faces):
    nrun = (T + run - 1)//run
    mid = np.minimum(np.arange(nrun)*run + run//2, T-1)
    h = 0.7*np.sqrt(cellsz_faces/ T)
    lo = tc.min(0)
    key = morton(((tc[mid,0]-lo[0])/h).astype(np.int64), ((tc[mid,1]-lo[1])/h).astype(np.int64))
    o = np.argsort(key, kind="stable")
    perm = (o[:,None]*run + np.arange(run)[None,:]).ravel()
    return perm[perm < T]
for run in (64, 16, 4, 1):
    for cs in (64, 256):
        block_stats(tiled(run, cs), "rows tiled run=%d cell=%d" % (run, cs))
# with morton col numbering
h = np.sqrt(8.0/S); lo = sc.min(0)
ck = morton(((sc[:,0]-lo[0])/h).astype(np.int64), ((sc[:,1]-lo[1])/h).astype(np.int64))
colorder = np.argsort(ck, kind="stable"); colmap = np.empty(S, np.int64); colmap[colorder] = np.arange(S)
block_stats(tiled(64,256), "tiled64 + morton cols  ", colmap)
block_stats(tiled(1,256), "tiled1 + morton cols  ", colmap)
