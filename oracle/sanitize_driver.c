/*
 * sanitize_driver.c -- runs the CPU oracle (test infrastructure, see xr_oracle.h) on small seeded inputs under
 * AddressSanitizer + UndefinedBehaviorSanitizer (the role numba's bounds checking / a race detector
 * would play for the reference).  Built and run by `make -C oracle sanitize` and by tests/test_oracle_goldens.py.
 * Exit code 0 = every call returned and the sanitizers stayed silent (-fno-sanitize-recover: any finding aborts).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "xr_oracle.h"

static uint64_t rng_state = 88172645463325252ull;
static double rnd(void) { /* xorshift64: seeded, reproducible */
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

/* m x m jittered lattice split into 2 (m-1)^2 triangles (+ a quad mesh view with -1 fill for the mixed case) */
static void make_mesh(int m, double jitter, double ox, double oy, double scale, double **xy_out, int64_t **faces_out,
                      int64_t *n_node, int64_t *n_face, int mm) {
    const int64_t nn = (int64_t)m * m, nf = 2 * (int64_t)(m - 1) * (m - 1);
    double *xy = malloc(sizeof(double) * 2 * (size_t)nn);
    int64_t *f = malloc(sizeof(int64_t) * (size_t)nf * (size_t)mm);
    const double h = 1.0 / (m - 1);
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) {
            xy[2 * (j * m + i)] = ox + scale * (i * h + jitter * h * (rnd() - 0.5));
            xy[2 * (j * m + i) + 1] = oy + scale * (j * h + jitter * h * (rnd() - 0.5));
        }
    int64_t k = 0;
    for (int j = 0; j < m - 1; j++)
        for (int i = 0; i < m - 1; i++) {
            const int64_t a = j * m + i, b = a + 1, c = a + m + 1, d = a + m;
            int64_t t1[3] = {a, b, c}, t2[3] = {a, c, d};
            for (int q = 0; q < mm; q++) f[k * mm + q] = q < 3 ? t1[q] : -1;
            k++;
            for (int q = 0; q < mm; q++) f[k * mm + q] = q < 3 ? t2[2 - q] : -1; /* clockwise on purpose */
            k++;
        }
    *xy_out = xy; *faces_out = f; *n_node = nn; *n_face = nf;
}

int main(void) {
    double *sxy, *txy;
    int64_t *sf, *tf, sn, sF, tn, tF;
    make_mesh(23, 0.6, 0.0, 0.0, 1.0, &sxy, &sf, &sn, &sF, 4);      /* triangles stored with a fill column */
    make_mesh(17, 0.5, 0.13, -0.07, 0.9, &txy, &tf, &tn, &tF, 3);
    xo_tree *tree = xo_tree_create(sxy, sn, sf, sF, 4, -1);
    if (!tree) return 2;

    /* overlap: tree search + SAT + clip, two-phase */
    int64_t nnz = 0, ncand = 0;
    if (xo_intersect_faces_count(tree, txy, tn, tf, tF, 3, -1, 1, &nnz, &ncand) != 0) return 3;
    int64_t *q = malloc(sizeof(int64_t) * (size_t)(nnz + 1)), *s = malloc(sizeof(int64_t) * (size_t)(nnz + 1));
    double *a = malloc(sizeof(double) * (size_t)(nnz + 1));
    xo_intersect_faces_fill(tree, q, s, a);
    int64_t *indptr = malloc(sizeof(int64_t) * (size_t)(tF + 1));
    xo_to_csr_indptr(q, nnz, tF, indptr);
    double total = 0.0;
    for (int64_t i = 0; i < nnz; i++) total += a[i];

    /* brute force on a prefix of the query faces */
    {
        const int64_t cap = 40 * sF;
        int64_t *bq = malloc(sizeof(int64_t) * (size_t)cap), *bs = malloc(sizeof(int64_t) * (size_t)cap), bn = 0;
        double *ba = malloc(sizeof(double) * (size_t)cap);
        if (xo_intersect_faces_bruteforce(tree, txy, tn, tf, 40, 3, -1, cap, bq, bs, ba, &bn) != 0) return 4;
        free(bq); free(bs); free(ba);
    }

    /* every reducer through the apply loop, K = 3, with NaNs, zeros and negatives */
    const int64_t K = 3;
    double *src = malloc(sizeof(double) * (size_t)(K * sF)), *out = malloc(sizeof(double) * (size_t)(K * tF));
    for (int64_t i = 0; i < K * sF; i++) {
        const double r = rnd();
        src[i] = r < 0.05 ? NAN : (r < 0.1 ? 0.0 : 4.0 * r - 2.0);
    }
    const double ps[] = {0.0, 5.0, 50.0, 95.0, 100.0};
    for (int method = 0; method <= 9; method++)
        for (int ip = 0; ip < (method == XO_PERCENTILE ? 5 : 1); ip++)
            for (int par = 0; par < 2; par++)
                if (xo_regrid_csr(method, ps[ip], src, K, sF, a, s, indptr, tF, out, par) != 0) return 5;
    /* COO scatter */
    xo_regrid_coo(src, K, sF, q, s, nnz, tF, out);

    /* geometry */
    double *area = malloc(sizeof(double) * (size_t)sF), *cen = malloc(sizeof(double) * 2 * (size_t)sF);
    xo_area(sxy, sf, sF, 4, area);
    xo_centroids(sxy, sf, sF, 4, cen);

    /* locate + barycentric, points inside, outside and ON vertices / edges */
    const int64_t np = 4000;
    double *pts = malloc(sizeof(double) * 2 * (size_t)np);
    for (int64_t i = 0; i < np; i++) {
        pts[2 * i] = 1.4 * rnd() - 0.2;
        pts[2 * i + 1] = 1.4 * rnd() - 0.2;
    }
    for (int64_t i = 0; i < 200 && i < sn; i++) { pts[2 * i] = sxy[2 * i]; pts[2 * i + 1] = sxy[2 * i + 1]; }
    int64_t *face = malloc(sizeof(int64_t) * (size_t)np);
    double *w = malloc(sizeof(double) * (size_t)np * 4);
    xo_locate_points(tree, pts, np, -1.0, face);
    xo_locate_points(tree, pts, np, 1e-3, face);
    xo_barycentric(tree, pts, np, -1.0, face, w);

    /* network edges */
    const int64_t ne = 500;
    double *edges = malloc(sizeof(double) * 4 * (size_t)ne);
    for (int64_t i = 0; i < 4 * ne; i++) edges[i] = 1.3 * rnd() - 0.15;
    int64_t nfound = 0;
    if (xo_intersect_edges_count(tree, edges, ne, &nfound) != 0) return 6;
    int64_t *ei = malloc(sizeof(int64_t) * (size_t)(nfound + 1)), *fi = malloc(sizeof(int64_t) * (size_t)(nfound + 1));
    double *pieces = malloc(sizeof(double) * 4 * (size_t)(nfound + 1));
    xo_intersect_edges_fill(tree, ei, fi, pieces);

    printf("sanitize_driver ok: %lld pairs of %lld candidates, overlap area %.12g, %lld edge pieces\n", (long long)nnz,
           (long long)ncand, total, (long long)nfound);
    free(ei); free(fi); free(pieces); free(edges); free(face); free(w); free(pts); free(area); free(cen); free(src); free(out);
    free(indptr); free(q); free(s); free(a);
    xo_tree_destroy(tree);
    free(sxy); free(sf); free(txy); free(tf);
    return 0;
}
