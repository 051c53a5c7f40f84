/*
 * xr_oracle.c -- CPU ORACLE (test infrastructure only; see xr_oracle.h for the rules and
 * the parity status of each part).  Plain C11 + OpenMP, float64, no FMA contraction
 * (built with -ffp-contract=off so that the HIP kernels, built the same way, can be compared
 * bit-for-bit).
 *
 * Part 1 restates code that is IN /root/reference (xugrid 0.15.3), citing file:line.
 * Part 2 restates the published algorithm of numba_celltree 0.4.2 (absent; pixi.lock:298).
 */
#include "xr_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define XO_FILL (-1)
#define XO_MAXV 64 /* max vertices of a clipped polygon (numba_celltree MAX_N_VERTEX=32, x2) */

void xo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int xo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* =====================================================================================
 * Part 1a: reducers -- xugrid/regrid/reduce.py, xugrid/regrid/nanpercentile.py
 * ===================================================================================== */

/* reduce.py:16-27 */
static double r_mean(const double *v, const double *w, int64_t n) {
    double vsum = 0.0, wsum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) continue;
        vsum += w[i] * v[i];
        wsum += w[i];
    }
    if (wsum == 0) return NAN;
    return vsum / wsum;
}

/* reduce.py:30-42 */
static double r_harmonic_mean(const double *v, const double *w, int64_t n) {
    double v_agg = 0.0, w_sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i]) || v[i] == 0) continue;
        if (w[i] > 0) {
            w_sum += w[i];
            v_agg += w[i] / v[i];
        }
    }
    if (v_agg == 0 || w_sum == 0) return NAN;
    return w_sum / v_agg;
}

/* reduce.py:45-72 */
static double r_geometric_mean(const double *v, const double *w, int64_t n) {
    double v_agg = 0.0, w_sum = 0.0, normsum = 0.0;
    for (int64_t i = 0; i < n; i++) normsum += w[i];
    if (normsum == 0) return NAN;
    for (int64_t i = 0; i < n; i++) {
        double wi = w[i] / normsum;
        if (v[i] > 0 && wi > 0) {
            v_agg += wi * log(fabs(v[i]));
            w_sum += wi;
        } else if (v[i] < 0) {
            return NAN;
        }
    }
    if (w_sum == 0) return NAN;
    return exp((1.0 / w_sum) * v_agg);
}

/* reduce.py:75-87 */
static double r_sum(const double *v, const double *w, int64_t n) {
    double v_sum = 0.0, w_sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) continue;
        v_sum += v[i];
        w_sum += w[i];
    }
    if (w_sum == 0) return NAN;
    return v_sum;
}

/* reduce.py:90-106 */
static double r_minimum(const double *v, const double *w, int64_t n) {
    double v_min = INFINITY, w_max = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) continue;
        if (v[i] < v_min) v_min = v[i];
        if (w[i] > w_max) w_max = w[i];
    }
    if (w_max == 0.0) return NAN;
    return v_min;
}

/* reduce.py:109-123 */
static double r_maximum(const double *v, const double *w, int64_t n) {
    double v_max = -INFINITY, w_max = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) continue;
        if (v[i] > v_max) v_max = v[i];
        if (w[i] > w_max) w_max = w[i];
    }
    if (w_max == 0.0) return NAN;
    return v_max;
}

/* reduce.py:126-158 */
static double r_mode(const double *v, const double *w, double *accum, int64_t n) {
    for (int64_t i = 0; i < n; i++) accum[i] = w[i];
    int64_t w_sum = 0;
    double w_max = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) continue;
        if (w[i] > w_max) w_max = w[i];
        w_sum += 1;
        for (int64_t j = 0; j < i; j++) {
            if (v[j] == v[i]) {
                accum[j] += w[i];
                break;
            }
        }
    }
    if (w_sum == 0 || w_max == 0.0) return NAN;
    w_max = 0;
    double mode_value = v[0];
    for (int64_t i = 0; i < n; i++) {
        if (!isnan(v[i])) {
            if ((accum[i] > w_max) || (accum[i] == w_max && v[i] > mode_value)) {
                w_max = accum[i];
                mode_value = v[i];
            }
        }
    }
    return mode_value;
}

/* nanpercentile.py:19-27 */
static inline int nan_le(double a, double b) {
    if (isnan(a)) return 0;
    if (isnan(b)) return 1;
    return a < b;
}
#define SWAP(A, i, j) do { double _t = A[i]; A[i] = A[j]; A[j] = _t; } while (0)

/* nanpercentile.py:30-63 */
static int64_t q_partition(double *A, int64_t low, int64_t high) {
    int64_t mid = (low + high) >> 1;
    if (nan_le(A[mid], A[low])) SWAP(A, low, mid);
    if (nan_le(A[high], A[mid])) SWAP(A, high, mid);
    if (nan_le(A[mid], A[low])) SWAP(A, low, mid);
    double pivot = A[mid];
    SWAP(A, high, mid);
    int64_t i = low, j = high - 1;
    for (;;) {
        while (i < high && nan_le(A[i], pivot)) i++;
        while (j >= low && nan_le(pivot, A[j])) j--;
        if (i >= j) break;
        SWAP(A, i, j);
        i++;
        j--;
    }
    SWAP(A, i, high);
    return i;
}

/* nanpercentile.py:66-77 */
static double q_select(double *A, int64_t k, int64_t low, int64_t high) {
    int64_t i = q_partition(A, low, high);
    while (i != k) {
        if (i < k) {
            low = i + 1;
            i = q_partition(A, low, high);
        } else {
            high = i - 1;
            i = q_partition(A, low, high);
        }
    }
    return A[k];
}

/* nanpercentile.py:80-102 */
static void q_select_two(double *A, int64_t k, int64_t low, int64_t high, double *lo, double *hi) {
    for (;;) {
        int64_t i = q_partition(A, low, high);
        if (i < k) {
            low = i + 1;
        } else if (i > k + 1) {
            high = i - 1;
        } else if (i == k) {
            q_select(A, k + 1, i + 1, high);
            break;
        } else {
            q_select(A, k, low, i - 1);
            break;
        }
    }
    *lo = A[k];
    *hi = A[k + 1];
}

/* reduce.py:161-203 */
static double r_percentile(const double *v, const double *w, double *ws, int64_t nn, double p) {
    double w_max = 0.0;
    for (int64_t i = 0; i < nn; i++)
        if (w[i] > w_max) w_max = w[i];
    if (w_max == 0.0) return NAN;
    if (p == 0) return r_minimum(v, w, nn);
    if (p == 100) return r_maximum(v, w, nn);
    int64_t n = 0;
    for (int64_t i = 0; i < nn; i++)
        if (!isnan(v[i])) ws[n++] = v[i];
    if (n == 0) return NAN;
    if (n == 1) return ws[0];
    double rank = 1 + (double)(n - 1) * p / 100.0;
    double f = floor(rank);
    double m = rank - f;
    double lower, upper;
    q_select_two(ws, (int64_t)(f - 1), 0, n - 1, &lower, &upper);
    return lower * (1 - m) + upper * m;
}

/* reduce.py:206-222 */
static double r_first_order_conservative(const double *v, const double *w, int64_t n) {
    double v_agg = 0.0, w_sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (isnan(v[i])) continue;
        v_agg += v[i] * w[i];
        w_sum += w[i];
    }
    if (w_sum == 0) return NAN;
    return v_agg;
}

/* reduce.py:225-238 */
static double r_max_overlap(const double *v, const double *w, int64_t n) {
    double w_max = 0.0, v_max = -INFINITY;
    for (int64_t i = 0; i < n; i++) {
        if (!isnan(v[i])) {
            if ((w[i] > w_max) || (w[i] == w_max && v[i] > v_max)) {
                w_max = w[i];
                v_max = v[i];
            }
        }
    }
    if (w_max == 0.0) return NAN;
    return v_max;
}

double xo_reduce(int method, double p, const double *v, const double *w, double *ws, int64_t n) {
    switch (method) {
    case XO_MEAN: return r_mean(v, w, n);
    case XO_HARMONIC_MEAN: return r_harmonic_mean(v, w, n);
    case XO_GEOMETRIC_MEAN: return r_geometric_mean(v, w, n);
    case XO_SUM: return r_sum(v, w, n);
    case XO_MINIMUM: return r_minimum(v, w, n);
    case XO_MAXIMUM: return r_maximum(v, w, n);
    case XO_MODE: return r_mode(v, w, ws, n);
    case XO_PERCENTILE: return r_percentile(v, w, ws, n, p);
    case XO_FIRST_ORDER_CONSERVATIVE: return r_first_order_conservative(v, w, n);
    case XO_MAX_OVERLAP: return r_max_overlap(v, w, n);
    default: return NAN;
    }
}

/* =====================================================================================
 * Part 1b: apply kernels -- xugrid/regrid/regridder.py:34-69 and :400-409
 * ===================================================================================== */

int xo_regrid_csr(int method, double p, const double *source, int64_t K, int64_t S,
                  const double *data, const int64_t *indices, const int64_t *indptr,
                  int64_t T, double *out, int parallel_rows) {
    /* regridder.py:44 out = np.full((n_extra, size), nan) */
    for (int64_t i = 0; i < K * T; i++) out[i] = NAN;
    /* regridder.py:48 n_work = np.diff(A.indptr).max() */
    int64_t n_work = 0;
    for (int64_t t = 0; t < T; t++)
        if (indptr[t + 1] - indptr[t] > n_work) n_work = indptr[t + 1] - indptr[t];
    if (n_work == 0) return 0;
    if (!parallel_rows) {
        /* regridder.py:50 for extra_index in numba.prange(n_extra) */
#pragma omp parallel
        {
            double *ws = (double *)malloc(sizeof(double) * 2 * (size_t)n_work);
#pragma omp for schedule(static)
            for (int64_t k = 0; k < K; k++) {
                const double *src = source + k * S;
                for (int64_t t = 0; t < T; t++) {
                    int64_t s = indptr[t], e = indptr[t + 1], n = e - s;
                    for (int64_t i = 0; i < n; i++) ws[i] = src[indices[s + i]];
                    if (n > 0) out[k * T + t] = xo_reduce(method, p, ws, data + s, ws + n_work, n);
                }
            }
            free(ws);
        }
    } else {
#pragma omp parallel
        {
            double *ws = (double *)malloc(sizeof(double) * 2 * (size_t)n_work);
#pragma omp for schedule(static) collapse(2)
            for (int64_t k = 0; k < K; k++) {
                for (int64_t t = 0; t < T; t++) {
                    const double *src = source + k * S;
                    int64_t s = indptr[t], e = indptr[t + 1], n = e - s;
                    for (int64_t i = 0; i < n; i++) ws[i] = src[indices[s + i]];
                    if (n > 0) out[k * T + t] = xo_reduce(method, p, ws, data + s, ws + n_work, n);
                }
            }
            free(ws);
        }
    }
    return 0;
}

/* regridder.py:400-409 */
int xo_regrid_coo(const double *source, int64_t K, int64_t S, const int64_t *row,
                  const int64_t *col, int64_t nnz, int64_t T, double *out) {
    for (int64_t i = 0; i < K * T; i++) out[i] = NAN;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < K; k++) {
        const double *src = source + k * S;
        for (int64_t i = 0; i < nnz; i++) out[k * T + row[i]] = src[col[i]];
    }
    return 0;
}

/* core/sparse.py:61-78: indptr = [0, cumsum(bincount(row, minlength=n))] */
int xo_to_csr_indptr(const int64_t *row, int64_t nnz, int64_t n, int64_t *indptr) {
    for (int64_t i = 0; i <= n; i++) indptr[i] = 0;
    for (int64_t i = 0; i < nnz; i++) {
        if (row[i] < 0 || row[i] >= n) return -1;
        indptr[row[i] + 1] += 1;
    }
    for (int64_t i = 0; i < n; i++) indptr[i + 1] += indptr[i];
    return 0;
}

/* =====================================================================================
 * Part 1c: mesh geometry -- xugrid/ugrid/connectivity.py
 * ===================================================================================== */

/* connectivity.py:372-382 close_polygons + :615-633 area:
 * closed polygon = nodes, with fill slots and the closing slot replaced by node 0;
 * a = c[:-1]-c0, b = c[1:]-c0, area = 0.5*abs(sum(cross(a,b))). */
int xo_area(const double *xy, const int64_t *faces, int64_t n_face, int64_t m, double *area) {
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < n_face; f++) {
        const int64_t *face = faces + f * m;
        double x0 = xy[2 * face[0]], y0 = xy[2 * face[0] + 1];
        double det = 0.0;
        for (int64_t i = 0; i < m; i++) {
            int64_t ia = face[i] == XO_FILL ? face[0] : face[i];
            int64_t ib = (i + 1 < m) ? (face[i + 1] == XO_FILL ? face[0] : face[i + 1]) : face[0];
            double ax = xy[2 * ia] - x0, ay = xy[2 * ia + 1] - y0;
            double bx = xy[2 * ib] - x0, by = xy[2 * ib + 1] - y0;
            det += ax * by - ay * bx;
        }
        area[f] = 0.5 * fabs(det);
    }
    return 0;
}

/* connectivity.py:636-664 */
int xo_centroids(const double *xy, const int64_t *faces, int64_t n_face, int64_t m, double *cxy) {
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < n_face; f++) {
        const int64_t *face = faces + f * m;
        if (m == 3) {
            /* :641-647 triangles: nanmean of the 3 vertices (numpy pairwise sum of 3 = left to right) */
            double sx = 0.0, sy = 0.0;
            for (int i = 0; i < 3; i++) {
                sx += xy[2 * face[i]];
                sy += xy[2 * face[i] + 1];
            }
            cxy[2 * f] = sx / 3.0;
            cxy[2 * f + 1] = sy / 3.0;
        } else {
            double x0 = xy[2 * face[0]], y0 = xy[2 * face[0] + 1];
            double det = 0.0, sx = 0.0, sy = 0.0;
            for (int64_t i = 0; i < m; i++) {
                int64_t ia = face[i] == XO_FILL ? face[0] : face[i];
                int64_t ib = (i + 1 < m) ? (face[i + 1] == XO_FILL ? face[0] : face[i + 1]) : face[0];
                double ax = xy[2 * ia] - x0, ay = xy[2 * ia + 1] - y0;
                double bx = xy[2 * ib] - x0, by = xy[2 * ib + 1] - y0;
                double d = ax * by - ay * bx;
                det += d;
                sx += (ax + bx) * d;
                sy += (ay + by) * d;
            }
            double aw = 1.0 / (3.0 * det);
            cxy[2 * f] = aw * sx + x0;
            cxy[2 * f + 1] = aw * sy + y0;
        }
    }
    return 0;
}

/* regrid/unstructured.py:17-57 */
int xo_replace_interpolated_weights(const double *vertices, const int64_t *faces, int64_t m,
                                    const int64_t *face_index, double *weights, int64_t n,
                                    const int64_t *node_to_node_map, int64_t node_index_threshold) {
    for (int64_t i = 0; i < n; i++) {
        if (face_index[i] < 0) continue; /* weights are all zero there; faces[-1] row never matters */
        const int64_t *face = faces + face_index[i] * m;
        double *wr = weights + i * m;
        for (int64_t j = 0; j < m; j++) {
            int64_t pidx = face[j];
            double w = wr[j];
            if (pidx < node_index_threshold || w <= 0) continue;
            int64_t index = pidx - node_index_threshold;
            int64_t q = node_to_node_map[2 * index], r = node_to_node_map[2 * index + 1];
            double px = vertices[2 * pidx], py = vertices[2 * pidx + 1];
            double qx = vertices[2 * q], qy = vertices[2 * q + 1];
            double rx = vertices[2 * r], ry = vertices[2 * r + 1];
            double p_q = sqrt((qx - px) * (qx - px) + (qy - py) * (qy - py));
            double p_r = sqrt((rx - px) * (rx - px) + (ry - py) * (ry - py));
            double total = p_q + p_r;
            double weight_q = (p_r / total) * w;
            double weight_r = (p_q / total) * w;
            wr[j] = 0.0;
            for (int64_t jj = 0; jj < m; jj++) {
                if (face[jj] == q) wr[jj] += weight_q;
                if (face[jj] == r) wr[jj] += weight_r;
            }
        }
    }
    return 0;
}

/* =====================================================================================
 * Part 2: numba_celltree 0.4.2 restatement (external package; see header)
 * ===================================================================================== */

typedef struct { double x, y; } P2;
typedef struct { double xmin, xmax, ymin, ymax; } Box;

typedef struct {
    int64_t child; /* left child, right = child + 1; -1 = leaf */
    double Lmax, Rmin;
    int64_t ptr, size;
    int dim;
} Node;

struct xo_tree {
    int64_t n_node, n_face, m;
    double *xy;     /* n_node x 2 */
    int64_t *faces; /* n_face x m, CCW, -1 fill */
    int32_t *len;   /* n_face */
    Box *bb;        /* n_face */
    int64_t *bb_indices;
    Node *nodes;
    int64_t n_tnodes, cap_tnodes;
    /* last intersect_faces result */
    int64_t res_n;
    int64_t *res_q, *res_t;
    double *res_a;
    /* last intersect_edges result */
    int64_t edg_n;
    int64_t *edg_e, *edg_f;
    double *edg_xy; /* edg_n x 2 x 2 */
};

/* polygon_length: a minimal polygon is a triangle; stops at the first fill value */
static inline int poly_len(const int64_t *face, int64_t m) {
    for (int64_t i = 3; i < m; i++)
        if (face[i] == XO_FILL) return (int)i;
    return (int)m;
}

static inline double cross2(double ux, double uy, double vx, double vy) { return ux * vy - uy * vx; }

/* counter_clockwise: the first non-collinear vertex triple decides; clockwise faces are
 * reversed in place (vertex order face[0..len) reversed). */
static void make_ccw(const double *xy, int64_t *face, int len) {
    for (int i = 0; i < len; i++) {
        int64_t ia = face[(i + len - 2) % len], ib = face[(i + len - 1) % len], ic = face[i];
        double ux = xy[2 * ib] - xy[2 * ia], uy = xy[2 * ib + 1] - xy[2 * ia + 1];
        double vx = xy[2 * ic] - xy[2 * ia], vy = xy[2 * ic + 1] - xy[2 * ia + 1];
        double prod = cross2(ux, uy, vx, vy);
        if (prod == 0) continue;
        if (prod < 0) {
            for (int a = 0, b = len - 1; a < b; a++, b--) {
                int64_t t = face[a];
                face[a] = face[b];
                face[b] = t;
            }
        }
        return;
    }
}

static void normalise_faces(const double *xy, const int64_t *faces_in, int64_t n_face, int64_t m,
                            int64_t fill, int64_t *faces, int32_t *len, Box *bb) {
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < n_face; f++) {
        int64_t *face = faces + f * m;
        for (int64_t j = 0; j < m; j++) {
            int64_t v = faces_in[f * m + j];
            face[j] = (v == fill) ? XO_FILL : v;
        }
        int n = poly_len(face, m);
        len[f] = n;
        make_ccw(xy, face, n);
        Box b = {INFINITY, -INFINITY, INFINITY, -INFINITY};
        for (int j = 0; j < n; j++) {
            double x = xy[2 * face[j]], y = xy[2 * face[j] + 1];
            if (x < b.xmin) b.xmin = x;
            if (x > b.xmax) b.xmax = x;
            if (y < b.ymin) b.ymin = y;
            if (y > b.ymax) b.ymax = y;
        }
        bb[f] = b;
    }
}

static inline int boxes_intersect(Box a, Box b) {
    return a.xmin < b.xmax && b.xmin < a.xmax && a.ymin < b.ymax && b.ymin < a.ymax;
}

/* ---- cell tree build (Garth & Joy 2010: bucketed split, n_buckets=4, cells_per_leaf=2) ---- */
#define N_BUCKETS 4
#define CELLS_PER_LEAF 2

static inline double bb_lo(const Box *b, int dim) { return dim ? b->ymin : b->xmin; }
static inline double bb_hi(const Box *b, int dim) { return dim ? b->ymax : b->xmax; }

static int64_t tree_new_nodes(xo_tree *t) {
    if (t->n_tnodes + 2 > t->cap_tnodes) {
        t->cap_tnodes = t->cap_tnodes * 2 + 16;
        t->nodes = (Node *)realloc(t->nodes, sizeof(Node) * (size_t)t->cap_tnodes);
    }
    int64_t i = t->n_tnodes;
    t->n_tnodes += 2;
    return i;
}

static void tree_build(xo_tree *t) {
    int64_t n = t->n_face;
    t->bb_indices = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) t->bb_indices[i] = i;
    t->cap_tnodes = 2 * n + 16;
    t->nodes = (Node *)malloc(sizeof(Node) * (size_t)t->cap_tnodes);
    t->n_tnodes = 1;
    Node root = {-1, 0, 0, 0, n, 0};
    t->nodes[0] = root;
    int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * n + 64));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        int64_t ni = stack[--sp];
        Node nd = t->nodes[ni];
        if (nd.size <= CELLS_PER_LEAF) continue;
        int64_t *idx = t->bb_indices + nd.ptr;
        int dim = nd.dim;
        double lo = 0, hi = 0;
        int ok = 0;
        for (int attempt = 0; attempt < 2 && !ok; attempt++) {
            lo = INFINITY;
            hi = -INFINITY;
            for (int64_t i = 0; i < nd.size; i++) {
                const Box *b = &t->bb[idx[i]];
                if (bb_lo(b, dim) < lo) lo = bb_lo(b, dim);
                if (bb_hi(b, dim) > hi) hi = bb_hi(b, dim);
            }
            if (hi > lo) ok = 1;
            else dim = !dim;
        }
        if (!ok) continue; /* all boxes degenerate to one point: keep as (large) leaf */
        int64_t count[N_BUCKETS] = {0};
        double bmax[N_BUCKETS], bmin[N_BUCKETS];
        for (int b = 0; b < N_BUCKETS; b++) {
            bmax[b] = -INFINITY;
            bmin[b] = INFINITY;
        }
        double scale = N_BUCKETS / (hi - lo);
        for (int64_t i = 0; i < nd.size; i++) {
            const Box *b = &t->bb[idx[i]];
            double c = 0.5 * (bb_lo(b, dim) + bb_hi(b, dim));
            int k = (int)((c - lo) * scale);
            if (k < 0) k = 0;
            if (k >= N_BUCKETS) k = N_BUCKETS - 1;
            count[k]++;
            if (bb_hi(b, dim) > bmax[k]) bmax[k] = bb_hi(b, dim);
            if (bb_lo(b, dim) < bmin[k]) bmin[k] = bb_lo(b, dim);
        }
        /* choose the split plane of minimum cost = left extent * n_left + right extent * n_right */
        int best = -1;
        double best_cost = INFINITY, best_L = 0, best_R = 0;
        for (int k = 1; k < N_BUCKETS; k++) {
            int64_t nl = 0, nr = 0;
            double L = -INFINITY, R = INFINITY;
            for (int b = 0; b < k; b++) {
                nl += count[b];
                if (bmax[b] > L) L = bmax[b];
            }
            for (int b = k; b < N_BUCKETS; b++) {
                nr += count[b];
                if (bmin[b] < R) R = bmin[b];
            }
            if (nl == 0 || nr == 0) continue;
            double cost = (L - lo) * (double)nl + (hi - R) * (double)nr;
            if (cost < best_cost) {
                best_cost = cost;
                best = k;
                best_L = L;
                best_R = R;
            }
        }
        int64_t n_left;
        if (best < 0) {
            /* all centroids in one bucket: split the index range evenly */
            n_left = nd.size / 2;
            best_L = -INFINITY;
            best_R = INFINITY;
            for (int64_t i = 0; i < nd.size; i++) {
                const Box *b = &t->bb[idx[i]];
                if (i < n_left) {
                    if (bb_hi(b, dim) > best_L) best_L = bb_hi(b, dim);
                } else {
                    if (bb_lo(b, dim) < best_R) best_R = bb_lo(b, dim);
                }
            }
        } else {
            /* stable partition by bucket < best */
            int64_t a = 0, c = 0;
            for (int64_t i = 0; i < nd.size; i++) {
                const Box *b = &t->bb[idx[i]];
                double cc = 0.5 * (bb_lo(b, dim) + bb_hi(b, dim));
                int k = (int)((cc - lo) * scale);
                if (k < 0) k = 0;
                if (k >= N_BUCKETS) k = N_BUCKETS - 1;
                if (k < best) idx[a++] = idx[i];
                else tmp[c++] = idx[i];
            }
            memcpy(idx + a, tmp, sizeof(int64_t) * (size_t)c);
            n_left = a;
        }
        int64_t ch = tree_new_nodes(t);
        Node left = {-1, 0, 0, nd.ptr, n_left, !dim};
        Node right = {-1, 0, 0, nd.ptr + n_left, nd.size - n_left, !dim};
        t->nodes[ch] = left;
        t->nodes[ch + 1] = right;
        t->nodes[ni].child = ch;
        t->nodes[ni].Lmax = best_L;
        t->nodes[ni].Rmin = best_R;
        t->nodes[ni].dim = dim;
        stack[sp++] = ch;
        stack[sp++] = ch + 1;
    }
    free(stack);
    free(tmp);
}

xo_tree *xo_tree_create(const double *node_xy, int64_t n_node, const int64_t *faces,
                        int64_t n_face, int64_t m, int64_t fill_value) {
    xo_tree *t = (xo_tree *)calloc(1, sizeof(xo_tree));
    t->n_node = n_node;
    t->n_face = n_face;
    t->m = m;
    t->xy = (double *)malloc(sizeof(double) * 2 * (size_t)(n_node > 0 ? n_node : 1));
    memcpy(t->xy, node_xy, sizeof(double) * 2 * (size_t)n_node);
    size_t nf = (size_t)(n_face > 0 ? n_face : 1);
    t->faces = (int64_t *)malloc(sizeof(int64_t) * nf * (size_t)m);
    t->len = (int32_t *)malloc(sizeof(int32_t) * nf);
    t->bb = (Box *)malloc(sizeof(Box) * nf);
    normalise_faces(t->xy, faces, n_face, m, fill_value, t->faces, t->len, t->bb);
    tree_build(t);
    return t;
}

void xo_tree_destroy(xo_tree *t) {
    if (!t) return;
    free(t->xy); free(t->faces); free(t->len); free(t->bb); free(t->bb_indices); free(t->nodes);
    free(t->res_q); free(t->res_t); free(t->res_a);
    free(t->edg_e); free(t->edg_f); free(t->edg_xy);
    free(t);
}

int64_t xo_tree_n_face(const xo_tree *t) { return t->n_face; }

int xo_tree_faces(const xo_tree *t, int64_t *out) {
    memcpy(out, t->faces, sizeof(int64_t) * (size_t)t->n_face * (size_t)t->m);
    return 0;
}

/* locate_boxes for one query box: explicit-stack traversal; emits tree face ids. */
static inline int boxes_touch(Box a, Box b) { /* closed boxes: a segment's box may have zero width */
    return a.xmin <= b.xmax && b.xmin <= a.xmax && a.ymin <= b.ymax && b.ymin <= a.ymax;
}

static int64_t tree_query_box_ex(const xo_tree *t, Box q, int64_t *out, int closed);
static int64_t tree_query_box(const xo_tree *t, Box q, int64_t *out /* may be NULL: count only */) {
    return tree_query_box_ex(t, q, out, 0);
}

static int64_t tree_query_box_ex(const xo_tree *t, Box q, int64_t *out, int closed) {
    int64_t stack[128];
    int sp = 0;
    int64_t cnt = 0;
    if (t->n_face == 0) return 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const Node *nd = &t->nodes[stack[--sp]];
        if (nd->child == -1) {
            for (int64_t i = nd->ptr; i < nd->ptr + nd->size; i++) {
                int64_t f = t->bb_indices[i];
                if (closed ? boxes_touch(q, t->bb[f]) : boxes_intersect(q, t->bb[f])) {
                    if (out) out[cnt] = f;
                    cnt++;
                }
            }
            continue;
        }
        double qlo = nd->dim ? q.ymin : q.xmin, qhi = nd->dim ? q.ymax : q.xmax;
        int left = qlo <= nd->Lmax, right = qhi >= nd->Rmin;
        if (left) stack[sp++] = nd->child;
        if (right) stack[sp++] = nd->child + 1;
    }
    return cnt;
}

static int cmp_i64(const void *x, const void *y) {
    const int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
    return (a > b) - (a < b);
}

static void isort64(int64_t *a, int64_t n) {
    if (n > 64) { /* hull slivers collect thousands of candidates: insertion sort would dominate the whole search */
        qsort(a, (size_t)n, sizeof(int64_t), cmp_i64);
        return;
    }
    for (int64_t i = 1; i < n; i++) {
        int64_t v = a[i], j = i - 1;
        while (j >= 0 && a[j] > v) {
            a[j + 1] = a[j];
            j--;
        }
        a[j + 1] = v;
    }
}

/* ---- separating axis test on two convex polygons (touching counts as intersecting) ---- */
static int sat_intersect(const P2 *a, int na, const P2 *b, int nb) {
    for (int pass = 0; pass < 2; pass++) {
        const P2 *p = pass ? b : a;
        int np_ = pass ? nb : na;
        for (int i = 0; i < np_; i++) {
            P2 v0 = p[i], v1 = p[(i + 1) % np_];
            double nx = -(v1.y - v0.y), ny = v1.x - v0.x; /* edge normal */
            if (nx == 0 && ny == 0) continue;
            double amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
            for (int j = 0; j < na; j++) {
                double d = nx * a[j].x + ny * a[j].y;
                if (d < amin) amin = d;
                if (d > amax) amax = d;
            }
            for (int j = 0; j < nb; j++) {
                double d = nx * b[j].x + ny * b[j].y;
                if (d < bmin) bmin = d;
                if (d > bmax) bmax = d;
            }
            if (amax < bmin || bmax < amin) return 0;
        }
    }
    return 1;
}

/* ---- Sutherland-Hodgman clip of `subject` (query polygon) by convex CCW `clipper` (tree
 * polygon) followed by the fan area of the clipped polygon.  THE reference arithmetic for
 * the HIP kernels (xugrid_amd/csrc/xr_clip_tri.h and the clip kernels of xr_overlap.hip mirror it operation for operation). ---- */
static inline int sh_inside(P2 p, P2 r, P2 U) { return U.x * (p.y - r.y) > U.y * (p.x - r.x); }

static inline int sh_intersection(P2 a, P2 V, P2 r, P2 N, P2 *out) {
    P2 W = {r.x - a.x, r.y - a.y};
    double nw = N.x * W.x + N.y * W.y;
    double nv = N.x * V.x + N.y * V.y;
    if (nv != 0) {
        double tt = nw / nv;
        out->x = a.x + tt * V.x;
        out->y = a.y + tt * V.y;
        return 1;
    }
    return 0;
}

static double sh_polygon_area(const P2 *poly, int n) {
    double area = 0.0;
    P2 a = poly[0];
    P2 U = {poly[1].x - a.x, poly[1].y - a.y};
    for (int i = 2; i < n; i++) {
        P2 c = poly[i];
        P2 V = {a.x - c.x, a.y - c.y};
        area += fabs(U.x * V.y - U.y * V.x);
        U = V;
    }
    return 0.5 * area;
}

static double clip_polygons(const P2 *polygon, int n_poly, const P2 *clipper, int n_clip) {
    P2 subject[XO_MAXV], output[XO_MAXV];
    int n_output = n_poly;
    for (int i = 0; i < n_poly; i++) output[i] = polygon[i];
    P2 r = clipper[n_clip - 1];
    for (int i = 0; i < n_clip; i++) {
        P2 s = clipper[i];
        P2 U = {s.x - r.x, s.y - r.y};
        if (U.x == 0 && U.y == 0) continue; /* zero-length clip edge (repeated vertex) */
        P2 N = {-U.y, U.x};
        int length = n_output;
        for (int j = 0; j < length; j++) subject[j] = output[j];
        n_output = 0;
        P2 a = subject[length - 1];
        int a_inside = sh_inside(a, r, U);
        for (int j = 0; j < length; j++) {
            P2 b = subject[j];
            P2 V = {b.x - a.x, b.y - a.y};
            if (V.x == 0 && V.y == 0) continue; /* zero-length subject edge */
            int b_inside = sh_inside(b, r, U);
            if (b_inside) {
                if (!a_inside) {
                    P2 pt;
                    if (sh_intersection(a, V, r, N, &pt)) output[n_output++] = pt;
                }
                output[n_output++] = b;
            } else if (a_inside) {
                P2 pt;
                if (sh_intersection(a, V, r, N, &pt)) {
                    output[n_output++] = pt;
                } else { /* parallel: floating point inconsistency, keep b */
                    b_inside = 1;
                    output[n_output++] = b;
                }
            }
            a = b;
            a_inside = b_inside;
        }
        if (n_output < 3) return 0.0;
        r = s;
    }
    return sh_polygon_area(output, n_output);
}

double xo_clip_area(const double *subject, int64_t ns, const double *clipper, int64_t nc) {
    P2 a[XO_MAXV / 2], b[XO_MAXV / 2];
    if (ns > XO_MAXV / 2 || nc > XO_MAXV / 2) return NAN;
    for (int64_t i = 0; i < ns; i++) { a[i].x = subject[2 * i]; a[i].y = subject[2 * i + 1]; }
    for (int64_t i = 0; i < nc; i++) { b[i].x = clipper[2 * i]; b[i].y = clipper[2 * i + 1]; }
    return clip_polygons(a, (int)ns, b, (int)nc);
}

static inline int load_poly(const double *xy, const int64_t *face, int n, P2 *out) {
    for (int i = 0; i < n; i++) {
        out[i].x = xy[2 * face[i]];
        out[i].y = xy[2 * face[i] + 1];
    }
    return n;
}

typedef struct {
    int64_t n_face, m;
    int64_t *faces;
    int32_t *len;
    Box *bb;
} QMesh;

static int qmesh_init(QMesh *q, const double *xy, const int64_t *faces, int64_t n_face, int64_t m,
                      int64_t fill) {
    size_t nf = (size_t)(n_face > 0 ? n_face : 1);
    q->n_face = n_face;
    q->m = m;
    q->faces = (int64_t *)malloc(sizeof(int64_t) * nf * (size_t)m);
    q->len = (int32_t *)malloc(sizeof(int32_t) * nf);
    q->bb = (Box *)malloc(sizeof(Box) * nf);
    normalise_faces(xy, faces, n_face, m, fill, q->faces, q->len, q->bb);
    return 0;
}
static void qmesh_free(QMesh *q) { free(q->faces); free(q->len); free(q->bb); }

static void store_result(xo_tree *t, int64_t n, int64_t *q, int64_t *s, double *a) {
    free(t->res_q); free(t->res_t); free(t->res_a);
    t->res_n = n; t->res_q = q; t->res_t = s; t->res_a = a;
}

int xo_intersect_faces_count(xo_tree *t, const double *q_xy, int64_t q_n_node,
                             const int64_t *q_faces, int64_t q_n_face, int64_t q_m,
                             int64_t q_fill, int use_sat, int64_t *nnz, int64_t *n_candidates) {
    (void)q_n_node;
    if (t->m > XO_MAXV / 2 || q_m > XO_MAXV / 2) return -2;
    QMesh q;
    const int timing = getenv("XO_TIMING") != NULL;
    double tt[8];
    tt[0] = omp_get_wtime();
    qmesh_init(&q, q_xy, q_faces, q_n_face, q_m, q_fill);
    tt[1] = omp_get_wtime();
    /* pass 1: count bbox candidates per query face (parallel over queries) */
    int64_t *off = (int64_t *)calloc((size_t)q_n_face + 1, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < q_n_face; i++) off[i + 1] = tree_query_box(t, q.bb[i], NULL);
    for (int64_t i = 0; i < q_n_face; i++) off[i + 1] += off[i];
    int64_t C = off[q_n_face];
    if (n_candidates) *n_candidates = C;
    tt[2] = omp_get_wtime();
    int64_t *cq = (int64_t *)malloc(sizeof(int64_t) * (size_t)(C > 0 ? C : 1));
    int64_t *cs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(C > 0 ? C : 1));
    double *ca = (double *)malloc(sizeof(double) * (size_t)(C > 0 ? C : 1));
    /* pass 2: fill, each query's candidates sorted by tree face index (canonical order) */
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < q_n_face; i++) {
        int64_t n = tree_query_box(t, q.bb[i], cs + off[i]);
        isort64(cs + off[i], n);
        for (int64_t j = 0; j < n; j++) cq[off[i] + j] = i;
    }
    tt[3] = omp_get_wtime();
    /* pass 3: SAT filter + clip + area, parallel over candidate pairs */
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t c = 0; c < C; c++) {
        P2 a[XO_MAXV / 2], b[XO_MAXV / 2];
        int na = load_poly(q_xy, q.faces + cq[c] * q_m, q.len[cq[c]], a);
        int nb = load_poly(t->xy, t->faces + cs[c] * t->m, t->len[cs[c]], b);
        if (use_sat && !sat_intersect(a, na, b, nb)) {
            ca[c] = 0.0;
            continue;
        }
        ca[c] = clip_polygons(a, na, b, nb);
    }
    tt[4] = omp_get_wtime();
    /* pass 4: keep area > 0 */
    int64_t P = 0;
    for (int64_t c = 0; c < C; c++)
        if (ca[c] > 0) P++;
    int64_t *rq = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P > 0 ? P : 1));
    int64_t *rs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P > 0 ? P : 1));
    double *ra = (double *)malloc(sizeof(double) * (size_t)(P > 0 ? P : 1));
    int64_t k = 0;
    for (int64_t c = 0; c < C; c++) {
        if (ca[c] > 0) {
            rq[k] = cq[c];
            rs[k] = cs[c];
            ra[k] = ca[c];
            k++;
        }
    }
    free(cq); free(cs); free(ca); free(off);
    qmesh_free(&q);
    store_result(t, P, rq, rs, ra);
    *nnz = P;
    tt[5] = omp_get_wtime();
    if (timing)
        fprintf(stderr, "[xo] intersect_faces %d threads: normalise %.3f count %.3f fill %.3f clip %.3f compact %.3f s\n",
                omp_get_max_threads(), tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4]);
    return 0;
}

int xo_intersect_faces_fill(xo_tree *t, int64_t *query_idx, int64_t *tree_idx, double *area) {
    memcpy(query_idx, t->res_q, sizeof(int64_t) * (size_t)t->res_n);
    memcpy(tree_idx, t->res_t, sizeof(int64_t) * (size_t)t->res_n);
    memcpy(area, t->res_a, sizeof(double) * (size_t)t->res_n);
    return 0;
}

int xo_intersect_faces_bruteforce(xo_tree *t, const double *q_xy, int64_t q_n_node,
                                  const int64_t *q_faces, int64_t q_n_face, int64_t q_m,
                                  int64_t q_fill, int64_t cap, int64_t *query_idx,
                                  int64_t *tree_idx, double *area, int64_t *nnz) {
    (void)q_n_node;
    if (t->m > XO_MAXV / 2 || q_m > XO_MAXV / 2) return -2;
    QMesh q;
    qmesh_init(&q, q_xy, q_faces, q_n_face, q_m, q_fill);
    int64_t k = 0;
    int rc = 0;
    for (int64_t i = 0; i < q_n_face && rc == 0; i++) {
        P2 a[XO_MAXV / 2], b[XO_MAXV / 2];
        int na = load_poly(q_xy, q.faces + i * q_m, q.len[i], a);
        for (int64_t s = 0; s < t->n_face; s++) {
            int nb = load_poly(t->xy, t->faces + s * t->m, t->len[s], b);
            double ar = clip_polygons(a, na, b, nb);
            if (ar > 0) {
                if (k >= cap) { rc = -3; break; }
                query_idx[k] = i;
                tree_idx[k] = s;
                area[k] = ar;
                k++;
            }
        }
    }
    qmesh_free(&q);
    *nnz = k;
    return rc;
}

/* ---- point location ---- */
double xo_default_tolerance(const xo_tree *t) {
    /* ugridbase.py:1165-1170: 1e-12 x the maximum bounding-box diagonal */
    double dmax = 0.0;
    for (int64_t f = 0; f < t->n_face; f++) {
        double dx = t->bb[f].xmax - t->bb[f].xmin, dy = t->bb[f].ymax - t->bb[f].ymin;
        double d = sqrt(dx * dx + dy * dy);
        if (d > dmax) dmax = d;
    }
    return 1e-12 * dmax;
}

/* point_in_polygon_or_on_edge: crossing-number test, plus "on edge" when the point is
 * strictly within `tol` (distance) of an edge's line, |cross(p-v0, v1-v0)| < tol*|v1-v0|, and its
 * projection falls on the segment (0 <= t <= 1).  Pinned by tests/test_ugrid2d.py:724-730 (a point
 * 0.01 outside is found with tolerance 0.011) and :771-791 (a point 0.01 beyond a corner along
 * the edge direction, or exactly tol away, is NOT found with tolerance 0.01). */
/* ---- CellTree2d.intersect_edges(edge_coords) -- called from UnstructuredGrid2d.intersection_length,
 * xugrid/regrid/unstructured.py:203-215 (NetworkGridder weights, regrid/gridder.py:66-73).
 * numba_celltree restatement: Cyrus-Beck parametric clip of the segment a + t (b - a), t in [0, 1], against the
 * half-planes of a CONVEX, CCW polygon.  For the polygon edge v0 -> v1 with inward normal n = (-(v1-v0).y, (v1-v0).x):
 * inside iff n.(a + t s - v0) >= 0, i.e. t den >= num with den = n.s, num = n.(v0 - a); den > 0 raises t0,
 * den < 0 lowers t1, den == 0 (parallel) rejects iff num > 0 (strictly outside).  A pair is an intersection iff
 * t0 < t1 and the clipped piece has positive length: touching a corner or an edge from outside gives none
 * (pinned by tests/test_regrid/test_network_gridder.py: 8 pairs, 11 empty cells, the listed cell means). */
static int cyrus_beck_clip(const P2 *poly, int n, P2 a, P2 b, P2 *c, P2 *d) {
    double sx = b.x - a.x, sy = b.y - a.y;
    double t0 = 0.0, t1 = 1.0;
    for (int i = 0; i < n; i++) {
        P2 v0 = poly[i], v1 = poly[(i + 1 < n) ? i + 1 : 0];
        double wx = v1.x - v0.x, wy = v1.y - v0.y;
        if (wx == 0.0 && wy == 0.0) continue;
        double nx = -wy, ny = wx;
        double den = nx * sx + ny * sy;
        double num = nx * (v0.x - a.x) + ny * (v0.y - a.y);
        if (den == 0.0) {
            if (num > 0.0) return 0;
            continue;
        }
        double t = num / den;
        if (den > 0.0) {
            if (t > t0) t0 = t;
        } else {
            if (t < t1) t1 = t;
        }
    }
    if (!(t0 < t1)) return 0;
    c->x = a.x + t0 * sx;
    c->y = a.y + t0 * sy;
    d->x = a.x + t1 * sx;
    d->y = a.y + t1 * sy;
    return 1;
}

/* length of an intersection as the reference computes it: norm(diff(intersections)), unstructured.py:212 */
static double piece_length(P2 c, P2 d) {
    double ex = d.x - c.x, ey = d.y - c.y;
    return sqrt(ex * ex + ey * ey);
}

/* edge_xy: (n_edge, 2, 2).  Result ordered by (edge, face); stored in the tree until _fill copies it out. */
int xo_intersect_edges_count(xo_tree *t, const double *edge_xy, int64_t n_edge, int64_t *n_found) {
    if (t->m > XO_MAXV / 2) return -2;
    int64_t *off = (int64_t *)calloc((size_t)n_edge + 1, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t e = 0; e < n_edge; e++) {
        const double *p = edge_xy + 4 * e;
        if (p[0] != p[0] || p[1] != p[1] || p[2] != p[2] || p[3] != p[3]) continue; /* NaN */
        Box q = {fmin(p[0], p[2]), fmax(p[0], p[2]), fmin(p[1], p[3]), fmax(p[1], p[3])};
        off[e + 1] = tree_query_box_ex(t, q, NULL, 1);
    }
    for (int64_t e = 0; e < n_edge; e++) off[e + 1] += off[e];
    int64_t C = off[n_edge];
    int64_t *cf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(C > 0 ? C : 1));
    double *cxy = (double *)malloc(sizeof(double) * 4 * (size_t)(C > 0 ? C : 1));
    char *keep = (char *)calloc((size_t)(C > 0 ? C : 1), 1);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t e = 0; e < n_edge; e++) {
        if (off[e + 1] == off[e]) continue;
        const double *p = edge_xy + 4 * e;
        Box q = {fmin(p[0], p[2]), fmax(p[0], p[2]), fmin(p[1], p[3]), fmax(p[1], p[3])};
        int64_t n = tree_query_box_ex(t, q, cf + off[e], 1);
        isort64(cf + off[e], n);
        P2 a = {p[0], p[1]}, b = {p[2], p[3]};
        for (int64_t j = 0; j < n; j++) {
            int64_t f = cf[off[e] + j];
            P2 poly[XO_MAXV / 2], c, d;
            int np = load_poly(t->xy, t->faces + f * t->m, t->len[f], poly);
            if (cyrus_beck_clip(poly, np, a, b, &c, &d) && piece_length(c, d) > 0.0) {
                double *o = cxy + 4 * (off[e] + j);
                o[0] = c.x; o[1] = c.y; o[2] = d.x; o[3] = d.y;
                keep[off[e] + j] = 1;
            }
        }
    }
    int64_t P = 0;
    for (int64_t c = 0; c < C; c++) P += keep[c];
    free(t->edg_e); free(t->edg_f); free(t->edg_xy);
    t->edg_n = P;
    t->edg_e = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P > 0 ? P : 1));
    t->edg_f = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P > 0 ? P : 1));
    t->edg_xy = (double *)malloc(sizeof(double) * 4 * (size_t)(P > 0 ? P : 1));
    int64_t k = 0;
    for (int64_t e = 0; e < n_edge; e++) {
        for (int64_t c = off[e]; c < off[e + 1]; c++) {
            if (!keep[c]) continue;
            t->edg_e[k] = e;
            t->edg_f[k] = cf[c];
            memcpy(t->edg_xy + 4 * k, cxy + 4 * c, sizeof(double) * 4);
            k++;
        }
    }
    free(off); free(cf); free(cxy); free(keep);
    *n_found = P;
    return 0;
}

int xo_intersect_edges_fill(xo_tree *t, int64_t *edge_idx, int64_t *face_idx, double *intersections) {
    memcpy(edge_idx, t->edg_e, sizeof(int64_t) * (size_t)t->edg_n);
    memcpy(face_idx, t->edg_f, sizeof(int64_t) * (size_t)t->edg_n);
    memcpy(intersections, t->edg_xy, sizeof(double) * 4 * (size_t)t->edg_n);
    return 0;
}

static int point_in_poly_or_on_edge(P2 p, const P2 *poly, int n, double tol) {
    int c = 0;
    P2 v0 = poly[n - 1];
    for (int i = 0; i < n; i++) {
        P2 v1 = poly[i];
        double wx = v1.x - v0.x, wy = v1.y - v0.y;
        double len2 = wx * wx + wy * wy;
        if (len2 > 0) {
            double ux = p.x - v0.x, uy = p.y - v0.y;
            double twice_area = fabs(wx * uy - wy * ux);
            double len = sqrt(len2);
            if (twice_area < tol * len) {
                double tpar = ux * wx + uy * wy; /* = t * len2 */
                if (tpar >= 0 && tpar <= len2) return 1;
            }
            if ((v0.y > p.y) != (v1.y > p.y)) {
                double xint = wx * (p.y - v0.y) / wy + v0.x;
                if (p.x < xint) c = !c;
            }
        }
        v0 = v1;
    }
    return c;
}

static int64_t locate_one(const xo_tree *t, P2 p, double tol) {
    /* tree traversal with the query box inflated by tol; LOWEST matching face index wins */
    int64_t stack[128];
    int sp = 0;
    int64_t best = -1;
    if (t->n_face == 0) return -1;
    stack[sp++] = 0;
    while (sp > 0) {
        const Node *nd = &t->nodes[stack[--sp]];
        if (nd->child == -1) {
            for (int64_t i = nd->ptr; i < nd->ptr + nd->size; i++) {
                int64_t f = t->bb_indices[i];
                if (best >= 0 && f > best) continue;
                const Box *b = &t->bb[f];
                if (p.x < b->xmin - tol || p.x > b->xmax + tol || p.y < b->ymin - tol || p.y > b->ymax + tol)
                    continue;
                P2 poly[XO_MAXV / 2];
                int n = load_poly(t->xy, t->faces + f * t->m, t->len[f], poly);
                if (point_in_poly_or_on_edge(p, poly, n, tol)) best = f;
            }
            continue;
        }
        double c = nd->dim ? p.y : p.x;
        if (c - tol <= nd->Lmax) stack[sp++] = nd->child;
        if (c + tol >= nd->Rmin) stack[sp++] = nd->child + 1;
    }
    return best;
}

int xo_locate_points(const xo_tree *t, const double *pts, int64_t n, double tolerance,
                     int64_t *face_index) {
    if (t->m > XO_MAXV / 2) return -2;
    double tol = tolerance < 0 ? xo_default_tolerance(t) : tolerance;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; i++) {
        P2 p = {pts[2 * i], pts[2 * i + 1]};
        face_index[i] = locate_one(t, p, tol);
    }
    return 0;
}

/* ---- generalized barycentric (Wachspress) weights of p in a convex CCW polygon ----
 * on-edge special case (within tol of an edge segment): linear interpolation along the edge.
 * triangles: plain area coordinates. */
static void bary_weights(P2 p, const P2 *poly, int n, double tol, double *w /* n */) {
    double A[XO_MAXV / 2]; /* A[i] = cross(v_i - p, v_{i+1} - p) = twice area(p, v_i, v_{i+1}) */
    for (int i = 0; i < n; i++) w[i] = 0.0;
    for (int i = 0; i < n; i++) {
        P2 v0 = poly[i], v1 = poly[(i + 1) % n];
        double wx = v1.x - v0.x, wy = v1.y - v0.y;
        double ux = p.x - v0.x, uy = p.y - v0.y;
        double a = wx * uy - wy * ux; /* = cross(v0-p, v1-p) */
        A[i] = a;
        double len2 = wx * wx + wy * wy;
        if (len2 > 0) {
            double len = sqrt(len2);
            if (fabs(a) < tol * len) {
                double tpar = ux * wx + uy * wy;
                if (tpar >= 0 && tpar <= len2) {
                    double tt = tpar / len2;
                    if (tt < 0) tt = 0;
                    if (tt > 1) tt = 1;
                    w[i] = 1.0 - tt;
                    w[(i + 1) % n] = tt;
                    return;
                }
            }
        }
    }
    if (n == 3) {
        double s = A[0] + A[1] + A[2];
        w[0] = A[1] / s; /* opposite edge (v1,v2) */
        w[1] = A[2] / s;
        w[2] = A[0] / s;
        return;
    }
    double wsum = 0.0;
    for (int i = 0; i < n; i++) {
        int ip = (i + n - 1) % n, in = (i + 1) % n;
        double cx = (poly[i].x - poly[ip].x) * (poly[in].y - poly[i].y) -
                    (poly[i].y - poly[ip].y) * (poly[in].x - poly[i].x);
        double wi = cx / (A[ip] * A[i]);
        w[i] = wi;
        wsum += wi;
    }
    for (int i = 0; i < n; i++) w[i] = w[i] / wsum;
}

int xo_barycentric(const xo_tree *t, const double *pts, int64_t n, double tolerance,
                   int64_t *face_index, double *weights) {
    if (t->m > XO_MAXV / 2) return -2;
    double tol = tolerance < 0 ? xo_default_tolerance(t) : tolerance;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; i++) {
        P2 p = {pts[2 * i], pts[2 * i + 1]};
        int64_t f = locate_one(t, p, tol);
        face_index[i] = f;
        double *w = weights + i * t->m;
        for (int64_t j = 0; j < t->m; j++) w[j] = 0.0;
        if (f >= 0) {
            P2 poly[XO_MAXV / 2];
            int len = load_poly(t->xy, t->faces + f * t->m, t->len[f], poly);
            bary_weights(p, poly, len, tol, w);
        }
    }
    return 0;
}
