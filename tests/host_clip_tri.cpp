// CPU harness of xugrid_amd/csrc/xr_clip_tri.h (test infrastructure): one "lane", BLOCK = 1.
#include "xr_clip_tri.h"

extern "C" double host_tri_clip_area(const double *tv_, const double *sv_) {
    static uint2 lut[xr::TRI_LUT];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < xr::TRI_LUT; i++) {
            threadIdx.x = i;
            xr::tri_lut_init(lut);
        }
        threadIdx.x = 0;
        init = true;
    }
    xr::P2 tv[3], sv[3];
    for (int j = 0; j < 3; j++) {
        tv[j] = xr::P2{tv_[2 * j], tv_[2 * j + 1]};
        sv[j] = xr::P2{sv_[2 * j], sv_[2 * j + 1]};
    }
    double2 col[xr::TRI_MAXV + 1];
    for (auto &c : col) c = double2{NAN, NAN};
    return xr::tri_clip_area(tv, sv, col, lut, true);
}

extern "C" void host_tri_clip_many(const double *tv, const double *sv, long n, double *out) {
    for (long i = 0; i < n; i++) out[i] = host_tri_clip_area(tv + 6 * i, sv + 6 * i);
}

// quadrilateral (or triangular: n0 = 3) subject against a triangle: the MAXV = 7 instantiation
extern "C" void host_quad_clip_many(const double *tv_, const int *n0, const double *sv_, long n, double *out) {
    static uint2 lut[xr::QUAD_LUT];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < xr::QUAD_LUT; i++) {
            threadIdx.x = i;
            xr::poly_lut_init<xr::QUAD_MAXV>(lut);
        }
        threadIdx.x = 0;
        init = true;
    }
    for (long i = 0; i < n; i++) {
        xr::P2 tv[4], sv[3];
        for (int j = 0; j < 4; j++) tv[j] = xr::P2{tv_[8 * i + 2 * j], tv_[8 * i + 2 * j + 1]};
        for (int j = 0; j < 3; j++) sv[j] = xr::P2{sv_[6 * i + 2 * j], sv_[6 * i + 2 * j + 1]};
        double2 col[xr::QUAD_MAXV + 2];
        for (auto &c : col) c = double2{NAN, NAN};
        out[i] = xr::poly_clip_area<xr::QUAD_MAXV, 4>(tv, n0[i], sv, col, lut, true);
    }
}
