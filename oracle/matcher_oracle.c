/*
 * oracle/matcher_oracle.c -- CPU restatement of the reference's correlative scan matcher.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The shipped library (libcgmr.so) never does.
 *
 * PARITY UNPINNED.  The matcher core of the reference (src/matcher/chargrid.cpp) needs Eigen, which
 * is not installed in this image; building it against a stand-in header is not allowed, the
 * reference has no tests or golden vectors, so this restatement cannot be
 * checked against outputs of the reference itself.  What it is checked against: the kernel tables,
 * grid sizes and the synthetic-offset recovery recorded from the reference in SURVEY.md Appendix C,
 * plus brute-force property tests (tests/test_oracle_matcher.py).
 *
 * Every function cites the reference lines it follows:
 *   cmo_make_kernel        ScanMatcher::initializeKernel      src/matcher/scan_matcher.cpp:38-61
 *   cmo_grid_*             _GridMap ctor/world2grid/grid2world/isInside
 *                                                             src/matcher/gridmap.h:24-58,196-214
 *   cmo_rasterize          resetGrid + addAndConvolvePoints + applyKernel
 *                          src/matcher/scan_matcher.cpp:68-76, chargrid.h:205-216, chargrid.cpp:132-161
 *   cmo_subsample          CharGrid::subsample                src/matcher/chargrid.cpp:61-122
 *   cmo_greedy_search      CharGrid::greedySearch (regions)   src/matcher/chargrid.cpp:208-308
 *                          + addToPrunedMap (36-46), DiscreteTriplet (chargrid.h:68-85)
 *   cmo_hierarchical_search CharGrid::hierarchicalSearch      src/matcher/chargrid.cpp:310-413
 *   cmo_cartesian          RawLaser::cartesian [g2o-recalled, SURVEY.md Appendix A]
 *   cmo_apply_transf       ScanMatcher::applyTransfToScan     src/matcher/scan_matcher.cpp:78-87
 *   cmo_close_scan_match   ScanMatcher::closeScanMatching     src/matcher/scan_matcher.cpp:112-189
 *
 * Deviations, stated once:
 *  (1) cos/sin of the search angle: the reference calls libm (chargrid.cpp:241); here a
 *      self-contained routine (Cody-Waite reduction + the classic 13th/14th-order kernels, < 1 ulp)
 *      is used so that this oracle and the HIP kernel, which carries the same routine, agree bit
 *      for bit on every platform.  A last-bit difference to libm can change a result only if a
 *      rotated coordinate times 1/res lands within one ulp of an integer (the value is truncated,
 *      chargrid.cpp:249); cmo_sincos_cell_differences below counts such events and
 *      tests/test_oracle_matcher.py runs it over the golden fixture's pairs and angles.
 *  (2) std::sort is unstable for equal scores (chargrid.cpp:307).  Here equal scores keep map order
 *      (thread, ix, iy, ith), which is what libstdc++ does for <= 16 results (insertion sort).
 *  (3) `char distance = K1*sqrt(...)` (scan_matcher.cpp:50) is signed char on x86; values stay below
 *      128 for every configuration the reference uses; larger kernels are rejected here.
 *
 * Plain C (gnu99), compiled with -ffp-contract=off: the reference is built without FMA contraction
 * (x86-64 -O3, CMakeLists.txt:4,121) and `c*x - s*y` must round twice.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ portable sin / cos */

/* The polynomial kernels k_sin / k_cos and the three-stage pi/2 reduction below follow FreeBSD/Sun fdlibm
 * (__kernel_sin, __kernel_cos, __ieee754_rem_pio2: same coefficients, same evaluation order), whose licence asks
 * that this notice be preserved:
 * ====================================================
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunPro, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ====================================================
 */
static double k_sin(double x, double y, int iy) {
  static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                      S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                      S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static double k_cos(double x, double y) {
  static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                      C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                      C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double ax = fabs(x);
  if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
  double qx;
  if (ax > 0.78125) qx = 0.28125;
  else {
    /* x/4 with the low 32 bits of the mantissa cleared */
    union { double d; uint64_t u; } c;
    c.d = ax * 0.25;
    c.u &= 0xffffffff00000000ULL;
    qx = c.d;
  }
  double hz = 0.5 * z - qx;
  double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

/* x = n*pi/2 + (y0 + y1), |y0| <= pi/4; three-stage Cody-Waite, valid for |x| < 2^19 * pi/2 */
static int rem_pio2(double x, double *y0, double *y1) {
  static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                      pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                      pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                      pio2_3t = 8.47842766036889956997e-32;
  double ax = fabs(x);
  int n = (int)(ax * invpio2 + 0.5);
  double fn = (double)n;
  double r = ax - fn * pio2_1;
  double w = fn * pio2_1t;
  double a0 = r - w;
  union { double d; uint64_t u; } cx, ca;
  cx.d = ax; ca.d = a0;
  int ex = (int)((cx.u >> 52) & 0x7ff), ea = (int)((ca.u >> 52) & 0x7ff);
  if (ex - ea > 16) {
    double t = r;
    w = fn * pio2_2;
    r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    a0 = r - w;
    ca.d = a0;
    ea = (int)((ca.u >> 52) & 0x7ff);
    if (ex - ea > 49) {
      t = r;
      w = fn * pio2_3;
      r = t - w;
      w = fn * pio2_3t - ((t - r) - w);
      a0 = r - w;
    }
  }
  double a1 = (r - a0) - w;
  if (x < 0) { *y0 = -a0; *y1 = -a1; return -n; }
  *y0 = a0; *y1 = a1;
  return n;
}

void cmo_sincos(double x, double *s, double *c) {
  if (fabs(x) <= 0.78539816339744830962) { *s = k_sin(x, 0.0, 0); *c = k_cos(x, 0.0); return; }
  double y0, y1;
  int n = rem_pio2(x, &y0, &y1);
  double sn = k_sin(y0, y1, 1), cs = k_cos(y0, y1);
  switch (n & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

/* Deviation (1) made checkable: the cells CharGrid::greedySearch truncates to (chargrid.cpp:246-250) with the search angle's
 * cos / sin from libm -- what the reference computes -- and from cmo_sincos, for every (point, angle) pair; returns how many
 * of the cells differ (and how many of the angles' cos / sin values differ in their last bit, for the record). */
long cmo_sincos_cell_differences(int npts, const double *pts, int nangles, const double *angles, float inv_res, long *angle_bits_differ) {
  long ndiff = 0, nbits = 0;
  for (int a = 0; a < nangles; a++) {
    const double t = angles[a];
    const double cl = cos(t), sl = sin(t);
    double sp, cp;
    cmo_sincos(t, &sp, &cp);
    if (cl != cp || sl != sp) nbits++;
    for (int i = 0; i < npts; i++) {
      const double x = pts[2 * i], y = pts[2 * i + 1];
      const int ixl = (int)((cl * x - sl * y) * inv_res), iyl = (int)((sl * x + cl * y) * inv_res);
      const int ixp = (int)((cp * x - sp * y) * inv_res), iyp = (int)((sp * x + cp * y) * inv_res);
      if (ixl != ixp || iyl != iyp) ndiff++;
    }
  }
  if (angle_bits_differ) *angle_bits_differ = nbits;
  return ndiff;
}

/* ------------------------------------------------------------------------ grid */

typedef struct {
  float ll_x, ll_y, ur_x, ur_y;
  float res, inv_res;
  int nx, ny;
  uint8_t *cells; /* cells[x * ny + y], the reference's rows[x][y] */
  int kscale;
} cmo_grid;

/* _GridMap(lowerLeft, upperRight, res) (gridmap.h:196-214) */
void cmo_grid_dims(float ll_x, float ll_y, float ur_x, float ur_y, float res, int *nx, int *ny, float *inv_res) {
  float ir = (float)(1. / res);
  float dx = (ur_x - ll_x) * ir, dy = (ur_y - ll_y) * ir;
  *nx = (int)dx;
  *ny = (int)dy;
  *inv_res = ir;
}

static void grid_init(cmo_grid *g, float ll_x, float ll_y, float ur_x, float ur_y, float res, int kscale) {
  g->ll_x = ll_x; g->ll_y = ll_y; g->ur_x = ur_x; g->ur_y = ur_y;
  g->res = res;
  cmo_grid_dims(ll_x, ll_y, ur_x, ur_y, res, &g->nx, &g->ny, &g->inv_res);
  g->cells = (uint8_t *)malloc((size_t)g->nx * g->ny);
  g->kscale = kscale;
}

static inline void world2grid(const cmo_grid *g, float wx, float wy, int *ix, int *iy) {
  *ix = (int)lrintf((wx - g->ll_x) * g->inv_res);
  *iy = (int)lrintf((wy - g->ll_y) * g->inv_res);
}
static inline int is_inside(const cmo_grid *g, int x, int y) { return x >= 0 && y >= 0 && x < g->nx && y < g->ny; }

/* initializeKernel (scan_matcher.cpp:38-61).  kernel is dim*dim, element (row i, col j) at j*dim+i.
 * Returns dim or -1 if a value would not fit a signed char. */
int cmo_make_kernel(double resolution, double kernel_range, int kscale, uint8_t *kernel, int cap) {
  int size = (int)(kernel_range / resolution);
  int center = size;
  int dim = 2 * size + 1;
  if (dim * dim > cap) return -1;
  int K1 = (int)(resolution * kscale);
  int K2 = (int)(kernel_range * kscale);
  if (K2 > 127) return -1;
  memset(kernel, (unsigned char)(char)K2, (size_t)dim * dim);
  for (int j = 0; j <= size; j++)
    for (int i = 0; i <= size; i++) {
      double dv = K1 * sqrt((double)(j * j + i * i));
      if (dv >= 128.0) continue;              /* (char) would wrap; the reference never gets here */
      char distance = (char)dv;
      if (distance > K2) continue;
      unsigned char d = (unsigned char)distance;
      kernel[(j + center) * dim + (i + center)] = d;
      kernel[(j + center) * dim + (center - i)] = d;
      kernel[(center - j) * dim + (i + center)] = d;
      kernel[(center - j) * dim + (center - i)] = d;
    }
  return dim;
}

/* resetGrid (fill with K2) + addAndConvolvePoints (chargrid.h:205-216) + applyKernel (chargrid.cpp:132-161) */
static void rasterize(cmo_grid *g, const uint8_t *kernel, int kdim, double kernel_range, int n, const double *pts) {
  int K2 = (int)(kernel_range * g->kscale);
  memset(g->cells, (unsigned char)K2, (size_t)g->nx * g->ny);
  int center = (kdim - 1) / 2;
  for (int p = 0; p < n; p++) {
    float px = (float)pts[2 * p], py = (float)pts[2 * p + 1];
    int r, c;
    world2grid(g, px, py, &r, &c);
    for (int i = 0; i < kdim; i++) {
      int io = r + i - center;
      if (io < 0 || io >= g->nx) continue;
      for (int j = 0; j < kdim; j++) {
        int jo = c + j - center;
        if (jo < 0 || jo >= g->ny) continue;
        uint8_t *v = &g->cells[(size_t)io * g->ny + jo];
        uint8_t k = kernel[j * kdim + i];
        if (k < *v) *v = k;
      }
    }
  }
}

/* exported: build a grid and return its cells (for tests of the device rasteriser) */
int cmo_rasterize(float ll_x, float ll_y, float ur_x, float ur_y, float res, double kernel_res,
                  double kernel_range, int kscale, int n, const double *pts, uint8_t *cells_out, int *nx, int *ny) {
  cmo_grid g;
  grid_init(&g, ll_x, ll_y, ur_x, ur_y, res, kscale);
  uint8_t kernel[64 * 64];
  int kdim = cmo_make_kernel(kernel_res, kernel_range, kscale, kernel, sizeof kernel);
  if (kdim < 0) { free(g.cells); return -1; }
  rasterize(&g, kernel, kdim, kernel_range, n, pts);
  if (cells_out) memcpy(cells_out, g.cells, (size_t)g.nx * g.ny);
  *nx = g.nx; *ny = g.ny;
  free(g.cells);
  return 0;
}

/* ------------------------------------------------------------------- subsample */

typedef struct { int kx, ky, idx; } sub_key;
static int sub_cmp(const void *a, const void *b) {
  const sub_key *p = (const sub_key *)a, *q = (const sub_key *)b;
  if (p->kx != q->kx) return p->kx < q->kx ? -1 : 1;
  if (p->ky != q->ky) return p->ky < q->ky ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

/* CharGrid::subsample (chargrid.cpp:61-122): bucket by (int)(ires*p), mean = acc * (1/count),
 * output in (kx, ky) order, members accumulated in input order.  Returns the number of buckets. */
int cmo_subsample(int n, const double *src, double res, double *dst) {
  double ires = 1. / res;
  sub_key *keys = (sub_key *)malloc(sizeof(sub_key) * (n ? n : 1));
  for (int i = 0; i < n; i++) {
    keys[i].kx = (int)(ires * src[2 * i]);
    keys[i].ky = (int)(ires * src[2 * i + 1]);
    keys[i].idx = i;
  }
  qsort(keys, n, sizeof(sub_key), sub_cmp);
  int m = 0;
  for (int i = 0; i < n;) {
    int j = i;
    double ax = 0, ay = 0;
    int cnt = 0;
    while (j < n && keys[j].kx == keys[i].kx && keys[j].ky == keys[i].ky) {
      ax += src[2 * keys[j].idx];
      ay += src[2 * keys[j].idx + 1];
      cnt++; j++;
    }
    double w = 1. / (double)cnt;
    dst[2 * m] = ax * w;
    dst[2 * m + 1] = ay * w;
    m++;
    i = j;
  }
  free(keys);
  return m;
}

/* --------------------------------------------------------------- greedy search */

typedef struct { double x, y, theta, score; } cmo_result;

typedef struct {       /* one entry of std::map<DiscreteTriplet, MatcherResult> */
  int thread;
  double ix, iy, ith;  /* ints stored as doubles, chargrid.h:84 */
  cmo_result r;
} map_entry;

typedef struct { map_entry *e; int n, cap; } result_map;

static int triplet_less(double ax, double ay, double at, double bx, double by, double bt) {
  if (ax < bx) return 1;
  if (ax == bx && ay < by) return 1;
  if (ax == bx && ay == by && at < bt) return 1;
  return 0;
}

/* addToPrunedMap (chargrid.cpp:36-46): replace only if the stored score is strictly greater */
static void add_pruned(result_map *m, int thread, const cmo_result *r, double dx, double dy, double dth) {
  double ix = (double)(int)(r->x / dx), iy = (double)(int)(r->y / dy), ith = (double)(int)(r->theta / dth);
  for (int k = 0; k < m->n; k++) {
    map_entry *e = &m->e[k];
    if (e->thread == thread && e->ix == ix && e->iy == iy && e->ith == ith) {
      if (e->r.score > r->score) e->r = *r;
      return;
    }
  }
  if (m->n == m->cap) { m->cap = m->cap ? 2 * m->cap : 64; m->e = (map_entry *)realloc(m->e, sizeof(map_entry) * m->cap); }
  map_entry *e = &m->e[m->n++];
  e->thread = thread; e->ix = ix; e->iy = iy; e->ith = ith; e->r = *r;
}

static int entry_order_cmp(const void *a, const void *b) {   /* thread, then map order */
  const map_entry *p = (const map_entry *)a, *q = (const map_entry *)b;
  if (p->thread != q->thread) return p->thread < q->thread ? -1 : 1;
  if (triplet_less(p->ix, p->iy, p->ith, q->ix, q->iy, q->ith)) return -1;
  if (triplet_less(q->ix, q->iy, q->ith, p->ix, p->iy, p->ith)) return 1;
  return 0;
}

/* CharGrid::greedySearch (chargrid.cpp:208-308).  regions: nreg * 6 floats (lower xyz, upper xyz).
 * Results ascending by score (stable w.r.t. map order), at most cap are written; returns the total. */
static int greedy(const cmo_grid *g, int npts, const double *pts, int nreg, const float *regions,
                  double step_x, double step_y, double theta_res, double max_score, double dx, double dy,
                  double dth, cmo_result *out, int cap) {
  int x_steps = (int)(step_x / g->res), y_steps = (int)(step_y / g->res);
  if (x_steps <= 0) x_steps = 1;
  if (y_steps <= 0) y_steps = 1;
  int max_threads = 4;
  int num_threads = nreg < max_threads ? nreg : max_threads;
  if (num_threads <= 0) return 0;
  int chunk = nreg / num_threads;
  result_map map = {0, 0, 0};
  int *ipx = (int *)malloc(sizeof(int) * (npts ? npts : 1)), *ipy = (int *)malloc(sizeof(int) * (npts ? npts : 1));
  for (int th = 0; th < num_threads; th++) {
    int imin = th * chunk, imax = (th == num_threads - 1) ? nreg : (th + 1) * chunk;
    for (int reg = imin; reg < imax; reg++) {
      const float *rg = regions + 6 * reg;
      int lo_x, lo_y, hi_x, hi_y;
      world2grid(g, rg[0], rg[1], &lo_x, &lo_y);
      world2grid(g, rg[3], rg[4], &hi_x, &hi_y);
      for (double t = rg[2]; t < rg[5]; t += theta_res) {
        double c, s;
        cmo_sincos(t, &s, &c);
        int prev_x = -10000, prev_y = -10000, k = 0;
        for (int i = 0; i < npts; i++) {
          double px = c * pts[2 * i] - s * pts[2 * i + 1];
          double py = s * pts[2 * i] + c * pts[2 * i + 1];
          int ix = (int)(px * g->inv_res), iy = (int)(py * g->inv_res);
          if (ix != prev_x || iy != prev_y) { ipx[k] = ix; ipy[k] = iy; k++; prev_x = ix; prev_y = iy; }
        }
        float ikscale = (float)(1. / (float)g->kscale);
        for (int i = lo_x; i < hi_x; i += x_steps)
          for (int j = lo_y; j < hi_y; j += y_steps) {
            int idsum = 0;
            for (int q = 0; q < k; q++) {
              int cx = ipx[q] + i, cy = ipy[q] + j;
              if (is_inside(g, cx, cy)) idsum += g->cells[(size_t)cx * g->ny + cy];
            }
            float dsum = (float)idsum * (float)ikscale;
            dsum = k ? (float)(dsum / (double)k) : (float)(max_score + 1);
            if (dsum < max_score) {
              cmo_result r;
              r.x = (double)(float)(g->ll_x + (g->res * (float)i));
              r.y = (double)(float)(g->ll_y + (g->res * (float)j));
              r.theta = t;
              r.score = dsum;
              add_pruned(&map, th, &r, dx, dy, dth);
            }
          }
      }
    }
  }
  free(ipx); free(ipy);
  /* concatenate thread maps in thread order (each map iterates in triplet order), sort by score */
  qsort(map.e, map.n, sizeof(map_entry), entry_order_cmp);
  /* stable insertion sort on score */
  for (int a = 1; a < map.n; a++) {
    map_entry v = map.e[a];
    int b = a - 1;
    while (b >= 0 && map.e[b].r.score > v.r.score) { map.e[b + 1] = map.e[b]; b--; }
    map.e[b + 1] = v;
  }
  int total = map.n;
  for (int a = 0; a < total && a < cap; a++) out[a] = map.e[a].r;
  free(map.e);
  return total;
}

/* exported: rasterise ref points into a fresh grid, then greedySearch. */
int cmo_greedy_search(float ll_x, float ll_y, float ur_x, float ur_y, float res, double kernel_res,
                      double kernel_range, int kscale, int nref, const double *ref_pts, int nq, const double *q_pts,
                      int nreg, const float *regions, double step_x, double step_y, double theta_res,
                      double max_score, double dx, double dy, double dth, cmo_result *out, int cap) {
  cmo_grid g;
  grid_init(&g, ll_x, ll_y, ur_x, ur_y, res, kscale);
  uint8_t kernel[64 * 64];
  int kdim = cmo_make_kernel(kernel_res, kernel_range, kscale, kernel, sizeof kernel);
  if (kdim < 0) { free(g.cells); return -1; }
  rasterize(&g, kernel, kdim, kernel_range, nref, ref_pts);
  int n = greedy(&g, nq, q_pts, nreg, regions, step_x, step_y, theta_res, max_score, dx, dy, dth, out, cap);
  free(g.cells);
  return n;
}

/* hierarchicalSearch with nLevels (chargrid.cpp:310-344, 376-400) */
int cmo_hierarchical_search(float ll_x, float ll_y, float ur_x, float ur_y, float res, double kernel_res,
                            double kernel_range, int kscale, int nref, const double *ref_pts, int nq,
                            const double *q_pts, int nreg, const float *regions, double theta_res, double max_score,
                            double dx, double dy, double dth, int n_levels, cmo_result *out, int cap) {
  cmo_grid g;
  grid_init(&g, ll_x, ll_y, ur_x, ur_y, res, kscale);
  uint8_t kernel[64 * 64];
  int kdim = cmo_make_kernel(kernel_res, kernel_range, kscale, kernel, sizeof kernel);
  if (kdim < 0) { free(g.cells); return -1; }
  rasterize(&g, kernel, kdim, kernel_range, nref, ref_pts);
  int max_res = 1 << 16;
  cmo_result *res_buf = (cmo_result *)malloc(sizeof(cmo_result) * max_res);
  float *cur = (float *)malloc(sizeof(float) * 6 * (nreg > 0 ? nreg : 1));
  memcpy(cur, regions, sizeof(float) * 6 * nreg);
  int ncur = nreg, nres = 0;
  for (int lv = 0; lv < n_levels; lv++) {
    int i = n_levels - 1 - lv;
    int m = 1 << i;                       /* pow(2,i) */
    int mtheta = (m / 2 < 1) ? m : m / 2;
    double sx = m * (double)g.res, sy = m * (double)g.res, st = mtheta * theta_res;
    double rdx = dx * m, rdy = dy * m, rdth = dth * m;
    int last = (lv == n_levels - 1);
    if (last && nres == 0) break;   /* chargrid.cpp:335: the last level runs only if mresvec is non-empty */
    nres = greedy(&g, nq, q_pts, ncur, cur, sx, sy, st, max_score, rdx, rdy, rdth, res_buf, max_res);
    if (nres > max_res) nres = max_res;
    if (last) break;
    if (nres == 0) break;
    cur = (float *)realloc(cur, sizeof(float) * 6 * nres);
    for (int k = 0; k < nres; k++) {
      double lx = -(rdx * .5) + res_buf[k].x, ly = -(rdy * .5) + res_buf[k].y, lt = -(rdth * .5) + res_buf[k].theta;
      double ux = (rdx * .5) + res_buf[k].x, uy = (rdy * .5) + res_buf[k].y, ut = (rdth * .5) + res_buf[k].theta;
      cur[6 * k] = (float)lx; cur[6 * k + 1] = (float)ly; cur[6 * k + 2] = (float)lt;
      cur[6 * k + 3] = (float)ux; cur[6 * k + 4] = (float)uy; cur[6 * k + 5] = (float)ut;
    }
    ncur = nres;
  }
  int n = nres;
  for (int k = 0; k < n && k < cap; k++) out[k] = res_buf[k];
  free(res_buf); free(cur); free(g.cells);
  return n;
}

/* --------------------------------------------------------- scans and close match */

/* RawLaser::cartesian [g2o-recalled]: beams with min_range < r < max_range, alpha = first + i*step */
int cmo_cartesian(int nbeams, const float *ranges, double angle_min, double angle_inc, double max_range,
                  double min_range, double *pts) {
  int n = 0;
  for (int i = 0; i < nbeams; i++) {
    double r = (double)ranges[i];
    if (r < max_range && r > min_range) {
      double alpha = angle_min + i * angle_inc;
      pts[2 * n] = cos(alpha) * r;
      pts[2 * n + 1] = sin(alpha) * r;
      n++;
    }
  }
  return n;
}

/* applyTransfToScan (scan_matcher.cpp:78-87): SE2(transf) * point = t + R(theta) p */
void cmo_apply_transf(const double *tr, int n, const double *in, double *out) {
  double c = cos(tr[2]), s = sin(tr[2]);
  for (int i = 0; i < n; i++) {
    double x = in[2 * i], y = in[2 * i + 1];
    out[2 * i] = (c * x - s * y) + tr[0];
    out[2 * i + 1] = (s * x + c * y) + tr[1];
  }
}

/* closeScanMatching (scan_matcher.cpp:112-189) for a single-scan reference set whose vertex is the
 * origin vertex: ref scan -> laser pose -> grid; query scan -> subsample(0.1) -> laser pose;
 * window guess -/+ (0.3, 0.3, 0.2) as floats; thetaRes 0.00625; bins (0.5, 0.5, 0.2).
 * guess = origin^-1 * current.  Returns 1 if found (out = mresvec[0]). */
int cmo_close_scan_match(int nbeams, const float *ranges_ref, const float *ranges_qry, double angle_min,
                         double angle_inc, double max_range, const double *laser_pose, const double *guess,
                         double resolution, double kernel_range, double max_score, double *out_xyt,
                         double *out_score, int *n_results) {
  cmo_grid g;
  grid_init(&g, -15.f, -15.f, 15.f, 15.f, (float)resolution, 128);
  uint8_t kernel[64 * 64];
  int kdim = cmo_make_kernel(resolution, kernel_range, 128, kernel, sizeof kernel);
  if (kdim < 0) { free(g.cells); return -1; }
  double *tmp = (double *)malloc(sizeof(double) * 2 * nbeams), *ref = (double *)malloc(sizeof(double) * 2 * nbeams);
  double *sub = (double *)malloc(sizeof(double) * 2 * nbeams), *qry = (double *)malloc(sizeof(double) * 2 * nbeams);
  int nref = cmo_cartesian(nbeams, ranges_ref, angle_min, angle_inc, max_range, 0.0, tmp);
  cmo_apply_transf(laser_pose, nref, tmp, ref);
  rasterize(&g, kernel, kdim, kernel_range, nref, ref);
  int nq0 = cmo_cartesian(nbeams, ranges_qry, angle_min, angle_inc, max_range, 0.0, tmp);
  int nq = cmo_subsample(nq0, tmp, 0.1, sub);
  cmo_apply_transf(laser_pose, nq, sub, qry);
  double theta_res = 0.0125 * .5;
  float region[6] = {(float)(-.3 + guess[0]), (float)(-.3 + guess[1]), (float)(-0.2 + guess[2]),
                     (float)(+.3 + guess[0]), (float)(.3 + guess[1]), (float)(0.2 + guess[2])};
  cmo_result resv[256];
  int n = greedy(&g, nq, qry, 1, region, (double)g.res, (double)g.res, theta_res, max_score, 0.5, 0.5, 0.2, resv, 256);
  if (n_results) *n_results = n;
  int found = n > 0;
  if (found) {
    out_xyt[0] = resv[0].x; out_xyt[1] = resv[0].y; out_xyt[2] = resv[0].theta;
    *out_score = resv[0].score;
  }
  free(tmp); free(ref); free(sub); free(qry); free(g.cells);
  return found;
}

/* batch wrapper used by the cpu_baseline leg: pairs are independent */
int cmo_close_scan_match_batch(int npairs, int nbeams, const float *ranges_ref, const float *ranges_qry,
                               double angle_min, double angle_inc, double max_range, const double *laser_pose,
                               const double *guess, double resolution, double kernel_range, double max_score,
                               double *out_xyt, double *out_score, uint8_t *out_found) {
  for (int p = 0; p < npairs; p++) {
    int f = cmo_close_scan_match(nbeams, ranges_ref + (size_t)p * nbeams, ranges_qry + (size_t)p * nbeams, angle_min,
                                 angle_inc, max_range, laser_pose, guess + 3 * p, resolution, kernel_range, max_score,
                                 out_xyt + 3 * p, out_score + p, NULL);
    if (f < 0) return f;
    out_found[p] = (uint8_t)f;
    if (!f) { out_xyt[3 * p] = out_xyt[3 * p + 1] = out_xyt[3 * p + 2] = 0; out_score[p] = 0; }
  }
  return 0;
}

/* ScanMatcher::verifyMatching numeric core (scan_matcher.cpp:430-505 with CharGrid::searchNonMatchedPoints,
 * chargrid.cpp:444-455, and CharGrid::countPoints, chargrid.cpp:417-441):
 *   grid  := reset + addAndConvolvePoints(pts2)            (local map of vset2 in the frame of reference vertex 1)
 *   nonmatched := points of pts1 inside the grid whose cell * (1/kscale) > nonmatched_score (0.3 in the reference)
 *   aux   := reset + addAndConvolvePoints(nonmatched)
 *   score := (float) sum of aux cells over [world2grid(lower), world2grid(upper)) / (float) visited cells
 * Returns the number of non-matched points; *score_out as the reference computes it (float -> double). */
int cmo_verify(float ll_x, float ll_y, float ur_x, float ur_y, float res, double kernel_res, double kernel_range,
               int kscale, int n2, const double *pts2, int n1, const double *pts1, double nonmatched_score,
               const float *lower_xy, const float *upper_xy, double *score_out) {
  cmo_grid g;
  grid_init(&g, ll_x, ll_y, ur_x, ur_y, res, kscale);
  uint8_t kernel[64 * 64];
  int kdim = cmo_make_kernel(kernel_res, kernel_range, kscale, kernel, sizeof kernel);
  if (kdim < 0) { free(g.cells); return -1; }
  rasterize(&g, kernel, kdim, kernel_range, n2, pts2);
  double *nm = (double *)malloc(sizeof(double) * 2 * (n1 ? n1 : 1));
  int nnm = 0;
  float ikscale = (float)(1. / (float)kscale);
  for (int i = 0; i < n1; i++) {
    int gx, gy;
    world2grid(&g, (float)pts1[2 * i], (float)pts1[2 * i + 1], &gx, &gy);
    if (is_inside(&g, gx, gy)) {
      double value = (float)g.cells[(size_t)gx * g.ny + gy] * ikscale;
      if (value > nonmatched_score) { nm[2 * nnm] = pts1[2 * i]; nm[2 * nnm + 1] = pts1[2 * i + 1]; nnm++; }
    }
  }
  rasterize(&g, kernel, kdim, kernel_range, nnm, nm);       /* auxGrid: reset + stamp the unexplained points */
  int lx, ly, ux, uy;
  world2grid(&g, lower_xy[0], lower_xy[1], &lx, &ly);
  world2grid(&g, upper_xy[0], upper_xy[1], &ux, &uy);
  int isum = 0;
  for (int i = lx; i < ux; i++)
    for (int j = ly; j < uy; j++)
      if (is_inside(&g, i, j)) isum += g.cells[(size_t)i * g.ny + j];
  int visited = (ux - lx) * (uy - ly);
  *score_out = (double)((float)isum / (float)visited);
  free(nm); free(g.cells);
  return nnm;
}
