/*
 * oracle/gn_oracle.c -- CPU restatement of the pose-graph Gauss-Newton hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.  The shipped library (libcgmr.so) never does.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in g2o (RainerKuemmerle/g2o,
 * pinned by the reference's README.md:15-19 to 4b9c2f5b68d14ad479457b18c5a2a0bce1541a90)
 * and CSparse, neither of which is vendored in /root/reference nor installed in
 * this image, and the reference holds no tests or golden vectors for it
 * (SURVEY.md section 4).  This file restates the *published* algorithms and is
 * anchored on the reference's own call sites:
 *   - solver configuration: GN + BlockSolver<-1,-1> + LinearSolverCSparse,
 *     scalar ordering            src/slam/graph_slam.cpp:44-56
 *   - GraphSLAM::optimize(n)    src/slam/graph_slam.cpp:561-575
 *   - GraphManipulator push/fixGauge/optimize/pop
 *                               src/slam/graph_manipulator.cpp:62-124
 *   - CovarianceEstimator::compute (marginal 3x3 blocks of H^-1)
 *                               src/slam/graph_manipulator.cpp:128-157
 *   - CondensedGraphCreator::compute (star edges + EdgeLabeler)
 *                               src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66
 * g2o-side behaviour ([g2o-recalled]):
 *   EdgeSE2::computeError / linearizeOplus (analytic), VertexSE2::oplusImpl,
 *   BaseBinaryEdge::constructQuadraticForm, OptimizationAlgorithmGaussNewton::solve,
 *   LinearSolverCSparse::solve = cs_schol(order=AMD) + up-looking cs_chol
 *   (T. Davis, "Direct Methods for Sparse Linear Systems", SIAM 2006, ch. 4),
 *   SparseOptimizer::computeInitialGuess (unit-cost spanning tree from the
 *   fixed vertices), g2o_hierarchical EdgeLabeler (unscented transform,
 *   alpha=1e-3, beta=2, lambda=alpha^2*n).
 * It is cross-checked against an independent numpy/SciPy implementation
 * (tests/ref_numpy.py) and the known-answer tests of SURVEY.md section 8(c).
 *
 * Plain C99, no dependencies beyond libm.  Single thread, like g2o's default.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ SE2 */

static double normalize_theta(double t) {
  if (t >= -M_PI && t < M_PI) return t;
  double m = floor((t + M_PI) / (2 * M_PI));
  return t - 2 * M_PI * m;
}

static void se2_mul(const double *a, const double *b, double *o) {
  double c = cos(a[2]), s = sin(a[2]);
  double x = a[0] + c * b[0] - s * b[1];
  double y = a[1] + s * b[0] + c * b[1];
  o[2] = normalize_theta(a[2] + b[2]);
  o[0] = x; o[1] = y;
}

static void se2_inv(const double *a, double *o) {
  double c = cos(a[2]), s = sin(a[2]);
  double x = -(c * a[0] + s * a[1]);
  double y = -(-s * a[0] + c * a[1]);
  o[0] = x; o[1] = y; o[2] = -a[2];
}

/* e = (z^-1 * (xi^-1 * xj)).toVector()   [EdgeSE2::computeError] */
static void edge_error(const double *xi, const double *xj, const double *z, double *e) {
  double ci = cos(xi[2]), si = sin(xi[2]);
  double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
  double rx = ci * dx + si * dy, ry = -si * dx + ci * dy;
  double rth = normalize_theta(xj[2] - xi[2]);
  double cz = cos(z[2]), sz = sin(z[2]);
  double tx = rx - z[0], ty = ry - z[1];
  e[0] = cz * tx + sz * ty;
  e[1] = -sz * tx + cz * ty;
  e[2] = normalize_theta(rth - z[2]);
}

/* analytic Jacobians  [EdgeSE2::linearizeOplus]; row-major 3x3 */
static void edge_jacobians(const double *xi, const double *xj, const double *z, double *Ji, double *Jj) {
  double c = cos(xi[2]), s = sin(xi[2]);
  double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
  double A[9] = {-c, -s, -s * dx + c * dy, s, -c, -c * dx - s * dy, 0, 0, -1};
  double B[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
  double cz = cos(z[2]), sz = sin(z[2]);
  for (int k = 0; k < 3; k++) {
    Ji[0 + k] = cz * A[0 + k] + sz * A[3 + k];
    Ji[3 + k] = -sz * A[0 + k] + cz * A[3 + k];
    Ji[6 + k] = A[6 + k];
    Jj[0 + k] = cz * B[0 + k] + sz * B[3 + k];
    Jj[3 + k] = -sz * B[0 + k] + cz * B[3 + k];
    Jj[6 + k] = B[6 + k];
  }
}

static void info_full(const double *u, double *O) {
  O[0] = u[0]; O[1] = u[1]; O[2] = u[2];
  O[3] = u[1]; O[4] = u[3]; O[5] = u[4];
  O[6] = u[2]; O[7] = u[4]; O[8] = u[5];
}

/* C = A^T * B (3x3 row-major) */
static void mat_atb(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[0 + i] * B[0 + j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
static void mat_ab(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

double cgo_chi2(int nE, const double *poses, const int32_t *ef, const int32_t *et,
                const double *meas, const double *info) {
  double acc = 0;
  for (int k = 0; k < nE; k++) {
    double e[3], O[9];
    edge_error(poses + 3 * ef[k], poses + 3 * et[k], meas + 3 * k, e);
    info_full(info + 6 * k, O);
    double t = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) t += e[i] * O[3 * i + j] * e[j];
    acc += t;
  }
  return acc;
}

/* per-edge chi2 and error, exported for unit tests */
void cgo_edge_terms(const double *xi, const double *xj, const double *z, double *e, double *Ji, double *Jj) {
  edge_error(xi, xj, z, e);
  edge_jacobians(xi, xj, z, Ji, Jj);
}

/* ------------------------------------------------------- sparse structures */

typedef struct {
  int n;        /* scalar dimension */
  int *p;       /* column pointers (n+1) */
  int *i;       /* row indices */
  double *x;    /* values */
} ccs_t;

static void ccs_free(ccs_t *A) { free(A->p); free(A->i); free(A->x); A->p = A->i = NULL; A->x = NULL; }

typedef struct {
  int nV, nE;
  int nfree;          /* number of free (non fixed, active) vertices */
  int *hidx;          /* vertex -> hessian block index or -1 */
  int *perm;          /* block permutation: new -> old block index */
  int *iperm;         /* old -> new */
  /* block pattern of the permuted upper triangle: for block column c (new index) the
     sorted block rows r<=c */
  int *bp, *bi;       /* (nfree+1), nnzb */
  ccs_t C;            /* scalar upper CCS of P H P^T, values refreshed every iteration */
  /* maps: for every edge the position (in C.x) of the first scalar of its blocks */
  int *pos_ii, *pos_jj, *pos_ij;  /* pos of block top-left scalar in column major; -1 if absent */
  uint8_t *ij_transposed;         /* 1 if block (i,j) is stored as (j,i)^T (because inew>jnew) */
  /* symbolic */
  int *parent;        /* etree (scalar) */
  int *Lp, *Li;       /* L structure (CCS, lower, diagonal first) */
  double *Lx;
  double *b;          /* rhs (permuted scalar order) */
  double *x;          /* solution */
} gn_sys_t;

/* ------------------------------------------------ minimum-degree ordering */

typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *a, int x) {
  if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->v = (int *)realloc(a->v, sizeof(int) * a->cap); }
  a->v[a->n++] = x;
}

/* Exact minimum (external) degree on the block graph with explicit elimination
 * graph.  The 3 scalars of one pose are indistinguishable nodes, so scalar AMD
 * (what cs_schol(order=1) runs because of setBlockOrdering(false),
 * src/slam/graph_slam.cpp:52) merges them into one supervariable; ordering the
 * block graph is the same thing up to tie breaking. */
static void min_degree_order(int n, const ivec *adj_in, int *perm) {
  ivec *adj = (ivec *)calloc(n, sizeof(ivec));
  for (int v = 0; v < n; v++) {
    adj[v].n = adj[v].cap = adj_in[v].n;
    adj[v].v = (int *)malloc(sizeof(int) * (adj[v].cap ? adj[v].cap : 1));
    memcpy(adj[v].v, adj_in[v].v, sizeof(int) * adj_in[v].n);
  }
  /* degree buckets (doubly linked) */
  int *head = (int *)malloc(sizeof(int) * (n + 1));
  int *next = (int *)malloc(sizeof(int) * n), *prev = (int *)malloc(sizeof(int) * n);
  int *deg = (int *)malloc(sizeof(int) * n);
  uint8_t *gone = (uint8_t *)calloc(n, 1);
  int *mark = (int *)malloc(sizeof(int) * n);
  for (int d = 0; d <= n; d++) head[d] = -1;
  for (int v = 0; v < n; v++) mark[v] = -1;
#define BUCKET_INSERT(v, d) do { next[v] = head[d]; prev[v] = -1; if (head[d] >= 0) prev[head[d]] = v; head[d] = v; } while (0)
#define BUCKET_REMOVE(v, d) do { if (prev[v] >= 0) next[prev[v]] = next[v]; else head[d] = next[v]; if (next[v] >= 0) prev[next[v]] = prev[v]; } while (0)
  for (int v = n - 1; v >= 0; v--) { deg[v] = adj[v].n; BUCKET_INSERT(v, deg[v]); }
  int mind = 0;
  int *tmp = (int *)malloc(sizeof(int) * n);
  for (int k = 0; k < n; k++) {
    while (head[mind] < 0) mind++;
    int v = head[mind];
    BUCKET_REMOVE(v, mind);
    gone[v] = 1;
    perm[k] = v;
    int nn = adj[v].n;
    int *N = adj[v].v;
    /* for each neighbour u: adj(u) = (adj(u) U N) \ {u, v} */
    for (int a = 0; a < nn; a++) {
      int u = N[a];
      int m = 0;
      for (int q = 0; q < adj[u].n; q++) {
        int w = adj[u].v[q];
        if (w == v) continue;
        mark[w] = u;
        tmp[m++] = w;
      }
      for (int q = 0; q < nn; q++) {
        int w = N[q];
        if (w == u || mark[w] == u) continue;
        mark[w] = u;
        tmp[m++] = w;
      }
      if (m > adj[u].cap) { adj[u].cap = m + m / 2; adj[u].v = (int *)realloc(adj[u].v, sizeof(int) * adj[u].cap); }
      memcpy(adj[u].v, tmp, sizeof(int) * m);
      adj[u].n = m;
      BUCKET_REMOVE(u, deg[u]);
      deg[u] = m;
      BUCKET_INSERT(u, m);
      if (m < mind) mind = m;
    }
    for (int a = 0; a < nn; a++) mark[N[a]] = -1;
    free(adj[v].v); adj[v].v = NULL; adj[v].n = 0;
  }
  for (int v = 0; v < n; v++) free(adj[v].v);
  free(adj); free(head); free(next); free(prev); free(deg); free(gone); free(mark); free(tmp);
#undef BUCKET_INSERT
#undef BUCKET_REMOVE
}

/* ----------------------------------------------------------- system setup */

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static void gn_sys_free(gn_sys_t *S) {
  free(S->hidx); free(S->perm); free(S->iperm); free(S->bp); free(S->bi);
  ccs_free(&S->C);
  free(S->pos_ii); free(S->pos_jj); free(S->pos_ij); free(S->ij_transposed);
  free(S->parent); free(S->Lp); free(S->Li); free(S->Lx); free(S->b); free(S->x);
  memset(S, 0, sizeof(*S));
}

/* etree of the upper-triangular CCS matrix (Davis, cs_etree with ata=0) */
static void etree(const ccs_t *A, int *parent) {
  int n = A->n;
  int *anc = (int *)malloc(sizeof(int) * n);
  for (int k = 0; k < n; k++) {
    parent[k] = -1; anc[k] = -1;
    for (int p = A->p[k]; p < A->p[k + 1]; p++) {
      int i = A->i[p];
      while (i != -1 && i < k) {
        int inext = anc[i];
        anc[i] = k;
        if (inext == -1) parent[i] = k;
        i = inext;
      }
    }
  }
  free(anc);
}

/* nonzero pattern of row k of L (Davis, cs_ereach).  s[top..n-1] holds it in
 * topological order; w is a flag array (w[i]>=0 unmarked convention replaced by
 * explicit mark stamp). */
static int ereach(const ccs_t *A, int k, const int *parent, int *s, int *stamp, int tag) {
  int n = A->n, top = n;
  stamp[k] = tag;
  for (int p = A->p[k]; p < A->p[k + 1]; p++) {
    int i = A->i[p];
    if (i > k) continue;
    int len = 0;
    for (; stamp[i] != tag; i = parent[i]) { s[len++] = i; stamp[i] = tag; }
    while (len > 0) s[--top] = s[--len];
  }
  return top;
}

/* Build structure: active/free vertices, ordering, block pattern, scalar CCS,
 * edge->slot maps, etree, L pattern. */
static int gn_sys_build(gn_sys_t *S, int nV, const uint8_t *fixed, int nE,
                        const int32_t *ef, const int32_t *et) {
  memset(S, 0, sizeof(*S));
  S->nV = nV; S->nE = nE;
  /* active vertices: touched by at least one edge */
  uint8_t *active = (uint8_t *)calloc(nV, 1);
  for (int k = 0; k < nE; k++) { active[ef[k]] = 1; active[et[k]] = 1; }
  S->hidx = (int *)malloc(sizeof(int) * nV);
  int nf = 0;
  for (int v = 0; v < nV; v++) S->hidx[v] = (active[v] && !fixed[v]) ? nf++ : -1;
  free(active);
  S->nfree = nf;
  if (nf == 0) return 0;
  /* block adjacency */
  ivec *adj = (ivec *)calloc(nf, sizeof(ivec));
  for (int k = 0; k < nE; k++) {
    int a = S->hidx[ef[k]], b = S->hidx[et[k]];
    if (a < 0 || b < 0 || a == b) continue;
    iv_push(&adj[a], b); iv_push(&adj[b], a);
  }
  for (int v = 0; v < nf; v++) {       /* dedupe */
    if (adj[v].n > 1) {
      qsort(adj[v].v, adj[v].n, sizeof(int), cmp_int);
      int m = 1;
      for (int q = 1; q < adj[v].n; q++) if (adj[v].v[q] != adj[v].v[m - 1]) adj[v].v[m++] = adj[v].v[q];
      adj[v].n = m;
    }
  }
  S->perm = (int *)malloc(sizeof(int) * nf);
  S->iperm = (int *)malloc(sizeof(int) * nf);
  min_degree_order(nf, adj, S->perm);
  for (int k = 0; k < nf; k++) S->iperm[S->perm[k]] = k;
  /* permuted upper block pattern */
  S->bp = (int *)calloc(nf + 1, sizeof(int));
  for (int v = 0; v < nf; v++) {
    int c = S->iperm[v];
    S->bp[c + 1]++;                                   /* diagonal */
    for (int q = 0; q < adj[v].n; q++) { int r = S->iperm[adj[v].v[q]]; if (r < c) S->bp[c + 1]++; }
  }
  for (int c = 0; c < nf; c++) S->bp[c + 1] += S->bp[c];
  int nnzb = S->bp[nf];
  S->bi = (int *)malloc(sizeof(int) * nnzb);
  int *fill = (int *)malloc(sizeof(int) * nf);
  for (int c = 0; c < nf; c++) fill[c] = S->bp[c];
  for (int v = 0; v < nf; v++) {
    int c = S->iperm[v];
    for (int q = 0; q < adj[v].n; q++) { int r = S->iperm[adj[v].v[q]]; if (r < c) S->bi[fill[c]++] = r; }
    S->bi[fill[c]++] = c;
  }
  for (int c = 0; c < nf; c++) qsort(S->bi + S->bp[c], S->bp[c + 1] - S->bp[c], sizeof(int), cmp_int);
  free(fill);
  for (int v = 0; v < nf; v++) free(adj[v].v);
  free(adj);
  /* scalar CCS (upper): block (r,c) r<c contributes a full 3x3; diagonal block its upper triangle */
  int n = 3 * nf;
  S->C.n = n;
  S->C.p = (int *)malloc(sizeof(int) * (n + 1));
  int *blk_off = (int *)malloc(sizeof(int) * nnzb);   /* offset of block's first scalar inside scalar column 3c */
  int nnz = 0;
  for (int c = 0; c < nf; c++) {
    int nb = S->bp[c + 1] - S->bp[c];                 /* incl. diagonal (last) */
    for (int q = 0; q < nb; q++) blk_off[S->bp[c] + q] = 3 * q;
    for (int kk = 0; kk < 3; kk++) {
      S->C.p[3 * c + kk] = nnz;
      nnz += 3 * (nb - 1) + kk + 1;
    }
  }
  S->C.p[n] = nnz;
  S->C.i = (int *)malloc(sizeof(int) * nnz);
  S->C.x = (double *)calloc(nnz, sizeof(double));
  for (int c = 0; c < nf; c++) {
    int nb = S->bp[c + 1] - S->bp[c];
    for (int kk = 0; kk < 3; kk++) {
      int p = S->C.p[3 * c + kk];
      for (int q = 0; q < nb - 1; q++) { int r = S->bi[S->bp[c] + q]; for (int rr = 0; rr < 3; rr++) S->C.i[p++] = 3 * r + rr; }
      for (int rr = 0; rr <= kk; rr++) S->C.i[p++] = 3 * c + rr;
    }
  }
  /* edge -> block slot maps */
  S->pos_ii = (int *)malloc(sizeof(int) * nE);
  S->pos_jj = (int *)malloc(sizeof(int) * nE);
  S->pos_ij = (int *)malloc(sizeof(int) * nE);
  S->ij_transposed = (uint8_t *)calloc(nE, 1);
  for (int k = 0; k < nE; k++) {
    int a = S->hidx[ef[k]], b = S->hidx[et[k]];
    S->pos_ii[k] = S->pos_jj[k] = S->pos_ij[k] = -1;
    if (a >= 0) { int c = S->iperm[a]; S->pos_ii[k] = S->bp[c + 1] - 1; }
    if (b >= 0) { int c = S->iperm[b]; S->pos_jj[k] = S->bp[c + 1] - 1; }
    if (a >= 0 && b >= 0 && a != b) {
      int ra = S->iperm[a], cb = S->iperm[b];
      int r = ra, c = cb;
      if (ra > cb) { r = cb; c = ra; S->ij_transposed[k] = 1; }
      int lo = S->bp[c], hi = S->bp[c + 1] - 1;       /* binary search r in column c */
      while (lo < hi) { int mid = (lo + hi) / 2; if (S->bi[mid] < r) lo = mid + 1; else hi = mid; }
      S->pos_ij[k] = lo;
    }
  }
  /* convert block slot -> we keep block index; scalar offset computed on the fly */
  free(blk_off);
  /* symbolic: etree + column counts via row patterns */
  S->parent = (int *)malloc(sizeof(int) * n);
  etree(&S->C, S->parent);
  int *cnt = (int *)calloc(n, sizeof(int));
  int *s = (int *)malloc(sizeof(int) * n), *stamp = (int *)malloc(sizeof(int) * n);
  for (int k = 0; k < n; k++) stamp[k] = -1;
  for (int k = 0; k < n; k++) {
    int top = ereach(&S->C, k, S->parent, s, stamp, k);
    for (int q = top; q < n; q++) cnt[s[q]]++;
    cnt[k]++;
  }
  S->Lp = (int *)malloc(sizeof(int) * (n + 1));
  S->Lp[0] = 0;
  for (int k = 0; k < n; k++) S->Lp[k + 1] = S->Lp[k] + cnt[k];
  S->Li = (int *)malloc(sizeof(int) * S->Lp[n]);
  S->Lx = (double *)malloc(sizeof(double) * S->Lp[n]);
  S->b = (double *)calloc(n, sizeof(double));
  S->x = (double *)calloc(n, sizeof(double));
  free(cnt); free(s); free(stamp);
  return 0;
}

/* scalar position of element (rr,cc) of block slot q in block column c (upper CCS) */
static inline int blk_scalar_pos(const gn_sys_t *S, int c, int q, int rr, int cc) {
  int nb = S->bp[c + 1] - S->bp[c];
  int qi = q - S->bp[c];
  /* column 3c+cc starts at C.p[3c+cc]; off-diagonal blocks take 3 rows each */
  (void)nb;
  return S->C.p[3 * c + cc] + 3 * qi + rr;
}

/* buildSystem: H (upper, permuted) and b from the current poses */
static void gn_sys_linearize(gn_sys_t *S, const double *poses, const int32_t *ef, const int32_t *et,
                             const double *meas, const double *info) {
  int n = S->C.n;
  memset(S->C.x, 0, sizeof(double) * S->C.p[n]);
  memset(S->b, 0, sizeof(double) * n);
  for (int k = 0; k < S->nE; k++) {
    const double *xi = poses + 3 * ef[k], *xj = poses + 3 * et[k];
    double e[3], Ji[9], Jj[9], O[9], JiO[9], JjO[9], H[9];
    edge_error(xi, xj, meas + 3 * k, e);
    edge_jacobians(xi, xj, meas + 3 * k, Ji, Jj);
    info_full(info + 6 * k, O);
    mat_atb(Ji, O, JiO);       /* Ji^T Omega */
    mat_atb(Jj, O, JjO);
    int a = S->hidx[ef[k]], b = S->hidx[et[k]];
    if (a >= 0) {
      int c = S->iperm[a];
      mat_ab(JiO, Ji, H);
      for (int cc = 0; cc < 3; cc++)
        for (int rr = 0; rr <= cc; rr++) S->C.x[blk_scalar_pos(S, c, S->pos_ii[k], rr, cc)] += H[3 * rr + cc];
      for (int rr = 0; rr < 3; rr++) S->b[3 * c + rr] -= JiO[3 * rr] * e[0] + JiO[3 * rr + 1] * e[1] + JiO[3 * rr + 2] * e[2];
    }
    if (b >= 0) {
      int c = S->iperm[b];
      mat_ab(JjO, Jj, H);
      for (int cc = 0; cc < 3; cc++)
        for (int rr = 0; rr <= cc; rr++) S->C.x[blk_scalar_pos(S, c, S->pos_jj[k], rr, cc)] += H[3 * rr + cc];
      for (int rr = 0; rr < 3; rr++) S->b[3 * c + rr] -= JjO[3 * rr] * e[0] + JjO[3 * rr + 1] * e[1] + JjO[3 * rr + 2] * e[2];
    }
    if (S->pos_ij[k] >= 0) {
      mat_ab(JiO, Jj, H);      /* H_ij = Ji^T Omega Jj, rows i, cols j */
      int ra = S->iperm[a], cb = S->iperm[b];
      if (!S->ij_transposed[k]) {
        for (int cc = 0; cc < 3; cc++)
          for (int rr = 0; rr < 3; rr++) S->C.x[blk_scalar_pos(S, cb, S->pos_ij[k], rr, cc)] += H[3 * rr + cc];
      } else {
        for (int cc = 0; cc < 3; cc++)
          for (int rr = 0; rr < 3; rr++) S->C.x[blk_scalar_pos(S, ra, S->pos_ij[k], rr, cc)] += H[3 * cc + rr];
      }
    }
  }
}

/* up-looking sparse Cholesky (Davis, cs_chol).  Returns 0 or -(k+1) on a non
 * positive pivot in column k. */
static int gn_sys_factor(gn_sys_t *S) {
  const ccs_t *C = &S->C;
  int n = C->n;
  int *c = (int *)malloc(sizeof(int) * n), *s = (int *)malloc(sizeof(int) * n), *stamp = (int *)malloc(sizeof(int) * n);
  double *x = (double *)calloc(n, sizeof(double));
  for (int k = 0; k < n; k++) { c[k] = S->Lp[k]; stamp[k] = -1; }
  int status = 0;
  for (int k = 0; k < n; k++) {
    int top = ereach(C, k, S->parent, s, stamp, k);
    x[k] = 0;
    for (int p = C->p[k]; p < C->p[k + 1]; p++) if (C->i[p] <= k) x[C->i[p]] = C->x[p];
    double d = x[k];
    x[k] = 0;
    for (; top < n; top++) {
      int i = s[top];
      double lki = x[i] / S->Lx[S->Lp[i]];
      x[i] = 0;
      for (int p = S->Lp[i] + 1; p < c[i]; p++) x[S->Li[p]] -= S->Lx[p] * lki;
      d -= lki * lki;
      int p = c[i]++;
      S->Li[p] = k;
      S->Lx[p] = lki;
    }
    if (d <= 0) { status = -(k + 1); break; }
    int p = c[k]++;
    S->Li[p] = k;
    S->Lx[p] = sqrt(d);
  }
  free(c); free(s); free(stamp); free(x);
  return status;
}

static void l_solve(const gn_sys_t *S, double *x) {        /* L x = b */
  int n = S->C.n;
  for (int j = 0; j < n; j++) {
    x[j] /= S->Lx[S->Lp[j]];
    double xj = x[j];
    for (int p = S->Lp[j] + 1; p < S->Lp[j + 1]; p++) x[S->Li[p]] -= S->Lx[p] * xj;
  }
}
static void lt_solve(const gn_sys_t *S, double *x) {       /* L^T x = b */
  int n = S->C.n;
  for (int j = n - 1; j >= 0; j--) {
    double t = x[j];
    for (int p = S->Lp[j] + 1; p < S->Lp[j + 1]; p++) t -= S->Lx[p] * x[S->Li[p]];
    x[j] = t / S->Lx[S->Lp[j]];
  }
}

/* ---------------------------------------------------------------- GN driver */

/* GraphSLAM::optimize(n) on flat arrays (src/slam/graph_slam.cpp:561-575).
 * chi2_out has iters+1 entries: chi2 before every iteration and after the last.
 * times_out (nullable, 4 doubles): structure+ordering+symbolic, linearise,
 * numeric factor, solve+update.  Returns 0, or -(it+1) if the Cholesky failed
 * in iteration it (poses keep the last successful update, like g2o's early
 * return). */
int cgo_gn_optimize(int nV, double *poses, const uint8_t *fixed, int nE, const int32_t *ef,
                    const int32_t *et, const double *meas, const double *info, int iters,
                    double *chi2_out, double *times_out) {
  gn_sys_t S;
  double t0 = now_s(), tl = 0, tf = 0, ts = 0;
  gn_sys_build(&S, nV, fixed, nE, ef, et);
  double tb = now_s() - t0;
  int status = 0;
  if (chi2_out) chi2_out[0] = cgo_chi2(nE, poses, ef, et, meas, info);
  for (int it = 0; it < iters; it++) {
    if (S.nfree == 0) { if (chi2_out) chi2_out[it + 1] = chi2_out[it]; continue; }
    double t1 = now_s();
    gn_sys_linearize(&S, poses, ef, et, meas, info);
    double t2 = now_s();
    int st = gn_sys_factor(&S);
    double t3 = now_s();
    if (st != 0) {
      status = -(it + 1);
      if (chi2_out) for (int q = it; q < iters; q++) chi2_out[q + 1] = chi2_out[it];
      break;
    }
    memcpy(S.x, S.b, sizeof(double) * S.C.n);
    l_solve(&S, S.x);
    lt_solve(&S, S.x);
    for (int v = 0; v < nV; v++) {                     /* VertexSE2::oplusImpl */
      int h = S.hidx[v];
      if (h < 0) continue;
      const double *d = S.x + 3 * S.iperm[h];
      poses[3 * v] += d[0];
      poses[3 * v + 1] += d[1];
      poses[3 * v + 2] = normalize_theta(poses[3 * v + 2] + d[2]);
    }
    double t4 = now_s();
    if (chi2_out) chi2_out[it + 1] = cgo_chi2(nE, poses, ef, et, meas, info);
    tl += t2 - t1; tf += t3 - t2; ts += t4 - t3;
  }
  if (times_out) { times_out[0] = tb; times_out[1] = tl; times_out[2] = tf; times_out[3] = ts; }
  if (S.nfree) gn_sys_free(&S); else free(S.hidx);
  return status;
}

/* nnz(L) (scalars) and nnz(H upper) for reporting */
int cgo_gn_symbolic_stats(int nV, const uint8_t *fixed, int nE, const int32_t *ef, const int32_t *et,
                          int64_t *nnzL, int64_t *nnzH) {
  gn_sys_t S;
  gn_sys_build(&S, nV, fixed, nE, ef, et);
  if (S.nfree == 0) { *nnzL = 0; *nnzH = 0; free(S.hidx); return 0; }
  *nnzL = S.Lp[S.C.n];
  *nnzH = S.C.p[S.C.n];
  gn_sys_free(&S);
  return 0;
}

/* ------------------------------------------------- spanning-tree initial guess */

/* SparseOptimizer::computeInitialGuess with unit edge cost [g2o-recalled]:
 * breadth-first from the fixed vertices over the given edges (in edge order for
 * ties), x_to = x_from * z or x_from = x_to * z^-1.  Vertices not reached keep
 * their estimate. */
void cgo_initial_guess(int nV, double *poses, const uint8_t *fixed, int nE, const int32_t *ef,
                       const int32_t *et, const double *meas) {
  int *deg = (int *)calloc(nV + 1, sizeof(int));
  for (int k = 0; k < nE; k++) { deg[ef[k] + 1]++; deg[et[k] + 1]++; }
  for (int v = 0; v < nV; v++) deg[v + 1] += deg[v];
  int *inc = (int *)malloc(sizeof(int) * (2 * nE > 0 ? 2 * nE : 1));
  int *pos = (int *)malloc(sizeof(int) * nV);
  for (int v = 0; v < nV; v++) pos[v] = deg[v];
  for (int k = 0; k < nE; k++) { inc[pos[ef[k]]++] = k; inc[pos[et[k]]++] = k; }
  uint8_t *seen = (uint8_t *)calloc(nV, 1);
  int *queue = (int *)malloc(sizeof(int) * nV);
  int qh = 0, qt = 0;
  for (int v = 0; v < nV; v++) if (fixed[v] && deg[v + 1] > deg[v]) { seen[v] = 1; queue[qt++] = v; }
  while (qh < qt) {
    int u = queue[qh++];
    for (int p = deg[u]; p < deg[u + 1]; p++) {
      int k = inc[p];
      int w = (ef[k] == u) ? et[k] : ef[k];
      if (seen[w]) continue;
      seen[w] = 1;
      if (ef[k] == u) se2_mul(poses + 3 * u, meas + 3 * k, poses + 3 * w);
      else { double zi[3]; se2_inv(meas + 3 * k, zi); se2_mul(poses + 3 * u, zi, poses + 3 * w); }
      queue[qt++] = w;
    }
  }
  free(deg); free(inc); free(pos); free(seen); free(queue);
}

/* -------------------------------------------------------- marginal covariances */

/* 3x3 diagonal blocks of H^-1 for the query vertices, H linearised at `poses`
 * (computeMarginals uses the Hessian of the last buildSystem, i.e. *before* the
 * update of that iteration [g2o-recalled]).  g2o evaluates the entries with the
 * memoised Takahashi recursion on L (MarginalCovarianceCholesky); here every
 * requested block column is obtained by two triangular solves with the same L,
 * which is the same quantity.  Fixed / inactive query vertices get zeros.
 * cov_out: nK*9 row-major. */
int cgo_marginals(int nV, const double *poses, const uint8_t *fixed, int nE, const int32_t *ef,
                  const int32_t *et, const double *meas, const double *info, int nK,
                  const int32_t *query, double *cov_out) {
  gn_sys_t S;
  gn_sys_build(&S, nV, fixed, nE, ef, et);
  memset(cov_out, 0, sizeof(double) * 9 * nK);
  if (S.nfree == 0) { free(S.hidx); return 0; }
  gn_sys_linearize(&S, poses, ef, et, meas, info);
  int st = gn_sys_factor(&S);
  if (st != 0) { gn_sys_free(&S); return -1; }
  int n = S.C.n;
  double *col = (double *)malloc(sizeof(double) * n);
  for (int k = 0; k < nK; k++) {
    int h = S.hidx[query[k]];
    if (h < 0) continue;
    int c = S.iperm[h];
    for (int cc = 0; cc < 3; cc++) {
      memset(col, 0, sizeof(double) * n);
      col[3 * c + cc] = 1.0;
      l_solve(&S, col);
      lt_solve(&S, col);
      for (int rr = 0; rr < 3; rr++) cov_out[9 * k + 3 * rr + cc] = col[3 * c + rr];
    }
  }
  free(col);
  gn_sys_free(&S);
  return 0;
}

/* ------------------------------------------------------ condensed measurements */

static int chol3(const double *A, double *L) {      /* lower Cholesky of 3x3 row-major */
  memset(L, 0, sizeof(double) * 9);
  for (int j = 0; j < 3; j++) {
    double d = A[3 * j + j];
    for (int k = 0; k < j; k++) d -= L[3 * j + k] * L[3 * j + k];
    if (!(d > 0)) return -1;
    L[3 * j + j] = sqrt(d);
    for (int i = j + 1; i < 3; i++) {
      double s = A[3 * i + j];
      for (int k = 0; k < j; k++) s -= L[3 * i + k] * L[3 * j + k];
      L[3 * i + j] = s / L[3 * j + j];
    }
  }
  return 0;
}

static int inv3(const double *A, double *B) {
  double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
  double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (det == 0 || det != det) return -1;
  double id = 1.0 / det;
  B[0] = (e * i - f * h) * id; B[1] = (c * h - b * i) * id; B[2] = (b * f - c * e) * id;
  B[3] = (f * g - d * i) * id; B[4] = (a * i - c * g) * id; B[5] = (c * d - a * f) * id;
  B[6] = (d * h - e * g) * id; B[7] = (b * g - a * h) * id; B[8] = (a * e - b * d) * id;
  return 0;
}

/* EdgeLabeler::labelEdge for a star edge gauge->v with the gauge fixed
 * [g2o-recalled, SURVEY.md Appendix A]: measurement := xg^-1 * xv, information
 * := (unscented-transform covariance of the edge error under N(0, cov_vv))^-1.
 * Returns 0, or -1 when the LLT of the scaled covariance fails (g2o then leaves
 * the edge unlabeled: identity information). */
int cgo_label_edge(const double *xg, const double *xv, const double *cov, double *meas_out,
                   double *info_upper_out) {
  const int dim = 3;
  const double alpha = 1e-3, beta = 2.0;
  const double lambda = alpha * alpha * dim;
  const double wi = 1.0 / (2.0 * (dim + lambda));
  const double wm0 = lambda / (dim + lambda);
  const double wc0 = wm0 + (1.0 - alpha * alpha + beta);
  double xgi[3], z[3];
  se2_inv(xg, xgi);
  se2_mul(xgi, xv, z);                       /* setMeasurementFromState */
  meas_out[0] = z[0]; meas_out[1] = z[1]; meas_out[2] = z[2];
  double sc[9], L[9];
  for (int k = 0; k < 9; k++) sc[k] = cov[k] * (dim + lambda);
  if (chol3(sc, L) != 0) {
    info_upper_out[0] = 1; info_upper_out[1] = 0; info_upper_out[2] = 0;
    info_upper_out[3] = 1; info_upper_out[4] = 0; info_upper_out[5] = 1;
    return -1;
  }
  double pts[7][3], wm[7], wc[7], err[7][3];
  pts[0][0] = pts[0][1] = pts[0][2] = 0; wm[0] = wm0; wc[0] = wc0;
  int k = 1;
  for (int i = 0; i < 3; i++) {
    for (int sgn = 0; sgn < 2; sgn++) {
      for (int r = 0; r < 3; r++) pts[k][r] = (sgn ? -1.0 : 1.0) * L[3 * r + i];
      wm[k] = wi; wc[k] = wi; k++;
    }
  }
  for (int q = 0; q < 7; q++) {
    double xs[3] = {xv[0] + pts[q][0], xv[1] + pts[q][1], normalize_theta(xv[2] + pts[q][2])};
    edge_error(xg, xs, z, err[q]);
  }
  double mean[3] = {0, 0, 0}, C[9] = {0};
  for (int q = 0; q < 7; q++) for (int r = 0; r < 3; r++) mean[r] += wm[q] * err[q][r];
  for (int q = 0; q < 7; q++) {
    double d[3] = {err[q][0] - mean[0], err[q][1] - mean[1], err[q][2] - mean[2]};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[3 * r + c] += wc[q] * d[r] * d[c];
  }
  double I[9];
  if (inv3(C, I) != 0) return -2;
  info_upper_out[0] = I[0]; info_upper_out[1] = I[1]; info_upper_out[2] = I[2];
  info_upper_out[3] = I[4]; info_upper_out[4] = I[5]; info_upper_out[5] = I[8];
  return 0;
}

/* CondensedGraphCreator::compute on flat arrays
 * (src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66 with
 * GraphManipulator::{pushState,fixGauge,optimize,popState},
 * src/slam/graph_manipulator.cpp:62-124):
 *   work on a copy of the poses (push/pop), fix exactly the gauge, spanning
 *   tree initial guess over the given (own) edges, one GN iteration, marginals
 *   of the Hessian of that iteration, then label the star edges gauge->query[k]
 *   at the post-update estimates.
 * query[] holds nK vertex indices including the gauge; outputs are written for
 * the nK-1 non-gauge entries in query order: est_out[(nK-1)*3],
 * info_out[(nK-1)*6], to_out[nK-1] (vertex index).  cov_out (nullable) receives
 * the nK-1 marginal blocks.  Returns the number of edges or <0. */
int cgo_condense(int nV, const double *poses_in, int nE, const int32_t *ef, const int32_t *et,
                 const double *meas, const double *info, int gauge, int nK, const int32_t *query,
                 int32_t *to_out, double *est_out, double *info_out, double *cov_out) {
  double *poses = (double *)malloc(sizeof(double) * 3 * nV);
  memcpy(poses, poses_in, sizeof(double) * 3 * nV);
  uint8_t *fixed = (uint8_t *)calloc(nV, 1);
  fixed[gauge] = 1;
  cgo_initial_guess(nV, poses, fixed, nE, ef, et, meas);
  double *lin = (double *)malloc(sizeof(double) * 3 * nV);
  memcpy(lin, poses, sizeof(double) * 3 * nV);       /* linearisation point of the iteration */
  int st = cgo_gn_optimize(nV, poses, fixed, nE, ef, et, meas, info, 1, NULL, NULL);
  int nout = 0;
  if (st == 0) {
    int32_t *q2 = (int32_t *)malloc(sizeof(int32_t) * nK);
    for (int k = 0; k < nK; k++) if (query[k] != gauge) q2[nout++] = query[k];
    double *cov = (double *)malloc(sizeof(double) * 9 * (nout ? nout : 1));
    st = cgo_marginals(nV, lin, fixed, nE, ef, et, meas, info, nout, q2, cov);
    if (st == 0) {
      for (int k = 0; k < nout; k++) {
        to_out[k] = q2[k];
        cgo_label_edge(poses + 3 * gauge, poses + 3 * q2[k], cov + 9 * k, est_out + 3 * k, info_out + 6 * k);
      }
      if (cov_out) memcpy(cov_out, cov, sizeof(double) * 9 * nout);
    }
    free(cov); free(q2);
  }
  free(poses); free(fixed); free(lin);
  return st == 0 ? nout : st;
}

/* CovarianceEstimator::compute (src/slam/graph_manipulator.cpp:128-145): gauge
 * fixed, all other vertices free, spanning-tree init over all edges, one GN
 * iteration, marginals of that iteration's Hessian. */
int cgo_covariance_estimate(int nV, const double *poses_in, int nE, const int32_t *ef, const int32_t *et,
                            const double *meas, const double *info, int gauge, int nK,
                            const int32_t *query, double *cov_out) {
  double *poses = (double *)malloc(sizeof(double) * 3 * nV);
  memcpy(poses, poses_in, sizeof(double) * 3 * nV);
  uint8_t *fixed = (uint8_t *)calloc(nV, 1);
  fixed[gauge] = 1;
  cgo_initial_guess(nV, poses, fixed, nE, ef, et, meas);
  int st = cgo_marginals(nV, poses, fixed, nE, ef, et, meas, info, nK, query, cov_out);
  free(poses); free(fixed);
  return st;
}
