/* lbm_fast.c -- blocked, SIMD-friendly OpenMP twin of the periodic-box sweep: the CPU baseline of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY (like lbm_oracle.c): used by tests/ and bench.py's cpu_baseline leg, never by the
 * product.  The reference has no CPU compute path (backend_dummy.py:31-50 is a stub), so the
 * reported baseline is this restatement ("kind": "port") of what its device code does for a fluid-only periodic
 * box: BGK collide-and-stream, in-place AA pattern (templates/propagation.mako:384-421, geo_helpers.mako:248-276;
 * relaxation.mako:99-181; sym_equilibrium.py:90-120), D3Q19 or D2Q9.
 *
 * Same memory layout (dist[q][z][y][x], x padded, one ghost layer) and the same IEEE operation order as the
 * table-driven oracle and the gfx950 kernels (-ffp-contract=off), so the three are bit-identical:
 * tests/test_cpu_twin.py.  What makes it fast: one (y, z) row per work item, per-direction row pointers, an
 * `omp simd` loop over the interior x with the direction loops fully unrolled (constant tables), the two wrapped
 * end nodes done separately; rows are distributed statically over the threads and the arrays are first touched
 * with the same distribution (fast_copy), so every thread streams through NUMA-local memory.
 */
#include <stddef.h>
#include <stdint.h>
#include <omp.h>

#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;

#define MAXQ 19

typedef struct {
  int dim, Q;
  int e[MAXQ][3];
  int opp[MAXQ];
  int wnum[MAXQ], wden[MAXQ];
} lat_t;

/* direction order / opposite table / weights: sailfish/sym.py:61-75 (D2Q9), sym.py:312-329 (D3Q19) */
static const lat_t D2Q9 = {2, 9,
    {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0}, {1, 1, 0}, {-1, 1, 0}, {-1, -1, 0}, {1, -1, 0}},
    {0, 3, 4, 1, 2, 7, 8, 5, 6},
    {4, 1, 1, 1, 1, 1, 1, 1, 1}, {9, 9, 9, 9, 9, 36, 36, 36, 36}};
static const lat_t D3Q19 = {3, 19,
    {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 1, 0}, {-1, 1, 0}, {1, -1, 0},
     {-1, -1, 0}, {0, 1, 1}, {0, -1, 1}, {0, 1, -1}, {0, -1, -1}, {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1}},
    {0, 2, 1, 4, 3, 6, 5, 10, 9, 8, 7, 14, 13, 12, 11, 18, 17, 16, 15},
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
    {3, 18, 18, 18, 18, 18, 18, 36, 36, 36, 36, 36, 36, 36, 36, 36, 36, 36, 36}};

typedef struct {
  int nx, ny, nz;            /* real nodes */
  long sx, sy, sz;           /* strides: 1, arr_nx, arr_nx * arr_ny */
  size_t ds;                 /* stride between direction arrays */
  real omega;
  real w[MAXQ];
} box_t;

/* one node, generic addressing (used for the two wrapped ends of a row); same operation order as the row loop */
#define NODE_BODY(Q_, L_)                                                                       \
  real rho = f[0];                                                                              \
  for (int i = 1; i < Q_; i++) rho = rho + f[i];                                                \
  real v[3] = {0, 0, 0};                                                                        \
  for (int d = 0; d < L_->dim; d++) {                                                           \
    real acc = (real)0;                                                                         \
    for (int i = 1; i < Q_; i++) {                                                              \
      if (L_->e[i][d] > 0) acc = acc + f[i];                                                    \
      if (L_->e[i][d] < 0) acc = acc - f[i];                                                    \
    }                                                                                           \
    v[d] = acc / rho;                                                                           \
  }                                                                                             \
  real s = v[0] * v[0] + v[1] * v[1];                                                           \
  if (L_->dim == 3) s = s + v[2] * v[2];                                                        \
  const real u15 = (real)1.5 * s;                                                               \
  for (int i = 0; i < Q_; i++) {                                                                \
    real eu = (real)0;                                                                          \
    for (int d = 0; d < L_->dim; d++) {                                                         \
      if (L_->e[i][d] > 0) eu = eu + v[d];                                                      \
      if (L_->e[i][d] < 0) eu = eu - v[d];                                                      \
    }                                                                                           \
    const real fe = b->w[i] * (rho + rho * (eu * ((real)3 + (real)4.5 * eu) - u15));            \
    f[i] = f[i] + b->omega * (fe - f[i]);                                                       \
  }

static inline int wrap1(int c, int n) { return c < 1 ? n : (c > n ? 1 : c); }

#define DEFINE_STEP(NAME, LAT, Q_)                                                                                  \
  static void NAME(const box_t* b, real* dist, int odd, real* orho, real* ovx, real* ovy, real* ovz) {            \
    const lat_t* L = &LAT;                                                                                          \
    const int nx = b->nx, ny = b->ny, nz = (L->dim == 3) ? b->nz : 1;                                               \
    const int zlo = (L->dim == 3) ? 1 : 0;                                                                          \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                                        \
    for (int zz = 0; zz < nz; zz++)                                                                                 \
      for (int y = 1; y <= ny; y++) {                                                                               \
        const int z = zz + zlo;                                                                                     \
        /* row pointers: in[i] = where f_i of node x is read, out[i] = where the post-collision f_i goes */       \
        const real* in[Q_];                                                                                         \
        real* out[Q_];                                                                                              \
        for (int i = 0; i < Q_; i++) {                                                                              \
          if (!odd) {                                                                                               \
            in[i] = dist + b->ds * (size_t)i + (size_t)(b->sy * y + b->sz * z);                                     \
            out[i] = dist + b->ds * (size_t)L->opp[i] + (size_t)(b->sy * y + b->sz * z);                            \
          } else {                                                                                                  \
            const int ys = wrap1(y - L->e[i][1], ny), yt = wrap1(y + L->e[i][1], ny);                               \
            const int zs = (L->dim == 3) ? wrap1(z - L->e[i][2], b->nz) : 0;                                        \
            const int zt = (L->dim == 3) ? wrap1(z + L->e[i][2], b->nz) : 0;                                        \
            in[i] = dist + b->ds * (size_t)L->opp[i] + (size_t)(b->sy * ys + b->sz * zs) - L->e[i][0];              \
            out[i] = dist + b->ds * (size_t)i + (size_t)(b->sy * yt + b->sz * zt) + L->e[i][0];                     \
          }                                                                                                         \
        }                                                                                                           \
        const size_t row = (size_t)(b->sy * y + b->sz * z);                                                         \
        /* interior x: no wrap, stride-1, vectorisable */                                                           \
        const int x0 = odd ? 2 : 1, x1 = odd ? nx - 1 : nx;                                                         \
        _Pragma("omp simd")                                                                                         \
        for (int x = x0; x <= x1; x++) {                                                                            \
          real f[Q_];                                                                                               \
          for (int i = 0; i < Q_; i++) f[i] = in[i][x];                                                             \
          NODE_BODY(Q_, L)                                                                                          \
          if (orho) {                                                                                               \
            orho[row + x] = rho;                                                                                    \
            ovx[row + x] = v[0];                                                                                    \
            ovy[row + x] = v[1];                                                                                    \
            if (L->dim == 3) ovz[row + x] = v[2];                                                                   \
          }                                                                                                         \
          for (int i = 0; i < Q_; i++) out[i][x] = f[i];                                                            \
        }                                                                                                           \
        if (odd) { /* the two ends of the row: x -+ e_x wraps around */                                             \
          for (int k = 0; k < 2; k++) {                                                                             \
            const int x = k ? nx : 1;                                                                               \
            if (k && nx == 1) break;                                                                                \
            real f[Q_];                                                                                             \
            for (int i = 0; i < Q_; i++) f[i] = (in[i] + L->e[i][0])[wrap1(x - L->e[i][0], nx)];                    \
            NODE_BODY(Q_, L)                                                                                        \
            if (orho) {                                                                                             \
              orho[row + x] = rho;                                                                                  \
              ovx[row + x] = v[0];                                                                                  \
              ovy[row + x] = v[1];                                                                                  \
              if (L->dim == 3) ovz[row + x] = v[2];                                                                 \
            }                                                                                                       \
            for (int i = 0; i < Q_; i++) (out[i] - L->e[i][0])[wrap1(x + L->e[i][0], nx)] = f[i];                   \
          }                                                                                                         \
        }                                                                                                           \
      }                                                                                                             \
  }

DEFINE_STEP(step_d3q19, D3Q19, 19)
DEFINE_STEP(step_d2q9, D2Q9, 9)

static void make_box(box_t* b, const lat_t* L, int nx, int ny, int nz, int arr_nx, double tau) {
  b->nx = nx; b->ny = ny; b->nz = nz;
  b->sx = 1; b->sy = arr_nx; b->sz = (long)arr_nx * (ny + 2);
  b->ds = (size_t)arr_nx * (size_t)(ny + 2) * (size_t)(L->dim == 3 ? nz + 2 : 1);
  b->omega = (real)(1.0 / tau);
  for (int i = 0; i < L->Q; i++) b->w[i] = (real)((double)L->wnum[i] / (double)L->wden[i]);
}

int fast_real_size(void) { return (int)sizeof(real); }
int fast_max_threads(void) { return omp_get_max_threads(); }
void fast_set_threads(int n) { omp_set_num_threads(n); }

/* `steps` AA steps starting at iteration `it0` (even iteration = in-place opposite-slot step, odd = pull + push);
 * lattice: 0 D2Q9, 1 D3Q19; the macroscopic fields of the LAST step are stored when rho != NULL. */
void fast_run(int lattice, int nx, int ny, int nz, int arr_nx, double tau, real* dist, int it0, int steps, real* rho,
              real* vx, real* vy, real* vz) {
  box_t b;
  make_box(&b, lattice ? &D3Q19 : &D2Q9, nx, ny, nz, arr_nx, tau);
  for (int s = 0; s < steps; s++) {
    const int last = (s == steps - 1) && rho;
    if (lattice) step_d3q19(&b, dist, (it0 + s) & 1, last ? rho : 0, vx, vy, vz);
    else step_d2q9(&b, dist, (it0 + s) & 1, last ? rho : 0, vx, vy, vz);
  }
}

/* dst = src over the whole padded array, rows distributed like the sweep (first touch = NUMA placement) */
void fast_copy(int lattice, int ny, int nz, int arr_nx, real* dst, const real* src) {
  const int Q = lattice ? 19 : 9;
  const long rows = (long)(ny + 2) * (lattice ? nz + 2 : 1);
  const size_t ds = (size_t)arr_nx * (size_t)rows;
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; r++)
    for (int q = 0; q < Q; q++) {
      const size_t o = ds * (size_t)q + (size_t)r * (size_t)arr_nx;
      for (int x = 0; x < arr_nx; x++) dst[o + x] = src[o + x];
    }
}
