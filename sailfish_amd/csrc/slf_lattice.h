// Lattice descriptors for the gfx950 kernels (compile-time tables).
//
// Direction order, opposite table and weights restate the reference's
// sailfish/sym.py:61-75 (D2Q9) and sym.py:312-329 (D3Q19); they are pinned by
// tests/golden/lattices.json (generated from the reference by
// tools/capture_goldens.py) through tests/test_lattice_tables.py.
//
// All tables are constexpr functions so that, with the static_for<> helper,
// every e_i component / weight is a literal inside the fully unrolled per-node
// code (no table loads, no 0*f multiplies surviving in floating point).
#pragma once
#include <stdint.h>
#include <type_traits>

#ifdef __HIPCC__
#define SLF_HD __host__ __device__ inline
#define SLF_D __device__ __forceinline__
#else
#define SLF_HD inline
#define SLF_D inline
#endif

namespace slf {

template <int I, int N, class F>
SLF_HD void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(static_cast<F&&>(f));
  }
}

struct D2Q9 {
  static constexpr int dim = 2;
  static constexpr int Q = 9;
  static constexpr int id = 0;
  // fC fE fN fW fS fNE fNW fSW fSE
  static constexpr int ex(int i) { constexpr int t[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}; return t[i]; }
  static constexpr int ey(int i) { constexpr int t[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1}; return t[i]; }
  static constexpr int ez(int) { return 0; }
  static constexpr int opp(int i) { constexpr int t[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6}; return t[i]; }
  // weights as rationals num/den
  static constexpr int wnum(int i) { return i == 0 ? 4 : 1; }
  static constexpr int wden(int i) { return i < 5 ? 9 : 36; }
  // orientation code (1..2*dim) -> basis index of the inward normal (sym.py:1013-1018)
  static constexpr int dir2vecidx(int o) { return o; }
  // MRT (sym.py:78-149): integer Gram-Schmidt rows, squared norms, fixed rates
  static constexpr int mrt(int k, int i) {
    constexpr int t[9][9] = {
        {1, 1, 1, 1, 1, 1, 1, 1, 1},      {-4, -1, -1, -1, -1, 2, 2, 2, 2}, {4, -2, -2, -2, -2, 1, 1, 1, 1},
        {0, 1, 0, -1, 0, 1, -1, -1, 1},   {0, -2, 0, 2, 0, 1, -1, -1, 1},   {0, 0, 1, 0, -1, 1, 1, -1, -1},
        {0, 0, -2, 0, 2, 1, 1, -1, -1},   {0, 1, -1, 1, -1, 0, 0, 0, 0},    {0, 0, 0, 0, 0, 1, -1, 1, -1}};
    return t[k][i];
  }
  static constexpr int mrt_norm(int k) { constexpr int t[9] = {9, 36, 36, 6, 12, 6, 12, 4, 4}; return t[k]; }
  // index of moments: rho en ens mx ex my ey pxx pxy
  static constexpr int M_RHO = 0, M_MX = 3, M_MY = 5, M_MZ = -1;
};

struct D3Q19 {
  static constexpr int dim = 3;
  static constexpr int Q = 19;
  static constexpr int id = 1;
  // fC fE fW fN fS fT fB fNE fNW fSE fSW fTN fTS fBN fBS fTE fTW fBE fBW
  static constexpr int ex(int i) {
    constexpr int t[19] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1};
    return t[i];
  }
  static constexpr int ey(int i) {
    constexpr int t[19] = {0, 0, 0, 1, -1, 0, 0, 1, 1, -1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
    return t[i];
  }
  static constexpr int ez(int i) {
    constexpr int t[19] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, 1, -1, -1, 1, 1, -1, -1};
    return t[i];
  }
  static constexpr int opp(int i) {
    constexpr int t[19] = {0, 2, 1, 4, 3, 6, 5, 10, 9, 8, 7, 14, 13, 12, 11, 18, 17, 16, 15};
    return t[i];
  }
  static constexpr int wnum(int) { return 1; }
  static constexpr int wden(int i) { return i == 0 ? 3 : (i < 7 ? 18 : 36); }
  static constexpr int dir2vecidx(int o) { return o; }
  static constexpr int mrt(int k, int i) {
    constexpr int t[19][19] = {
        {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
        {-30, -11, -11, -11, -11, -11, -11, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8},
        {12, -4, -4, -4, -4, -4, -4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
        {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1},
        {0, -4, 4, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1},
        {0, 0, 0, 1, -1, 0, 0, 1, 1, -1, -1, 1, -1, 1, -1, 0, 0, 0, 0},
        {0, 0, 0, -4, 4, 0, 0, 1, 1, -1, -1, 1, -1, 1, -1, 0, 0, 0, 0},
        {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, 1, -1, -1, 1, 1, -1, -1},
        {0, 0, 0, 0, 0, -4, 4, 0, 0, 0, 0, 1, 1, -1, -1, 1, 1, -1, -1},
        {0, 2, 2, -1, -1, -1, -1, 1, 1, 1, 1, -2, -2, -2, -2, 1, 1, 1, 1},
        {0, -4, -4, 2, 2, 2, 2, 1, 1, 1, 1, -2, -2, -2, -2, 1, 1, 1, 1},
        {0, 0, 0, 1, 1, -1, -1, 1, 1, 1, 1, 0, 0, 0, 0, -1, -1, -1, -1},
        {0, 0, 0, -2, -2, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, -1, -1, -1, -1},
        {0, 0, 0, 0, 0, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 0, 0, 0, 0},
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0},
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, -1, -1, 1},
        {0, 0, 0, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 0, 0, -1, 1, -1, 1},
        {0, 0, 0, 0, 0, 0, 0, -1, -1, 1, 1, 1, -1, 1, -1, 0, 0, 0, 0},
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, -1, 1, 1, 1, 1, -1, -1}};
    return t[k][i];
  }
  static constexpr int mrt_norm(int k) {
    constexpr int t[19] = {19, 2394, 252, 10, 40, 10, 40, 10, 40, 36, 72, 12, 24, 4, 4, 4, 8, 8, 8};
    return t[k];
  }
  // rho en eps mx ex my ey mz ez pxx3 pixx3 pww piww pxy pyz pzx m3x m3y m3z
  static constexpr int M_RHO = 0, M_MX = 3, M_MY = 5, M_MZ = 7;
};

template <class L>
constexpr int e_comp(int i, int d) {
  return d == 0 ? L::ex(i) : (d == 1 ? L::ey(i) : L::ez(i));
}

}  // namespace slf
