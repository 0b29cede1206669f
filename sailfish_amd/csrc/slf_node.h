// Per-node arithmetic of the collide-and-stream sweep (registers only).
//
// This is the "arithmetic contract" of DESIGN.md §4: every function below has a
// fixed floating-point operation order, compiled with -ffp-contract=off, which
// the CPU oracle (oracle/lbm_oracle.c) restates independently with table-driven
// loops.  What is computed follows the reference:
//   moments        sailfish/templates/boundary.mako:267-319, sym.py:573-682
//   equilibrium    sailfish/sym_equilibrium.py:90-120
//   BGK            sailfish/templates/relaxation.mako:99-181
//   Guo forcing    sailfish/templates/relaxation_common.mako:56-64,110-149, sym_force.py:121-160
//   MRT            sailfish/templates/relaxation_mrt.mako:31-97, sym.py:78-149,331-406,716-735
//   BCs            sailfish/templates/boundary.mako:255-264,330-340,425-459,784-835
#pragma once
#include "slf_lattice.h"

namespace slf {

// Canonical node kinds the kernels understand.  The per-subdomain dense type
// ids produced by the geometry encoder (reference geo_encoder.py:83-91) are
// mapped to these through a 4-bit LUT passed with the kernel parameters.
enum NodeKind : int {
  NK_FLUID = 0,
  NK_GHOST = 1,             // excluded
  NK_UNUSED = 2,            // excluded
  NK_PROPAGATION_ONLY = 3,  // no work in this design (no LDS hand-shake needed)
  NK_FULL_BB = 4,
  NK_HALF_BB = 5,
  NK_REGULARIZED_VELOCITY = 6,
  NK_EQUILIBRIUM_DENSITY = 7,
  NK_EQUILIBRIUM_VELOCITY = 8,
  NK_ZOUHE_VELOCITY = 9,
  NK_ZOUHE_DENSITY = 10,
  NK_REGULARIZED_DENSITY = 11,
  NK_COPY = 12,
  NK_YU_OUTFLOW = 13,
  NK_DO_NOTHING = 14,       // in-place pattern: keeps its unknown populations (slf_sweep.h); two-copy pattern: a fluid node
  NK_SLIP = 15,             // dry: specular reflection (slip_reflect)
  NK_COUNT = 16
};

SLF_HD bool kind_is_wet(int k) {
  return k == NK_FLUID || k == NK_HALF_BB || k == NK_REGULARIZED_VELOCITY || k == NK_EQUILIBRIUM_DENSITY ||
         k == NK_EQUILIBRIUM_VELOCITY || k == NK_ZOUHE_VELOCITY || k == NK_ZOUHE_DENSITY ||
         k == NK_REGULARIZED_DENSITY || k == NK_COPY || k == NK_YU_OUTFLOW || k == NK_DO_NOTHING;
}
SLF_HD bool kind_is_excluded(int k) { return k == NK_GHOST || k == NK_UNUSED || k == NK_PROPAGATION_ONLY; }

template <class L, class R>
struct Weights {
  static constexpr R w(int i) { return (R)((double)L::wnum(i) / (double)L::wden(i)); }
  static constexpr R w6(int i) { return (R)(6.0 * (double)L::wnum(i) / (double)L::wden(i)); }
  static constexpr R w45(int i) { return (R)(4.5 * (double)L::wnum(i) / (double)L::wden(i)); }
};

// C1: rho = ((f0 + f1) + f2) + ...
template <class L, class R>
SLF_D R density(const R (&f)[L::Q]) {
  R rho = f[0];
  static_for<1, L::Q>([&](auto I) { rho = rho + f[I]; });
  return rho;
}

// C2: j_d = sum_i e_id f_i, index order, +-1 as add/sub.
template <class L, class R, int D>
SLF_D R momentum(const R (&f)[L::Q]) {
  R acc = (R)0;
  static_for<1, L::Q>([&](auto I) {
    constexpr int e = e_comp<L>(I, D);
    if constexpr (e > 0) acc = acc + f[I];
    if constexpr (e < 0) acc = acc - f[I];
  });
  return acc;
}

// C1-C3: standard macroscopic quantities.
template <class L, class R>
SLF_D void macro_standard(const R (&f)[L::Q], bool incompressible, R& rho, R (&v)[3]) {
  rho = density<L, R>(f);
  v[0] = momentum<L, R, 0>(f);
  v[1] = momentum<L, R, 1>(f);
  v[2] = (R)0;
  if constexpr (L::dim == 3) v[2] = momentum<L, R, 2>(f);
  if (!incompressible) {
    v[0] = v[0] / rho;
    v[1] = v[1] / rho;
    if constexpr (L::dim == 3) v[2] = v[2] / rho;
  }
}

template <class L, class R>
SLF_D R usq15(const R (&v)[3]) {
  R s = v[0] * v[0] + v[1] * v[1];
  if constexpr (L::dim == 3) s = s + v[2] * v[2];
  return (R)1.5 * s;
}

// e_i . v with the components taken in x,y,z order.
template <class L, class R, int I>
SLF_D R edotv(const R (&v)[3]) {
  R acc = (R)0;
  static_for<0, L::dim>([&](auto D) {
    constexpr int e = e_comp<L>(I, D);
    if constexpr (e > 0) acc = acc + v[D];
    if constexpr (e < 0) acc = acc - v[D];
  });
  return acc;
}

// C4: feq_i = w_i (rho + rho0 (eu (3 + 4.5 eu) - 1.5 u^2))
template <class L, class R, int I>
SLF_D R feq(R rho, R rho0, const R (&v)[3], R u15) {
  constexpr bool rest = (L::ex(I) == 0 && L::ey(I) == 0 && L::ez(I) == 0);
  if constexpr (rest) {
    return Weights<L, R>::w(I) * (rho + rho0 * ((R)0 - u15));
  } else {
    const R eu = edotv<L, R, I>(v);
    return Weights<L, R>::w(I) * (rho + rho0 * (eu * ((R)3 + (R)4.5 * eu) - u15));
  }
}

template <class L, class R>
struct CollideParams {
  R omega;        // 1/tau
  R mrt_s[L::Q];  // MRT relaxation rates (0 for conserved moments)
  R accel[3];     // body-force acceleration
  R guo_pref;     // 3 (1 - 1/(2 tau))
  int incompressible;
  int has_force;
  int force_edm;  // exact difference method instead of Guo forcing
};

// C5 + C6: BGK relaxation with optional Guo forcing by the acceleration a.  v is updated to the
// velocity used for the equilibrium (u + a/2), which is also the reference's output velocity.
template <class L, class R>
SLF_D void bgk_relax_accel(R (&f)[L::Q], R rho, R (&v)[3], R omega, R guo_pref, bool incompressible, bool has_force,
                           const R (&a)[3], bool edm = false) {
  const R rho0 = incompressible ? (R)1 : rho;
  if (has_force && edm) {
    // exact difference method (relaxation_common.mako:66-73, sym_force.py:184-193): equilibrium at the
    // unshifted velocity, then f_i += feq_i(rho, u + a) - feq_i(rho, u); output velocity u + a/2
    R vs[3] = {v[0] + a[0], v[1] + a[1], (R)0};
    if constexpr (L::dim == 3) vs[2] = v[2] + a[2];
    const R u15 = usq15<L, R>(v);
    const R u15s = usq15<L, R>(vs);
    static_for<0, L::Q>([&](auto I) {
      const R fe = feq<L, R, I>(rho, rho0, v, u15);
      const R fs = feq<L, R, I>(rho, rho0, vs, u15s);
      f[I] = f[I] + omega * (fe - f[I]);
      f[I] = f[I] + (fs - fe);
    });
    static_for<0, L::dim>([&](auto D) { v[D] = v[D] + (R)0.5 * a[D]; });
    return;
  }
  if (has_force) {
    static_for<0, L::dim>([&](auto D) { v[D] = v[D] + (R)0.5 * a[D]; });
  }
  const R u15 = usq15<L, R>(v);
  static_for<0, L::Q>([&](auto I) {
    const R fe = feq<L, R, I>(rho, rho0, v, u15);
    f[I] = f[I] + omega * (fe - f[I]);
  });
  if (has_force) {
    const R pref = rho * guo_pref;
    R va = v[0] * a[0] + v[1] * a[1];
    if constexpr (L::dim == 3) va = va + v[2] * a[2];
    static_for<0, L::Q>([&](auto I) {
      const R eu = edotv<L, R, I>(v);
      const R ea = edotv<L, R, I>(a);
      const R t = (ea - va) + (R)3 * eu * ea;
      f[I] = f[I] + pref * Weights<L, R>::w(I) * t;
    });
  }
}

// --minimize_roundoff (reference lb_base.py:72-76, "BGK-like models only"): the arrays hold f_i - w_i, so that the
// O(1) rest-state part never meets the O(Ma) part in one floating-point sum.  The model's density variable is
// drho = rho - 1 = sum_i (f_i - w_i) (sym.py:573-597); velocities divide by drho + 1 (sym.py:654-661); the equilibrium
// is w_i (drho + (drho + 1) h_i(u)) (sym_equilibrium.py:100-118: rho0 = rho + 1); the Guo prefactor carries drho + 1
// (sym_force.py:147-160); SetInitialConditions subtracts 1 from the density field (lb_single_fluid.mako:113) and the
// density output is drho (kernel_common.mako:216-224).
template <class L, class R>
SLF_D void macro_roundoff(const R (&f)[L::Q], R& drho, R (&v)[3]) {
  drho = density<L, R>(f);
  const R rho0 = drho + (R)1;
  v[0] = momentum<L, R, 0>(f) / rho0;
  v[1] = momentum<L, R, 1>(f) / rho0;
  v[2] = (R)0;
  if constexpr (L::dim == 3) v[2] = momentum<L, R, 2>(f) / rho0;
}

template <class L, class R>
SLF_D void bgk_relax_roundoff(R (&f)[L::Q], R drho, R (&v)[3], R omega, R guo_pref, bool has_force, const R (&a)[3],
                              bool edm) {
  const R rho0 = drho + (R)1;
  if (has_force && edm) {
    R vs[3] = {v[0] + a[0], v[1] + a[1], (R)0};
    if constexpr (L::dim == 3) vs[2] = v[2] + a[2];
    const R u15 = usq15<L, R>(v);
    const R u15s = usq15<L, R>(vs);
    static_for<0, L::Q>([&](auto I) {
      const R fe = feq<L, R, I>(drho, rho0, v, u15);
      const R fs = feq<L, R, I>(drho, rho0, vs, u15s);
      f[I] = f[I] + omega * (fe - f[I]);
      f[I] = f[I] + (fs - fe);
    });
    static_for<0, L::dim>([&](auto D) { v[D] = v[D] + (R)0.5 * a[D]; });
    return;
  }
  if (has_force) {
    static_for<0, L::dim>([&](auto D) { v[D] = v[D] + (R)0.5 * a[D]; });
  }
  const R u15 = usq15<L, R>(v);
  static_for<0, L::Q>([&](auto I) {
    const R fe = feq<L, R, I>(drho, rho0, v, u15);
    f[I] = f[I] + omega * (fe - f[I]);
  });
  if (has_force) {
    const R pref = rho0 * guo_pref;
    R va = v[0] * a[0] + v[1] * a[1];
    if constexpr (L::dim == 3) va = va + v[2] * a[2];
    static_for<0, L::Q>([&](auto I) {
      const R eu = edotv<L, R, I>(v);
      const R ea = edotv<L, R, I>(a);
      const R t = (ea - va) + (R)3 * eu * ea;
      f[I] = f[I] + pref * Weights<L, R>::w(I) * t;
    });
  }
}

// FORCE: what the instantiation knows about the body force at compile time.
//   0  the module has none: the Guo / exact-difference code (and the registers its merge points cost: 98 -> 48 VGPRs in
//      the whole-row kernel, 4 -> 8 resident waves per SIMD) does not exist in the instantiation;
//   1  the module has one (the launchers pick this instantiation iff Physics::has_force): no run-time "is there a
//      force" test either -- 95-100 -> 74-82 VGPRs in the node-map row kernels;
//   2  decided at run time (the per-node kernels, which exist once per module kind).
constexpr int FORCE_RUNTIME = 2;
template <class L, class R, int FORCE = FORCE_RUNTIME>
SLF_D void bgk_relax(R (&f)[L::Q], R rho, R (&v)[3], const CollideParams<L, R>& cp) {
  if constexpr (FORCE == 0) {
    bgk_relax_accel<L, R>(f, rho, v, cp.omega, cp.guo_pref, cp.incompressible != 0, false, cp.accel, false);
  } else {
    const bool has_force = (FORCE == 1) ? true : (cp.has_force != 0);
    bgk_relax_accel<L, R>(f, rho, v, cp.omega, cp.guo_pref, cp.incompressible != 0, has_force, cp.accel,
                          cp.force_edm != 0);
  }
}

// C7 helpers: one row of the integer moment matrix applied to a vector.
template <class L, class R, int K>
SLF_D R mrt_row(const R (&f)[L::Q]) {
  R acc = (R)0;
  static_for<0, L::Q>([&](auto I) {
    constexpr int c = L::mrt(K, I);
    if constexpr (c == 1) acc = acc + f[I];
    else if constexpr (c == -1) acc = acc - f[I];
    else if constexpr (c != 0) acc = acc + (R)c * f[I];
  });
  return acc;
}
template <class L, class R, int I>
SLF_D R mrt_col(const R (&m)[L::Q]) {
  R acc = (R)0;
  static_for<0, L::Q>([&](auto K) {
    constexpr int c = L::mrt(K, I);
    if constexpr (c == 1) acc = acc + m[K];
    else if constexpr (c == -1) acc = acc - m[K];
    else if constexpr (c != 0) acc = acc + (R)c * m[K];
  });
  return acc;
}

// D3Q19: the same two products M f and M^T (m / |row|^2), evaluated through the sums and differences of the nine pairs of
// opposite directions -- the even moments (rho, en, eps, the stresses) need only the sums, the odd ones (j, q, the
// third-order m) only the differences: 62 + 85 operations instead of 2 x 270.  (The sweep with the row-by-row product
// kept the vector ALUs 72 % busy on a 512^3 cavity, profiles/r03/sq_summary_cfg2.txt: the collision, not HBM, set its
// pace.)  The order of the operations below is the arithmetic contract with oracle/lbm_oracle.c (mrt_forward_d3q19 /
// mrt_inverse_d3q19): identical, with the same fused multiply-adds; tests/test_oracle_golden.py holds it against the
// matrix form and the reference's sympy values.
template <class R>
SLF_D R fma_(R a, R b, R c) {
  if constexpr (sizeof(R) == 4) return __builtin_fmaf(a, b, c);
  else return __builtin_fma(a, b, c);
}

template <class R>
SLF_D void mrt_forward_d3q19(const R (&f)[19], R (&m)[19]) {
  const R sx = f[1] + f[2], dx = f[1] - f[2], sy = f[3] + f[4], dy = f[3] - f[4], sz = f[5] + f[6], dz = f[5] - f[6];
  const R sA = f[7] + f[10], dA = f[7] - f[10], sB = f[8] + f[9], dB = f[9] - f[8];
  const R sC = f[11] + f[14], dC = f[11] - f[14], sD = f[12] + f[13], dD = f[13] - f[12];
  const R sE = f[15] + f[18], dE = f[15] - f[18], sF = f[16] + f[17], dF = f[17] - f[16];
  // odd moments
  const R xab = dA + dB, yab = dA - dB, ycd = dC + dD, zcd = dC - dD, xef = dE + dF, zef = dE - dF;
  const R X = xab + xef, Y = yab + ycd, Z = zcd + zef;
  m[3] = dx + X;  m[4] = fma_<R>((R)-4, dx, X);  m[16] = xab - xef;
  m[5] = dy + Y;  m[6] = fma_<R>((R)-4, dy, Y);  m[17] = ycd - yab;
  m[7] = dz + Z;  m[8] = fma_<R>((R)-4, dz, Z);  m[18] = zef - zcd;
  // even moments
  const R Sxy = sA + sB, Syz = sC + sD, Szx = sE + sF;
  const R A1 = (sx + sy) + sz, B1 = (Sxy + Syz) + Szx;
  m[0] = (f[0] + A1) + B1;
  m[1] = fma_<R>((R)8, B1, fma_<R>((R)-11, A1, (R)-30 * f[0]));
  m[2] = fma_<R>((R)-4, A1, fma_<R>((R)12, f[0], B1));
  const R P = fma_<R>((R)2, sx, (R)0 - (sy + sz)), Qd = fma_<R>((R)-2, Syz, Sxy + Szx);
  m[9] = P + Qd;   m[10] = fma_<R>((R)-2, P, Qd);
  const R W = sy - sz, V = Sxy - Szx;
  m[11] = W + V;   m[12] = fma_<R>((R)-2, W, V);
  m[13] = sA - sB; m[14] = sC - sD; m[15] = sE - sF;
}

// f = M^T m for moments m already divided by the squared row norms.
template <class R>
SLF_D void mrt_inverse_d3q19(const R (&m)[19], R (&f)[19]) {
  const R K0 = fma_<R>((R)-4, m[2], fma_<R>((R)-11, m[1], m[0]));
  const R K1 = fma_<R>((R)8, m[1], m[0]) + m[2];
  f[0] = fma_<R>((R)12, m[2], fma_<R>((R)-30, m[1], m[0]));
  const R G = fma_<R>((R)2, m[10], (R)0 - m[9]), H = fma_<R>((R)-2, m[12], m[11]);
  const R Ex = fma_<R>((R)-2, G, K0), KG = K0 + G, Ey = KG + H, Ez = KG - H;
  const R Ox = fma_<R>((R)-4, m[4], m[3]), Oy = fma_<R>((R)-4, m[6], m[5]), Oz = fma_<R>((R)-4, m[8], m[7]);
  f[1] = Ex + Ox; f[2] = Ex - Ox; f[3] = Ey + Oy; f[4] = Ey - Oy; f[5] = Ez + Oz; f[6] = Ez - Oz;
  const R S9 = m[9] + m[10], S11 = m[11] + m[12];
  const R K19 = K1 + S9, Txy = K19 + S11, Tzx = K19 - S11, Tyz = fma_<R>((R)-2, S9, K1);
  const R ax = m[3] + m[4], ay = m[5] + m[6], az = m[7] + m[8];
  R ep = Txy + m[13], em = Txy - m[13];
  const R oA = (ax + ay) + (m[16] - m[17]), oB = (ax - ay) + (m[16] + m[17]);
  f[7] = ep + oA; f[10] = ep - oA; f[9] = em + oB; f[8] = em - oB;
  ep = Tyz + m[14]; em = Tyz - m[14];
  const R oC = (ay + az) + (m[17] - m[18]), oD = (ay - az) + (m[17] + m[18]);
  f[11] = ep + oC; f[14] = ep - oC; f[13] = em + oD; f[12] = em - oD;
  ep = Tzx + m[15]; em = Tzx - m[15];
  const R oE = (ax + az) + (m[18] - m[16]), oF = (ax - az) - (m[16] + m[18]);
  f[15] = ep + oE; f[18] = ep - oE; f[17] = em + oF; f[16] = em - oF;
}

template <class L, class R>
SLF_D void mrt_equilibrium(const R (&m)[L::Q], R inv_rho, R (&meq)[L::Q]) {
  static_for<0, L::Q>([&](auto K) { meq[K] = (R)0; });
  const R rho = m[L::M_RHO];
  const R mx = m[L::M_MX], my = m[L::M_MY];
  if constexpr (L::id == D2Q9::id) {
    // sym.py:101-149: en = -2 rho + 3 j^2, ens = rho - 3 j^2, ex = -mx, ey = -my,
    // pxx = mx^2 - my^2, pxy = mx my  (no 1/rho in the reference's D2Q9 basis).
    const R jsq = mx * mx + my * my;
    meq[1] = (R)3 * jsq - (R)2 * rho;
    meq[2] = rho - (R)3 * jsq;
    meq[4] = (R)0 - mx;
    meq[6] = (R)0 - my;
    meq[7] = mx * mx - my * my;
    meq[8] = mx * my;
  } else {
    // sym.py:380-406 (d'Humieres 2002).
    const R mz = m[L::M_MZ];
    const R jsq = (mx * mx + my * my) + mz * mz;
    const R irj = inv_rho * jsq;
    meq[1] = (R)19 * irj - (R)11 * rho;
    meq[2] = (R)(-475.0 / 63.0) * irj;
    meq[4] = (R)(-2.0 / 3.0) * mx;
    meq[6] = (R)(-2.0 / 3.0) * my;
    meq[8] = (R)(-2.0 / 3.0) * mz;
    meq[9] = inv_rho * ((R)2 * (mx * mx) - ((my * my) + (mz * mz)));
    meq[11] = inv_rho * ((my * my) - (mz * mz));
    meq[13] = inv_rho * (mx * my);
    meq[14] = inv_rho * (my * mz);
    meq[15] = inv_rho * (mx * mz);
  }
}

// C7: relaxation in moment space.  force_eq: equilibrium-type node (moments
// forced to equilibrium, relaxation_mrt.mako:58-77).
template <class L, class R, int FORCE = FORCE_RUNTIME>
SLF_D void mrt_relax(R (&f)[L::Q], R (&v)[3], const CollideParams<L, R>& cp, bool force_eq) {
  R m[L::Q];
  if constexpr (L::id == D3Q19::id) mrt_forward_d3q19<R>(f, m);
  else static_for<0, L::Q>([&](auto K) { m[K] = mrt_row<L, R, K>(f); });
  const bool has_force = (FORCE == 1) ? true : ((FORCE == 0) ? false : (cp.has_force != 0));
  if (has_force) {
    m[L::M_MX] = m[L::M_MX] + (R)0.5 * cp.accel[0];
    m[L::M_MY] = m[L::M_MY] + (R)0.5 * cp.accel[1];
    if constexpr (L::dim == 3) m[L::M_MZ] = m[L::M_MZ] + (R)0.5 * cp.accel[2];
  }
  const R inv_rho = cp.incompressible ? (R)1 : (R)1 / m[L::M_RHO];
  R meq[L::Q];
  mrt_equilibrium<L, R>(m, inv_rho, meq);
  static_for<0, L::Q>([&](auto K) {
    constexpr bool conserved =
        (K == L::M_RHO || K == L::M_MX || K == L::M_MY || (L::dim == 3 && K == L::M_MZ));
    if constexpr (!conserved) {
      if (force_eq) m[K] = meq[K];
      m[K] = m[K] - cp.mrt_s[K] * (m[K] - meq[K]);
    }
  });
  if (has_force) {
    m[L::M_MX] = m[L::M_MX] + (R)0.5 * cp.accel[0];
    m[L::M_MY] = m[L::M_MY] + (R)0.5 * cp.accel[1];
    if constexpr (L::dim == 3) m[L::M_MZ] = m[L::M_MZ] + (R)0.5 * cp.accel[2];
  }
  static_for<0, L::Q>([&](auto K) { m[K] = m[K] * (R)(1.0 / (double)L::mrt_norm(K)); });
  if constexpr (L::id == D3Q19::id) mrt_inverse_d3q19<R>(m, f);
  else static_for<0, L::Q>([&](auto I) { f[I] = mrt_col<L, R, I>(m); });
  if (has_force) {
    static_for<0, L::dim>([&](auto D) { v[D] = v[D] + (R)0.5 * cp.accel[D]; });
  }
}

// C8: full-way bounce-back (boundary.mako:255-264).
template <class L, class R>
SLF_D void bounce_back(R (&f)[L::Q]) {
  static_for<1, L::Q>([&](auto I) {
    constexpr int o = L::opp(I);
    if constexpr (I < o) {
      const R t = f[I];
      f[I] = f[o];
      f[o] = t;
    }
  });
}

// The direction whose vector is (x, y, z); -1: the lattice has none.
template <class L>
constexpr int dir_of(int x, int y, int z) {
  for (int i = 0; i < L::Q; i++)
    if (L::ex(i) == x && L::ey(i) == y && L::ez(i) == z) return i;
  return -1;
}

// Full-slip node (NTSlip, boundary.mako:837-855; the pairs are sym.py:481-497 slip_bb_swap_pairs): every population
// with a component along the normal of orientation o swaps with its mirror image -- the direction with the normal
// component reversed and the tangential ones kept; o = 0 (no case in the reference's switch): nothing.
// Written as selects between VALUES, one straight line for all orientations: swaps of f[i] and f[j] inside one branch per
// orientation get their common code sunk by the compiler into one swap with the indices as phi nodes, and an array
// indexed by a run-time value lives in scratch memory (every kernel that contains the node code then keeps its
// populations there: 48-160 bytes of scratch per lane in the first version of this function).
template <class L, int I, int O>
constexpr int slip_image() {
  constexpr int n = L::dir2vecidx(O);
  if ((L::ex(I) * L::ex(n) + L::ey(I) * L::ey(n) + L::ez(I) * L::ez(n)) == 0) return I;
  return dir_of<L>(L::ex(n) != 0 ? -L::ex(I) : L::ex(I), L::ey(n) != 0 ? -L::ey(I) : L::ey(I),
                   L::ez(n) != 0 ? -L::ez(I) : L::ez(I));
}

template <class L, class R>
SLF_D void slip_reflect(R (&f)[L::Q], int o) {
  R g[L::Q];
  static_for<0, L::Q>([&](auto I) { g[I] = f[I]; });
  static_for<1, 2 * L::dim + 1>([&](auto O) {
    const bool on = o == O;
    static_for<1, L::Q>([&](auto I) {
      constexpr int J = slip_image<L, I, O>();
      static_assert(J > 0, "the mirror image of a lattice direction is a lattice direction");
      if constexpr (J != I) g[I] = on ? f[J] : g[I];
    });
  });
  static_for<1, L::Q>([&](auto I) { f[I] = g[I]; });
}

// Is population I unknown at a node whose inward normal is orientation O (1..2 dim)?
template <class L, int I, int O>
constexpr bool is_missing() {
  constexpr int n = L::dir2vecidx(O);
  return (L::ex(I) * L::ex(n) + L::ey(I) * L::ey(n) + L::ez(I) * L::ez(n)) > 0;
}

// boundary.mako:418-422: replace unknown populations by their opposites.
template <class L, class R, int O>
SLF_D void fill_missing_with_opposite(R (&f)[L::Q]) {
  static_for<1, L::Q>([&](auto I) {
    if constexpr (is_missing<L, I, O>()) f[I] = f[L::opp(I)];
  });
}

// Signed normal component n.v for orientation O.
template <class L, class R, int O>
SLF_D R ndotv(const R (&v)[3]) {
  return edotv<L, R, L::dir2vecidx(O)>(v);
}

// C9a: macroscopic quantities on a velocity-BC node (boundary.mako:443-459).
template <class L, class R, int O>
SLF_D void macro_velocity_bc(R (&f)[L::Q], const R* par, bool incompressible, R& rho, R (&v)[3]) {
  fill_missing_with_opposite<L, R, O>(f);
  const R rs = density<L, R>(f);
  v[0] = par[0];
  v[1] = par[1];
  v[2] = (R)0;
  if constexpr (L::dim == 3) v[2] = par[2];
  const R nv = ndotv<L, R, O>(v);
  rho = incompressible ? (rs + nv) : (rs / ((R)1 - nv));
}

// C10a: macroscopic quantities on a density-BC node (boundary.mako:425-441).
template <class L, class R, int O>
SLF_D void macro_density_bc(R (&f)[L::Q], R par_rho, R& rho, R (&v)[3]) {
  fill_missing_with_opposite<L, R, O>(f);
  const R rs = density<L, R>(f);
  const R t = (rs - par_rho) / par_rho;
  constexpr int n = L::dir2vecidx(O);
  v[0] = v[1] = v[2] = (R)0;
  static_for<0, L::dim>([&](auto D) {
    constexpr int e = e_comp<L>(n, D);
    if constexpr (e > 0) v[D] = (R)0 - t;
    if constexpr (e < 0) v[D] = t;
  });
  rho = par_rho;
}

template <class L, class R>
SLF_D void set_equilibrium(R (&f)[L::Q], R rho, R rho0, const R (&v)[3]) {
  const R u15 = usq15<L, R>(v);
  static_for<0, L::Q>([&](auto I) { f[I] = feq<L, R, I>(rho, rho0, v, u15); });
}

// Index of the (a, b), a <= b, component in the order xx xy [xz] yy [yz zz].
template <class L>
constexpr int flux_index(int a, int b) {
  return a * L::dim - a * (a - 1) / 2 + (b - a);
}

// C9b: regularized velocity/density node, pre-collision (boundary.mako:817-835,
// sym.py:750-766 non-equilibrium bounce-back, sym.py:882-891 regularisation).
template <class L, class R, int O>
SLF_D void regularized_bc(R (&f)[L::Q], R rho, R rho0, const R (&v)[3]) {
  // f_i = f_opp(i) + 6 w_i rho0 (e_i . v) for the unknown populations.
  static_for<1, L::Q>([&](auto I) {
    if constexpr (is_missing<L, I, O>()) {
      f[I] = f[L::opp(I)] + Weights<L, R>::w6(I) * (rho0 * edotv<L, R, I>(v));
    }
  });
  // Non-equilibrium momentum flux, components xx xy [xz] yy [yz zz].
  constexpr int NP = L::dim * (L::dim + 1) / 2;
  R P[NP];
  {
    static_for<0, L::dim>([&](auto A) {
      static_for<A, L::dim>([&](auto B) {
        constexpr int idx = flux_index<L>(A, B);
        R acc = (R)0;
        static_for<1, L::Q>([&](auto I) {
          constexpr int c = e_comp<L>(I, A) * e_comp<L>(I, B);
          if constexpr (c > 0) acc = acc + f[I];
          if constexpr (c < 0) acc = acc - f[I];
        });
        if constexpr (A == B) acc = acc - rho * (v[A] * v[A] + (R)(1.0 / 3.0));
        else acc = acc - rho * (v[A] * v[B]);
        P[idx] = acc;
      });
    });
  }
  const R u15 = usq15<L, R>(v);
  static_for<0, L::Q>([&](auto I) {
    R acc = (R)0;
    static_for<0, L::dim>([&](auto A) {
      static_for<A, L::dim>([&](auto B) {
        constexpr int idx = flux_index<L>(A, B);
        constexpr int ea = e_comp<L>(I, A), eb = e_comp<L>(I, B);
        if constexpr (A == B) {
          if constexpr (ea != 0) acc = acc + (R)(2.0 / 3.0) * P[idx];
          else acc = acc + (R)(-1.0 / 3.0) * P[idx];
        } else {
          if constexpr (ea * eb > 0) acc = acc + (R)2 * P[idx];
          if constexpr (ea * eb < 0) acc = acc - (R)2 * P[idx];
        }
      });
    });
    const R val = feq<L, R, I>(rho, rho0, v, u15) + Weights<L, R>::w45(I) * acc;
    f[I] = val > (R)1e-7 ? val : (R)1e-7;
  });
}

SLF_D float slf_sqrt(float x) { return sqrtf(x); }
SLF_D double slf_sqrt(double x) { return sqrt(x); }

// --regularized / --subgrid=les-smagorinsky (reference lb_single.py:27-42; bgk_relaxation_preamble and
// update_relaxation_time, relaxation_common.mako:166-237; BGK_relaxate, relaxation.mako:124-132).  Both work on the
// non-equilibrium momentum flux of the populations as read, at the equilibrium's velocity (u + a/2 under Guo forcing):
//   subgrid (flags bit 1): tau0 = 1/2 + 3 visc, tau0 += (sqrt(tau0^2 + 36 C^2 sqrt(Q)) - tau0) / 2 with Q = T_ab T_ab
//     (off-diagonal components twice), omega = 1 / tau0, the Guo prefactor at that tau0 (sym_force.py:156-160);
//   regularized (bit 0): f_i = feq_i + w_i / (2 cs^4) (e_ia e_ib - cs^2 delta_ab) T_ab (sym.reglb_flux_tensor) before the
//     relaxation.
// Operation for operation oracle/lbm_oracle.c: bgk_relax.  Guo forcing or none (the exact difference method is refused at
// module creation); standard and incompressible density models.
template <class L, class R>
SLF_D void bgk_relax_turb(R (&f)[L::Q], R rho, R (&v)[3], const CollideParams<L, R>& cp, int flags, R visc, R c2x36) {
  const R rho0 = cp.incompressible != 0 ? (R)1 : rho;
  const bool has_force = cp.has_force != 0;
  if (has_force) {
    static_for<0, L::dim>([&](auto D) { v[D] = v[D] + (R)0.5 * cp.accel[D]; });
  }
  const R u15 = usq15<L, R>(v);
  constexpr int NP = L::dim * (L::dim + 1) / 2;
  R P[NP];
  static_for<0, L::dim>([&](auto A) {
    static_for<A, L::dim>([&](auto B) {
      constexpr int idx = flux_index<L>(A, B);
      R acc = (R)0;
      static_for<1, L::Q>([&](auto I) {
        constexpr int c = e_comp<L>(I, A) * e_comp<L>(I, B);
        if constexpr (c > 0) acc = acc + f[I];
        if constexpr (c < 0) acc = acc - f[I];
      });
      if constexpr (A == B) acc = acc - rho * (v[A] * v[A] + (R)(1.0 / 3.0));
      else acc = acc - rho * (v[A] * v[B]);
      P[idx] = acc;
    });
  });
  R omega = cp.omega, guo_pref = cp.guo_pref;
  if (flags & 2) {
    R strain = (R)0;
    static_for<0, L::dim>([&](auto A) {
      static_for<A + 1, L::dim>([&](auto B) {
        const R t = P[flux_index<L>(A, B)];
        strain = strain + (R)2 * t * t;
      });
    });
    static_for<0, L::dim>([&](auto A) {
      const R t = P[flux_index<L>(A, A)];
      strain = strain + t * t;
    });
    R tau0 = (R)0.5 + (R)3 * visc;
    tau0 = tau0 + (R)0.5 * (slf_sqrt(tau0 * tau0 + c2x36 * slf_sqrt(strain)) - tau0);
    omega = (R)1 / tau0;
    guo_pref = (R)3 * ((R)1 - (R)0.5 / tau0);
  }
  if (flags & 1) {
    static_for<0, L::Q>([&](auto I) {
      R acc = (R)0;
      static_for<0, L::dim>([&](auto A) {
        static_for<A, L::dim>([&](auto B) {
          constexpr int idx = flux_index<L>(A, B);
          constexpr int ea = e_comp<L>(I, A), eb = e_comp<L>(I, B);
          if constexpr (A == B) {
            if constexpr (ea != 0) acc = acc + (R)(2.0 / 3.0) * P[idx];
            else acc = acc + (R)(-1.0 / 3.0) * P[idx];
          } else {
            if constexpr (ea * eb > 0) acc = acc + (R)2 * P[idx];
            if constexpr (ea * eb < 0) acc = acc - (R)2 * P[idx];
          }
        });
      });
      f[I] = feq<L, R, I>(rho, rho0, v, u15) + Weights<L, R>::w45(I) * acc;
    });
  }
  static_for<0, L::Q>([&](auto I) {
    const R fe = feq<L, R, I>(rho, rho0, v, u15);
    f[I] = f[I] + omega * (fe - f[I]);
  });
  if (has_force) {
    const R pref = rho * guo_pref;
    R va = v[0] * cp.accel[0] + v[1] * cp.accel[1];
    if constexpr (L::dim == 3) va = va + v[2] * cp.accel[2];
    static_for<0, L::Q>([&](auto I) {
      const R eu = edotv<L, R, I>(v);
      const R ea = edotv<L, R, I>(cp.accel);
      const R t = (ea - va) + (R)3 * eu * ea;
      f[I] = f[I] + pref * Weights<L, R>::w(I) * t;
    });
  }
}

// Number of unknown populations, other than the one along the normal n, with a component along axis d.
template <class L>
constexpr int zouhe_count(int n, int d) {
  int c = 0;
  for (int i = 1; i < L::Q; i++) {
    const int sp = L::ex(i) * L::ex(n) + L::ey(i) * L::ey(n) + L::ez(i) * L::ez(n);
    const int ed = (d == 0) ? L::ex(i) : ((d == 1) ? L::ey(i) : L::ez(i));
    if (sp > 0 && i != n && ed != 0) c++;
  }
  return c;
}

// C11: Zou-He node (boundary.mako:343-382): bounce-back of the non-equilibrium part of the unknown
// populations (sym.py:750-766), then the tangential momentum excess  rho v - sum e_i f_i  is spread over
// the unknown populations that are not along the normal (sym.py:768-815).
template <class L, class R, int O>
SLF_D void zouhe_bb(R (&f)[L::Q], R rho, R rho0, const R (&v)[3]) {
  static_for<1, L::Q>([&](auto I) {
    if constexpr (is_missing<L, I, O>()) {
      f[I] = f[L::opp(I)] + Weights<L, R>::w6(I) * (rho0 * edotv<L, R, I>(v));
    }
  });
  constexpr int n = L::dir2vecidx(O);
  static_for<0, L::dim>([&](auto D) {
    if constexpr (e_comp<L>(n, D) == 0) {
      constexpr int cnt = zouhe_count<L>(n, D);
      const R md = (rho * v[D] - momentum<L, R, D>(f)) / (R)cnt;
      static_for<1, L::Q>([&](auto I) {
        if constexpr (is_missing<L, I, O>() && I != n) {
          constexpr int e = e_comp<L>(I, D);
          if constexpr (e > 0) f[I] = f[I] + md;
          if constexpr (e < 0) f[I] = f[I] - md;
        }
      });
    }
  });
}

// Dispatch a functor templated on the orientation constant (1..2 dim).
template <class L, class F>
SLF_D void with_orientation(int o, F&& fn) {
  static_for<1, 2 * L::dim + 1>([&](auto O) {
    if (o == O) fn(O);
  });
}

}  // namespace slf
