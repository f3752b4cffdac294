// nmf_gram.h — reverse-mode HALS rank-1 NMF for a NON-NEGATIVE 8×N matrix, iterated in the 8-dimensional row space.
//
// The reference iteration (factorization/matrix_factorization.py:210-229, CoordinateDescent with project = ReLU, R = 1;
// alternation U then V with the new U, :122-136; reconstruct u vᵀ, :532-533; grad through the last `num_grad_steps`
// iterations, :506-512) is
//      u_i = relu((X v_{i-1} + ε) / (v_{i-1}ᵀv_{i-1} + ε)),     v_i = relu((Xᵀu_i + ε) / (u_iᵀu_i + ε)).
// For X ≥ 0 (the FactMixer applies ReLU in front of the factorisation, factorizer.py:44) and the uniform non-negative
// initial factors (matrix_factorization.py:44-50) no ReLU ever clips: every quotient is ≥ ε / (…) > 0.  Then v_{i-1} can
// be eliminated:  with  K = X Xᵀ (8×8),  s = X·1,  d = uᵀu,  ρ = 1/(d + ε),  p = K u
//      X v     = (p + ε s) ρ                                      (a of the next U half-step)
//      vᵀ v    = (uᵀp + 2ε uᵀs + N ε²) ρ²                         (b of the next U half-step)
// — the SAME function of X, term for term including every ε (an all-zero matrix still gives v = ε/(d+ε) ≈ 1, b = N),
// evaluated with X touched three times (K and s; the first half-step X v_start; the last V half-step Xᵀu_T) instead of
// 2T times, and no per-column history.  Reverse mode: everything between the first and the last half-step is a chain
// of 8-vectors; dL/dX = u_T gcᵀ + gs·1ᵀ + ga₁ v_startᵀ + S X with S = dL/dK + dL/dKᵀ (one 8×8 by 8×N product at the end).
// Checked against autograd through the reference's update rule in float64 (2e-15; tools/probes/gram_proto.py), through
// the host emulation (tests/test_wave_program_emul.py) and on the device against the oracle and the reference goldens.
//
// Lane layout (wave context W as in nmf_core.h, with its distributed-row capability): per-column values are lane-local
// (NPL columns per lane); 8-vectors live DISTRIBUTED — lane group g (8 lanes) holds element g; scalars are uniform.
#pragma once

#include "nmf_core.h"

namespace fz {

// floats of wave-private LDS history for `steps` graded Gram iterations
// (64 floats of scratch for the one-time re-ordering of K, then 12 per step: u (8), rho, nb, q)
FZ_HD int gram_hist_floats(int steps) { return 64 + (steps > 0 ? steps : 0) * 12; }

template <int NPL, class W>
struct GramBwd {
  using F = typename W::F;
  F vs[NPL];     // v_start: the V factor the graded part starts from (v0, or v_{T-G} recomputed), a constant
  F gc[NPL];     // dL/dc of the output layer, c = Xᵀu_T
  F vT[NPL];     // v_T (alive while the rows of gY stream in)
  F gv[NPL];     // gYᵀ u_T
  F col[8];      // per-lane partial sums of gY v_T, one per row
  F Kd[8];       // K[g][k]           (distributed: row g in lane group g)
  F gKd[8];      // S[g][k] = dL/dK[g][k] + dL/dK[k][g]
  F sd;          // s_g
  F ud;          // current u_g
  F gsd, ga1d;   // dL/ds_g, dL/da_g of the first graded half-step
  F uT[8];       // u_T, uniform
  F r;           // 1/(u_Tᵀu_T + ε)
  F binv;        // 1/(v_startᵀv_start + ε)
  float eps, neps2;
  float* hist;   // wave-private LDS
  int steps;     // graded Gram iterations (G - 1)

  // u' = (a + ε)/(b + ε) for an explicit V factor (per-column vcol)
  template <int M>
  FZ_HD void explicit_step(W& w, const F (&x)[M][NPL], const F (&vcol)[NPL], int mreal) {
    F c8[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      F acc = x[m][0] * vcol[0];
#pragma unroll
      for (int j = 1; j < NPL; ++j) acc = acc + x[m][j] * vcol[j];
      c8[m] = acc;
    }
    F ad = w.sum8_dist(c8);
    F bb = vcol[0] * vcol[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) bb = bb + vcol[j] * vcol[j];
    bb = w.sum(bb);
    binv = fz_rcp(bb + eps);
    ud = (ad + eps) * binv;
    ud = w.mask_rows(ud, mreal);
  }

  // one iteration in the row space; rec >= 0: record it as graded step `rec`
  FZ_HD void gram_step(W& w, int mreal, int rec) {
    F uk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) uk[k] = w.grp_take(ud, k);
    F d = uk[0] * uk[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) d = d + uk[k] * uk[k];
    const F rho = fz_rcp(d + eps);
    F p = Kd[0] * uk[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) p = p + Kd[k] * uk[k];
    const F nb = w.grp_sum(ud * (p + (2.0f * eps) * sd)) + neps2;
    const F a = (p + eps * sd) * rho;
    const F b = nb * rho * rho;
    const F q = fz_rcp(b + eps);
    if (rec >= 0) {
      w.st_grp(hist, rec * 12, 1, ud);
      w.st_uni(hist, rec * 12 + 8, rho);
      w.st_uni(hist, rec * 12 + 9, nb);
      w.st_uni(hist, rec * 12 + 10, q);
    }
    ud = (a + eps) * q;
    ud = w.mask_rows(ud, mreal);
  }

  // ---- phase A: K, s, the forward iterations; leaves u_T, r, v_T --------------------------------------------------
  //  x: the matrix (X ≥ 0), v0: the RandomInit buffer (N×1), T iterations of which the last G carry gradient (1 ≤ G ≤ T)
  template <class Hook>
  FZ_HD void forward(W& w, const F (&x)[8][NPL], const float* v0, int mreal, int nreal, int T, int G, float eps_,
                     float* hist_, Hook&& after_reductions) {
    eps = eps_;
    neps2 = (float)nreal * eps_ * eps_;
    hist = hist_ + 64;
    steps = G - 1;
    // K = X Xᵀ (upper triangle) and s = X·1, lane-local partial sums
    F kp[36];
    F sp[8];
#pragma unroll
    for (int m = 0, i = 0; m < 8; ++m) {
      F acc = x[m][0];
#pragma unroll
      for (int j = 1; j < NPL; ++j) acc = acc + x[m][j];
      sp[m] = acc;
#pragma unroll
      for (int k = m; k < 8; ++k, ++i) {
        F a2 = x[m][0] * x[k][0];
#pragma unroll
        for (int j = 1; j < NPL; ++j) a2 = a2 + x[m][j] * x[k][j];
        kp[i] = a2;
      }
    }
    // wave totals, row g of K into lane group g: call c delivers K[g][(g + c) & 7]; lane i of every group keeps call i,
    // one LDS round trip turns the rotated order into Kd[k] = K[g][k]
    F mine = F(0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      F c8[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int a = g, b = (g + c) & 7;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        c8[g] = kp[lo * 8 - lo * (lo - 1) / 2 + (hi - lo)];
      }
      const F tot = w.sum8_dist(c8);
      mine = w.pick_call(c, tot, mine);
    }
    w.k_unrotate(hist_, mine, Kd);
    sd = w.sum8_dist(sp);
    after_reductions();

    const int t0 = T - G;
#pragma unroll
    for (int j = 0; j < NPL; ++j) vs[j] = w.ld_v0(v0, j, 0, 1);
    if (t0 > 0) {
      explicit_step<8>(w, x, vs, mreal);
      for (int i = 2; i <= t0; ++i) gram_step(w, mreal, -1);
      // v_start = v_{t0} = (Xᵀu + ε)/(uᵀu + ε)
      F uk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) uk[k] = w.grp_take(ud, k);
      F d = uk[0] * uk[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) d = d + uk[k] * uk[k];
      const F rho = fz_rcp(d + eps);
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        F acc = x[0][j] * uk[0];
#pragma unroll
        for (int m = 1; m < 8; ++m) acc = acc + x[m][j] * uk[m];
        vs[j] = w.keep_col(j, (acc + eps) * rho);
      }
    }
    explicit_step<8>(w, x, vs, mreal);
    for (int i = 0; i < steps; ++i) gram_step(w, mreal, i);
    // output layer operands: u_T (uniform), r, v_T = (Xᵀu_T + ε) r
#pragma unroll
    for (int k = 0; k < 8; ++k) uT[k] = w.grp_take(ud, k);
    F d = uT[0] * uT[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) d = d + uT[k] * uT[k];
    r = fz_rcp(d + eps);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      F acc = x[0][j] * uT[0];
#pragma unroll
      for (int m = 1; m < 8; ++m) acc = acc + x[m][j] * uT[m];
      vT[j] = w.keep_col(j, (acc + eps) * r);
    }
  }

  // ---- phase B: one row of gY = dL/dY (row 0 FIRST — it initialises gYᵀu_T — then any order, each row once) -----------
  FZ_HD void out_row(int m, const F (&grow)[NPL]) {
    F acc = grow[0] * vT[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) acc = acc + grow[j] * vT[j];
    col[m] = acc;
#pragma unroll
    for (int j = 0; j < NPL; ++j) gv[j] = m == 0 ? grow[j] * uT[0] : gv[j] + grow[j] * uT[m];
  }

  // ---- phase C: the reverse sweep in the row space; gscale multiplies dL/dY (1 / number of windows) ----------------
  //  (after_columns runs when the per-column part is done and only the row-space chain is left: the cheapest point, in
  //  registers, to request data for phase D)
  template <class Hook>
  FZ_HD void reverse(W& w, const F (&x)[8][NPL], float gscale, Hook&& after_columns) {
    F gud = w.sum8_dist(col) * gscale;
    F gvv = F(0.f);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      gv[j] = gv[j] * gscale;
      gc[j] = gv[j] * r;
      gvv = gvv + gv[j] * vT[j];
    }
    const F gdT = F(0.f) - w.sum(gvv) * r;
    F c8[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      F acc = x[m][0] * gc[0];
#pragma unroll
      for (int j = 1; j < NPL; ++j) acc = acc + x[m][j] * gc[j];
      c8[m] = acc;
    }
    gud = gud + w.sum8_dist(c8) + (2.0f * ud) * gdT;
    after_columns();
#pragma unroll
    for (int k = 0; k < 8; ++k) gKd[k] = F(0.f);
    gsd = F(0.f);
    F un = ud;   // u_i of the step being undone
    for (int i = steps - 1; i >= 0; --i) {
      const F uo = w.ld_grp(hist, i * 12, 1);
      const F rho = w.ld_uni(hist, i * 12 + 8);
      const F nb = w.ld_uni(hist, i * 12 + 9);
      const F q = w.ld_uni(hist, i * 12 + 10);
      F uk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) uk[k] = w.grp_take(uo, k);
      F p = Kd[0] * uk[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) p = p + Kd[k] * uk[k];
      const F ga = gud * q;
      const F gb = F(0.f) - w.grp_sum(gud * un) * q;
      const F rr = rho * rho;
      const F t1 = (gb * rr) * uo;
      const F gp = ga * rho + t1;
      gsd = gsd + eps * (gp + t1);
      const F grho = w.grp_sum(ga * (p + eps * sd)) + (2.0f * rho) * gb * nb;
      const F gd = F(0.f) - rr * grho;
      F gpk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) gpk[k] = w.grp_take(gp, k);
      F acc = Kd[0] * gpk[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) acc = acc + Kd[k] * gpk[k];
      gud = acc + (gb * rr) * (p + (2.0f * eps) * sd) + (2.0f * uo) * gd;
#pragma unroll
      for (int k = 0; k < 8; ++k) gKd[k] = gKd[k] + gp * uk[k] + uo * gpk[k];
      un = uo;
    }
    ga1d = gud * binv;
  }

  // ---- phase D: row m of dL/dX (ungated) ----------------------------------------------------------------------------
  //  (srow[k] = S[m][k], gsm = dL/ds_m, ga1m = ga₁[m]: uniform values of row m, fetched by the caller with row_coeffs)
  FZ_HD void row_coeffs(W& w, int m, F (&srow)[8], F& gsm, F& ga1m) const {
#pragma unroll
    for (int k = 0; k < 8; ++k) srow[k] = w.grp_take(gKd[k], m);
    gsm = w.grp_take(gsd, m);
    ga1m = w.grp_take(ga1d, m);
  }
  FZ_HD void gx_row(int m, const F (&x)[8][NPL], const F (&srow)[8], const F& gsm, const F& ga1m, F (&out)[NPL]) const {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      F acc = uT[m] * gc[j] + gsm;
      acc = acc + ga1m * vs[j];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = acc + srow[k] * x[k][j];
      out[j] = acc;
    }
  }
};

}  // namespace fz
