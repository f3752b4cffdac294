// nmf_core.h — the per-wavefront NMF program (forward iterations + reverse sweep).
//
// One wave64 owns one M×N matrix: every lane keeps NPL = N/64 columns of X (all M rows) in
// registers, its NPL rows of V, and a replicated copy of U.  X·V and VᵀV are lane-local partial
// sums followed by a wave reduction; XᵀU and UᵀU are purely lane-local.  X is read from HBM
// exactly once for all T iterations (the reference re-reads it in ~4T bmm launches,
// factorization/matrix_factorization.py:213,243).
//
// The program is written once against a "wave context" W so that the same source compiles
//   * for gfx950 (W::F = float, one lane per thread; csrc/nmf.hip), and
//   * for the host lock-step emulation used by the CPU tests (W::F = 64-lane vector;
//     tests/emul/emul.cpp) — that build exists only to test this file without a GPU.
//
// Update rules restated from the reference (paths relative to the reference root):
//   MU   factorization/matrix_factorization.py:241-247
//   HALS factorization/matrix_factorization.py:210-229 (CoordinateDescent, project=ReLU)
//   alternation U then V with the new U: :122-136 ; reconstruct u @ v.mT: :532-533
// Reverse sweep: SURVEY.md Appendix A (hand-derived; checked against autograd by the oracle).
#pragma once

#if defined(__HIPCC__)
#define FZ_HD __host__ __device__ __forceinline__
#else
#define FZ_HD inline
#endif

namespace fz {

constexpr int SOLVER_MU = 0;
constexpr int SOLVER_HALS = 1;

// reciprocal: v_rcp_f32 on device (1 ulp) instead of the ~10-instruction IEEE division sequence —
// the HALS/MU ratios need 1e-4 parity, not correct rounding, and the backward kernel is VALU-bound
#if defined(__HIP_DEVICE_COMPILE__)
FZ_HD float fz_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
#else
FZ_HD float fz_rcp(float v) { return 1.0f / v; }
#endif
FZ_HD float fz_relu(float v) { return v > 0.f ? v : 0.f; }
// gate(w, g) = g where w > 0 else 0  (ReLU mask taken from the forward value)
FZ_HD float fz_gate(float w, float g) { return w > 0.f ? g : 0.f; }

// ---- optional policy capability: factor rows DISTRIBUTED over lane groups ---------------------------------
// A policy with `static constexpr bool kDistRows = true` provides, for M = 8 rows:
//   F    sum8_dist(const F (&v)[8])             eight wave totals, total m left in every lane of lane group m
//   F    grp_take(F d, int m)                   the value lane group m holds, uniform
//   bool grp_below(int n)                       this lane's group index < n
//   void st_grp(float* base, int i0, int stride, F d)    base[i0 + group·stride] = d   (one lane per group)
//   F    ld_grp_global(const float* p, int i0, int stride)  p[i0 + group·stride]
//   F    ld_grp(const float* base, int i0, int stride)      the same from the wave's LDS history
//   F    grp_sum(F d)                                        total over the eight groups of a group-constant value, uniform
// The U half-step then runs ONCE with one row per lane group instead of eight times on wave-uniform values in all 64
// lanes (a third of the VALU instructions of a rank-2 iteration).  The arithmetic of every row is unchanged, but the
// compiler contracts and schedules the one-row and the eight-row code separately: the two forms agree to the last bit
// or two (4e-7 relative, tools/probes/nmf_ab.py), not bit for bit.  Policies without the flag (the host emulation among
// them) take the uniform form.
template <class W, class = void>
struct DistRows { static constexpr bool value = false; };
template <class W>
struct DistRows<W, decltype((void)W::kDistRows)> { static constexpr bool value = W::kDistRows; };

// ---- wave totals of K independent per-lane partial sums, eight at a time where K allows -----------
template <int K, class W, class F>
FZ_HD void sum_all(W& w, F (&acc)[K]) {
  if constexpr (K % 8 == 0) {
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 8) {
      F t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = acc[k0 + i];
      w.sum8(t);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[k0 + i] = t[i];
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = w.sum(acc[k]);
  }
}

// ---- one half-step on K independent rows of a factor:  w' = update(w; a, b) ------------
template <int K, int R, int SOLVER, class F>
FZ_HD void update_rows(F (&w)[K][R], const F (&a)[K][R], const F (&b)[R][R], float eps) {
  if (SOLVER == SOLVER_MU) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      F nw[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        F dn = w[k][0] * b[0][r];
#pragma unroll
        for (int q = 1; q < R; ++q) dn = dn + w[k][q] * b[q][r];
        nw[r] = (w[k][r] * a[k][r] + eps) * fz_rcp(dn + eps);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) w[k][r] = nw[r];
    }
  } else {
    // the denominators b[r][r] + eps are shared by all K rows: one reciprocal per column
    F inv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) inv[r] = fz_rcp(b[r][r] + eps);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (R == 1) {
        w[k][0] = fz_relu((a[k][0] + eps) * inv[0]);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          F s = F(0.f);
          bool first = true;
#pragma unroll
          for (int q = 0; q < R; ++q) {
            if (q == r) continue;
            F t = w[k][q] * b[q][r];  // columns q < r are already updated (Gauss–Seidel)
            s = first ? t : s + t;
            first = false;
          }
          w[k][r] = fz_relu((a[k][r] - s + eps) * inv[r]);
        }
      }
    }
  }
}

// ---- reverse of one half-step for ONE row --------------------------------------------
//  in : wold, wnew (forward values), a (row of Z s), b = sᵀs, gwn = dL/dwnew (clobbered)
//  out: gwo = dL/dwold, ga = dL/da (row), gb += this row's contribution to dL/db
template <int R, int SOLVER, class F>
FZ_HD void half_bwd_row(const F (&wold)[R], const F (&wnew)[R], const F (&a)[R],
                        const F (&b)[R][R], F (&gwn)[R], F (&gwo)[R], F (&ga)[R],
                        F (&gb)[R][R], float eps) {
  if (SOLVER == SOLVER_MU) {
    F gn[R], gdn[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      F dn = wold[0] * b[0][r];
#pragma unroll
      for (int q = 1; q < R; ++q) dn = dn + wold[q] * b[q][r];
      const F idn = fz_rcp(dn + eps);
      gn[r] = gwn[r] * idn;
      gdn[r] = F(0.f) - gwn[r] * wnew[r] * idn;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      F t = gn[r] * a[r];
#pragma unroll
      for (int q = 0; q < R; ++q) t = t + gdn[q] * b[r][q];
      gwo[r] = t;
      ga[r] = gn[r] * wold[r];
#pragma unroll
      for (int q = 0; q < R; ++q) gb[q][r] = gb[q][r] + wold[q] * gdn[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) gwo[r] = F(0.f);
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      const F iden = fz_rcp(b[r][r] + eps);
      F gq = fz_gate(wnew[r], gwn[r]);
      F gnum = gq * iden;
      gb[r][r] = gb[r][r] - (gq * wnew[r]) * iden;
      ga[r] = gnum;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        if (q == r) continue;
        if (q < r) {
          gb[q][r] = gb[q][r] - gnum * wnew[q];
          gwn[q] = gwn[q] - gnum * b[q][r];
        } else {
          gb[q][r] = gb[q][r] - gnum * wold[q];
          gwo[q] = gwo[q] - gnum * b[q][r];
        }
      }
    }
  }
}

// ---- per-wave history kept in LDS for the reverse sweep --------------------------------
// state s = 0..G holds (u, v) after iteration T-G+s; step s = 0..G-1 holds the (a, b) of the
// U-update that produced state s+1.
template <int M, int NPL, int R>
struct Hist {
  float* vh;  // lane-private: [(G+1)][R][NPL] × 64 lanes
  float* uh;  // uniform:      [(G+1)][M][R]
  float* ah;  // uniform:      [G][M][R]
  float* bh;  // uniform:      [G][R][R]
  static FZ_HD int floats(int G) {
    return (G + 1) * R * NPL * 64 + (G + 1) * M * R + G * (M * R + R * R);
  }
  FZ_HD void carve(float* base, int G) {
    vh = base;
    uh = vh + (G + 1) * R * NPL * 64;
    ah = uh + (G + 1) * M * R;
    bh = ah + G * M * R;
  }
};

// waves per workgroup of a backward launch whose history needs `per_wave` bytes of LDS per wave: the count (1..4) that
// puts the most waves on a CU (160 KB of LDS), larger workgroups on ties.  (A fixed 64 KB budget per workgroup left
// the R = 2, T = 10 history of BASELINE configs[4] — 18.4 KB per wave — at 3 waves x 2 workgroups = 6 waves per CU; 4 x 2 = 8.)
static inline int fz_hist_waves_per_block(int per_wave) {
  int best = 1, best_waves = 0;
  for (int wv = 1; wv <= 4; ++wv) {
    const int lds = per_wave * wv;
    if (lds > 160 * 1024) break;
    const int waves = (160 * 1024 / lds) * wv;
    if (waves >= best_waves) { best_waves = waves; best = wv; }
  }
  return best;
}

template <int M, int NPL, int R, class W>
FZ_HD void save_state(W& w, const Hist<M, NPL, R>& h, int s, const typename W::F (&u)[M][R],
                      const typename W::F (&v)[NPL][R]) {
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) w.st_uni(h.uh, (s * M + m) * R + r, u[m][r]);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < NPL; ++j) w.st_priv(h.vh, (s * R + r) * NPL + j, v[j][r]);
}

template <int M, int NPL, int R, class W>
FZ_HD void save_state_dist(W& w, const Hist<M, NPL, R>& h, int s, const typename W::F (&ud)[R],
                           const typename W::F (&v)[NPL][R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) w.st_grp(h.uh, s * M * R + r, R, ud[r]);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < NPL; ++j) w.st_priv(h.vh, (s * R + r) * NPL + j, v[j][r]);
}

// ---- one full iteration: U-update then V-update ---------------------------------------
// step >= 0: also record (a, b) of the U-update as step `step` and the new state as step+1.
// ud (policies with kDistRows, M = 8): the CURRENT u in distributed form, ud[r] of lane group m = u[m][r]; kept in step
// with the uniform copy `u` that the column work reads.
template <int M, int NPL, int R, int SOLVER, class W>
FZ_HD void nmf_step(W& w, const typename W::F (&x)[M][NPL], typename W::F (&u)[M][R],
                    typename W::F (&v)[NPL][R], int mreal, float eps,
                    const Hist<M, NPL, R>* h, int step, typename W::F (*ud)[R] = nullptr) {
  using F = typename W::F;
  if constexpr (DistRows<W>::value && M == 8) {
    F ad[1][R];
    F b[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      F col[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        F acc = x[m][0] * v[0][r];
#pragma unroll
        for (int j = 1; j < NPL; ++j) acc = acc + x[m][j] * v[j][r];
        col[m] = acc;
      }
      ad[0][r] = w.sum8_dist(col);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = r; q < R; ++q) {
        F acc = v[0][r] * v[0][q];
#pragma unroll
        for (int j = 1; j < NPL; ++j) acc = acc + v[j][r] * v[j][q];
        F t = w.sum(acc);
        b[r][q] = t;
        b[q][r] = t;
      }
    if (h != nullptr && step >= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) w.st_grp(h->ah, step * M * R + r, R, ad[0][r]);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < R; ++q) w.st_uni(h->bh, (step * R + r) * R + q, b[r][q]);
    }
    update_rows<1, R, SOLVER, F>(*reinterpret_cast<F (*)[1][R]>(ud), ad, b, eps);
#pragma unroll
    for (int r = 0; r < R; ++r) (*ud)[r] = w.grp_below(mreal) ? (*ud)[r] : F(0.f);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) u[m][r] = w.grp_take((*ud)[r], m);
  } else {
    F a[M][R];
    F b[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      F col[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        F acc = x[m][0] * v[0][r];
#pragma unroll
        for (int j = 1; j < NPL; ++j) acc = acc + x[m][j] * v[j][r];
        col[m] = acc;
      }
      sum_all<M, W, F>(w, col);
#pragma unroll
      for (int m = 0; m < M; ++m) a[m][r] = col[m];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = r; q < R; ++q) {
        F acc = v[0][r] * v[0][q];
#pragma unroll
        for (int j = 1; j < NPL; ++j) acc = acc + v[j][r] * v[j][q];
        F t = w.sum(acc);
        b[r][q] = t;
        b[q][r] = t;
      }
    if (h != nullptr && step >= 0) {
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) w.st_uni(h->ah, (step * M + m) * R + r, a[m][r]);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < R; ++q) w.st_uni(h->bh, (step * R + r) * R + q, b[r][q]);
    }
    update_rows<M, R, SOLVER, F>(u, a, b, eps);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) u[m][r] = (m < mreal) ? u[m][r] : F(0.f);
  }
  {
    F ap[NPL][R];
    F bp[R][R];
#pragma unroll
    for (int j = 0; j < NPL; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        F acc = x[0][j] * u[0][r];
#pragma unroll
        for (int m = 1; m < M; ++m) acc = acc + x[m][j] * u[m][r];
        ap[j][r] = acc;
      }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = r; q < R; ++q) {
        F acc = u[0][r] * u[0][q];
#pragma unroll
        for (int m = 1; m < M; ++m) acc = acc + u[m][r] * u[m][q];
        bp[r][q] = acc;
        bp[q][r] = acc;
      }
    update_rows<NPL, R, SOLVER, F>(v, ap, bp, eps);
#pragma unroll
    for (int j = 0; j < NPL; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r) v[j][r] = w.keep_col(j, v[j][r]);
  }
  if (h != nullptr && step >= 0) {
    if constexpr (DistRows<W>::value && M == 8) save_state_dist<M, NPL, R, W>(w, *h, step + 1, *ud, v);
    else save_state<M, NPL, R, W>(w, *h, step + 1, u, v);
  }
}

// ---- initial factors: the broadcast RandomInit buffers (matrix_factorization.py:52-58) --
template <int M, int NPL, int R, class W>
FZ_HD void nmf_init(W& w, const float* u0, const float* v0, typename W::F (&u)[M][R],
                    typename W::F (&v)[NPL][R], int mreal) {
  using F = typename W::F;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) u[m][r] = (m < mreal) ? w.ld_uni_global(u0, m * R + r) : F(0.f);
#pragma unroll
  for (int j = 0; j < NPL; ++j)
#pragma unroll
    for (int r = 0; r < R; ++r) v[j][r] = w.ld_v0(v0, j, r, R);
}

// distributed copy of the initial u (policies with kDistRows)
template <int M, int R, class W>
FZ_HD void nmf_init_dist(W& w, const float* u0, typename W::F (&ud)[R], int mreal) {
  using F = typename W::F;
  if constexpr (DistRows<W>::value && M == 8) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const F t = w.ld_grp_global(u0, r, R);
      ud[r] = w.grp_below(mreal) ? t : F(0.f);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) ud[r] = F(0.f);
  }
}

// ---- forward: T iterations, then y = u vᵀ ---------------------------------------------
template <int M, int NPL, int R, int SOLVER, class W>
FZ_HD void nmf_forward_wave(W& w, const float* u0, const float* v0,
                            typename W::F (&x)[M][NPL], typename W::F (&u)[M][R],
                            typename W::F (&v)[NPL][R], int mreal, int T, float eps) {
  nmf_init<M, NPL, R, W>(w, u0, v0, u, v, mreal);
  typename W::F ud[R];
  nmf_init_dist<M, R, W>(w, u0, ud, mreal);
  for (int t = 0; t < T; ++t)
    nmf_step<M, NPL, R, SOLVER, W>(w, x, u, v, mreal, eps, nullptr, -1, &ud);
  // x becomes y
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      typename W::F acc = u[m][0] * v[j][0];
#pragma unroll
      for (int r = 1; r < R; ++r) acc = acc + u[m][r] * v[j][r];
      x[m][j] = acc;
    }
}

// ---- backward: recompute forward with history, then the reverse sweep ------------------
//  x: the matrix;  g: on entry dL/dy (zeros if only gu/gv are given), on exit dL/dx.
//  gu_ext / gv_ext: optional extra gradients of the decompose() outputs (device: global
//  pointers for this matrix, may be nullptr).
template <int M, int NPL, int R, int SOLVER, class W>
FZ_HD void nmf_backward_wave(W& w, const float* u0, const float* v0,
                             const typename W::F (&x)[M][NPL], typename W::F (&g)[M][NPL],
                             const Hist<M, NPL, R>& h, int mreal, int T, int G, float eps,
                             const float* gu_ext, const float* gv_ext) {
  using F = typename W::F;
  // The reverse sweep on distributed rows adds the rows' contributions to dL/db in a tree over the lane groups instead of
  // row by row.  MU keeps the row-by-row (uniform) reverse sweep (its ε⁻¹-scaled gradients are the most sensitive to
  // rounding order in the oracle tests); its forward recomputation is distributed where the policy asks for it.
  constexpr bool kDistBwd = DistRows<W>::value && M == 8 && SOLVER != SOLVER_MU;
  F gu[M][R];
  F gud[R];      // (kDistBwd) dL/du in distributed form: lane group m holds row m
  F gv[NPL][R];
  {
    F u[M][R];
    F v[NPL][R];
    nmf_init<M, NPL, R, W>(w, u0, v0, u, v, mreal);
    F ud[R];
    nmf_init_dist<M, R, W>(w, u0, ud, mreal);
    for (int t = 0; t < T - G; ++t)
      nmf_step<M, NPL, R, SOLVER, W>(w, x, u, v, mreal, eps, nullptr, -1, &ud);
    save_state<M, NPL, R, W>(w, h, 0, u, v);
    for (int s = 0; s < G; ++s) nmf_step<M, NPL, R, SOLVER, W>(w, x, u, v, mreal, eps, &h, s, &ud);
    // output layer  y = u_T v_Tᵀ :  gu = gY v_T ,  gv = gYᵀ u_T
#pragma unroll
    for (int r = 0; r < R; ++r) {
      F col[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        F acc = g[m][0] * v[0][r];
#pragma unroll
        for (int j = 1; j < NPL; ++j) acc = acc + g[m][j] * v[j][r];
        col[m] = acc;
      }
      if constexpr (kDistBwd) {
        gud[r] = w.sum8_dist(col);
        if (gu_ext != nullptr) gud[r] = gud[r] + (w.grp_below(mreal) ? w.ld_grp_global(gu_ext, r, R) : F(0.f));
      } else {
      sum_all<M, W, F>(w, col);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        gu[m][r] = col[m];
        if (gu_ext != nullptr && m < mreal) gu[m][r] = gu[m][r] + w.ld_uni_global(gu_ext, m * R + r);
      }
      }
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        F acc = g[0][j] * u[0][r];
#pragma unroll
        for (int m = 1; m < M; ++m) acc = acc + g[m][j] * u[m][r];
        gv[j][r] = acc;
        if (gv_ext != nullptr) gv[j][r] = gv[j][r] + w.ld_v0(gv_ext, j, r, R);
      }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int j = 0; j < NPL; ++j) g[m][j] = F(0.f);
  w.fence();

  for (int s = G - 1; s >= 0; --s) {
    // ---- undo the V-update of this iteration: v_{s+1} = upd(xᵀ; v_s, u_{s+1}) ----
    F un[M][R];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) un[m][r] = w.ld_uni(h.uh, ((s + 1) * M + m) * R + r);
    F bp[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = r; q < R; ++q) {
        F acc = un[0][r] * un[0][q];
#pragma unroll
        for (int m = 1; m < M; ++m) acc = acc + un[m][r] * un[m][q];
        bp[r][q] = acc;
        bp[q][r] = acc;
      }
    F gb[R][R];
    F gup[M][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) gb[r][q] = F(0.f);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) gup[m][r] = F(0.f);
    F gvold[NPL][R];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      F vold[R], vnew[R], apj[R], gwn[R], gwo[R], gaj[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        vold[r] = w.ld_priv(h.vh, (s * R + r) * NPL + j);
        vnew[r] = w.ld_priv(h.vh, ((s + 1) * R + r) * NPL + j);
        F acc = x[0][j] * un[0][r];
#pragma unroll
        for (int m = 1; m < M; ++m) acc = acc + x[m][j] * un[m][r];
        apj[r] = acc;
        gwn[r] = gv[j][r];
      }
      half_bwd_row<R, SOLVER, F>(vold, vnew, apj, bp, gwn, gwo, gaj, gb, eps);
#pragma unroll
      for (int r = 0; r < R; ++r) gvold[j][r] = gwo[r];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        F acc = g[m][j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          acc = acc + un[m][r] * gaj[r];
          gup[m][r] = gup[m][r] + x[m][j] * gaj[r];
        }
        g[m][j] = acc;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) gb[r][q] = w.sum(gb[r][q]);
    F und[R];    // (kDistRows) u_{s+1} of this lane group's row
    if constexpr (kDistBwd) {
#pragma unroll
      for (int r = 0; r < R; ++r) und[r] = w.ld_grp(h.uh, (s + 1) * M * R + r, R);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        F col[M];
#pragma unroll
        for (int m = 0; m < M; ++m) col[m] = gup[m][r];
        F acc = gud[r] + w.sum8_dist(col);
#pragma unroll
        for (int q = 0; q < R; ++q) acc = acc + und[q] * (gb[q][r] + gb[r][q]);
        gud[r] = acc;
      }
    } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      F col[M];
#pragma unroll
      for (int m = 0; m < M; ++m) col[m] = gup[m][r];
      sum_all<M, W, F>(w, col);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        F acc = gu[m][r] + col[m];
#pragma unroll
        for (int q = 0; q < R; ++q) acc = acc + un[m][q] * (gb[q][r] + gb[r][q]);
        gu[m][r] = acc;
      }
    }
    }

    // ---- undo the U-update: u_{s+1} = upd(x; u_s, v_s) ----
    F bs[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) bs[r][q] = w.ld_uni(h.bh, (s * R + r) * R + q);
    F ga[M][R];
    F gbu[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) gbu[r][q] = F(0.f);
    if constexpr (kDistBwd) {
      // one row per lane group: the reverse half-step once instead of eight times on uniform values; the rows'
      // contributions to dL/db are added over the groups in a fixed tree (the uniform form adds them row by row: the
      // results differ by rounding only, both orders are deterministic)
      F uold[R], as[R], gwn[R], gwo[R], gam[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uold[r] = w.ld_grp(h.uh, s * M * R + r, R);
        as[r] = w.ld_grp(h.ah, s * M * R + r, R);
        gwn[r] = gud[r];
      }
      half_bwd_row<R, SOLVER, F>(uold, und, as, bs, gwn, gwo, gam, gbu, eps);
#pragma unroll
      for (int r = 0; r < R; ++r) gud[r] = gwo[r];  // gradient handed to u_s
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < R; ++q) gbu[r][q] = w.grp_sum(gbu[r][q]);
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) ga[m][r] = w.grp_take(gam[r], m);
    } else {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      F uold[R], as[R], gwn[R], gwo[R], gam[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uold[r] = w.ld_uni(h.uh, (s * M + m) * R + r);
        as[r] = w.ld_uni(h.ah, (s * M + m) * R + r);
        gwn[r] = gu[m][r];
      }
      half_bwd_row<R, SOLVER, F>(uold, un[m], as, bs, gwn, gwo, gam, gbu, eps);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        ga[m][r] = gam[r];
        gu[m][r] = gwo[r];  // gradient handed to u_s
      }
    }
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      F vold[R];
#pragma unroll
      for (int r = 0; r < R; ++r) vold[r] = w.ld_priv(h.vh, (s * R + r) * NPL + j);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        F acc = g[m][j];
#pragma unroll
        for (int r = 0; r < R; ++r) acc = acc + ga[m][r] * vold[r];
        g[m][j] = acc;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        F acc = gvold[j][r];
#pragma unroll
        for (int m = 0; m < M; ++m) acc = acc + x[m][j] * ga[m][r];
#pragma unroll
        for (int q = 0; q < R; ++q) acc = acc + vold[q] * (gbu[q][r] + gbu[r][q]);
        gv[j][r] = acc;
      }
    }
  }
}

}  // namespace fz
