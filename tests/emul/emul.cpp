// TEST INFRASTRUCTURE — host lock-step emulation of the per-wave NMF program.
// Compiles factorizer_amd/csrc/nmf_core.h with F = 64-lane vector so the exact source that
// runs on gfx950 can be checked against the oracle on a machine without a GPU (and under
// host sanitizers).  Never shipped, never used by the product path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

struct V64 {
  float a[64];
  V64() {}
  V64(float s) { for (int i = 0; i < 64; ++i) a[i] = s; }
};
#define BINOP(op)                                                                       \
  inline V64 operator op(const V64& x, const V64& y) { V64 r; for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] op y.a[i]; return r; } \
  inline V64 operator op(const V64& x, float y) { V64 r; for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] op y; return r; }           \
  inline V64 operator op(float x, const V64& y) { V64 r; for (int i = 0; i < 64; ++i) r.a[i] = x op y.a[i]; return r; }
BINOP(+) BINOP(-) BINOP(*) BINOP(/)

namespace fz {
inline V64 fz_rcp(const V64& v) { V64 r; for (int i = 0; i < 64; ++i) r.a[i] = 1.0f / v.a[i]; return r; }
inline V64 fz_relu(const V64& v) { V64 r; for (int i = 0; i < 64; ++i) r.a[i] = v.a[i] > 0.f ? v.a[i] : 0.f; return r; }
inline V64 fz_gate(const V64& w, const V64& g) { V64 r; for (int i = 0; i < 64; ++i) r.a[i] = w.a[i] > 0.f ? g.a[i] : 0.f; return r; }
}  // namespace fz

#include "../../factorizer_amd/csrc/nmf_core.h"
#include "../../factorizer_amd/csrc/nmf_gram.h"


template <int M, int NPL>
struct EmuWave {
  using F = V64;
  int mreal, nreal;
  int col(int j, int lane) const { return j * 64 + lane; }
  // same reduction tree as the device: xor 1,2 ; half-mirror ; mirror ; row bcast
  F sum(const F& v) const {
    float t[64];
    std::memcpy(t, v.a, sizeof(t));
    float n[64];
    for (int i = 0; i < 64; ++i) n[i] = t[i] + t[i ^ 1];
    for (int i = 0; i < 64; ++i) t[i] = n[i] + n[i ^ 2];
    for (int i = 0; i < 64; ++i) n[i] = t[i] + t[(i & ~7) | (7 - (i & 7))];
    for (int i = 0; i < 64; ++i) t[i] = n[i] + n[(i & ~15) | (15 - (i & 15))];
    float r0 = t[15], r1 = t[31], r2 = t[47], r3 = t[63];
    float tot = (r3 + r2) + (r1 + r0);
    return F(tot);
  }
  // eight sums at once: the device's multi-value butterfly (fz_common.h wave_sum8) adds, per value,
  // x[j]+x[j+32] ; +[j+16] ; +[j+8] ; xor 1 ; xor 2 ; half-mirror
  void sum8(F (&v)[8]) const {
    for (int i = 0; i < 8; ++i) {
      float f32[32], f16[16], f8[8], g[8], h[8];
      for (int j = 0; j < 32; ++j) f32[j] = v[i].a[j] + v[i].a[j + 32];
      for (int j = 0; j < 16; ++j) f16[j] = f32[j] + f32[j + 16];
      for (int j = 0; j < 8; ++j) f8[j] = f16[j] + f16[j + 8];
      for (int j = 0; j < 8; ++j) g[j] = f8[j] + f8[j ^ 1];
      for (int j = 0; j < 8; ++j) h[j] = g[j] + g[j ^ 2];
      v[i] = F(h[0] + h[7]);
    }
  }
  void st_priv(float* base, int idx, const F& v) const { std::memcpy(base + idx * 64, v.a, 256); }
  F ld_priv(const float* base, int idx) const { F r; std::memcpy(r.a, base + idx * 64, 256); return r; }
  void st_uni(float* base, int idx, const F& v) const { base[idx] = v.a[0]; }
  F ld_uni(const float* base, int idx) const { return F(base[idx]); }
  F ld_uni_global(const float* p, int idx) const { return F(p[idx]); }
  F ld_v0(const float* v0, int j, int r, int R) const {
    F o;
    for (int l = 0; l < 64; ++l) { int n = col(j, l); o.a[l] = n < nreal ? v0[n * R + r] : 0.f; }
    return o;
  }
  F keep_col(int j, const F& v) const {
    F o;
    for (int l = 0; l < 64; ++l) o.a[l] = col(j, l) < nreal ? v.a[l] : 0.f;
    return o;
  }
  void fence() const {}
  void load_mat(const float* X, F (&x)[M][NPL]) const {
    for (int m = 0; m < M; ++m)
      for (int j = 0; j < NPL; ++j)
        for (int l = 0; l < 64; ++l) {
          int n = col(j, l);
          x[m][j].a[l] = (m < mreal && n < nreal) ? X[m * nreal + n] : 0.f;
        }
  }
  void store_mat(float* Y, const F (&x)[M][NPL]) const {
    for (int m = 0; m < mreal; ++m)
      for (int j = 0; j < NPL; ++j)
        for (int l = 0; l < 64; ++l) {
          int n = col(j, l);
          if (n < nreal) Y[m * nreal + n] = x[m][j].a[l];
        }
  }
};

template <int M, int NPL, int R, int S>
static void run_fwd(const float* x, const float* u0, const float* v0, float* y, float* uo, float* vo,
                    int64_t nmat, int mreal, int nreal, int T, float eps) {
  for (int64_t k = 0; k < nmat; ++k) {
    EmuWave<M, NPL> w{mreal, nreal};
    V64 xm[M][NPL], u[M][R], v[NPL][R];
    w.load_mat(x + k * mreal * nreal, xm);
    fz::nmf_forward_wave<M, NPL, R, S>(w, u0, v0, xm, u, v, mreal, T, eps);
    w.store_mat(y + k * mreal * nreal, xm);
    if (uo) for (int m = 0; m < mreal; ++m) for (int r = 0; r < R; ++r) uo[(k * mreal + m) * R + r] = u[m][r].a[0];
    if (vo) for (int j = 0; j < NPL; ++j) for (int l = 0; l < 64; ++l) { int n = j * 64 + l; if (n < nreal) for (int r = 0; r < R; ++r) vo[(k * nreal + n) * R + r] = v[j][r].a[l]; }
  }
}

template <int M, int NPL, int R, int S>
static void run_bwd(const float* x, const float* u0, const float* v0, const float* gy, const float* gu,
                    const float* gv, float* gx, int64_t nmat, int mreal, int nreal, int T, int G, float eps) {
  std::vector<float> lds(fz::Hist<M, NPL, R>::floats(G));
  std::vector<float> zeros((size_t)mreal * nreal, 0.f);
  for (int64_t k = 0; k < nmat; ++k) {
    EmuWave<M, NPL> w{mreal, nreal};
    fz::Hist<M, NPL, R> h;
    h.carve(lds.data(), G);
    V64 xm[M][NPL], g[M][NPL];
    w.load_mat(x + k * mreal * nreal, xm);
    w.load_mat(gy ? gy + k * mreal * nreal : zeros.data(), g);
    fz::nmf_backward_wave<M, NPL, R, S>(w, u0, v0, xm, g, h, mreal, T, G, eps,
                                        gu ? gu + k * mreal * R : nullptr, gv ? gv + k * nreal * R : nullptr);
    w.store_mat(gx + k * mreal * nreal, g);
  }
}


// ---- the row-space (Gram) reverse mode of nmf_gram.h: HALS rank 1 on non-negative matrices -------------------------
// The distributed-row capabilities of the device policy (CfWave in csrc/nmf_cf.hip) restated for 64-lane vectors:
// lane group g = lanes 8g .. 8g+7 holds element g of an 8-vector.
template <int NPL>
struct EmuWaveDist : EmuWave<8, NPL> {
  using F = V64;
  using Base = EmuWave<8, NPL>;
  F sum8_dist(const F (&v)[8]) const {
    F tmp[8];
    for (int i = 0; i < 8; ++i) tmp[i] = v[i];
    Base::sum8(tmp);
    F o;
    for (int l = 0; l < 64; ++l) o.a[l] = tmp[l >> 3].a[0];
    return o;
  }
  F grp_take(const F& d, int m) const { return F(d.a[8 * m]); }
  // device order (fz_common.h wave_group_sum): distance 8, 16, 32
  F grp_sum(const F& d) const {
    float g[8], h[8];
    for (int i = 0; i < 8; ++i) g[i] = d.a[8 * i];
    for (int i = 0; i < 8; ++i) h[i] = g[i] + g[i ^ 1];
    for (int i = 0; i < 8; ++i) g[i] = h[i] + h[i ^ 2];
    for (int i = 0; i < 8; ++i) h[i] = g[i] + g[i ^ 4];
    return F(h[0]);
  }
  F mask_rows(const F& d, int mreal) const { F o; for (int l = 0; l < 64; ++l) o.a[l] = (l >> 3) < mreal ? d.a[l] : 0.f; return o; }
  F pick_call(int c, const F& tot, const F& mine) const { F o; for (int l = 0; l < 64; ++l) o.a[l] = (l & 7) == c ? tot.a[l] : mine.a[l]; return o; }
  void k_unrotate(float* scratch, const F& mine, F (&Kd)[8]) const {
    for (int l = 0; l < 64; ++l) { const int g = l >> 3, i = l & 7; scratch[g * 8 + ((g + i) & 7)] = mine.a[l]; }
    for (int k = 0; k < 8; ++k) for (int l = 0; l < 64; ++l) Kd[k].a[l] = scratch[(l >> 3) * 8 + k];
  }
  void st_grp(float* base, int i0, int stride, const F& d) const { for (int g = 0; g < 8; ++g) base[i0 + g * stride] = d.a[8 * g]; }
  F ld_grp(const float* base, int i0, int stride) const { F o; for (int l = 0; l < 64; ++l) o.a[l] = base[i0 + (l >> 3) * stride]; return o; }
};

template <int NPL>
static void run_gram_bwd(const float* x, const float* v0, const float* gy, float* gx, int64_t nmat, int mreal, int nreal,
                         int T, int G, float eps, float gscale) {
  std::vector<float> lds(fz::gram_hist_floats(G - 1));
  for (int64_t k = 0; k < nmat; ++k) {
    EmuWaveDist<NPL> w;
    w.mreal = mreal; w.nreal = nreal;
    V64 xm[8][NPL], g[8][NPL];
    w.load_mat(x + k * mreal * nreal, xm);
    w.load_mat(gy + k * mreal * nreal, g);
    fz::GramBwd<NPL, EmuWaveDist<NPL>> P;
    P.forward(w, xm, v0, mreal, nreal, T, G, eps, lds.data(), [] {});
    for (int m = 0; m < 8; ++m) P.out_row(m, g[m]);
    P.reverse(w, xm, gscale, [] {});
    for (int m = 0; m < 8; ++m) {
      V64 srow[8], gsm, ga1m;
      P.row_coeffs(w, m, srow, gsm, ga1m);
      P.gx_row(m, xm, srow, gsm, ga1m, g[m]);
    }
    w.store_mat(gx + k * mreal * nreal, g);
  }
}

extern "C" int emu_gram_bwd(const float* x, const float* v0, const float* gy, float* gx, int64_t nmat, int M, int N, int T,
                            int G, float eps, float gscale) {
  if (M > 8 || N > 512 || G < 1 || G > T) return -2;
  switch ((N + 63) / 64) {
    case 1: run_gram_bwd<1>(x, v0, gy, gx, nmat, M, N, T, G, eps, gscale); return 0;
    case 2: run_gram_bwd<2>(x, v0, gy, gx, nmat, M, N, T, G, eps, gscale); return 0;
    case 3: run_gram_bwd<3>(x, v0, gy, gx, nmat, M, N, T, G, eps, gscale); return 0;
    case 4: run_gram_bwd<4>(x, v0, gy, gx, nmat, M, N, T, G, eps, gscale); return 0;
    case 8: run_gram_bwd<8>(x, v0, gy, gx, nmat, M, N, T, G, eps, gscale); return 0;
    default: return -2;
  }
}

#define DISPATCH_R(FN, M, NPL, ...)                                            \
  switch (R * 2 + solver) {                                                    \
    case 2: FN<M, NPL, 1, 0>(__VA_ARGS__); return 0;                           \
    case 3: FN<M, NPL, 1, 1>(__VA_ARGS__); return 0;                           \
    case 4: FN<M, NPL, 2, 0>(__VA_ARGS__); return 0;                           \
    case 5: FN<M, NPL, 2, 1>(__VA_ARGS__); return 0;                           \
    case 6: FN<M, NPL, 3, 0>(__VA_ARGS__); return 0;                           \
    case 7: FN<M, NPL, 3, 1>(__VA_ARGS__); return 0;                           \
    case 8: FN<M, NPL, 4, 0>(__VA_ARGS__); return 0;                           \
    case 9: FN<M, NPL, 4, 1>(__VA_ARGS__); return 0;                           \
    default: return -2;                                                        \
  }

extern "C" int emu_nmf_fwd(const float* x, const float* u0, const float* v0, float* y, float* uo, float* vo,
                           int64_t nmat, int M, int N, int R, int T, int solver, float eps) {
  if (M <= 8 && N <= 64) { DISPATCH_R(run_fwd, 8, 1, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 8 && N <= 128) { DISPATCH_R(run_fwd, 8, 2, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 8 && N <= 192) { DISPATCH_R(run_fwd, 8, 3, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 8 && N <= 256) { DISPATCH_R(run_fwd, 8, 4, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 8 && N <= 512) { DISPATCH_R(run_fwd, 8, 8, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 16 && N <= 64) { DISPATCH_R(run_fwd, 16, 1, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 16 && N <= 256) { DISPATCH_R(run_fwd, 16, 4, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 32 && N <= 64) { DISPATCH_R(run_fwd, 32, 1, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  if (M <= 32 && N <= 128) { DISPATCH_R(run_fwd, 32, 2, x, u0, v0, y, uo, vo, nmat, M, N, T, eps) }
  return -2;
}

extern "C" int emu_nmf_bwd(const float* x, const float* u0, const float* v0, const float* gy, const float* gu,
                           const float* gv, float* gx, int64_t nmat, int M, int N, int R, int T, int G,
                           int solver, float eps) {
  if (M <= 8 && N <= 64) { DISPATCH_R(run_bwd, 8, 1, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 8 && N <= 128) { DISPATCH_R(run_bwd, 8, 2, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 8 && N <= 192) { DISPATCH_R(run_bwd, 8, 3, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 8 && N <= 256) { DISPATCH_R(run_bwd, 8, 4, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 8 && N <= 512) { DISPATCH_R(run_bwd, 8, 8, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 16 && N <= 64) { DISPATCH_R(run_bwd, 16, 1, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 16 && N <= 256) { DISPATCH_R(run_bwd, 16, 4, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 32 && N <= 64) { DISPATCH_R(run_bwd, 32, 1, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  if (M <= 32 && N <= 128) { DISPATCH_R(run_bwd, 32, 2, x, u0, v0, gy, gu, gv, gx, nmat, M, N, T, G, eps) }
  return -2;
}
