// nmf_global.hip — split-N NMF: matrices too wide for one wavefront.
//
// The reference's DEFAULT FactMixer reshape is the global Matricize(num_heads=1, grid_size=1)
// (factorizer/factorizer.py:17; its own tests run it: tests/test_factorizer.py:25 — x (1,16,64³) →
// ONE 16 x 262 144 matrix), and `num_heads=` forms give M != 8 (tests/test_factorizer.py:123).  The
// wave-resident kernels (nmf_kernels.inc: M <= 32, N <= 512) cannot hold such a matrix, so the columns are
// split over workgroups ("slabs" of 1024 columns) and one iteration of MatrixFactorization.decompose
// (matrix_factorization.py:514-530) becomes
//
//     gn_u_kernel   (1 workgroup / matrix)   a = Σ_slabs partial(X v),  b = Σ_slabs partial(vᵀv)   [fixed order]
//                                            u ← update(u; a, b)                                   (:241-247, :210-229)
//     gn_fwd_kernel (1 workgroup / slab)     v ← update(v; Xᵀu, uᵀu) on its columns (local: a column's
//                                            update needs no other column), then the partial sums of X v and
//                                            vᵀv of the NEW v for the next iteration; the last one writes
//                                            y = u vᵀ (:532-533)
//
// i.e. 2T + 1 launches per forward instead of ~60 framework launches, X read twice per iteration (16 MB for
// the test shape: it lives in the 256 MB Infinity Cache), no float atomics: the cross-workgroup reductions
// are two-stage sums in a fixed order, so results are bitwise reproducible.  The kernel boundary is the
// inter-workgroup synchronisation (1.5-2 us each; a software grid barrier costs 4-10 us on this chip,
// MI355X_MICROARCH.md "barrier-xcd").
//
// Backward: the forward is recomputed with the (u_t, v_t, a_t, b_t) history kept in the workspace (v_t: N·R
// floats per state), then the reverse sweep of SURVEY.md Appendix A runs as the same alternation: the column-
// local parts (V-update undo, gX, gV) in gn_bwd_kernel, the M x R parts (gU, U-update undo) in gn_bwd_u_kernel,
// with `half_bwd_row` of nmf_core.h — the very code the wave kernels run — for the per-row algebra.
#include "fz_common.h"
#include "nmf_core.h"

namespace fz {

constexpr int kGnThreads = 256;
constexpr int kGnMaxM = 64;
constexpr int kGnMaxR = 4;

// sum of a per-lane partial over the wave, fixed order (DPP tree of fz_common.h)
__device__ __forceinline__ float gn_wsum(float v) { return wave_sum(v); }

// ---- forward ------------------------------------------------------------------------------------------
// u == nullptr: no V-update (v = v_in): the initial partial sums, or (with Y) a bare reconstruction.
template <int R, int SOLVER, int CV, bool XREG>
__global__ __launch_bounds__(kGnThreads) void gn_fwd_kernel(const float* __restrict__ X, int M, int64_t N,
                                                            const float* __restrict__ u, int64_t u_stride,
                                                            int do_update, const float* __restrict__ v_in,
                                                            int64_t v_in_stride, float* __restrict__ v_out,
                                                            float* __restrict__ part, float* __restrict__ Y, float eps) {
  __shared__ float su[kGnMaxM * kGnMaxR];
  __shared__ float sb[kGnMaxR * kGnMaxR];
  __shared__ float red[4][kGnMaxM * kGnMaxR + kGnMaxR * kGnMaxR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t mat = blockIdx.y;
  const int slab = blockIdx.x, nslab = gridDim.x;
  const int64_t n0 = ((int64_t)slab * kGnThreads + tid) * CV;
  const float* Xm = X + mat * M * N + n0;
  bool ok[CV];
#pragma unroll
  for (int c = 0; c < CV; ++c) ok[c] = n0 + c < N;
  const bool any = n0 < N;   // CV == 4 requires N % 4 == 0: a thread's columns are all in or all out

  float v[CV][R];
  {
    const float* vp = v_in + mat * v_in_stride + n0 * R;
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) v[c][r] = ok[c] ? vp[c * R + r] : 0.f;
  }
  if (u != nullptr) {
    for (int e = tid; e < M * R; e += kGnThreads) su[e] = u[mat * u_stride + e];
    __syncthreads();
    if (tid < R * R) {
      const int r = tid / R, q = tid % R;
      float acc = su[r] * su[q];
      for (int m = 1; m < M; ++m) acc = acc + su[m * R + r] * su[m * R + q];
      sb[tid] = acc;
    }
    __syncthreads();
  }
  float xr[XREG ? 16 : 1][CV];
  auto load_x = [&](int m, float (&x)[CV]) {
    if (CV == 4) {
      const float4 t = any ? *reinterpret_cast<const float4*>(Xm + (int64_t)m * N) : make_float4(0.f, 0.f, 0.f, 0.f);
      x[0] = t.x; x[1 % CV] = t.y; x[2 % CV] = t.z; x[3 % CV] = t.w;
    } else {
#pragma unroll
      for (int c = 0; c < CV; ++c) x[c] = ok[c] ? Xm[(int64_t)m * N + c] : 0.f;
    }
  };
  if (XREG) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < M) load_x(m, xr[XREG ? m : 0]);
      else {
#pragma unroll
        for (int c = 0; c < CV; ++c) xr[XREG ? m : 0][c] = 0.f;
      }
    }
  }
  if (u != nullptr && do_update) {
    // ---- V-update on this thread's columns: a' = Xᵀu (per column), b' = uᵀu ----
    float ap[CV][R];
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) ap[c][r] = 0.f;
    if (XREG) {
#pragma unroll
      for (int m = 0; m < 16; ++m)
        if (m < M) {
#pragma unroll
          for (int c = 0; c < CV; ++c)
#pragma unroll
            for (int r = 0; r < R; ++r) ap[c][r] = ap[c][r] + xr[XREG ? m : 0][c] * su[m * R + r];
        }
    } else {
      for (int m0 = 0; m0 < M; m0 += 8) {   // 8 row loads in flight
        float xb[8][CV];
#pragma unroll
        for (int i = 0; i < 8; ++i) load_x(m0 + i < M ? m0 + i : M - 1, xb[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (m0 + i < M) {
#pragma unroll
            for (int c = 0; c < CV; ++c)
#pragma unroll
              for (int r = 0; r < R; ++r) ap[c][r] = ap[c][r] + xb[i][c] * su[(m0 + i) * R + r];
          }
      }
    }
    float bp[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) bp[r][q] = sb[r * R + q];
    update_rows<CV, R, SOLVER, float>(v, ap, bp, eps);
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) v[c][r] = ok[c] ? v[c][r] : 0.f;
    if (v_out != nullptr) {
      float* vo = v_out + mat * N * R + n0 * R;
#pragma unroll
      for (int c = 0; c < CV; ++c)
        if (ok[c]) {
#pragma unroll
          for (int r = 0; r < R; ++r) vo[c * R + r] = v[c][r];
        }
    }
  }
  if (Y != nullptr && any) {
    // y = u vᵀ (matrix_factorization.py:532-533)
    float* Ym = Y + mat * M * N + n0;
    for (int m = 0; m < M; ++m) {
      float o[CV];
#pragma unroll
      for (int c = 0; c < CV; ++c) {
        float acc = su[m * R] * v[c][0];
#pragma unroll
        for (int r = 1; r < R; ++r) acc = acc + su[m * R + r] * v[c][r];
        o[c] = acc;
      }
      if (CV == 4) *reinterpret_cast<float4*>(Ym + (int64_t)m * N) = make_float4(o[0], o[1 % CV], o[2 % CV], o[3 % CV]);
      else {
#pragma unroll
        for (int c = 0; c < CV; ++c)
          if (ok[c]) Ym[(int64_t)m * N + c] = o[c];
      }
    }
  }
  if (part == nullptr) return;
  // ---- partial sums of X v and vᵀv over this slab (for the next U-update) ----
  auto row_partial = [&](int m, const float (&x)[CV]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float t = x[0] * v[0][r];
#pragma unroll
      for (int c = 1; c < CV; ++c) t = t + x[c] * v[c][r];
      t = gn_wsum(t);
      if (lane == 0) red[wave][m * R + r] = t;
    }
  };
  if (XREG) {
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (m < M) row_partial(m, xr[XREG ? m : 0]);
  } else {
    for (int m0 = 0; m0 < M; m0 += 8) {
      float xb[8][CV];
#pragma unroll
      for (int i = 0; i < 8; ++i) load_x(m0 + i < M ? m0 + i : M - 1, xb[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m0 + i < M) row_partial(m0 + i, xb[i]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < R; ++q) {
      float t = v[0][r] * v[0][q];
#pragma unroll
      for (int c = 1; c < CV; ++c) t = t + v[c][r] * v[c][q];
      t = gn_wsum(t);
      if (lane == 0) red[wave][M * R + r * R + q] = t;
    }
  __syncthreads();
  const int PE = M * R + R * R;
  for (int e = tid; e < PE; e += kGnThreads)
    part[(mat * nslab + slab) * PE + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// out[e] = Σ_slabs part[mat][slab][e] in a fixed order: lane l adds slabs l, l+64, ..., then the DPP tree
__device__ __forceinline__ void gn_reduce_partials(const float* __restrict__ part, int64_t mat, int nslab, int PE,
                                                   float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* p = part + mat * nslab * PE;
  for (int e = wave; e < PE; e += 4) {
    float s = 0.f;
    for (int k = lane; k < nslab; k += 64) s = s + p[(int64_t)k * PE + e];
    s = wave_sum(s);
    if (lane == 0) out[e] = s;
  }
}

// U-update: u_new = update(u_prev; a, b) with a, b reduced from the slab partials; keeps (a, b) for the backward
template <int R, int SOLVER>
__global__ __launch_bounds__(kGnThreads) void gn_u_kernel(const float* __restrict__ part, int nslab, int M,
                                                          const float* __restrict__ u_prev, int64_t u_prev_stride,
                                                          float* __restrict__ u_new, float* __restrict__ a_hist,
                                                          float* __restrict__ b_hist, float eps) {
  __shared__ float sa[kGnMaxM * kGnMaxR + kGnMaxR * kGnMaxR];
  const int64_t mat = blockIdx.x;
  const int PE = M * R + R * R;
  gn_reduce_partials(part, mat, nslab, PE, sa);
  __syncthreads();
  const int m = threadIdx.x;
  if (m < M) {
    float w[1][R], a[1][R], b[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      w[0][r] = u_prev[mat * u_prev_stride + m * R + r];
      a[0][r] = sa[m * R + r];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) b[r][q] = sa[M * R + r * R + q];
    update_rows<1, R, SOLVER, float>(w, a, b, eps);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      u_new[(mat * M + m) * R + r] = w[0][r];
      if (a_hist != nullptr) a_hist[(mat * M + m) * R + r] = a[0][r];
    }
  }
  if (b_hist != nullptr && threadIdx.x < R * R) b_hist[mat * R * R + threadIdx.x] = sa[M * R + threadIdx.x];
}

// ---- backward -----------------------------------------------------------------------------------------
// One launch = [pending column-local part of the U-update undo of step s+1]  (PEND)
//            + [V-update undo of step s]                                      (VUNDO)
// FIRST (s = T-1): the output layer y = u_T v_Tᵀ feeds gv = gYᵀu_T (local) and gu = gY v_T (partials).
template <int R, int SOLVER, int CV, bool XREG, bool FIRST, bool PEND, bool VUNDO>
__global__ __launch_bounds__(kGnThreads) void gn_bwd_kernel(
    const float* __restrict__ X, int M, int64_t N, const float* __restrict__ GY, const float* __restrict__ gv_ext,
    float* __restrict__ GX, float* __restrict__ gvbuf, const float* __restrict__ ga_p, const float* __restrict__ gbu_p,
    const float* __restrict__ u_n, const float* __restrict__ v_old, int64_t v_old_stride,
    const float* __restrict__ v_new, int64_t v_new_stride, float* __restrict__ part, float eps) {
  __shared__ float su[kGnMaxM * kGnMaxR];    // u_{s+1}
  __shared__ float sg[kGnMaxM * kGnMaxR];    // pending ga of step s+1
  __shared__ float sb[kGnMaxR * kGnMaxR];    // u_{s+1}ᵀu_{s+1}
  __shared__ float sgb[kGnMaxR * kGnMaxR];   // pending gbu + gbuᵀ
  __shared__ float red[4][2 * kGnMaxM * kGnMaxR + kGnMaxR * kGnMaxR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t mat = blockIdx.y;
  const int slab = blockIdx.x, nslab = gridDim.x;
  const int64_t n0 = ((int64_t)slab * kGnThreads + tid) * CV;
  const float* Xm = X + mat * M * N + n0;
  bool ok[CV];
#pragma unroll
  for (int c = 0; c < CV; ++c) ok[c] = n0 + c < N;
  const bool any = n0 < N;

  if (FIRST || VUNDO)
    for (int e = tid; e < M * R; e += kGnThreads) su[e] = u_n[mat * M * R + e];
  if (PEND) {
    for (int e = tid; e < M * R; e += kGnThreads) sg[e] = ga_p[mat * M * R + e];
    if (tid < R * R) {
      const int q = tid / R, r = tid % R;
      sgb[tid] = gbu_p[mat * R * R + q * R + r] + gbu_p[mat * R * R + r * R + q];
    }
  }
  __syncthreads();
  if ((FIRST || VUNDO) && tid < R * R) {
    const int r = tid / R, q = tid % R;
    float acc = su[r] * su[q];
    for (int m = 1; m < M; ++m) acc = acc + su[m * R + r] * su[m * R + q];
    sb[tid] = acc;
  }
  __syncthreads();

  float vn[CV][R], vo[CV][R], gv[CV][R];
  {
    const float* vp = v_new + mat * v_new_stride + n0 * R;   // (stride 0: the broadcast v0 of state 0)
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) vn[c][r] = ok[c] ? vp[c * R + r] : 0.f;
  }
  if (VUNDO) {
    const float* vp = v_old + mat * v_old_stride + n0 * R;
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) vo[c][r] = ok[c] ? vp[c * R + r] : 0.f;
  }
  {
    const float* gp = FIRST ? gv_ext : gvbuf;
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) gv[c][r] = (gp != nullptr && ok[c]) ? gp[mat * N * R + (n0 + c) * R + r] : 0.f;
  }

  auto load_row = [&](const float* base, int m, float (&x)[CV]) {
    if (CV == 4) {
      const float4 t = any ? *reinterpret_cast<const float4*>(base + (int64_t)m * N) : make_float4(0.f, 0.f, 0.f, 0.f);
      x[0] = t.x; x[1 % CV] = t.y; x[2 % CV] = t.z; x[3 % CV] = t.w;
    } else {
#pragma unroll
      for (int c = 0; c < CV; ++c) x[c] = ok[c] ? base[(int64_t)m * N + c] : 0.f;
    }
  };
  const float* GYm = (FIRST && GY != nullptr) ? GY + mat * M * N + n0 : nullptr;

  // ---- pass A over the rows: everything that must be complete before the per-column undo ----
  float ap[CV][R];
#pragma unroll
  for (int c = 0; c < CV; ++c)
#pragma unroll
    for (int r = 0; r < R; ++r) ap[c][r] = 0.f;
  float xr[XREG ? 16 : 1][CV];
  auto pass_a_row = [&](int m, const float (&x)[CV]) {
    if (FIRST && GYm != nullptr) {
      float gy[CV];
      load_row(GYm, m, gy);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float t = gy[0] * vn[0][r];
#pragma unroll
        for (int c = 0; c < CV; ++c) {
          gv[c][r] = gv[c][r] + gy[c] * su[m * R + r];
          if (c > 0) t = t + gy[c] * vn[c][r];
        }
        t = gn_wsum(t);
        if (lane == 0) red[wave][M * R + R * R + m * R + r] = t;   // gu0 = gY v_T
      }
    }
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (PEND) gv[c][r] = gv[c][r] + x[c] * sg[m * R + r];
        if (VUNDO) ap[c][r] = ap[c][r] + x[c] * su[m * R + r];
      }
  };
  if (FIRST && GYm == nullptr) {
    for (int e = lane; e < M * R; e += 64) red[wave][M * R + R * R + e] = 0.f;
  }
  if (XREG) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < M) {
        load_row(Xm, m, xr[XREG ? m : 0]);
        pass_a_row(m, xr[XREG ? m : 0]);
      }
    }
  } else {
    for (int m0 = 0; m0 < M; m0 += 8) {
      float xb[8][CV];
#pragma unroll
      for (int i = 0; i < 8; ++i) load_row(Xm, m0 + i < M ? m0 + i : M - 1, xb[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m0 + i < M) pass_a_row(m0 + i, xb[i]);
    }
  }
  if (PEND) {
    // gv_{s+1} += v_{s+1} (gbu + gbuᵀ)   (the old factor of the U-update of step s+1 is v_{s+1})
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = gv[c][r];
#pragma unroll
        for (int q = 0; q < R; ++q) acc = acc + vn[c][q] * sgb[q * R + r];
        gv[c][r] = acc;
      }
  }
  // ---- V-update undo, one column at a time (half_bwd_row: the wave kernels' code) ----
  float gaj[CV][R];
  float gb[R][R];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < R; ++q) gb[r][q] = 0.f;
  if (VUNDO) {
    float bp[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) bp[r][q] = sb[r * R + q];
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      float gwn[R], gwo[R], ga1[R];
#pragma unroll
      for (int r = 0; r < R; ++r) gwn[r] = gv[c][r];
      half_bwd_row<R, SOLVER, float>(vo[c], vn[c], ap[c], bp, gwn, gwo, ga1, gb, eps);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        gaj[c][r] = ok[c] ? ga1[r] : 0.f;
        gv[c][r] = gwo[r];
      }
    }
    float* go = gvbuf + mat * N * R + n0 * R;
#pragma unroll
    for (int c = 0; c < CV; ++c)
      if (ok[c]) {
#pragma unroll
        for (int r = 0; r < R; ++r) go[c * R + r] = gv[c][r];
      }
  }
  // ---- pass B over the rows: gX and the partial sums of X·ga ----
  float* GXm = GX + mat * M * N + n0;
  auto pass_b_row = [&](int m, const float (&x)[CV]) {
    float g[CV];
    if (FIRST) {
#pragma unroll
      for (int c = 0; c < CV; ++c) g[c] = 0.f;
    } else {
      load_row(GXm, m, g);
    }
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      float acc = g[c];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (PEND) acc = acc + sg[m * R + r] * vn[c][r];
        if (VUNDO) acc = acc + su[m * R + r] * gaj[c][r];
      }
      g[c] = acc;
    }
    if (any) {
      if (CV == 4) *reinterpret_cast<float4*>(GXm + (int64_t)m * N) = make_float4(g[0], g[1 % CV], g[2 % CV], g[3 % CV]);
      else {
#pragma unroll
        for (int c = 0; c < CV; ++c)
          if (ok[c]) GXm[(int64_t)m * N + c] = g[c];
      }
    }
    if (VUNDO) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float t = x[0] * gaj[0][r];
#pragma unroll
        for (int c = 1; c < CV; ++c) t = t + x[c] * gaj[c][r];
        t = gn_wsum(t);
        if (lane == 0) red[wave][m * R + r] = t;
      }
    }
  };
  if (XREG) {
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (m < M) pass_b_row(m, xr[XREG ? m : 0]);
  } else {
    for (int m0 = 0; m0 < M; m0 += 8) {
      float xb[8][CV];
      if (VUNDO) {
#pragma unroll
        for (int i = 0; i < 8; ++i) load_row(Xm, m0 + i < M ? m0 + i : M - 1, xb[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m0 + i < M) pass_b_row(m0 + i, xb[VUNDO ? i : 0]);
    }
  }
  if (!VUNDO) return;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const float t = gn_wsum(any ? gb[r][q] : 0.f);
      if (lane == 0) red[wave][M * R + r * R + q] = t;
    }
  __syncthreads();
  const int PE = M * R + R * R + (FIRST ? M * R : 0);
  for (int e = tid; e < PE; e += kGnThreads)
    part[(mat * nslab + slab) * PE + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// gu (dL/du_{s+1}) ← gu + Σ partial(X ga) + u_{s+1}(gb + gbᵀ); then the U-update undo of step s:
// ga (M x R), gbu (R x R) for the column-local part, gu ← dL/du_s.
template <int R, int SOLVER, bool FIRST>
__global__ __launch_bounds__(kGnThreads) void gn_bwd_u_kernel(const float* __restrict__ part, int nslab, int M,
                                                              float* __restrict__ gu, const float* __restrict__ gu_ext,
                                                              const float* __restrict__ u_old, int64_t u_old_stride,
                                                              const float* __restrict__ u_new,
                                                              const float* __restrict__ a_s, const float* __restrict__ b_s,
                                                              float* __restrict__ ga, float* __restrict__ gbu, float eps) {
  __shared__ float sp[2 * kGnMaxM * kGnMaxR + kGnMaxR * kGnMaxR];
  __shared__ float sgbu[kGnMaxM][kGnMaxR * kGnMaxR];
  const int64_t mat = blockIdx.x;
  const int PE = M * R + R * R + (FIRST ? M * R : 0);
  gn_reduce_partials(part, mat, nslab, PE, sp);
  __syncthreads();
  const int m = threadIdx.x;
  if (m < M) {
    float un[R], uo[R], as[R], bs[R][R], gwn[R], gwo[R], gam[R], gbt[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      un[r] = u_new[(mat * M + m) * R + r];
      uo[r] = u_old[mat * u_old_stride + m * R + r];
      as[r] = a_s[(mat * M + m) * R + r];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) {
        bs[r][q] = b_s[mat * R * R + r * R + q];
        gbt[r][q] = 0.f;
      }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float acc = FIRST ? sp[M * R + R * R + m * R + r] + (gu_ext != nullptr ? gu_ext[(mat * M + m) * R + r] : 0.f)
                        : gu[(mat * M + m) * R + r];
      acc = acc + sp[m * R + r];
#pragma unroll
      for (int q = 0; q < R; ++q) acc = acc + un[q] * (sp[M * R + q * R + r] + sp[M * R + r * R + q]);
      gwn[r] = acc;
    }
    half_bwd_row<R, SOLVER, float>(uo, un, as, bs, gwn, gwo, gam, gbt, eps);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ga[(mat * M + m) * R + r] = gam[r];
      gu[(mat * M + m) * R + r] = gwo[r];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) sgbu[m][r * R + q] = gbt[r][q];
  }
  __syncthreads();
  if (threadIdx.x < R * R) {
    float acc = sgbu[0][threadIdx.x];
    for (int i = 1; i < M; ++i) acc = acc + sgbu[i][threadIdx.x];   // rows in order: reproducible
    gbu[mat * R * R + threadIdx.x] = acc;
  }
}

// ---- workspace layout (floats) ------------------------------------------------------------------------
struct GnWs {
  int64_t uh, vh, ah, bh, part, gvbuf, gu, ga, gbu, total;
};
static GnWs gn_layout(int64_t nmat, int M, int64_t N, int R, int T, bool backward) {
  GnWs w;
  const int64_t nslab = (N + 255) / 256;  // upper bound: the scalar-column variant (CV = 1) has 256-column slabs
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
  const int Tn = T > 0 ? T : 1;
  w.uh = take((int64_t)Tn * nmat * M * R);                                  // u_1..u_T
  w.vh = take((backward ? (int64_t)Tn : 2) * nmat * N * R);                 // v_1..v_T (or a ping-pong pair)
  w.ah = take((int64_t)Tn * nmat * M * R);
  w.bh = take((int64_t)Tn * nmat * R * R);
  w.part = take(nmat * nslab * (2 * (int64_t)M * R + R * R));
  w.gvbuf = take(backward ? nmat * N * R : 0);
  w.gu = take(backward ? nmat * M * R : 0);
  w.ga = take(backward ? nmat * M * R : 0);
  w.gbu = take(backward ? nmat * R * R : 0);
  w.total = o;
  return w;
}

struct GnCall {
  const float *x, *u0, *v0;
  int64_t nmat;
  int M;
  int64_t N;
  int R, T, solver;
  float eps;
  float* ws;
  hipStream_t st;
};

template <int R, int SOLVER, int CV>
static int gn_forward_t(const GnCall& c, const GnWs& L, bool hist, float* y, float* u_out, float* v_out) {
  const int cols = kGnThreads * CV;
  const int nslab = (int)((c.N + cols - 1) / cols);
  dim3 grid(nslab, (unsigned)c.nmat), block(kGnThreads);
  const bool xreg = c.M <= 16;
  float* part = c.ws + L.part;
  const int64_t MR = (int64_t)c.M * R, NR = c.N * R;
  auto U = [&](int t) { return c.ws + L.uh + (int64_t)(t - 1) * c.nmat * MR; };                       // t = 1..T
  auto V = [&](int t) { return c.ws + L.vh + (int64_t)(hist ? t - 1 : (t & 1)) * c.nmat * NR; };      // t = 1..T
#define FZ_GN_FWD(...)                                                                                        \
  do {                                                                                                        \
    if (xreg) hipLaunchKernelGGL((gn_fwd_kernel<R, SOLVER, CV, true>), grid, block, 0, c.st, __VA_ARGS__);    \
    else hipLaunchKernelGGL((gn_fwd_kernel<R, SOLVER, CV, false>), grid, block, 0, c.st, __VA_ARGS__);        \
    FZ_LAUNCH_CHECK();                                                                                        \
  } while (0)
  if (c.T == 0) {  // y = u0 v0ᵀ
    FZ_GN_FWD(c.x, c.M, c.N, c.u0, (int64_t)0, 0, c.v0, (int64_t)0, (float*)nullptr, (float*)nullptr, y, c.eps);
    return FZ_OK;
  }
  // partial sums of X v_0, v_0ᵀ v_0
  FZ_GN_FWD(c.x, c.M, c.N, (const float*)nullptr, (int64_t)0, 0, c.v0, (int64_t)0, (float*)nullptr, part, (float*)nullptr, c.eps);
  for (int t = 1; t <= c.T; ++t) {
    const float* up = t == 1 ? c.u0 : U(t - 1);
    float* un = (t == c.T && u_out != nullptr) ? u_out : U(t);
    hipLaunchKernelGGL((gn_u_kernel<R, SOLVER>), dim3((unsigned)c.nmat), block, 0, c.st, part, nslab, c.M, up,
                       (int64_t)(t == 1 ? 0 : MR), un, hist ? c.ws + L.ah + (int64_t)(t - 1) * c.nmat * MR : (float*)nullptr,
                       hist ? c.ws + L.bh + (int64_t)(t - 1) * c.nmat * R * R : (float*)nullptr, c.eps);
    FZ_LAUNCH_CHECK();
    const float* vp = t == 1 ? c.v0 : V(t - 1);
    float* vn = (t == c.T && v_out != nullptr) ? v_out : V(t);
    FZ_GN_FWD(c.x, c.M, c.N, (const float*)un, MR, 1, vp, (int64_t)(t == 1 ? 0 : NR), vn,
              t < c.T ? part : (float*)nullptr, t == c.T ? y : (float*)nullptr, c.eps);
  }
#undef FZ_GN_FWD
  return FZ_OK;
}

template <int R, int SOLVER, int CV>
static int gn_backward_t(const GnCall& c, const GnWs& L, int G, const float* gy, const float* gu_ext, const float* gv_ext,
                         float* gx) {
  int rc = gn_forward_t<R, SOLVER, CV>(c, L, true, nullptr, nullptr, nullptr);
  if (rc != FZ_OK) return rc;
  const int cols = kGnThreads * CV;
  const int nslab = (int)((c.N + cols - 1) / cols);
  dim3 grid(nslab, (unsigned)c.nmat), block(kGnThreads);
  const bool xreg = c.M <= 16;
  const int64_t MR = (int64_t)c.M * R, NR = c.N * R;
  float* part = c.ws + L.part;
  float* gvbuf = c.ws + L.gvbuf;
  float* gu = c.ws + L.gu;
  float* ga = c.ws + L.ga;
  float* gbu = c.ws + L.gbu;
  auto U = [&](int t) { return t == 0 ? c.u0 : c.ws + L.uh + (int64_t)(t - 1) * c.nmat * MR; };
  auto V = [&](int t) { return t == 0 ? c.v0 : c.ws + L.vh + (int64_t)(t - 1) * c.nmat * NR; };
#define FZ_GN_BWD(FI, PE_, VU, ...)                                                                                     \
  do {                                                                                                                  \
    if (xreg) hipLaunchKernelGGL((gn_bwd_kernel<R, SOLVER, CV, true, FI, PE_, VU>), grid, block, 0, c.st, __VA_ARGS__); \
    else hipLaunchKernelGGL((gn_bwd_kernel<R, SOLVER, CV, false, FI, PE_, VU>), grid, block, 0, c.st, __VA_ARGS__);     \
    FZ_LAUNCH_CHECK();                                                                                                  \
  } while (0)
  for (int s = c.T - 1; s >= c.T - G; --s) {
    const bool first = s == c.T - 1;
    // state s+1 = (U(s+1), V(s+1)), state s = (U(s), V(s)); a_s, b_s recorded by the U-update of step s
    if (first)
      FZ_GN_BWD(true, false, true, c.x, c.M, c.N, gy, gv_ext, gx, gvbuf, (const float*)nullptr, (const float*)nullptr,
                U(s + 1), V(s), (int64_t)(s == 0 ? 0 : NR), V(s + 1), NR, part, c.eps);
    else
      FZ_GN_BWD(false, true, true, c.x, c.M, c.N, (const float*)nullptr, (const float*)nullptr, gx, gvbuf,
                (const float*)ga, (const float*)gbu, U(s + 1), V(s), (int64_t)(s == 0 ? 0 : NR), V(s + 1), NR, part, c.eps);
    const float* as = c.ws + L.ah + (int64_t)s * c.nmat * MR;
    const float* bs = c.ws + L.bh + (int64_t)s * c.nmat * R * R;
    if (first)
      hipLaunchKernelGGL((gn_bwd_u_kernel<R, SOLVER, true>), dim3((unsigned)c.nmat), block, 0, c.st, part, nslab, c.M, gu,
                         gu_ext, U(s), (int64_t)(s == 0 ? 0 : MR), U(s + 1), as, bs, ga, gbu, c.eps);
    else
      hipLaunchKernelGGL((gn_bwd_u_kernel<R, SOLVER, false>), dim3((unsigned)c.nmat), block, 0, c.st, part, nslab, c.M, gu,
                         (const float*)nullptr, U(s), (int64_t)(s == 0 ? 0 : MR), U(s + 1), as, bs, ga, gbu, c.eps);
    FZ_LAUNCH_CHECK();
  }
  // column-local part of the last U-update undo (step T-G): gX += ga v_{T-G}ᵀ
  const int s0 = c.T - G;
  FZ_GN_BWD(false, true, false, c.x, c.M, c.N, (const float*)nullptr, (const float*)nullptr, gx, gvbuf, (const float*)ga,
            (const float*)gbu, (const float*)nullptr, (const float*)nullptr, (int64_t)0,
            s0 == 0 ? c.v0 : (const float*)V(s0), (int64_t)(s0 == 0 ? 0 : NR), part, c.eps);
#undef FZ_GN_BWD
  return FZ_OK;
}

}  // namespace fz

using namespace fz;

extern "C" int fz_gnmf_supported(int M, int64_t N, int R, int T, int Tgrad) {
  (void)Tgrad;
  return (M >= 1 && M <= kGnMaxM && N >= 1 && R >= 1 && R <= kGnMaxR && T >= 0) ? 1 : 0;
}

extern "C" int64_t fz_gnmf_workspace_bytes(int64_t nmat, int M, int64_t N, int R, int T, int backward) {
  if (!fz_gnmf_supported(M, N, R, T, T) || nmat < 0) return -1;
  return gn_layout(nmat, M, N, R, T, backward != 0).total * (int64_t)sizeof(float);
}

// number of kernel launches of one call (tests assert it against fz_launch_count)
extern "C" int fz_gnmf_launches(int T, int Tgrad, int backward) {
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int fwd = T == 0 ? 1 : 2 * T + 1;
  return backward ? fwd + 2 * G + 1 : fwd;
}

#define FZ_GN_DISPATCH(CALL)                                                                         \
  do {                                                                                               \
    const bool v4 = (N % 4) == 0;                                                                    \
    if (solver == FZ_SOLVER_MU) {                                                                    \
      switch (R) {                                                                                   \
        case 1: return v4 ? CALL(1, SOLVER_MU, 4) : CALL(1, SOLVER_MU, 1);                           \
        case 2: return v4 ? CALL(2, SOLVER_MU, 4) : CALL(2, SOLVER_MU, 1);                           \
        case 3: return v4 ? CALL(3, SOLVER_MU, 4) : CALL(3, SOLVER_MU, 1);                           \
        default: return v4 ? CALL(4, SOLVER_MU, 4) : CALL(4, SOLVER_MU, 1);                          \
      }                                                                                              \
    }                                                                                                \
    switch (R) {                                                                                     \
      case 1: return v4 ? CALL(1, SOLVER_HALS, 4) : CALL(1, SOLVER_HALS, 1);                         \
      case 2: return v4 ? CALL(2, SOLVER_HALS, 4) : CALL(2, SOLVER_HALS, 1);                         \
      case 3: return v4 ? CALL(3, SOLVER_HALS, 4) : CALL(3, SOLVER_HALS, 1);                         \
      default: return v4 ? CALL(4, SOLVER_HALS, 4) : CALL(4, SOLVER_HALS, 1);                        \
    }                                                                                                \
  } while (0)

extern "C" int fz_gnmf_fwd(const float* x, const float* u0, const float* v0, float* y, float* u_out, float* v_out,
                           int64_t nmat, int M, int64_t N, int R, int T, int solver, float eps, void* workspace,
                           fz_stream_t stream) {
  if (!fz_gnmf_supported(M, N, R, T, T)) return fail(FZ_E_UNSUPPORTED, "fz_gnmf_fwd: needs 1 <= M <= 64, 1 <= R <= 4");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_gnmf_fwd: bad solver");
  if (nmat < 0 || nmat > 65535) return fail(FZ_E_SHAPE, "fz_gnmf_fwd: 0 <= nmat <= 65535");
  if (!x || !u0 || !v0 || !y || !workspace) return fail(FZ_E_ARG, "fz_gnmf_fwd: null pointer");
  if (nmat == 0) return FZ_OK;
  const GnWs L = gn_layout(nmat, M, N, R, T, false);
  const GnCall c{x, u0, v0, nmat, M, N, R, T, solver, eps, (float*)workspace, (hipStream_t)stream};
#define FZ_GN_F(RR, SS, CC) gn_forward_t<RR, SS, CC>(c, L, false, y, u_out, v_out)
  FZ_GN_DISPATCH(FZ_GN_F);
#undef FZ_GN_F
}

extern "C" int fz_gnmf_bwd(const float* x, const float* u0, const float* v0, const float* gy, const float* gu,
                           const float* gv, float* gx, int64_t nmat, int M, int64_t N, int R, int T, int Tgrad,
                           int solver, float eps, void* workspace, fz_stream_t stream) {
  if (!fz_gnmf_supported(M, N, R, T, Tgrad)) return fail(FZ_E_UNSUPPORTED, "fz_gnmf_bwd: needs 1 <= M <= 64, 1 <= R <= 4");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_gnmf_bwd: bad solver");
  if (nmat < 0 || nmat > 65535) return fail(FZ_E_SHAPE, "fz_gnmf_bwd: 0 <= nmat <= 65535");
  if (!x || !u0 || !v0 || !gx || !workspace || (!gy && !gu && !gv)) return fail(FZ_E_ARG, "fz_gnmf_bwd: null pointer");
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  if (G < 1) return fail(FZ_E_ARG, "fz_gnmf_bwd: no iteration carries gradient (the caller returns zeros)");
  if (nmat == 0) return FZ_OK;
  const GnWs L = gn_layout(nmat, M, N, R, T, true);
  const GnCall c{x, u0, v0, nmat, M, N, R, T, solver, eps, (float*)workspace, (hipStream_t)stream};
#define FZ_GN_B(RR, SS, CC) gn_backward_t<RR, SS, CC>(c, L, G, gy, gu, gv, gx)
  FZ_GN_DISPATCH(FZ_GN_B);
#undef FZ_GN_B
}
