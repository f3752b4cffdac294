// swm.hip — shifted-window matricize / inverse for gfx950 (C ABI: fz_swm_fwd, fz_swm_inv).
//
// Replaces, in ONE pass per window, the reference's torch.roll (3 sequential 1-D roll copies)
// + einops rearrange copy + torch.cat copy (factorization/operations.py:266-272, 321-325,
// 417-421) and, for the inverse, rearrange⁻¹ + roll back + add + div (:274-280, 423-434).
//
// Work decomposition: one workgroup per (window, b, head, g0, g1) "pencil" = all G2 patches
// along W for one pair of patch rows.  Inside, threads are laid out as (g2, p1, p2-chunk):
//   * 16 consecutive lanes = one patch row-block (p1 × p2 floats) = 256 B contiguous in the
//     matricized tensor y  → full 128-B lines on the y side;
//   * the same 64 lanes cover p1 rows × (4 patches × p2) = 128 B contiguous per row of the
//     channels-first tensor x → full lines on the x side too (unshifted; a shifted window
//     straddles two lines per row, the neighbouring wave of the same workgroup takes the rest).
// All index divisions are hoisted out of the (channel, p0) inner loop; data moves as 16-byte
// (or 8/4-byte, when shifts/patches are not multiples of 4) vectors, so the kernels are pure
// HBM streams: algorithmic bytes = (1 + nshift)·B·C·V·s each way.
#include "fz_common.h"

namespace fz {

constexpr int kMaxShifts = 8;

struct SwmGeom {
  int B, C, D, H, W;     // channels-first tensor
  int d, h;              // head_dim, heads
  int p0, p1, p2;        // patch
  int G0, G1, G2;        // grid
  int nshift;
  int s[kMaxShifts][3];  // normalised to [0, S_i)
};

template <int VB>
struct VecT;
template <>
struct VecT<16> { using type = uint4; };
template <>
struct VecT<8> { using type = uint2; };
template <>
struct VecT<4> { using type = uint32_t; };
template <>
struct VecT<2> { using type = uint16_t; };

__device__ __forceinline__ float relu_f(float v) { return v > 0.f ? v : (v != v ? v : 0.f); }

// ------------------------------------------------------------------------------------------
// forward: x -> y, element size ES bytes, vector VB bytes (VB/ES elements along p2)
// ------------------------------------------------------------------------------------------
template <int ES, int VB, bool RELU, bool DIV>
__global__ __launch_bounds__(256) void swm_fwd_kernel(const char* __restrict__ x, char* __restrict__ y,
                                                      SwmGeom q, float divisor) {
  using V = typename VecT<VB>::type;
  constexpr int VE = VB / ES;  // elements per vector
  // block -> (w, b, hh, g0, g1)
  int bid = blockIdx.x;
  const int g1 = bid % q.G1; bid /= q.G1;
  const int g0 = bid % q.G0; bid /= q.G0;
  const int hh = bid % q.h;  bid /= q.h;
  const int b = bid % q.B;
  const int w = bid / q.B;
  const int s0 = q.s[w][0], s1 = q.s[w][1], s2 = q.s[w][2];
  const int CP = q.p2 / VE;
  const int P = q.p0 * q.p1 * q.p2;
  const int64_t HW = (int64_t)q.H * q.W;
  const int64_t V3 = (int64_t)q.D * HW;
  const int G = q.G0 * q.G1 * q.G2;
  const int64_t xbase = ((int64_t)b * q.C + (int64_t)hh * q.d) * V3;
  const int64_t ybase = ((((int64_t)w * q.B + b) * q.h + hh) * G + ((int64_t)g0 * q.G1 + g1) * q.G2) * q.d * P;
  const int items = q.G2 * q.p1 * CP;
  for (int f = threadIdx.x; f < items; f += blockDim.x) {
    const int p2c = f % CP;
    const int t = f / CP;
    const int p1i = t % q.p1;
    const int g2 = t / q.p1;
    int zh = g1 * q.p1 + p1i - s1; if (zh < 0) zh += q.H;
    int zw = g2 * q.p2 + p2c * VE - s2; if (zw < 0) zw += q.W;
    const int64_t xoff = (int64_t)zh * q.W + zw;
    const int64_t yoff = (int64_t)g2 * q.d * P + (p1i * q.p2 + p2c * VE);
    for (int dd = 0; dd < q.d; ++dd) {
      for (int p0i = 0; p0i < q.p0; ++p0i) {
        int zd = g0 * q.p0 + p0i - s0; if (zd < 0) zd += q.D;
        const int64_t xi = xbase + (int64_t)dd * V3 + (int64_t)zd * HW + xoff;
        const int64_t yi = ybase + (int64_t)dd * P + (int64_t)p0i * q.p1 * q.p2 + yoff;
        V v = *reinterpret_cast<const V*>(x + xi * ES);
        if (RELU || DIV) {
          // fp32 only (ES == 4)
          float* fv = reinterpret_cast<float*>(&v);
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            float a = fv[e];
            if (RELU) a = relu_f(a);
            if (DIV) a = a / divisor;
            fv[e] = a;
          }
        }
        *reinterpret_cast<V*>(y + yi * ES) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// inverse: y (all windows) -> x ;  x = (((0 + z_0) + z_1) + ...) [/ nshift]   (fp32)
// ------------------------------------------------------------------------------------------
// AT: storage type (float, or bf16 with the window sum / average taken in fp32 and rounded once)
template <int VE, bool GATE, typename AT>
__global__ __launch_bounds__(256) void swm_inv_kernel(const AT* __restrict__ y, AT* __restrict__ x,
                                                      const AT* __restrict__ gate, SwmGeom q,
                                                      int average) {
  struct Vf { float e[VE]; };
  // block -> (b, hh, g0, g1) of the UNSHIFTED tiling of x
  int bid = blockIdx.x;
  const int g1 = bid % q.G1; bid /= q.G1;
  const int g0 = bid % q.G0; bid /= q.G0;
  const int hh = bid % q.h;
  const int b = bid / q.h;
  const int CP = q.p2 / VE;
  const int P = q.p0 * q.p1 * q.p2;
  const int64_t HW = (int64_t)q.H * q.W;
  const int64_t V3 = (int64_t)q.D * HW;
  const int G = q.G0 * q.G1 * q.G2;
  const int64_t xbase = ((int64_t)b * q.C + (int64_t)hh * q.d) * V3;
  const int64_t wstride = (int64_t)q.B * q.h * G * q.d * P;  // one window of y
  const int64_t ybh = ((int64_t)b * q.h + hh) * G * q.d * P;
  const float nw = (float)q.nshift;
  const int items = q.G2 * q.p1 * CP;
  for (int f = threadIdx.x; f < items; f += blockDim.x) {
    const int p2c = f % CP;
    const int t = f / CP;
    const int p1i = t % q.p1;
    const int g2 = t / q.p1;
    const int zh = g1 * q.p1 + p1i;
    const int zw = g2 * q.p2 + p2c * VE;
    const int64_t xoff = (int64_t)zh * q.W + zw;
    // per-window source offsets that do not depend on (dd, p0i)
    int64_t yoff[kMaxShifts];
#pragma unroll
    for (int w = 0; w < kMaxShifts; ++w) {
      if (w < q.nshift) {
        int ch = zh + q.s[w][1]; if (ch >= q.H) ch -= q.H;
        int cw = zw + q.s[w][2]; if (cw >= q.W) cw -= q.W;
        const int g1s = ch / q.p1, p1s = ch % q.p1;
        const int g2s = cw / q.p2, p2s = cw % q.p2;
        yoff[w] = w * wstride + ybh + ((int64_t)g1s * q.G2 + g2s) * q.d * P + (p1s * q.p2 + p2s);
      }
    }
    for (int p0i = 0; p0i < q.p0; ++p0i) {
      const int zd = g0 * q.p0 + p0i;
      int64_t yrow[kMaxShifts];
#pragma unroll
      for (int w = 0; w < kMaxShifts; ++w) {
        if (w < q.nshift) {
          int cd = zd + q.s[w][0]; if (cd >= q.D) cd -= q.D;
          const int g0s = cd / q.p0, p0s = cd % q.p0;
          yrow[w] = yoff[w] + (int64_t)g0s * q.G1 * q.G2 * q.d * P + (int64_t)p0s * q.p1 * q.p2;
        }
      }
      for (int dd = 0; dd < q.d; ++dd) {
        Vf acc;
#pragma unroll
        for (int e = 0; e < VE; ++e) acc.e[e] = 0.0f;
#pragma unroll
        for (int w = 0; w < kMaxShifts; ++w) {
          if (w < q.nshift) {
            const int64_t yi = yrow[w] + (int64_t)dd * P;
            Vf z;
            aload<VE>(y + yi, z.e);
            if (GATE) {
              Vf gt;
              aload<VE>(gate + yi, gt.e);
#pragma unroll
              for (int e = 0; e < VE; ++e) z.e[e] = gt.e[e] > 0.f ? z.e[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) acc.e[e] = acc.e[e] + z.e[e];
          }
        }
        if (average) {
#pragma unroll
          for (int e = 0; e < VE; ++e) acc.e[e] = acc.e[e] / nw;
        }
        const int64_t xi = xbase + (int64_t)dd * V3 + (int64_t)zd * HW + xoff;
        astore<VE>(x + xi, acc.e);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward, x-centric: every element of x is read ONCE and stored to its position in each of the
// nshift windows (the index math of the inverse kernel with loads and stores swapped), so the
// HBM traffic equals the algorithmic (1 + nshift)·B·C·V·s.  fp32, 16/8/4-byte vectors.
// ------------------------------------------------------------------------------------------
template <int VE, bool RELU, bool DIV, typename AT>
__global__ __launch_bounds__(256) void swm_fwd_x_kernel(const AT* __restrict__ x, AT* __restrict__ y, SwmGeom q,
                                                        float divisor) {
  struct Vf { float e[VE]; };
  int bid = blockIdx.x;
  const int g1 = bid % q.G1; bid /= q.G1;
  const int g0 = bid % q.G0; bid /= q.G0;
  const int hh = bid % q.h;
  const int b = bid / q.h;
  const int CP = q.p2 / VE;
  const int P = q.p0 * q.p1 * q.p2;
  const int64_t HW = (int64_t)q.H * q.W;
  const int64_t V3 = (int64_t)q.D * HW;
  const int G = q.G0 * q.G1 * q.G2;
  const int64_t xbase = ((int64_t)b * q.C + (int64_t)hh * q.d) * V3;
  const int64_t wstride = (int64_t)q.B * q.h * G * q.d * P;
  const int64_t ybh = ((int64_t)b * q.h + hh) * G * q.d * P;
  const int items = q.G2 * q.p1 * CP;
  for (int f = threadIdx.x; f < items; f += blockDim.x) {
    const int p2c = f % CP;
    const int t = f / CP;
    const int p1i = t % q.p1;
    const int g2 = t / q.p1;
    const int zh = g1 * q.p1 + p1i;
    const int zw = g2 * q.p2 + p2c * VE;
    const int64_t xoff = (int64_t)zh * q.W + zw;
    int64_t yoff[kMaxShifts];
#pragma unroll
    for (int w = 0; w < kMaxShifts; ++w) {
      if (w < q.nshift) {
        int ch = zh + q.s[w][1]; if (ch >= q.H) ch -= q.H;
        int cw = zw + q.s[w][2]; if (cw >= q.W) cw -= q.W;
        const int g1s = ch / q.p1, p1s = ch % q.p1;
        const int g2s = cw / q.p2, p2s = cw % q.p2;
        yoff[w] = w * wstride + ybh + ((int64_t)g1s * q.G2 + g2s) * q.d * P + (p1s * q.p2 + p2s);
      }
    }
    for (int p0i = 0; p0i < q.p0; ++p0i) {
      const int zd = g0 * q.p0 + p0i;
      int64_t yrow[kMaxShifts];
#pragma unroll
      for (int w = 0; w < kMaxShifts; ++w) {
        if (w < q.nshift) {
          int cd = zd + q.s[w][0]; if (cd >= q.D) cd -= q.D;
          const int g0s = cd / q.p0, p0s = cd % q.p0;
          yrow[w] = yoff[w] + (int64_t)g0s * q.G1 * q.G2 * q.d * P + (int64_t)p0s * q.p1 * q.p2;
        }
      }
      for (int dd = 0; dd < q.d; ++dd) {
        Vf v;
        aload<VE>(x + xbase + (int64_t)dd * V3 + (int64_t)zd * HW + xoff, v.e);
        if (RELU || DIV) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            float a = v.e[e];
            if (RELU) a = relu_f(a);
            if (DIV) a = a / divisor;
            v.e[e] = a;
          }
        }
#pragma unroll
        for (int w = 0; w < kMaxShifts; ++w) {
          if (w < q.nshift) astore<VE>(y + yrow[w] + (int64_t)dd * P, v.e);
        }
      }
    }
  }
}

static int make_geom(SwmGeom& q, int B, int C, int D, int H, int W, int d, int pd, int ph, int pw,
                     int nshift, const int* shifts, const char* who) {
  if (B < 0 || C < 1 || D < 1 || H < 1 || W < 1 || d < 1 || pd < 1 || ph < 1 || pw < 1)
    return fail(FZ_E_SHAPE, "fz_swm: non-positive size");
  if (C % d != 0) return fail(FZ_E_SHAPE, "fz_swm: channels not divisible by head_dim");
  if (D % pd != 0 || H % ph != 0 || W % pw != 0)
    return fail(FZ_E_SHAPE, "fz_swm: spatial size not divisible by patch size");
  if (nshift < 1 || nshift > kMaxShifts) return fail(FZ_E_UNSUPPORTED, "fz_swm: 1..8 shift windows supported");
  if (shifts == nullptr) return fail(FZ_E_ARG, "fz_swm: shifts is null");
  q.B = B; q.C = C; q.D = D; q.H = H; q.W = W; q.d = d; q.h = C / d;
  q.p0 = pd; q.p1 = ph; q.p2 = pw; q.G0 = D / pd; q.G1 = H / ph; q.G2 = W / pw; q.nshift = nshift;
  const int S[3] = {D, H, W};
  for (int w = 0; w < kMaxShifts; ++w)
    for (int i = 0; i < 3; ++i) {
      int v = w < nshift ? shifts[w * 3 + i] % S[i] : 0;
      if (v < 0) v += S[i];
      q.s[w][i] = v;
    }
  (void)who;
  return FZ_OK;
}

// widest vector (in elements, <= maxve) that keeps every p2-run chunk contiguous in x
static int pick_ve(const SwmGeom& q, int maxve) {
  int ve = maxve;
  while (ve > 1) {
    bool ok = (q.p2 % ve == 0);
    for (int w = 0; w < q.nshift && ok; ++w) ok = (q.s[w][2] % ve == 0);
    if (ok) break;
    ve >>= 1;
  }
  return ve;
}

}  // namespace fz

using namespace fz;

extern "C" int fz_swm_fwd(const void* x, void* y, int B, int C, int D, int H, int W, int d, int pd, int ph,
                          int pw, int nshift, const int* shifts, int elem_bytes, int relu, int div,
                          fz_stream_t stream) {
  SwmGeom q;
  int rc = make_geom(q, B, C, D, H, W, d, pd, ph, pw, nshift, shifts, "fz_swm_fwd");
  if (rc != FZ_OK) return rc;
  if (!x || !y) return fail(FZ_E_ARG, "fz_swm_fwd: null pointer");
  if (elem_bytes != 4 && elem_bytes != 2) return fail(FZ_E_UNSUPPORTED, "fz_swm_fwd: elem_bytes must be 4 or 2");
  if (B == 0) return FZ_OK;
  const int64_t nblk = (int64_t)nshift * B * q.h * q.G0 * q.G1;
  if (nblk > 0x7fffffff) return fail(FZ_E_UNSUPPORTED, "fz_swm_fwd: grid too large");
  dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = (hipStream_t)stream;
  const float fdiv = (float)(div > 1 ? div : 1);
  const bool R = relu != 0, Dv = div > 1;
#define FZ_FWD(ES, VB, RR, DD) \
  hipLaunchKernelGGL((swm_fwd_kernel<ES, VB, RR, DD>), grid, block, 0, st, (const char*)x, (char*)y, q, fdiv)
  if (elem_bytes == 2 && (R || Dv)) {
    // 16-bit words with arithmetic asked for: they are bf16 (the mixed-precision activation type)
    const int ve = pick_ve(q, 4);
    const int64_t nb2 = (int64_t)B * q.h * q.G0 * q.G1;
    dim3 grid2((unsigned)nb2);
#define FZ_FWXB(VE, RR, DD) hipLaunchKernelGGL((swm_fwd_x_kernel<VE, RR, DD, bf16>), grid2, block, 0, st, (const bf16*)x, (bf16*)y, q, fdiv)
#define FZ_FWXB4(VE)                                 \
  do {                                               \
    if (R && Dv) FZ_FWXB(VE, true, true);            \
    else if (R) FZ_FWXB(VE, true, false);            \
    else FZ_FWXB(VE, false, true);                   \
  } while (0)
    if (ve == 4) FZ_FWXB4(4);
    else if (ve == 2) FZ_FWXB4(2);
    else FZ_FWXB4(1);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  if (elem_bytes == 4) {
    const int ve = pick_ve(q, 4);
    {
      // x-centric single-read kernel (one workgroup per unshifted pencil)
      const int64_t nb2 = (int64_t)B * q.h * q.G0 * q.G1;
      dim3 grid2((unsigned)nb2);
#define FZ_FWX(VE, RR, DD) hipLaunchKernelGGL((swm_fwd_x_kernel<VE, RR, DD, float>), grid2, block, 0, st, (const float*)x, (float*)y, q, fdiv)
#define FZ_FWX4(VE)                                  \
  do {                                               \
    if (R && Dv) FZ_FWX(VE, true, true);             \
    else if (R) FZ_FWX(VE, true, false);             \
    else if (Dv) FZ_FWX(VE, false, true);            \
    else FZ_FWX(VE, false, false);                   \
  } while (0)
      if (ve == 4) FZ_FWX4(4);
      else if (ve == 2) FZ_FWX4(2);
      else FZ_FWX4(1);
      FZ_LAUNCH_CHECK();
      return FZ_OK;
    }
#define FZ_FWD4(VB)                                  \
  do {                                               \
    if (R && Dv) FZ_FWD(4, VB, true, true);          \
    else if (R) FZ_FWD(4, VB, true, false);          \
    else if (Dv) FZ_FWD(4, VB, false, true);         \
    else FZ_FWD(4, VB, false, false);                \
  } while (0)
    if (ve == 4) FZ_FWD4(16);
    else if (ve == 2) FZ_FWD4(8);
    else FZ_FWD4(4);
  } else {
    const int ve = pick_ve(q, 8);
    if (ve == 8) FZ_FWD(2, 16, false, false);
    else if (ve == 4) FZ_FWD(2, 8, false, false);
    else if (ve == 2) FZ_FWD(2, 4, false, false);
    else FZ_FWD(2, 2, false, false);
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_swm_inv(const void* y, void* x, int B, int C, int D, int H, int W, int d, int pd, int ph,
                          int pw, int nshift, const int* shifts, int average, const void* gate,
                          int act_dtype, fz_stream_t stream) {
  SwmGeom q;
  int rc = make_geom(q, B, C, D, H, W, d, pd, ph, pw, nshift, shifts, "fz_swm_inv");
  if (rc != FZ_OK) return rc;
  if (!x || !y) return fail(FZ_E_ARG, "fz_swm_inv: null pointer");
  if (act_dtype != FZ_STORE_F32 && act_dtype != FZ_STORE_BF16) return fail(FZ_E_ARG, "fz_swm_inv: bad act_dtype");
  if (B == 0) return FZ_OK;
  const int64_t nblk = (int64_t)B * q.h * q.G0 * q.G1;
  if (nblk > 0x7fffffff) return fail(FZ_E_UNSUPPORTED, "fz_swm_inv: grid too large");
  dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int ve = pick_ve(q, 4);
#define FZ_INV_T(VE, AT)                                                                                 \
  do {                                                                                                   \
    if (gate) hipLaunchKernelGGL((swm_inv_kernel<VE, true, AT>), grid, block, 0, st, (const AT*)y,       \
                                 (AT*)x, (const AT*)gate, q, average);                                   \
    else hipLaunchKernelGGL((swm_inv_kernel<VE, false, AT>), grid, block, 0, st, (const AT*)y, (AT*)x,   \
                            (const AT*)nullptr, q, average);                                             \
  } while (0)
#define FZ_INV(VE) do { if (act_dtype == FZ_STORE_BF16) FZ_INV_T(VE, bf16); else FZ_INV_T(VE, float); } while (0)
  if (ve == 4) FZ_INV(4);
  else if (ve == 2) FZ_INV(2);
  else FZ_INV(1);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
