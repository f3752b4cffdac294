// nmf_pcf.hip — FactMixer core on channels-first tensors for ANY patch size with head_dim 8 and at most 256 voxels
// per patch: shifted-window matricize → NMF → inverse matricize in one kernel per window, like nmf_cf.hip (which keeps
// the 8x8x8 hot shape and its line-coalesced exchange).
//
// Replaces SWMatricize.forward → NMF.forward → SWMatricize.inverse_forward (factorizer/factorizer.py:41-50;
// operations.py:417-434; matrix_factorization.py:514-546) where the patch is not 8³ — BASELINE configs[4]
// (160×192×160 is not divisible by 8: patch (5,6,5), 8×150 matrices), the p = 4 / p = 5 configurations of the
// reference's tests.  The modular chain it replaces moved every activation five times per window (matricize write +
// read, NMF result write + read, inverse); here a wave gathers its 8×P matrix straight from t with the window's cyclic
// shift, runs the wave program of nmf_core.h (column n = 64·j + lane, masked beyond P) and scatters u vᵀ into the
// averaged output — window 0 stores 0 + z_0, window w > 0 adds z_w, the last window divides by the number of windows
// (operations.py:426-433 order).  Windows are separate launches on one stream: the accumulation is deterministic.
// Accesses are one element per lane: runs of p2 contiguous voxels per patch row (20 B at p2 = 5) — not the 16-byte
// lanes of the 8³ kernels, but 2 passes over t instead of 5 over t-sized tensors.
#include <cstdlib>

#include "fz_common.h"
#include "nmf_core.h"

namespace fz {

// two matrices per wave for 129..160-voxel patches (probe builds: FZ_PCF_HALF=0 restores the one-matrix kernels)
static inline bool pcf_half_on() { return !(FZ_KNOB("FZ_PCF_HALF").set && FZ_KNOB("FZ_PCF_HALF").val == 0); }


struct PcfGeom {
  int B, C, D, H, W;
  int h;                // heads (C / 8)
  int p0, p1, p2, P;    // patch, voxels per patch
  int G0, G1, G2;       // patch grid
  int s0, s1, s2;       // this window's shift, normalised to [0, S)
  int accumulate;       // add to the existing output (windows > 0)
  int divisor;          // > 1: divide the result by it (last window, forward)
  float gscale_div;     // backward: gY = gather(ga) / gscale_div
};

template <int NPL>
struct PcfWave {
  using F = float;
  int lane, nreal;
  __device__ __forceinline__ int col(int j) const { return j * 64 + lane; }
  __device__ __forceinline__ float sum(float v) const { return wave_sum(v); }
  __device__ __forceinline__ void sum8(float (&v)[8]) const { wave_sum8(v, lane); }
  __device__ __forceinline__ void st_priv(float* base, int idx, float v) const { base[idx * 64 + lane] = v; }
  __device__ __forceinline__ float ld_priv(const float* base, int idx) const { return base[idx * 64 + lane]; }
  __device__ __forceinline__ void st_uni(float* base, int idx, float v) const {
    if (lane == 0) base[idx] = v;
  }
  __device__ __forceinline__ float ld_uni(const float* base, int idx) const { return base[idx]; }
  __device__ __forceinline__ float ld_uni_global(const float* p, int idx) const { return p[idx]; }
  __device__ __forceinline__ float ld_v0(const float* v0, int j, int r, int R) const {
    const int n = col(j);
    return n < nreal ? v0[n * R + r] : 0.f;
  }
  __device__ __forceinline__ float keep_col(int j, float v) const { return col(j) < nreal ? v : 0.f; }
  __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  // factor rows distributed over the eight 8-lane groups (nmf_core.h DistRows): the rank-2 kernels of this file are
  // VALU-bound and a third of their instructions was row arithmetic repeated in all 64 lanes
  static constexpr bool kDistRows = true;
  __device__ __forceinline__ float sum8_dist(const float (&v)[8]) const { return wave_sum8_dist(v, lane); }
  __device__ __forceinline__ float grp_take(float d, int m) const {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 8 * m));
  }
  __device__ __forceinline__ bool grp_below(int n) const { return (lane >> 3) < n; }
  __device__ __forceinline__ void st_grp(float* base, int i0, int stride, float d) const {
    if ((lane & 7) == 0) base[i0 + (lane >> 3) * stride] = d;
  }
  __device__ __forceinline__ float ld_grp_global(const float* p, int i0, int stride) const { return p[i0 + (lane >> 3) * stride]; }
  __device__ __forceinline__ float ld_grp(const float* base, int i0, int stride) const { return base[i0 + (lane >> 3) * stride]; }
  __device__ __forceinline__ float grp_sum(float d) const { return wave_group_sum(d); }
};

// plane base of the wave's (sample, head) and the in-plane element offset of each of the lane's columns
template <int NPL, int LW = 64>   // LW: lanes per matrix (64, or 32 for the two-matrices-per-wave kernel: `lane` is then lane & 31)
__device__ __forceinline__ void pcf_decode(const PcfGeom& q, int64_t mat, int lane, int64_t& base, int64_t& V,
                                           int (&voff)[NPL], bool (&ok)[NPL]) {
  unsigned t = (unsigned)mat;  // the host rejects > 2^31 matrices
  const int g2 = (int)(t % (unsigned)q.G2); t /= (unsigned)q.G2;
  const int g1 = (int)(t % (unsigned)q.G1); t /= (unsigned)q.G1;
  const int g0 = (int)(t % (unsigned)q.G0); t /= (unsigned)q.G0;
  const int hh = (int)(t % (unsigned)q.h);
  const int b = (int)(t / (unsigned)q.h);
  V = (int64_t)q.D * q.H * q.W;
  base = ((int64_t)b * q.C + (int64_t)hh * 8) * V;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int n = j * LW + lane;
    ok[j] = n < q.P;
    const int nn = ok[j] ? n : 0;
    const int c2 = nn % q.p2, c1 = (nn / q.p2) % q.p1, c0 = nn / (q.p2 * q.p1);
    int z0 = g0 * q.p0 + c0 - q.s0; if (z0 < 0) z0 += q.D;
    int z1 = g1 * q.p1 + c1 - q.s1; if (z1 < 0) z1 += q.H;
    int z2 = g2 * q.p2 + c2 - q.s2; if (z2 < 0) z2 += q.W;
    voff[j] = (z0 * q.H + z1) * q.W + z2;   // < 2^31: checked by the host
  }
}

template <int NPL, typename AT>
__device__ __forceinline__ void pcf_load(const AT* __restrict__ t, int64_t base, int64_t V, const int (&voff)[NPL],
                                         const bool (&ok)[NPL], float (&x)[8][NPL]) {
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float v = aget(t + base + m * V + voff[j]);   // masked lanes read voxel 0 of the patch: always valid
      x[m][j] = ok[j] ? v : 0.f;
    }
}

// window w > 0 adds to what the earlier windows left: the old values of FOUR rows are requested together, then added and
// stored (a load per store, in program order, is a chain of 8·NPL dependent round trips per lane — the compiler may not move
// a load above a store it cannot prove disjoint — and cost the shifted windows up to 25 % of their launch)
template <int NPL, typename AT, class Fin>
__device__ __forceinline__ void pcf_rmw_store(AT* __restrict__ dst, int64_t base, int64_t V, const int (&voff)[NPL],
                                              const bool (&ok)[NPL], bool live, bool accumulate, float (&val)[8][NPL], Fin finish) {
#pragma unroll
  for (int m0 = 0; m0 < 8; m0 += 4) {
    float old[4][NPL];
#pragma unroll
    for (int mm = 0; mm < 4; ++mm)
#pragma unroll
      for (int j = 0; j < NPL; ++j)
        old[mm][j] = (accumulate && ok[j] && live) ? aget(dst + base + (m0 + mm) * V + voff[j]) : 0.0f;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm)
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        if (!ok[j] || !live) continue;
        aput(dst + base + (m0 + mm) * V + voff[j], finish(old[mm][j] + val[m0 + mm][j]));
      }
  }
}

template <int NPL, int R, int SOLVER, typename AT>
__global__ __launch_bounds__(256) void nmf_pcf_fwd_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                          const float* __restrict__ v0, AT* __restrict__ out, PcfGeom q,
                                                          int64_t nmat, int T, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t mat = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (mat >= nmat) return;
  PcfWave<NPL> w{lane, q.P};
  int64_t base, V;
  int voff[NPL];
  bool ok[NPL];
  pcf_decode<NPL>(q, mat, lane, base, V, voff, ok);
  float x[8][NPL], u[8][R], v[NPL][R];
  pcf_load<NPL>(t, base, V, voff, ok, x);
  nmf_forward_wave<8, NPL, R, SOLVER>(w, u0, v0, x, u, v, 8, T, eps);
  const float dv = (float)q.divisor;
  const bool pow2 = (__float_as_uint(dv) & 0x007fffffu) == 0u;   // 2 or 4 windows: exact scaling, no IEEE division
  const float inv = 1.0f / dv;
  pcf_rmw_store<NPL>(out, base, V, voff, ok, true, q.accumulate != 0, x,
                     [&](float o) { return q.divisor > 1 ? (pow2 ? o * inv : o / dv) : o; });
}

// ---- TWO matrices per wave (forward) ------------------------------------------------------------------------
// 150 columns (patch (5,6,5), BASELINE configs[4]) on 64 lanes are three passes at 78 % lane use; on the 32 lanes of a wave
// HALF they are five at 94 %, and everything the wave program does per ROW — the eight-row reductions, the U half-step on
// distributed rows, the Gram matrix — serves the two matrices of the wave in one instruction stream.  Policy of nmf_core.h
// with every "uniform" value uniform per half: reductions over 32 lanes (fz_common.h half_*), factor row m of a half in
// its 4-lane group m, row values fetched through the LDS crossbar (ds_bpermute) instead of v_readlane.
template <int NPL>
struct PcfHalf {
  using F = float;
  int lane, nreal;
  __device__ __forceinline__ int l32() const { return lane & 31; }
  __device__ __forceinline__ int grp() const { return (lane >> 2) & 7; }
  __device__ __forceinline__ int col(int j) const { return j * 32 + l32(); }
  __device__ __forceinline__ float sum(float v) const { return half_sum_all(v); }
  __device__ __forceinline__ void sum8(float (&v)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = half_sum_all(v[i]);
  }
  // (history accessors: the caller carves one history per HALF; only the backward uses them)
  __device__ __forceinline__ void st_priv(float* base, int idx, float v) const { base[idx * 32 + l32()] = v; }
  __device__ __forceinline__ float ld_priv(const float* base, int idx) const { return base[idx * 32 + l32()]; }
  __device__ __forceinline__ void st_uni(float* base, int idx, float v) const {
    if (l32() == 0) base[idx] = v;
  }
  __device__ __forceinline__ float ld_uni(const float* base, int idx) const { return base[idx]; }
  __device__ __forceinline__ float ld_uni_global(const float* p, int idx) const { return p[idx]; }
  __device__ __forceinline__ float ld_v0(const float* v0, int j, int r, int R) const {
    const int n = col(j);
    return n < nreal ? v0[n * R + r] : 0.f;
  }
  __device__ __forceinline__ float keep_col(int j, float v) const { return col(j) < nreal ? v : 0.f; }
  __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  static constexpr bool kDistRows = true;
  __device__ __forceinline__ float sum8_dist(const float (&v)[8]) const { return half_sum8_dist(v, lane); }
  __device__ __forceinline__ float grp_take(float d, int m) const { return half_take(d, lane, 4 * m); }
  __device__ __forceinline__ bool grp_below(int n) const { return grp() < n; }
  __device__ __forceinline__ void st_grp(float* base, int i0, int stride, float d) const {
    if ((lane & 3) == 0) base[i0 + grp() * stride] = d;
  }
  __device__ __forceinline__ float ld_grp_global(const float* p, int i0, int stride) const { return p[i0 + grp() * stride]; }
  __device__ __forceinline__ float ld_grp(const float* base, int i0, int stride) const { return base[i0 + grp() * stride]; }
  __device__ __forceinline__ float grp_sum(float d) const {   // total over the eight 4-lane groups of a group-constant value
    d += dpp_take<0x141, 0xf>(d);                 // the other quad of the 8-lane group
    d += dpp_take<0x128, 0xf>(d);                 // row_ror:8
    const fz_u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(d), __float_as_uint(d), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
  }
};

template <int NPL, int R, int SOLVER, typename AT>
__global__ __launch_bounds__(256) void nmf_pcf_fwd2_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                           const float* __restrict__ v0, AT* __restrict__ out, PcfGeom q,
                                                           int64_t nmat, int T, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (2 * pair >= nmat) return;
  int64_t mat = 2 * pair + (lane >> 5);
  const bool live = mat < nmat;          // an odd last matrix leaves the upper half idle: it repeats the lower half's work
  if (!live) mat = nmat - 1;             // (the exchanges of the wave program need all 64 lanes) and stores nothing
  PcfHalf<NPL> w{lane, q.P};
  int64_t base, V;
  int voff[NPL];
  bool ok[NPL];
  pcf_decode<NPL, 32>(q, mat, lane & 31, base, V, voff, ok);
  float x[8][NPL], u[8][R], v[NPL][R];
  pcf_load<NPL>(t, base, V, voff, ok, x);
  nmf_forward_wave<8, NPL, R, SOLVER>(w, u0, v0, x, u, v, 8, T, eps);
  const float dv = (float)q.divisor;
  const bool pow2 = (__float_as_uint(dv) & 0x007fffffu) == 0u;
  const float inv = 1.0f / dv;
  pcf_rmw_store<NPL>(out, base, V, voff, ok, live, q.accumulate != 0, x,
                     [&](float o) { return q.divisor > 1 ? (pow2 ? o * inv : o / dv) : o; });
}

// backward: gY = gather_w(ga) / W ; gt (+)= [t > 0] ∘ scatter_w(gX)
// (the R = 2 multiplicative-update kernels of 3-4 columns per lane and the HALS one of 4 need more than the 256 registers
// of two waves per SIMD: they run one wave per SIMD rather than spill — no scratch, see the kernel body)
template <int NPL, int R, int SOLVER>
constexpr int pcf_bwd_waves_per_simd() { return (R >= 2 && NPL >= 3 && !(NPL == 3 && SOLVER == 1)) ? 1 : 2; }

template <int NPL, int R, int SOLVER, typename AT>
__global__ __launch_bounds__(256, (pcf_bwd_waves_per_simd<NPL, R, SOLVER>())) void nmf_pcf_bwd_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                             const float* __restrict__ v0, const AT* __restrict__ ga,
                                                             AT* __restrict__ gt, PcfGeom q, int64_t nmat, int T, int G,
                                                             float eps, int relu_gate) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_pcf[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t mat = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (mat >= nmat) return;
  PcfWave<NPL> w{lane, q.P};
  Hist<8, NPL, R> h;
  h.carve(fz_lds_pcf + wave * Hist<8, NPL, R>::floats(G), G);
  float x[8][NPL], g[8][NPL];
  {
    // The voxel offsets are decoded twice — here for the loads and again for the stores — instead of living in 2·NPL + 4
    // registers across the whole wave program: the R = 2 kernels sat 3-23 registers over the 256 of two waves per SIMD
    // and spilled.  NO kernel of this library may use scratch: a kernel with a private segment that ran while the
    // weight-gradient kernels of the second stream were in flight returned (rarely, one matrix at a time) slightly
    // different values from run to run on ROCm 7.2 / gfx950 (tools/probes/bf16_replay_trace.py: bitwise reproducible with the
    // spill removed, or with the streams serialised); tests/test_no_spills.py keeps every kernel at zero scratch.
    int64_t base, V;
    int voff[NPL];
    bool ok[NPL];
    pcf_decode<NPL>(q, mat, lane, base, V, voff, ok);
    pcf_load<NPL>(t, base, V, voff, ok, x);
    pcf_load<NPL>(ga, base, V, voff, ok, g);
  }
  if (q.gscale_div != 1.0f) {
    const float dv = q.gscale_div;
    const bool pow2 = (__float_as_uint(dv) & 0x007fffffu) == 0u;
    const float inv = 1.0f / dv;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int j = 0; j < NPL; ++j) g[m][j] = pow2 ? g[m][j] * inv : g[m][j] / dv;
  }
  nmf_backward_wave<8, NPL, R, SOLVER>(w, u0, v0, x, g, h, 8, T, G, eps, nullptr, nullptr);
  int64_t base, V;
  int voff[NPL];
  bool ok[NPL];
  {
    int64_t mat2 = mat;
    asm volatile("" : "+v"(mat2));   // opaque: keeps the compiler from carrying the first decode across the wave program
    pcf_decode<NPL>(q, mat2, lane, base, V, voff, ok);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int j = 0; j < NPL; ++j) g[m][j] = (!relu_gate || x[m][j] > 0.f) ? g[m][j] : 0.f;
  pcf_rmw_store<NPL>(gt, base, V, voff, ok, true, q.accumulate != 0, g, [](float r) { return r; });
}

// backward, two matrices per wave: one history per HALF (lane-private part 32 lanes wide)
static inline int pcf_half_hist_floats(int NPL, int R, int G) {
  return (G + 1) * R * NPL * 32 + (G + 1) * 8 * R + G * (8 * R + R * R);
}
template <int NPL, int R, int SOLVER, typename AT>
__global__ __launch_bounds__(256, 2) void nmf_pcf_bwd2_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                                const float* __restrict__ v0, const AT* __restrict__ ga,
                                                                AT* __restrict__ gt, PcfGeom q, int64_t nmat, int T, int G,
                                                                float eps, int relu_gate) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_pcf2[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (2 * pair >= nmat) return;
  int64_t mat = 2 * pair + (lane >> 5);
  const bool live = mat < nmat;
  if (!live) mat = nmat - 1;
  PcfHalf<NPL> w{lane, q.P};
  Hist<8, NPL, R> h;
  {
    const int per_half = (G + 1) * R * NPL * 32 + (G + 1) * 8 * R + G * (8 * R + R * R);
    float* base = fz_lds_pcf2 + (wave * 2 + (lane >> 5)) * per_half;
    h.vh = base;
    h.uh = h.vh + (G + 1) * R * NPL * 32;
    h.ah = h.uh + (G + 1) * 8 * R;
    h.bh = h.ah + G * 8 * R;
  }
  float x[8][NPL], g[8][NPL];
  {
    int64_t base, V;
    int voff[NPL];
    bool ok[NPL];
    pcf_decode<NPL, 32>(q, mat, lane & 31, base, V, voff, ok);
    pcf_load<NPL>(t, base, V, voff, ok, x);
    pcf_load<NPL>(ga, base, V, voff, ok, g);
  }
  if (q.gscale_div != 1.0f) {
    const float dv = q.gscale_div;
    const bool pow2 = (__float_as_uint(dv) & 0x007fffffu) == 0u;
    const float inv = 1.0f / dv;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int j = 0; j < NPL; ++j) g[m][j] = pow2 ? g[m][j] * inv : g[m][j] / dv;
  }
  nmf_backward_wave<8, NPL, R, SOLVER>(w, u0, v0, x, g, h, 8, T, G, eps, nullptr, nullptr);
  int64_t base, V;
  int voff[NPL];
  bool ok[NPL];
  {
    int64_t mat2 = mat;
    asm volatile("" : "+v"(mat2));   // as in nmf_pcf_bwd_kernel: the decode is repeated, not carried across the wave program
    pcf_decode<NPL, 32>(q, mat2, lane & 31, base, V, voff, ok);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int j = 0; j < NPL; ++j) g[m][j] = (!relu_gate || x[m][j] > 0.f) ? g[m][j] : 0.f;
  pcf_rmw_store<NPL>(gt, base, V, voff, ok, live, q.accumulate != 0, g, [](float r) { return r; });
}

static int pcf_geom(PcfGeom& q, int B, int C, int D, int H, int W, int pd, int ph, int pw, const int* shift, int accumulate,
                    int divisor) {
  if (B < 0 || C < 8 || (C % 8) || pd < 1 || ph < 1 || pw < 1 || D < pd || H < ph || W < pw || (D % pd) || (H % ph) || (W % pw))
    return fail(FZ_E_SHAPE, "fz_nmf_pcf: needs C % 8 == 0 and spatial dims multiples of the patch");
  if ((int64_t)pd * ph * pw > 256) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf: more than 256 voxels per patch");
  if ((int64_t)D * H * W >= ((int64_t)1 << 31)) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf: more than 2^31 voxels per channel plane");
  if (!shift) return fail(FZ_E_ARG, "fz_nmf_pcf: shift is null");
  q.B = B; q.C = C; q.D = D; q.H = H; q.W = W; q.h = C / 8;
  q.p0 = pd; q.p1 = ph; q.p2 = pw; q.P = pd * ph * pw;
  q.G0 = D / pd; q.G1 = H / ph; q.G2 = W / pw;
  const int S[3] = {D, H, W};
  int s[3];
  for (int i = 0; i < 3; ++i) { s[i] = shift[i] % S[i]; if (s[i] < 0) s[i] += S[i]; }
  q.s0 = s[0]; q.s1 = s[1]; q.s2 = s[2];
  q.accumulate = accumulate; q.divisor = divisor; q.gscale_div = 1.0f;
  return FZ_OK;
}

static int pcf_npl(int P) { return P <= 64 ? 1 : (P <= 128 ? 2 : (P <= 192 ? 3 : 4)); }

static int pcf_per_wave(int P, int R, int G) {
  const int npl = pcf_npl(P);
  return ((G + 1) * R * npl * 64 + (G + 1) * 8 * R + G * (8 * R + R * R)) * (int)sizeof(float);
}

template <typename AT>
static int pcf_fwd_launch(const AT* t, const float* u0, const float* v0, AT* out, const PcfGeom& q, int R, int T, int solver,
                          float eps, hipStream_t st) {
  const int64_t nmat = (int64_t)q.B * q.h * q.G0 * q.G1 * q.G2;
  if (nmat >= (int64_t)1 << 31) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf: more than 2^31 matrices");
  dim3 grid((unsigned)((nmat + 3) / 4)), block(256);
  // 129..160 voxels per patch: two matrices per wave, five columns per lane (FZ_PCF_HALF=0: the one-matrix form, diagnostics)
  const bool half_on = pcf_half_on();
  if (half_on && q.P > 128 && q.P <= 160) {
    dim3 grid2((unsigned)(((nmat + 1) / 2 + 3) / 4));
#define FZ_PCF_FWD2(RR, SS) hipLaunchKernelGGL((nmf_pcf_fwd2_kernel<5, RR, SS, AT>), grid2, block, 0, st, t, u0, v0, out, q, nmat, T, eps)
    if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_PCF_FWD2(1, SOLVER_MU); else FZ_PCF_FWD2(1, SOLVER_HALS); }
    else { if (solver == FZ_SOLVER_MU) FZ_PCF_FWD2(2, SOLVER_MU); else FZ_PCF_FWD2(2, SOLVER_HALS); }
#undef FZ_PCF_FWD2
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
#define FZ_PCF_FWD(NN, RR, SS) hipLaunchKernelGGL((nmf_pcf_fwd_kernel<NN, RR, SS, AT>), grid, block, 0, st, t, u0, v0, out, q, nmat, T, eps)
#define FZ_PCF_FWD_RS(NN)                                                                                    \
  do {                                                                                                       \
    if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_PCF_FWD(NN, 1, SOLVER_MU); else FZ_PCF_FWD(NN, 1, SOLVER_HALS); } \
    else { if (solver == FZ_SOLVER_MU) FZ_PCF_FWD(NN, 2, SOLVER_MU); else FZ_PCF_FWD(NN, 2, SOLVER_HALS); }  \
  } while (0)
  switch (pcf_npl(q.P)) {
    case 1: FZ_PCF_FWD_RS(1); break;
    case 2: FZ_PCF_FWD_RS(2); break;
    case 3: FZ_PCF_FWD_RS(3); break;
    default: FZ_PCF_FWD_RS(4); break;
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

template <typename AT>
static int pcf_bwd_launch(const AT* t, const float* u0, const float* v0, const AT* ga, AT* gt, const PcfGeom& q, int R, int T,
                          int G, int solver, float eps, int relu_gate, hipStream_t st) {
  const int64_t nmat = (int64_t)q.B * q.h * q.G0 * q.G1 * q.G2;
  if (nmat >= (int64_t)1 << 31) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf: more than 2^31 matrices");
  const bool half_on = pcf_half_on();
  // (windows w > 0 — read-modify-write of the running gradient — keep the one-matrix form: at the two-matrix form's 5 waves per
  // CU the extra scattered read costs more than the form saves, 9.5 against 8.9 ms at the cfg-5 stage-0 launch; window 0: 6.3 / 7.1)
  if (half_on && !q.accumulate && q.P > 128 && q.P <= 160 && 2 * pcf_half_hist_floats(5, R, G) * (int)sizeof(float) <= 160 * 1024) {
    const int per_wave2 = 2 * pcf_half_hist_floats(5, R, G) * (int)sizeof(float);
    const int wpb2 = fz_hist_waves_per_block(per_wave2);
    const int lds2 = per_wave2 * wpb2;
    const int64_t npair = (nmat + 1) / 2;
    dim3 grid2((unsigned)((npair + wpb2 - 1) / wpb2)), block2(64 * wpb2);
#define FZ_PCF_BWD2(RR, SS)                                                                                   \
  do {                                                                                                        \
    auto kern = nmf_pcf_bwd2_kernel<5, RR, SS, AT>;                                                            \
    if (lds2 > 65536)                                                                                         \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds2)); \
    hipLaunchKernelGGL(kern, grid2, block2, lds2, st, t, u0, v0, ga, gt, q, nmat, T, G, eps, relu_gate);      \
  } while (0)
    if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_PCF_BWD2(1, SOLVER_MU); else FZ_PCF_BWD2(1, SOLVER_HALS); }
    else { if (solver == FZ_SOLVER_MU) FZ_PCF_BWD2(2, SOLVER_MU); else FZ_PCF_BWD2(2, SOLVER_HALS); }
#undef FZ_PCF_BWD2
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  const int per_wave = pcf_per_wave(q.P, R, G);
  if (per_wave > 160 * 1024) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf_bwd: history exceeds LDS");
  int wpb = fz_hist_waves_per_block(per_wave);
  const int lds = per_wave * wpb;
  dim3 grid((unsigned)((nmat + wpb - 1) / wpb)), block(64 * wpb);
#define FZ_PCF_BWD(NN, RR, SS)                                                                                \
  do {                                                                                                        \
    auto kern = nmf_pcf_bwd_kernel<NN, RR, SS, AT>;                                                           \
    if (lds > 65536)                                                                                          \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(kern, grid, block, lds, st, t, u0, v0, ga, gt, q, nmat, T, G, eps, relu_gate);         \
  } while (0)
#define FZ_PCF_BWD_RS(NN)                                                                                     \
  do {                                                                                                        \
    if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_PCF_BWD(NN, 1, SOLVER_MU); else FZ_PCF_BWD(NN, 1, SOLVER_HALS); } \
    else { if (solver == FZ_SOLVER_MU) FZ_PCF_BWD(NN, 2, SOLVER_MU); else FZ_PCF_BWD(NN, 2, SOLVER_HALS); }   \
  } while (0)
  switch (pcf_npl(q.P)) {
    case 1: FZ_PCF_BWD_RS(1); break;
    case 2: FZ_PCF_BWD_RS(2); break;
    case 3: FZ_PCF_BWD_RS(3); break;
    default: FZ_PCF_BWD_RS(4); break;
  }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// dst += src, elementwise on activations (16 bytes per lane and step; the sum is formed in fp32 and stored once).  Used where a
// window w > 0 of the generic-patch backward is cheaper into its own buffer than as a scattered read-modify-write (below).
template <typename AT>
__global__ __launch_bounds__(256) void act_add_kernel(AT* __restrict__ dst, const AT* __restrict__ src, int64_t n) {
  constexpr int VEC = 16 / (int)sizeof(AT);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * VEC;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride) {
    if (i + VEC <= n) {
      float a[VEC], b[VEC];
      aload<VEC>(dst + i, a);
      aload<VEC>(src + i, b);
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] += b[e];
      astore<VEC>(dst + i, a);
    } else {
      for (int64_t e = i; e < n; ++e) aput(dst + e, aget(dst + e) + aget(src + e));
    }
  }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_nmf_pcf_supported(int C, int D, int H, int W, int d, int pd, int ph, int pw, int R, int T, int Tgrad) {
  if (d != 8 || (C % 8) || pd < 1 || ph < 1 || pw < 1 || (D % pd) || (H % ph) || (W % pw)) return 0;
  if ((int64_t)pd * ph * pw > 256 || (int64_t)D * H * W >= ((int64_t)1 << 31)) return 0;
  if (R < 1 || R > 2 || T < 0) return 0;
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  return pcf_per_wave(pd * ph * pw, R, G) <= 160 * 1024 ? 1 : 0;
}

extern "C" int fz_nmf_pcf_fwd(const void* t, const float* u0, const float* v0, void* out, int B, int C, int D, int H, int W,
                              int pd, int ph, int pw, const int* shift, int accumulate, int divisor, int R, int T, int solver,
                              float eps, int act_dtype, fz_stream_t stream) {
  PcfGeom q;
  const int rc = pcf_geom(q, B, C, D, H, W, pd, ph, pw, shift, accumulate, divisor);
  if (rc != FZ_OK) return rc;
  if (!t || !u0 || !v0 || !out) return fail(FZ_E_ARG, "fz_nmf_pcf_fwd: null pointer");
  if (R < 1 || R > 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf_fwd: rank 1..2");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_pcf_fwd: bad solver");
  if (B == 0) return FZ_OK;
  hipStream_t st = (hipStream_t)stream;
  if (act_dtype == FZ_STORE_F32) return pcf_fwd_launch<float>((const float*)t, u0, v0, (float*)out, q, R, T, solver, eps, st);
  if (act_dtype == FZ_STORE_BF16) return pcf_fwd_launch<bf16>((const bf16*)t, u0, v0, (bf16*)out, q, R, T, solver, eps, st);
  return fail(FZ_E_ARG, "fz_nmf_pcf_fwd: bad act_dtype");
}

extern "C" int fz_nmf_pcf_bwd(const void* t, const float* u0, const float* v0, const void* ga, void* gt, int B, int C, int D,
                              int H, int W, int pd, int ph, int pw, const int* shift, int accumulate, int nshift, int relu_gate,
                              int R, int T, int Tgrad, int solver, float eps, int act_dtype, fz_stream_t stream) {
  PcfGeom q;
  const int rc = pcf_geom(q, B, C, D, H, W, pd, ph, pw, shift, accumulate, 1);
  if (rc != FZ_OK) return rc;
  if (!t || !u0 || !v0 || !ga || !gt) return fail(FZ_E_ARG, "fz_nmf_pcf_bwd: null pointer");
  if (R < 1 || R > 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_pcf_bwd: rank 1..2");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_pcf_bwd: bad solver");
  if (B == 0) return FZ_OK;
  q.gscale_div = (float)(nshift > 1 ? nshift : 1);
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  hipStream_t st = (hipStream_t)stream;
  if (act_dtype == FZ_STORE_F32)
    return pcf_bwd_launch<float>((const float*)t, u0, v0, (const float*)ga, (float*)gt, q, R, T, G, solver, eps, relu_gate, st);
  if (act_dtype == FZ_STORE_BF16)
    return pcf_bwd_launch<bf16>((const bf16*)t, u0, v0, (const bf16*)ga, (bf16*)gt, q, R, T, G, solver, eps, relu_gate, st);
  return fail(FZ_E_ARG, "fz_nmf_pcf_bwd: bad act_dtype");
}

/* 1 when a window w > 0 of the backward is faster written to its OWN buffer (accumulate = 0) and added with fz_act_add than
 * as the read-modify-write of accumulate = 1: the two-matrices-per-wave shapes (129..160 voxels per patch) in bf16 storage
 * (cfg-5 step, B = 4: 100.3 -> 96.2 ms; fp32 storage, where the add moves twice the bytes: 116.9 -> 120.9, so not there). */
extern "C" int fz_nmf_pcf_bwd_prefers_separate(int pd, int ph, int pw, int act_dtype) {
  const bool half_on = pcf_half_on();
  const bool sep_on = !(FZ_KNOB("FZ_PCF_SEPARATE").set && FZ_KNOB("FZ_PCF_SEPARATE").val == 0);   // probe builds: 0 = read-modify-write
  const int64_t P = (int64_t)pd * ph * pw;
  return (half_on && sep_on && act_dtype == FZ_STORE_BF16 && P > 128 && P <= 160) ? 1 : 0;
}

extern "C" int fz_act_add(void* dst, const void* src, int64_t n, int act_dtype, fz_stream_t stream) {
  if (!dst || !src) return fail(FZ_E_ARG, "fz_act_add: null pointer");
  if (n < 0) return fail(FZ_E_ARG, "fz_act_add: negative length");
  if (n == 0) return FZ_OK;
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) return fail(FZ_E_ARG, "fz_act_add: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int vec = act_dtype == FZ_STORE_F32 ? 4 : 8;
  int64_t nb = (n / vec + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 256 * 8) nb = 256 * 8;
  if (act_dtype == FZ_STORE_F32) hipLaunchKernelGGL(act_add_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (float*)dst, (const float*)src, n);
  else if (act_dtype == FZ_STORE_BF16) hipLaunchKernelGGL(act_add_kernel<bf16>, dim3((unsigned)nb), dim3(256), 0, st, (bf16*)dst, (const bf16*)src, n);
  else return fail(FZ_E_ARG, "fz_act_add: bad act_dtype");
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
