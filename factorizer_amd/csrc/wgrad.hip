// wgrad.hip — weight gradients of the channels-first GEMM family on the fp32 matrix cores.
//
//   GW[m, k] = Σ_{b, n}  P[b, m, n] · Q(In)[b, k, n]          (reduction over voxels n)
//
// where P is the output-side gradient (plain rows) and Q the layer's input operand, produced
// by the same loaders as the forward (plain / space-to-depth / 3x3x3 taps) with the same
// prologues (LayerNorm with saved statistics, GELU, window average, two-source concat).
// This is what autograd's conv1d/conv3d weight-gradient kernels compute for
// layers/linear.py:53-58, unet.py:53,123,231 — MIOpen's naive conv wrw kernel took 76 % of a
// training step on MI355X (profiles/r01_p1), hence a dedicated kernel.
//
// Both MFMA operands need "lane = channel, k-step = voxel pair", the transpose of the
// coalesced global layout, so every wave stages 32-voxel tiles of P and Q in its own LDS
// region (row stride 33 → conflict-free column reads), with no workgroup barriers.  Voxel
// ranges are split over waves and workgroups; partial sums go to a workspace and a second
// kernel reduces them in a fixed order (bitwise reproducible, no float atomics).
#include <type_traits>

#include "fz_common.h"
#include "finish.h"

namespace fz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 -> three bf16 levels (hi, mid, lo), round-to-nearest at each: x = hi + mid + lo exactly (8 + 8 + 8 significand
// bits).  The products keep the six terms hi·hi + (hi·mid + mid·hi) + (hi·lo + mid·mid + lo·hi): what is dropped is
// <= 2^-23 |pq| — an fp32 product in 6/16 of the fp32-MFMA time (gemm_bx.hip; tools/probes/bx6_accuracy.hip).
// v_cvt_pk_bf16_f32 converts two floats per instruction: 5.5 VALU instructions per element.
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 v = {x[e], x[e + 1]};
    const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(h2, f32x2);
    const bf16x2 m2 = __builtin_convertvector(r, bf16x2);
    const f32x2 r2 = r - __builtin_convertvector(m2, f32x2);
    const bf16x2 l2 = __builtin_convertvector(r2, bf16x2);
    hi[e] = h2[0]; hi[e + 1] = h2[1];
    mid[e] = m2[0]; mid[e + 1] = m2[1];
    lo[e] = l2[0]; lo[e + 1] = l2[1];
  }
}

enum { QL_PLAIN = 0, QL_S2D = 1, QL_K3 = 2 };

// AT = storage type of the activation tensors p, pmul, q (float or bf16); everything else is fp32
template <typename AT>
struct WgradArgsT {
  const AT* p;         // (B, M, Np) output-side gradient rows
  int M;
  const AT* pmul;      // optional: P *= act'(pmul) (same shape as p)
  int pmul_kind;
  const AT* q[4];      // input sources
  int nsrc, src_mode, c0;
  int Cin;             // input channels
  int K;               // Q rows: Cin (plain), 8*Cin (s2d), 27*Cin (k3)
  int64_t Vq;          // voxels per sample of the input tensor
  int D, H, W;         // input spatial dims (s2d: fine grid; k3: grid)
  int64_t N;           // reduction columns per sample (= Np): Vq (plain/k3) or Vq/8 (s2d)
  int Ho, Wo;          // coarse grid (s2d)
  const float* stats;  // (B, 2, Vq) mean/rstd: Q = (x - mean) * rstd   (LN prologue), or null
  int qact;            // activation on Q
  float* part;         // workspace: [nchunk][M][K] partial sums
  float* part_bias;    // workspace: [nchunk][M] partial row sums of P (or null)
  int B;
  int tiles_per_chunk; // 32-column tiles per (workgroup, wave) unit
};

constexpr int kTile = 32;       // voxels per staged tile
constexpr int kStride = 33;     // LDS row stride (floats)

__device__ __forceinline__ float gelu_w(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_w(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// MBP x MBQ blocks of 32x32 outputs per workgroup; 4 waves split the voxel tiles.
// BF: 0 = fp32 MFMAs; 6 = six-term split-bf16 products (fp32 accuracy, the default: see wgrad_fast_kernel); 1 = plain bf16 MFMAs
// (operands rounded to bf16, fp32 accumulation) — the mixed-precision mode, where P and Q are bf16 in HBM
// anyway and only a prologue (LayerNorm / GELU) result takes one extra rounding.
template <int MBP, int MBQ, int QL, int BF = 0, typename AT = float>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgsT<AT> a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PR = 32 * MBP, QR = 32 * MBQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* Pt = lds + wave * (PR + QR) * kStride;
  float* Qt = Pt + PR * kStride;
  const int m0 = blockIdx.y * PR;
  const int k0 = blockIdx.z * QR;
  const int64_t tiles_per_sample = (a.N + kTile - 1) / kTile;
  const int64_t total_tiles = tiles_per_sample * a.B;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t t_begin = unit * a.tiles_per_chunk;
  const int64_t t_end = min(t_begin + a.tiles_per_chunk, total_tiles);

  f32x16 acc[MBP][MBQ];
#pragma unroll
  for (int i = 0; i < MBP; ++i)
#pragma unroll
    for (int jq = 0; jq < MBQ; ++jq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jq][r] = 0.f;
  float psum[MBP];
#pragma unroll
  for (int i = 0; i < MBP; ++i) psum[i] = 0.f;

  const int c = lane & 31, h = lane >> 5;

  // Staging is BRANCH-FREE (clamped addresses + selects) so that the global loads of a tile are
  // issued back to back; for the plain loaders the next tile's loads are issued BEFORE the MFMA
  // loop of the current tile and land in registers while the matrix cores run.
  constexpr int NP = PR * 8 / 64;  // 16-byte chunks per lane for the P tile
  constexpr int NQ = QR * 8 / 64;  // ... for a plain Q tile
  const int cq = lane & 7;          // column chunk of this lane (same for every chunk it stages)
  const int r0 = lane >> 3;         // first row of this lane; chunk i covers row r0 + 8*i
  float4 pv[NP], qv[NQ], mu4, rs4;

  auto issue_loads = [&](int64_t t) {
    const int b = (int)(t / tiles_per_sample);
    const int64_t n = (t % tiles_per_sample) * kTile + cq * 4;
    const int64_t nc = n < a.N ? n : 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int m = m0 + r0 + 8 * i;
      const int mc = m < a.M ? m : a.M - 1;
      pv[i] = ld4(a.p + ((int64_t)b * a.M + mc) * a.N + nc);
    }
    if (QL == QL_PLAIN) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int k = k0 + r0 + 8 * i;
        const int kc = k < a.K ? k : a.K - 1;
        const bool first = kc < a.c0;
        const AT* base = first ? a.q[0] : a.q[1];
        const int cs = first ? a.c0 : a.Cin - a.c0;
        const int ci = first ? kc : kc - a.c0;
        qv[i] = ld4(base + ((int64_t)b * cs + ci) * a.Vq + nc);
      }
      // (without statistics: a harmless read of the first floats of the partial workspace, never used)
      const float* st = a.stats ? a.stats + (int64_t)b * 2 * a.Vq + nc : a.part;
      mu4 = *reinterpret_cast<const float4*>(st);
      rs4 = *reinterpret_cast<const float4*>(st + (a.stats ? a.Vq : 0));
    }
  };

  auto commit_tiles = [&](int64_t t) {
    const int64_t n = (t % tiles_per_sample) * kTile + cq * 4;
    const bool nok = n < a.N;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = r0 + 8 * i;
      const bool ok = nok && (m0 + r) < a.M;
      float* d = Pt + r * kStride + cq * 4;
      d[0] = ok ? pv[i].x : 0.f; d[1] = ok ? pv[i].y : 0.f; d[2] = ok ? pv[i].z : 0.f; d[3] = ok ? pv[i].w : 0.f;
    }
    if (QL == QL_PLAIN) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int r = r0 + 8 * i;
        const bool ok = nok && (k0 + r) < a.K;
        float v[4] = {qv[i].x, qv[i].y, qv[i].z, qv[i].w};
        if (a.stats) {
          v[0] = (v[0] - mu4.x) * rs4.x; v[1] = (v[1] - mu4.y) * rs4.y;
          v[2] = (v[2] - mu4.z) * rs4.z; v[3] = (v[3] - mu4.w) * rs4.w;
        }
        if (a.qact == 2) { v[0] = gelu_w(v[0]); v[1] = gelu_w(v[1]); v[2] = gelu_w(v[2]); v[3] = gelu_w(v[3]); }
        else if (a.qact == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        float* d = Qt + r * kStride + cq * 4;
        d[0] = ok ? v[0] : 0.f; d[1] = ok ? v[1] : 0.f; d[2] = ok ? v[2] : 0.f; d[3] = ok ? v[3] : 0.f;
      }
    }
  };

  if (t_begin < t_end) issue_loads(t_begin);
  for (int64_t t = t_begin; t < t_end; ++t) {
    const int b = (int)(t / tiles_per_sample);
    const int64_t n0 = (t % tiles_per_sample) * kTile;
    // ---- P (and plain Q) tiles: registers -> LDS ----
    if (a.pmul) {
      // ReLU / GELU' gate of the output-side gradient (modular path only)
      const int64_t n = n0 + cq * 4;
      const int64_t nc = n < a.N ? n : 0;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int m = m0 + r0 + 8 * i;
        const int mc = m < a.M ? m : a.M - 1;
        const float4 e = ld4(a.pmul + ((int64_t)b * a.M + mc) * a.N + nc);
        if (a.pmul_kind == 2) { pv[i].x *= gelu_grad_w(e.x); pv[i].y *= gelu_grad_w(e.y); pv[i].z *= gelu_grad_w(e.z); pv[i].w *= gelu_grad_w(e.w); }
        else { pv[i].x = e.x > 0.f ? pv[i].x : 0.f; pv[i].y = e.y > 0.f ? pv[i].y : 0.f; pv[i].z = e.z > 0.f ? pv[i].z : 0.f; pv[i].w = e.w > 0.f ? pv[i].w : 0.f; }
      }
    }
    commit_tiles(t);
    if (QL == QL_S2D) {
      // rows k = (ci, td, th, tw); one 16-byte load gives (tw0,tw1) of two coarse voxels
#pragma unroll
      for (int i = 0; i < (QR / 2) * 16 / 64; ++i) {
        const int idx = lane + 64 * i;
        const int rp = idx >> 4, cp = idx & 15;      // row pair (tw = 0/1), coarse column pair
        const int k = k0 + 2 * rp;                     // tw = 0 row
        const int64_t n = n0 + 2 * cp;
        const bool ok = k < a.K && n < a.N;
        const int kc = k < a.K ? k : 0;
        const int64_t nn = n < a.N ? n : 0;
        const int ci = kc >> 3, td = (kc >> 2) & 1, th = (kc >> 1) & 1;
        const int wo = (int)(nn % a.Wo);
        const int64_t t2 = nn / a.Wo;
        const int ho = (int)(t2 % a.Ho);
        const int dz = (int)(t2 / a.Ho);
        const int64_t fo = ((int64_t)(2 * dz + td) * a.H + (2 * ho + th)) * a.W + 2 * wo;
        const float4 v = ld4(a.q[0] + ((int64_t)b * a.Cin + ci) * a.Vq + fo);
        float* d0 = Qt + (2 * rp) * kStride + 2 * cp;
        float* d1 = d0 + kStride;
        d0[0] = ok ? v.x : 0.f; d1[0] = ok ? v.y : 0.f; d0[1] = ok ? v.z : 0.f; d1[1] = ok ? v.w : 0.f;
      }
    } else if (QL == QL_K3) {
      // QL_K3: rows k = ci*27 + (kd*9 + kh*3 + kw); zero padding 1.  Lane = (column, row parity):
      // the voxel coordinates are decoded once per tile, the tap once per (uniform) row.
      const int col = lane & 31;
      const int64_t n = n0 + col;
      const bool nok = n < a.N;
      const int w = (int)((nok ? n : 0) % a.W);
      const int64_t t2 = (nok ? n : 0) / a.W;
      const int hh = (int)(t2 % a.H);
      const int dz = (int)(t2 / a.H);
#pragma unroll 8
      for (int r = lane >> 5; r < QR; r += 2) {
        const int k = k0 + r;
        const int kc = k < a.K ? k : 0;
        const int ci = kc / 27, tap = kc % 27;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const int zd = dz + kd - 1, zh = hh + kh - 1, zw = w + kw - 1;
        const bool ok = nok && k < a.K && zd >= 0 && zd < a.D && zh >= 0 && zh < a.H && zw >= 0 && zw < a.W;
        const int zdc = min(max(zd, 0), a.D - 1), zhc = min(max(zh, 0), a.H - 1), zwc = min(max(zw, 0), a.W - 1);
        const float v = aget(a.q[0] + ((int64_t)b * a.Cin + ci) * a.Vq + ((int64_t)zdc * a.H + zhc) * a.W + zwc);
        Qt[r * kStride + col] = ok ? v : 0.f;
      }
    }
    if (t + 1 < t_end) issue_loads(t + 1);  // in flight during the MFMA loop below
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (BF != 0) {
      // ---- 2 K-steps of 16 voxels: lane half h, element e <-> voxel 16·t + 8·h + e ----
#pragma unroll
      for (int t16 = 0; t16 < 2; ++t16) {
        bf16x8 ph[MBP], pm[MBP], pl[MBP], qh[MBQ], qm[MBQ], ql[MBQ];
#pragma unroll
        for (int i = 0; i < MBP; ++i) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { x8[e] = Pt[(i * 32 + c) * kStride + 16 * t16 + 8 * h + e]; psum[i] += x8[e]; }
          split_bf16x8(x8, ph[i], pm[i], pl[i]);
        }
#pragma unroll
        for (int jq = 0; jq < MBQ; ++jq) {
          float x8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x8[e] = Qt[(jq * 32 + c) * kStride + 16 * t16 + 8 * h + e];
          split_bf16x8(x8, qh[jq], qm[jq], ql[jq]);
        }
#pragma unroll
        for (int i = 0; i < MBP; ++i)
#pragma unroll
          for (int jq = 0; jq < MBQ; ++jq) {
            if (BF == 6) {   // smallest terms first
              acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl[i], qh[jq], acc[i][jq], 0, 0, 0);
              acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[i], qm[jq], acc[i][jq], 0, 0, 0);
              acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[i], ql[jq], acc[i][jq], 0, 0, 0);
              acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[i], qh[jq], acc[i][jq], 0, 0, 0);
              acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[i], qm[jq], acc[i][jq], 0, 0, 0);
            }
            acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[i], qh[jq], acc[i][jq], 0, 0, 0);
          }
      }
    } else
    // ---- 16 K-steps of 2 voxels ----
#pragma unroll 4
    for (int s = 0; s < kTile / 2; ++s) {
      float av[MBP], bv[MBQ];
#pragma unroll
      for (int i = 0; i < MBP; ++i) {
        av[i] = Pt[(i * 32 + c) * kStride + 2 * s + h];
        psum[i] += av[i];
      }
#pragma unroll
      for (int jq = 0; jq < MBQ; ++jq) bv[jq] = Qt[(jq * 32 + c) * kStride + 2 * s + h];
#pragma unroll
      for (int i = 0; i < MBP; ++i)
#pragma unroll
        for (int jq = 0; jq < MBQ; ++jq)
          acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jq], acc[i][jq], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }

  // ---- reduce the 4 waves through LDS, then write this workgroup's partial block ----
  __syncthreads();
  float* red = lds;  // [4][PR*QR] would not fit for big blocks: reduce block by block
#pragma unroll
  for (int i = 0; i < MBP; ++i) {
#pragma unroll
    for (int jq = 0; jq < MBQ; ++jq) {
      // wave w writes its 32x32 block (1024 floats) at red + w*1024 ; layout [row][col]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[wave * 1024 + row * 32 + c] = acc[i][jq][r];
      }
      __syncthreads();
      for (int e = threadIdx.x; e < 1024; e += 256) {
        const float s = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        const int row = e >> 5, col = e & 31;
        const int m = m0 + i * 32 + row, k = k0 + jq * 32 + col;
        if (m < a.M && k < a.K) a.part[((int64_t)blockIdx.x * a.M + m) * a.K + k] = s;
      }
      __syncthreads();
    }
  }
  if (a.part_bias != nullptr && blockIdx.z == 0) {
#pragma unroll
    for (int i = 0; i < MBP; ++i) {
      float s = psum[i] + __shfl_xor(psum[i], 32, 64);
      if (h == 0) red[wave * 32 + c] = s;
      __syncthreads();
      if (threadIdx.x < 32) {
        const float tot = (red[threadIdx.x] + red[32 + threadIdx.x]) + (red[64 + threadIdx.x] + red[96 + threadIdx.x]);
        const int m = m0 + i * 32 + threadIdx.x;
        if (m < a.M) a.part_bias[(int64_t)blockIdx.x * a.M + m] = tot;
      }
      __syncthreads();
    }
  }
}

// ---- fast path: whole 32-row blocks, N % 32 == 0, plain loader, no P gate ------------------------
// Same decomposition as wgrad_kernel, with the per-tile overheads taken out:
//   * K-step s pairs voxel s (lane half 0) with voxel 16+s (half 1) instead of (2s, 2s+1) — the
//     order of a reduction is free — so a lane needs 16 CONTIGUOUS voxels of its channel: four
//     ds_read_b128 per 32-row block and tile replace sixteen ds_read_b32, all issued before the
//     MFMA loop, which then runs on register operands only;
//   * LDS rows are 36 floats (16-byte aligned): one ds_write_b128 per staged chunk instead of four
//     scalar writes;
//   * the Q prologue (LayerNorm statistics / GELU / ReLU) is a template parameter: no branches
//     between the chunks;
//   * addresses = uniform row base (scalar registers) + one 32-bit lane offset.
enum { QP_NONE = 0, QP_STATS = 1, QP_GELU = 2, QP_RELU = 3 };
constexpr int kStrideF = 36;

// BF = 6: the products run on the bf16 matrix pipe as the six leading terms of a three-level bf16 split of both operands
// (split_bf16x8) — 16 voxels per MFMA at twice the issue rate of the fp32 MFMA's 2, i.e. 6 instead of 16 MFMA slots per
// 16 voxels, with every kept term exact in the fp32 accumulator and <= 2^-23 |pq| dropped: fp32 accuracy.  Round 2
// carried a three-term variant (relative error 3·2^-18 per product) as an opt-in; the six-term form has no such caveat and
// is the default (fz_gemm_bx_enable(0) returns to v_mfma_f32_32x32x2_f32).
// The k index of an MFMA operand element is (lane half, element): ANY assignment of voxels to it is
// valid as long as both operands use the same one — half h, element e <-> voxel 16·t + 8·h + e of K-step t.
// (the body lives in wgrad_fast_body.inc so that wgrad_fast_group_kernel can run several problems in one grid)
template <int MBP, int MBQ, int QPRO, int BF, typename AT>
__device__ __forceinline__ void wgrad_fast_body(const WgradArgsT<AT>& a, float* lds, const int bx, const int by, const int bz) {
#define WG_BX bx
#define WG_BY by
#define WG_BZ bz
#include "wgrad_fast_body.inc"
#undef WG_BX
#undef WG_BY
#undef WG_BZ
}

template <int MBP, int MBQ, int QPRO, int BF = 0, typename AT = float>
__global__ __launch_bounds__(256, 2) void wgrad_fast_kernel(WgradArgsT<AT> a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#define WG_BX blockIdx.x
#define WG_BY blockIdx.y
#define WG_BZ blockIdx.z
#include "wgrad_fast_body.inc"
#undef WG_BX
#undef WG_BY
#undef WG_BZ
}

// Several weight-gradient problems of ONE layer block in one grid: the four dense layers of a FactorizerBlock at
// C >= 64 (fc2, fc1 behind LayerNorm, out_proj, in_proj behind LayerNorm) each fill one workgroup per CU for 20-80 us;
// launched back to back they leave the second resident workgroup slot of every CU empty and pay four launch
// boundaries.  A workgroup finds its problem from a prefix table (wave-uniform) and runs the unchanged body on that
// problem's descriptor; partial blocks, the order of every sum and the finish launches are those of the single
// launches — bitwise the same gradients.
constexpr int kWgGroupMax = 4;
template <typename AT>
struct WgradGroupT {
  WgradArgsT<AT> a[kWgGroupMax];
  int start[kWgGroupMax + 1];   // first workgroup of each problem
  int gx[kWgGroupMax], gy[kWgGroupMax];
  int qpro[kWgGroupMax];
  int n;
};

template <int MBP, int MBQ, int BF, typename AT>
__global__ __launch_bounds__(256, 2) void wgrad_fast_group_kernel(WgradGroupT<AT> g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int i = 0;
#pragma unroll
  for (int t = 1; t < kWgGroupMax; ++t)
    if (t < g.n && (int)blockIdx.x >= g.start[t]) i = t;
  int local = (int)blockIdx.x - g.start[i];
  const int bx = local % g.gx[i];
  local /= g.gx[i];
  const int by = local % g.gy[i], bz = local / g.gy[i];
  const WgradArgsT<AT>& a = g.a[i];
  switch (g.qpro[i]) {
    case QP_STATS: wgrad_fast_body<MBP, MBQ, QP_STATS, BF, AT>(a, lds, bx, by, bz); break;
    case QP_GELU: wgrad_fast_body<MBP, MBQ, QP_GELU, BF, AT>(a, lds, bx, by, bz); break;
    case QP_RELU: wgrad_fast_body<MBP, MBQ, QP_RELU, BF, AT>(a, lds, bx, by, bz); break;
    default: wgrad_fast_body<MBP, MBQ, QP_NONE, BF, AT>(a, lds, bx, by, bz); break;
  }
}

// (the fixed-order reductions of the partial blocks — chunk sums, one weight gradient with its bias sums and LayerNorm
// fold — are jobs of the finish kernel: finish.h, FK_CHUNK / FK_WGRAD)

// LayerNorm affine folded into the weight gradient: gw[m][k] = γ_k · acc[m][k] + β_k · gb[m]
__global__ __launch_bounds__(256) void ln_fold_kernel(float* __restrict__ gw, const float* __restrict__ acc,
                                                      const float* __restrict__ gb, const float* __restrict__ ln_g,
                                                      const float* __restrict__ ln_b, int M, int K, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)M * K) return;
  const int m = (int)(e / K), k = (int)(e % K);
  const float v = ln_g[k] * acc[e] + ln_b[k] * gb[m];
  gw[e] = accumulate ? gw[e] + v : v;
}

}  // namespace fz

using namespace fz;

// Register-operand kernel (wgrad_fast_kernel): whole 64x64 blocks, whole tiles, at most two concatenated
// sources split on a multiple of 8 (measured, round-1/2 probe `wgrad_probe`: 8-13 % faster for 64x64
// blocks, slower for the HBM-bound 32-row blocks of stage 0, which keep the generic kernel)
static bool use_fast(const fz_wgrad_desc* d) {
  const int PR = d->M > 32 ? 64 : 32, QR = d->K > 32 ? 64 : 32;
  const int c0 = d->c0 > 0 ? d->c0 : d->Cin;
  bool fast = PR == 64 && QR == 64 && d->loader == QL_PLAIN && !d->pmul && (d->M % PR) == 0 && (d->K % QR) == 0 &&
              (d->N % kTile) == 0 && d->N == d->Vq && (c0 % 8) == 0 && d->src_mode == 0 && !(d->stats && d->qact) &&
              (int64_t)8 * d->N < ((int64_t)1 << 30);
  { const auto& k = FZ_KNOB("FZ_WGRAD_FAST"); if (k.set && k.val == 0) fast = false; }   // probe builds only
  return fast;
}

static int pick_chunks(int64_t total_tiles, int out_blocks, bool fast, int* tiles_per_chunk) {
  // (workgroup, wave) units over the whole grid, at least 1 tile each.  Measured sweep on MI355X
  // (round-1/2 probe `wgrad_probe2`, FZ_WGRAD_UNITS): every workgroup pays a fixed prologue (first tile's
  // round trip) and epilogue (4-wave reduction, partial block) of a few microseconds, and a grid that is
  // not a whole number of workgroups per CU leaves a tail — ONE workgroup per CU (1024 units) is the
  // optimum for the register-operand kernel (64x64 at 64^3: 83 us against 107 with 4 per CU), two per CU
  // for the HBM-bound generic kernel (32x64 at 128^3: 370 against 400 us); 1.5 per CU loses to both.
  int64_t total_units = fast ? 1024 : 2048;
  int64_t units_target = total_units / (out_blocks > 0 ? out_blocks : 1);
  if (units_target < 16) units_target = 16;
  int64_t tpc = (total_tiles + units_target - 1) / units_target;
  if (tpc < 1) tpc = 1;
  *tiles_per_chunk = (int)tpc;
  int64_t units = (total_tiles + tpc - 1) / tpc;
  return (int)((units + 3) / 4);  // workgroups (4 waves each)
}

extern "C" int64_t fz_wgrad_workspace_bytes(const fz_wgrad_desc* d) {
  if (!d || d->M < 1 || d->K < 1) return -1;
  const int PR = d->M > 32 ? 64 : 32, QR = d->K > 32 ? 64 : 32;
  const int out_blocks = ((d->M + PR - 1) / PR) * ((d->K + QR - 1) / QR);
  const int64_t total_tiles = ((d->N + kTile - 1) / kTile) * d->B;
  int tpc;
  const int nchunk = pick_chunks(total_tiles, out_blocks, use_fast(d), &tpc);
  // partial blocks + partial row sums + (LN fold) reduced accumulator and row sums
  return ((int64_t)nchunk * d->M * d->K + (int64_t)nchunk * d->M + (int64_t)d->M * d->K + d->M) *
         (int64_t)sizeof(float);
}

static bool fz_wgrad_group_enabled() {   // FZ_WGRAD_GROUP=0: diagnostics (the single launches)
  const bool v = !(FZ_KNOB("FZ_WGRAD_GROUP").set && FZ_KNOB("FZ_WGRAD_GROUP").val == 0);   // probe builds: 0 = one launch per layer
  return v;
}

template <typename AT>
struct WgradPlan {
  WgradArgsT<AT> a;
  dim3 grid;
  int PR, QR, nchunk, qp;
  bool fast;
};

template <typename AT>
static int wgrad_plan(const fz_wgrad_desc* d, void* workspace, WgradPlan<AT>& pl) {
  if (!d->p || !d->q[0] || !d->gw) return fail(FZ_E_ARG, "fz_wgrad: null pointer");
  if (d->loader < 0 || d->loader > 2) return fail(FZ_E_ARG, "fz_wgrad: bad loader");
  if (d->N % 4 != 0) return fail(FZ_E_UNSUPPORTED, "fz_wgrad: column count must be a multiple of 4");
  if (d->loader == QL_S2D && ((d->Wo & 1) || d->K != 8 * d->Cin)) return fail(FZ_E_SHAPE, "fz_wgrad: s2d shape");
  if (d->loader == QL_K3 && d->K != 27 * d->Cin) return fail(FZ_E_SHAPE, "fz_wgrad: k3 shape");
  const int PR = d->M > 32 ? 64 : 32, QR = d->K > 32 ? 64 : 32;
  const int gy = (d->M + PR - 1) / PR, gz = (d->K + QR - 1) / QR;
  const int64_t total_tiles = ((d->N + kTile - 1) / kTile) * d->B;
  int tpc;
  pl.fast = use_fast(d);
  pl.PR = PR; pl.QR = QR;
  pl.nchunk = pick_chunks(total_tiles, gy * gz, pl.fast, &tpc);
  pl.qp = d->stats ? QP_STATS : (d->qact == 2 ? QP_GELU : (d->qact == 1 ? QP_RELU : QP_NONE));
  WgradArgsT<AT>& a = pl.a;
  a.p = (const AT*)d->p; a.M = d->M; a.pmul = (const AT*)d->pmul; a.pmul_kind = d->pmul_kind;
  for (int i = 0; i < 4; ++i) a.q[i] = (const AT*)d->q[i];
  a.nsrc = d->nsrc; a.src_mode = d->src_mode; a.c0 = d->c0 > 0 ? d->c0 : d->Cin; a.Cin = d->Cin; a.K = d->K;
  a.Vq = d->Vq; a.D = d->D; a.H = d->H; a.W = d->W; a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo;
  a.stats = d->stats; a.qact = d->qact; a.B = d->B; a.tiles_per_chunk = tpc;
  a.part = (float*)workspace;
  a.part_bias = a.part + (int64_t)pl.nchunk * d->M * d->K;
  pl.grid = dim3(pl.nchunk, gy, gz);
  return FZ_OK;
}

static size_t wgrad_fast_lds(int PR, int QR) {
  const size_t ldsf = (size_t)4 * (PR + QR) * kStrideF * sizeof(float);
  const size_t ldsr = (size_t)4 * 1024 * sizeof(float);
  return ldsf > ldsr ? ldsf : ldsr;
}

// the partial-sum launch of one problem
template <typename AT>
static int wgrad_main_launch(const fz_wgrad_desc* d, const WgradPlan<AT>& pl, hipStream_t st) {
  const WgradArgsT<AT>& a = pl.a;
  const int PR = pl.PR, QR = pl.QR;
  const dim3 grid = pl.grid, block(256);
  const size_t lds = (size_t)4 * (PR + QR) * kStride * sizeof(float);
  constexpr bool kBf16 = !std::is_same<AT, float>::value;  // bf16 activations: plain bf16 MFMAs
  // fp32 activations: six-term split-bf16 products unless the split-bf16 family is switched off (fz_gemm_bx_enable)
  const int bf3g = products_split(d->products);
#define FZ_WG(MBP, MBQ, QL)                                                                        \
  do {                                                                                             \
    if (kBf16) hipLaunchKernelGGL((wgrad_kernel<MBP, MBQ, QL, 1>), grid, block, lds, st, a);       \
    else if (bf3g) hipLaunchKernelGGL((wgrad_kernel<MBP, MBQ, QL, 6>), grid, block, lds, st, a);   \
    else hipLaunchKernelGGL((wgrad_kernel<MBP, MBQ, QL, 0>), grid, block, lds, st, a);             \
  } while (0)
#define FZ_WG_SHAPES(QL)                                 \
  do {                                                   \
    if (PR == 64 && QR == 64) FZ_WG(2, 2, QL);           \
    else if (PR == 64) FZ_WG(2, 1, QL);                  \
    else if (QR == 64) FZ_WG(1, 2, QL);                  \
    else FZ_WG(1, 1, QL);                                \
  } while (0)
  if (pl.fast) {
    const int qp = pl.qp;
    const size_t ldsz = wgrad_fast_lds(PR, QR);
    const int bf3 = bf3g;
#define FZ_WGF(MBP, MBQ, QP)                                                                              \
  do {                                                                                                    \
    if (kBf16) hipLaunchKernelGGL((wgrad_fast_kernel<MBP, MBQ, QP, 1>), grid, block, ldsz, st, a);        \
    else if (bf3) hipLaunchKernelGGL((wgrad_fast_kernel<MBP, MBQ, QP, 6>), grid, block, ldsz, st, a);     \
    else hipLaunchKernelGGL((wgrad_fast_kernel<MBP, MBQ, QP, 0>), grid, block, ldsz, st, a);              \
  } while (0)
#define FZ_WGF_SHAPES(QP)                                \
  do {                                                   \
    if (PR == 64 && QR == 64) FZ_WGF(2, 2, QP);          \
    else if (PR == 64) FZ_WGF(2, 1, QP);                 \
    else if (QR == 64) FZ_WGF(1, 2, QP);                 \
    else FZ_WGF(1, 1, QP);                               \
  } while (0)
    if (qp == QP_STATS) FZ_WGF_SHAPES(QP_STATS);
    else if (qp == QP_GELU) FZ_WGF_SHAPES(QP_GELU);
    else if (qp == QP_RELU) FZ_WGF_SHAPES(QP_RELU);
    else FZ_WGF_SHAPES(QP_NONE);
  } else if (d->loader == QL_PLAIN) FZ_WG_SHAPES(QL_PLAIN);
  else if (d->loader == QL_S2D) FZ_WG_SHAPES(QL_S2D);
  else FZ_WG_SHAPES(QL_K3);
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

// the fixed-order reduction of one problem's partial blocks into gw / gbias
// the fixed-order reduction of one problem's partial blocks into gw / gbias, as a finish job
template <typename AT>
static FinishJob wgrad_finish_job(const fz_wgrad_desc* d, const WgradPlan<AT>& pl) {
  const int64_t MK = (int64_t)d->M * d->K;
  // few partial blocks of a large weight: one thread per element; up to 256 of a mid-sized one: 64 elements x 4 chunk slices
  int wide = (pl.nchunk <= 64 && MK >= 16384) ? 1 : ((pl.nchunk <= 256 && MK >= 4096) ? 2 : 0);
  // ... four elements per thread where the 16-byte accesses are aligned (partial rows, outputs, LayerNorm vectors)
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (wide == 1 && MK >= 65536 && (d->M % 4) == 0 && (d->K % 4) == 0 && al16(pl.a.part) && al16(d->gw) &&
      (d->gbias == nullptr || (al16(pl.a.part_bias) && al16(d->gbias))) && (d->ln_g == nullptr || (al16(d->ln_g) && al16(d->ln_b))))
    wide = 3;
  const int per = wide == 3 ? 1024 : (wide == 1 ? 256 : (wide == 2 ? 64 : 8));
  const int nbw = (int)((MK + per - 1) / per);
  const int nbb = d->gbias != nullptr ? (d->M + per - 1) / per : 0;
  FinishJob j = finish_job(FK_WGRAD, nbw + nbb);
  j.u.wg = WgradFinishOne{pl.a.part, pl.a.part_bias, d->gw, d->gbias, d->ln_g != nullptr ? d->ln_g : (const float*)nullptr, d->ln_b,
                          pl.nchunk, d->M, d->K, d->accumulate, nbw, wide};
  return j;
}
template <typename AT>
static int wgrad_finish_launch(const fz_wgrad_desc* d, const WgradPlan<AT>& pl, hipStream_t st) {
  const FinishJob j = wgrad_finish_job<AT>(d, pl);
  return finish_run(&j, 1, st);
}

template <typename AT>
static int wgrad_launch(const fz_wgrad_desc* d, void* workspace, fz_stream_t stream) {
  WgradPlan<AT> pl;
  const int rc = wgrad_plan<AT>(d, workspace, pl);
  if (rc != FZ_OK) return rc;
  if (d->B == 0) return FZ_OK;
  const int rc2 = wgrad_main_launch<AT>(d, pl, (hipStream_t)stream);
  if (rc2 != FZ_OK) return rc2;
  return wgrad_finish_launch<AT>(d, pl, (hipStream_t)stream);
}

// n <= kWgGroupMax problems: ONE partial-sum grid when every problem takes the register-operand kernel at 64 x 64
// blocks (the dense layers of a C >= 64 block), else the single launches in order; then the finish launches
template <typename AT>
static int wgrad_group_launch(const fz_wgrad_desc* const* ds, void* const* ws, int n, fz_stream_t stream) {
  WgradPlan<AT> pl[kWgGroupMax];
  bool groupable = n >= 2 && fz_wgrad_group_enabled();
  for (int i = 0; i < n; ++i) {
    const int rc = wgrad_plan<AT>(ds[i], ws[i], pl[i]);
    if (rc != FZ_OK) return rc;
    groupable = groupable && pl[i].fast && pl[i].PR == 64 && pl[i].QR == 64 && ds[i]->B > 0 &&
                products_split(ds[i]->products) == products_split(ds[0]->products);   // one kernel instantiation per grid
  }
  hipStream_t st = (hipStream_t)stream;
  if (!groupable) {
    for (int i = 0; i < n; ++i) {
      if (ds[i]->B == 0) continue;
      int rc = wgrad_main_launch<AT>(ds[i], pl[i], st);
      if (rc == FZ_OK) rc = wgrad_finish_launch<AT>(ds[i], pl[i], st);
      if (rc != FZ_OK) return rc;
    }
    return FZ_OK;
  }
  WgradGroupT<AT> g;
  int total = 0;
  for (int i = 0; i < kWgGroupMax; ++i) {
    const int j = i < n ? i : n - 1;
    g.a[i] = pl[j].a;
    g.start[i] = total;
    g.gx[i] = (int)pl[j].grid.x; g.gy[i] = (int)pl[j].grid.y;
    g.qpro[i] = pl[j].qp;
    if (i < n) total += (int)(pl[i].grid.x * pl[i].grid.y * pl[i].grid.z);
  }
  g.start[kWgGroupMax] = total;
  g.n = n;
  const size_t ldsz = wgrad_fast_lds(64, 64);
  constexpr bool kBf16 = !std::is_same<AT, float>::value;
  const int bf3 = products_split(ds[0]->products);
  if (kBf16) hipLaunchKernelGGL((wgrad_fast_group_kernel<2, 2, 1, AT>), dim3(total), dim3(256), ldsz, st, g);
  else if (bf3) hipLaunchKernelGGL((wgrad_fast_group_kernel<2, 2, 6, AT>), dim3(total), dim3(256), ldsz, st, g);
  else hipLaunchKernelGGL((wgrad_fast_group_kernel<2, 2, 0, AT>), dim3(total), dim3(256), ldsz, st, g);
  FZ_LAUNCH_CHECK();
  FinishJob fj[kWgGroupMax];
  for (int i = 0; i < n; ++i) fj[i] = wgrad_finish_job<AT>(ds[i], pl[i]);
  return finish_run(fj, n, st);
}

extern "C" int fz_wgrad(const fz_wgrad_desc* d, void* workspace, fz_stream_t stream) {
  if (!d || !workspace) return fail(FZ_E_ARG, "fz_wgrad: null descriptor/workspace");
  if (d->act_dtype == FZ_STORE_F32) return wgrad_launch<float>(d, workspace, stream);
  if (d->act_dtype == FZ_STORE_BF16) return wgrad_launch<bf16>(d, workspace, stream);
  return fail(FZ_E_ARG, "fz_wgrad: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}

extern "C" int fz_wgrad_group(const fz_wgrad_desc* const* descs, void* const* workspaces, int n, fz_stream_t stream) {
  if (!descs || !workspaces || n < 1 || n > kWgGroupMax) return fail(FZ_E_ARG, "fz_wgrad_group: 1..4 descriptors");
  for (int i = 0; i < n; ++i) {
    if (!descs[i] || !workspaces[i]) return fail(FZ_E_ARG, "fz_wgrad_group: null descriptor/workspace");
    if (descs[i]->act_dtype != descs[0]->act_dtype) return fail(FZ_E_ARG, "fz_wgrad_group: one activation storage type per group");
  }
  if (descs[0]->act_dtype == FZ_STORE_F32) return wgrad_group_launch<float>(descs, workspaces, n, stream);
  if (descs[0]->act_dtype == FZ_STORE_BF16) return wgrad_group_launch<bf16>(descs, workspaces, n, stream);
  return fail(FZ_E_ARG, "fz_wgrad_group: act_dtype must be FZ_STORE_F32 or FZ_STORE_BF16");
}

// out[e] (+)= Σ_chunks part[chunk][e], fixed order — exposed for kernels that produce their own
// per-workgroup partial sums (conv3.hip).
extern "C" int fz_chunk_reduce_ld(const float* part, int nchunk, int64_t n, int64_t ld, float* out, int accumulate,
                                  fz_stream_t stream) {
  if (!part || !out || nchunk < 1 || n < 1 || ld < n) return fail(FZ_E_ARG, "fz_chunk_reduce: bad arguments");
  FinishJob j = finish_job(FK_CHUNK, (int)((n + 7) / 8));
  j.u.chunk = FinChunk{part, out, (long long)n, (long long)ld, nchunk, accumulate};
  return finish_run(&j, 1, (hipStream_t)stream);
}

extern "C" int fz_chunk_reduce(const float* part, int nchunk, int64_t n, float* out, int accumulate,
                               fz_stream_t stream) {
  return fz_chunk_reduce_ld(part, nchunk, n, n, out, accumulate, stream);
}
