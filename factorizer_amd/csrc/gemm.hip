// gemm.hip — channels-first GEMM family on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   Out[m, n] = epilogue( Σ_k A[m, k] · prologue(In)[k, n] )        n = voxel (column)
//
// One kernel template serves every dense layer of the Factorizer U-shape and their input
// gradients (reference call sites, paths relative to the reference root):
//   * Linear / in_proj / out_proj / adapter / MLP / head : Conv1d(k=1) on flatten(2)
//       layers/linear.py:53-58, factorizer.py:38,53,116, layers/mlp.py:54-60, unet.py:253
//   * downsample Conv3d(k=2, stride=2)        unet.py:53   (space-to-depth loader)
//   * upsample  ConvTranspose3d(k=2, stride=2) unet.py:123 (depth-to-space epilogue)
//   * LayerNorm over channels fused as a prologue (layers/norm.py:29-34), ReLU/GELU, bias,
//     residual add (factorizer.py:75-76) and window averaging (operations.py:426-433) fused
//     as prologue/epilogue so each full-resolution tensor is read/written once per layer.
//
// MFMA mapping (wave64, 32x32x2 f32): lane l = (j = l&31, h = l>>5).  The B operand of K-step
// s is In[k = 2s+h][column j]; each lane loads ONE 16-byte vector = 4 consecutive voxels of
// channel 2s+h, and the 4 components feed 4 MFMAs (column groups q = 0..3, voxel 4j+q), so
// global loads are 1 KiB-per-wave coalesced and no LDS staging of activations is needed.
// Weights (the A operand) are staged once per workgroup in LDS in operand order.
// The accumulator tile has its row in (register, h) and its column in j, so the epilogue
// stores 16-byte vectors (4 voxels) per register: 512 B contiguous per output row.
#include "fz_common.h"

namespace fz {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { LOAD_PLAIN = 0, LOAD_S2D = 1, LOAD_K3 = 2 };
enum { EPI_PLAIN = 0, EPI_D2S = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct GemmArgs {
  const float* x[4];   // input sources
  int nsrc;            // number of sources
  int src_mode;        // 0: channel concat (x[0]: c0 channels, x[1]: Cin-c0)   1: average of sources
  int c0;
  int Cin;             // input channels
  int64_t Vin;         // voxels per sample of the input tensor(s)
  int Di, Hi, Wi;      // input D, H, W (LOAD_S2D: fine tensor; LOAD_K3: the grid)
  const float* w;      // weights
  int w_t, ldw;        // A[m][k] = w_t ? w[k*ldw + m] : w[m*ldw + k]
  int M, K;            // output rows, reduction length
  const float* bias;   // [M] (EPI_PLAIN) or [M/8] (EPI_D2S), may be null
  int ln;              // LayerNorm prologue over the Cin channels
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  float* stats_out;    // [B][2][Vin] (mean, rstd) or null
  int bact;            // activation applied to the B operand
  const float* bmul;   // B operand *= act'(bmul) (same shape as the input) or null
  int bmul_kind;
  int eact;            // activation applied to the result
  const float* res;    // residual added in the epilogue (same shape as y) or null
  const float* emul;   // epilogue multiply by act'(emul) (same shape as y) or null
  int emul_kind;
  float* y;
  int64_t Ncol;        // columns per sample (= Vin for LOAD_PLAIN, coarse voxels for LOAD_S2D)
  int Ho, Wo;          // coarse H, W (LOAD_S2D columns / EPI_D2S input grid)
  int B;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float act_f(int kind, float v) {
  if (kind == ACT_RELU) return v > 0.f ? v : 0.f;
  if (kind == ACT_GELU) return gelu_f(v);
  return v;
}
__device__ __forceinline__ float act_grad_f(int kind, float v) {
  if (kind == ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (kind == ACT_GELU) return gelu_grad_f(v);
  return 1.f;
}

// k index of A-operand step `a` for lane half h
template <int LOADER>
__device__ __forceinline__ int a_k(int a, int h) {
  if (LOADER == LOAD_S2D) return (2 * (a >> 3) + h) * 8 + (a & 7);
  if (LOADER == LOAD_K3) return (2 * (a / 27) + h) * 27 + (a % 27);
  return 2 * a + h;
}

constexpr int kAChunk = 64;  // A-operand steps staged in LDS at a time

template <int MB, int LOADER, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int NACC = (LOADER == LOAD_S2D) ? 2 : 4;  // column groups per lane
  constexpr int TN = 32 * NACC;                         // columns per wave tile
  __shared__ float As[kAChunk * MB * 64];
  __shared__ float sW[32 * MB];
  __shared__ float tW[32 * MB];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int tiles_per_sample = (int)((p.Ncol + TN * 4 - 1) / (TN * 4));
  const int b = blockIdx.x / tiles_per_sample;
  const int64_t n0 = ((int64_t)(blockIdx.x % tiles_per_sample) * 4 + wave) * TN;
  const int m0 = blockIdx.y * 32 * MB;
  const int nA = (p.K + 1) / 2;  // number of A-operand steps (odd K: last step half-masked)

  // ---- per-row constants for the LN prologue: s[m] = Σ_k W[m][k]γ_k ; t[m] = Σ_k W[m][k]β_k
  if (p.ln) {
    for (int r = threadIdx.x; r < 32 * MB; r += blockDim.x) {
      const int m = m0 + r;
      float s = 0.f, t = 0.f;
      if (m < p.M) {
        for (int k = 0; k < p.K; ++k) {
          const float wv = p.w_t ? p.w[(int64_t)k * p.ldw + m] : p.w[(int64_t)m * p.ldw + k];
          s += wv * p.ln_g[k];
          t += wv * p.ln_b[k];
        }
      }
      sW[r] = s;
      tW[r] = t;
    }
  }

  // ---- per-lane input addressing ----
  int64_t col_off;       // offset of this lane's first column inside a channel plane
  bool col_ok;           // lane's columns are inside the tensor
  int kw0 = 0, kh0 = 0, kd0 = 0;  // LOAD_K3: (w, h, d) of this lane's first voxel
  if (LOADER == LOAD_PLAIN || LOADER == LOAD_K3) {
    col_off = n0 + 4 * j;
    col_ok = col_off < p.Ncol;
    if (LOADER == LOAD_K3) {
      const int64_t nn = col_ok ? col_off : 0;
      kw0 = (int)(nn % p.Wi);
      kh0 = (int)((nn / p.Wi) % p.Hi);
      kd0 = (int)(nn / ((int64_t)p.Wi * p.Hi));
    }
  } else {
    const int64_t n = n0 + 2 * j;  // coarse voxel pair (wo even)
    col_ok = n < p.Ncol;
    const int64_t nn = col_ok ? n : 0;
    const int wo = (int)(nn % p.Wo);
    const int64_t t2 = nn / p.Wo;
    const int ho = (int)(t2 % p.Ho);
    const int dz = (int)(t2 / p.Ho);
    col_off = ((int64_t)(2 * dz) * p.Hi + 2 * ho) * p.Wi + 2 * wo;
  }

  f32x16 acc[MB][NACC];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, shift[4] = {0.f, 0.f, 0.f, 0.f};

  // ---- B-operand fetch for load step s (4 floats per lane) ----
  auto fetch = [&](int s, float (&v)[4]) {
    int c;
    int64_t off;
    if (LOADER == LOAD_PLAIN) {
      c = 2 * s + h;
      off = col_off;
    } else if (LOADER == LOAD_S2D) {
      c = 2 * (s >> 2) + h;
      off = col_off + (int64_t)((s >> 1) & 1) * p.Hi * p.Wi + (int64_t)(s & 1) * p.Wi;
    } else {
      c = 2 * (s / 27) + h;
      off = 0;
    }
    if (!col_ok || c >= p.Cin) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
      return;
    }
    if (LOADER == LOAD_K3) {
      const int tap = s % 27;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int zd = kd0 + kd - 1, zh = kh0 + kh - 1;
      v[0] = v[1] = v[2] = v[3] = 0.f;
      if (zd < 0 || zd >= p.Di || zh < 0 || zh >= p.Hi) return;
      const float* row = p.x[0] + ((int64_t)b * p.Cin + c) * p.Vin + ((int64_t)zd * p.Hi + zh) * p.Wi;
      const float4 t = *reinterpret_cast<const float4*>(row + kw0);
      if (kw == 1) {
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else if (kw == 0) {
        v[0] = kw0 > 0 ? row[kw0 - 1] : 0.f; v[1] = t.x; v[2] = t.y; v[3] = t.z;
      } else {
        v[0] = t.y; v[1] = t.z; v[2] = t.w; v[3] = (kw0 + 4 < p.Wi) ? row[kw0 + 4] : 0.f;
      }
      return;
    }
    if (p.src_mode == 0) {
      const float* src;
      if (c < p.c0) src = p.x[0] + ((int64_t)b * p.c0 + c) * p.Vin;
      else src = p.x[1] + ((int64_t)b * (p.Cin - p.c0) + (c - p.c0)) * p.Vin;
      const float4 t = *reinterpret_cast<const float4*>(src + off);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (p.bmul != nullptr) {
        const float4 e = *reinterpret_cast<const float4*>(p.bmul + ((int64_t)b * p.Cin + c) * p.Vin + off);
        v[0] *= act_grad_f(p.bmul_kind, e.x); v[1] *= act_grad_f(p.bmul_kind, e.y);
        v[2] *= act_grad_f(p.bmul_kind, e.z); v[3] *= act_grad_f(p.bmul_kind, e.w);
      }
    } else {
      const int64_t o = ((int64_t)b * p.Cin + c) * p.Vin + off;
      float4 t = *reinterpret_cast<const float4*>(p.x[0] + o);
      v[0] = 0.0f + t.x; v[1] = 0.0f + t.y; v[2] = 0.0f + t.z; v[3] = 0.0f + t.w;
      for (int i = 1; i < p.nsrc; ++i) {
        t = *reinterpret_cast<const float4*>(p.x[i] + o);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      const float nw = (float)p.nsrc;
      v[0] /= nw; v[1] /= nw; v[2] /= nw; v[3] /= nw;
    }
  };

  const int nload = (LOADER == LOAD_S2D) ? nA / 2 : nA;  // load steps
  float cur[4], nxt[4];
  fetch(0, cur);
  if (p.ln) {
    // shift = channel-0 value of each voxel (held by half 0 at step 0): a cheap, well-conditioned
    // pivot for the single-pass variance
#pragma unroll
    for (int e = 0; e < 4; ++e) shift[e] = __shfl(cur[e], j, 64);
  }

  for (int a0 = 0; a0 < nA; a0 += kAChunk) {
    const int an = min(kAChunk, nA - a0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < an * MB * 64; idx += blockDim.x) {
      const int l = idx & 63;
      const int mb = (idx >> 6) % MB;
      const int a = a0 + idx / (64 * MB);
      const int m = m0 + mb * 32 + (l & 31);
      const int k = a_k<LOADER>(a, l >> 5);
      float wv = 0.f;
      if (m < p.M && k < p.K) {
        wv = p.w_t ? p.w[(int64_t)k * p.ldw + m] : p.w[(int64_t)m * p.ldw + k];
        if (p.ln) wv *= p.ln_g[k];
      }
      As[idx] = wv;
    }
    __syncthreads();
    if (LOADER != LOAD_S2D) {
      for (int al = 0; al < an; ++al) {
        const int s = a0 + al;
        if (s + 1 < nload) fetch(s + 1, nxt);
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = cur[e];
          if (p.ln) {
            t -= shift[e];
            s1[e] += t;
            s2[e] += t * t;
          }
          bv[e] = p.bact ? act_f(p.bact, t) : t;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float av = As[(al * MB + mb) * 64 + lane];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[q], acc[mb][q], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
      }
    } else {
      for (int al = 0; al < an; al += 2) {
        const int s = (a0 + al) >> 1;
        if (s + 1 < nload) fetch(s + 1, nxt);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float av0 = As[(al * MB + mb) * 64 + lane];        // tw = 0
          const float av1 = As[((al + 1) * MB + mb) * 64 + lane];  // tw = 1
          acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, cur[0], acc[mb][0], 0, 0, 0);
          acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, cur[2], acc[mb][1], 0, 0, 0);
          acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, cur[1], acc[mb][0], 0, 0, 0);
          acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, cur[3], acc[mb][1], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
      }
    }
  }

  // ---- LayerNorm statistics of this lane's 4 voxels ----
  float mu_d[4], rstd[4];
  if (p.ln) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t1 = s1[e] + __shfl_xor(s1[e], 32, 64);
      const float t2 = s2[e] + __shfl_xor(s2[e], 32, 64);
      const float inv = 1.0f / (float)p.Cin;
      const float md = t1 * inv;
      float var = t2 * inv - md * md;
      var = var > 0.f ? var : 0.f;
      mu_d[e] = md;
      rstd[e] = 1.0f / sqrtf(var + p.ln_eps);
    }
    if (p.stats_out != nullptr && blockIdx.y == 0 && h == 0 && col_ok) {
      float4 mean4 = make_float4(shift[0] + mu_d[0], shift[1] + mu_d[1], shift[2] + mu_d[2], shift[3] + mu_d[3]);
      float4 rs4 = make_float4(rstd[0], rstd[1], rstd[2], rstd[3]);
      float* so = p.stats_out + (int64_t)b * 2 * p.Vin;
      *reinterpret_cast<float4*>(so + col_off) = mean4;
      *reinterpret_cast<float4*>(so + p.Vin + col_off) = rs4;
    }
  }

  // ---- epilogue ----
  if (!col_ok) return;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;  // row inside the 32-row block
      const int m = m0 + mb * 32 + rl;
      if (m >= p.M) continue;
      float v[NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q) v[q] = acc[mb][q][r];
      if (p.ln) {
        const float sw = sW[mb * 32 + rl], tw = tW[mb * 32 + rl];
#pragma unroll
        for (int q = 0; q < NACC; ++q) v[q] = rstd[q] * (v[q] - mu_d[q] * sw) + tw;
      }
      if (EPI == EPI_PLAIN) {
        const float bs = p.bias ? p.bias[m] : 0.f;
        const int64_t o = ((int64_t)b * p.M + m) * p.Ncol + (LOADER == LOAD_S2D ? n0 + 2 * j : n0 + 4 * j);
#pragma unroll
        for (int q = 0; q < NACC; ++q) v[q] += bs;
        if (p.eact) {
#pragma unroll
          for (int q = 0; q < NACC; ++q) v[q] = act_f(p.eact, v[q]);
        }
        if (NACC == 4) {
          if (p.emul) {
            const float4 e4 = *reinterpret_cast<const float4*>(p.emul + o);
            v[0] *= act_grad_f(p.emul_kind, e4.x); v[1] *= act_grad_f(p.emul_kind, e4.y);
            v[2] *= act_grad_f(p.emul_kind, e4.z); v[3] *= act_grad_f(p.emul_kind, e4.w);
          }
          if (p.res) {
            const float4 r4 = *reinterpret_cast<const float4*>(p.res + o);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
          }
          *reinterpret_cast<float4*>(p.y + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          if (p.res) {
            const float2 r2 = *reinterpret_cast<const float2*>(p.res + o);
            v[0] += r2.x; v[1] += r2.y;
          }
          *reinterpret_cast<float2*>(p.y + o) = make_float2(v[0], v[1]);
        }
      }
    }
    if (EPI == EPI_D2S) {
      // rows are (o, tap): m = o*8 + td*4 + th*2 + tw ; inside a 32-row block td = h,
      // th = (r>>1)&1, tw = r&1, o_local = r>>2.  Columns 4j..4j+3 are coarse voxels; the
      // fine tensor gets 8 consecutive voxels (tw pairs) per (o, td, th).
      const int Wf = 2 * p.Wo, Hf = 2 * p.Ho;
      const int64_t Vf = 8 * p.Ncol;
      const int Mo = p.M >> 3;
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const int r0 = 2 * rp;  // registers r0 (tw=0), r0+1 (tw=1)
        const int o = ((m0 + mb * 32) >> 3) + (r0 >> 2);
        if (o >= Mo) continue;
        const int th = (r0 >> 1) & 1, td = h;
        const float bs = p.bias ? p.bias[o] : 0.f;
        const int64_t n = n0 + 4 * j;
        float* ybase = p.y + ((int64_t)b * Mo + o) * Vf;
        if ((p.Wo & 3) == 0) {
          const int wo = (int)(n % p.Wo);
          const int64_t t2 = n / p.Wo;
          const int ho = (int)(t2 % p.Ho);
          const int dz = (int)(t2 / p.Ho);
          const int64_t fo = ((int64_t)(2 * dz + td) * Hf + (2 * ho + th)) * Wf + 2 * wo;
          float4 lo = make_float4(acc[mb][0][r0] + bs, acc[mb][0][r0 + 1] + bs, acc[mb][1][r0] + bs,
                                  acc[mb][1][r0 + 1] + bs);
          float4 hi = make_float4(acc[mb][2][r0] + bs, acc[mb][2][r0 + 1] + bs, acc[mb][3][r0] + bs,
                                  acc[mb][3][r0 + 1] + bs);
          *reinterpret_cast<float4*>(ybase + fo) = lo;
          *reinterpret_cast<float4*>(ybase + fo + 4) = hi;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t nq = n + q;
            const int wo = (int)(nq % p.Wo);
            const int64_t t2 = nq / p.Wo;
            const int ho = (int)(t2 % p.Ho);
            const int dz = (int)(t2 / p.Ho);
            const int64_t fo = ((int64_t)(2 * dz + td) * Hf + (2 * ho + th)) * Wf + 2 * wo;
            ybase[fo] = acc[mb][q][r0] + bs;
            ybase[fo + 1] = acc[mb][q][r0 + 1] + bs;
          }
        }
      }
    }
  }
}

}  // namespace fz

using namespace fz;

// Flat C view of GemmArgs for the ABI (see include/factorizer_hip.h: fz_gemm_desc).
extern "C" int fz_gemm(const fz_gemm_desc* d, fz_stream_t stream) {
  if (!d) return fail(FZ_E_ARG, "fz_gemm: null descriptor");
  if (!d->x[0] || !d->w || !d->y) return fail(FZ_E_ARG, "fz_gemm: null pointer");
  if (d->loader < LOAD_PLAIN || d->loader > LOAD_K3) return fail(FZ_E_ARG, "fz_gemm: bad loader");
  if (d->loader == LOAD_K3 && (d->epilogue != EPI_PLAIN || d->ln || d->src_mode != 0 || d->nsrc != 1 ||
                               (d->Wi & 3) || d->K != 27 * d->Cin || (d->Cin & 1)))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: k3 loader needs W % 4 == 0, even Cin, K = 27*Cin, plain epilogue");
  if (d->bmul && (d->loader != LOAD_PLAIN || d->src_mode != 0 || d->nsrc != 1))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: bmul needs the plain single-source loader");
  if (d->epilogue != EPI_PLAIN && d->epilogue != EPI_D2S) return fail(FZ_E_ARG, "fz_gemm: bad epilogue");
  if (d->B < 0 || d->Cin < 1 || d->M < 1 || d->K < 1)
    return fail(FZ_E_SHAPE, "fz_gemm: sizes must be positive");
  if ((d->K & 1) && (d->loader != LOAD_PLAIN || d->ln))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: odd K only with the plain loader and no LayerNorm prologue");
  if (d->Ncol % 4 != 0 && d->loader != LOAD_S2D) return fail(FZ_E_UNSUPPORTED, "fz_gemm: voxel count must be a multiple of 4");
  if (d->loader == LOAD_S2D && ((d->Wo & 1) || d->epilogue != EPI_PLAIN || d->ln || d->src_mode != 0 || d->nsrc != 1))
    return fail(FZ_E_UNSUPPORTED, "fz_gemm: space-to-depth loader needs even coarse width, plain epilogue");
  if (d->epilogue == EPI_D2S && (d->M % 8 != 0)) return fail(FZ_E_SHAPE, "fz_gemm: depth-to-space rows must be 8*C");
  if (d->nsrc < 1 || d->nsrc > 4) return fail(FZ_E_ARG, "fz_gemm: 1..4 sources");
  if (d->B == 0) return FZ_OK;
  GemmArgs a;
  for (int i = 0; i < 4; ++i) a.x[i] = d->x[i];
  a.nsrc = d->nsrc; a.src_mode = d->src_mode; a.c0 = d->c0 > 0 ? d->c0 : d->Cin; a.Cin = d->Cin;
  a.Vin = d->Vin; a.Di = d->Di; a.Hi = d->Hi; a.Wi = d->Wi; a.bmul = d->bmul; a.bmul_kind = d->bmul_kind; a.w = d->w; a.w_t = d->w_t; a.ldw = d->ldw; a.M = d->M; a.K = d->K;
  a.bias = d->bias; a.ln = d->ln; a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.ln_eps = d->ln_eps;
  a.stats_out = d->stats_out; a.bact = d->bact; a.eact = d->eact; a.res = d->res; a.emul = d->emul;
  a.emul_kind = d->emul_kind; a.y = d->y; a.Ncol = d->Ncol; a.Ho = d->Ho; a.Wo = d->Wo; a.B = d->B;
  const int TN = d->loader == LOAD_S2D ? 64 : 128;
  const int64_t tiles = (d->Ncol + TN * 4 - 1) / (TN * 4);
  const int MBsel = d->M > 32 ? 2 : 1;
  dim3 grid((unsigned)(tiles * d->B), (unsigned)((d->M + 32 * MBsel - 1) / (32 * MBsel))), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FZ_GEMM(MB, L, E) hipLaunchKernelGGL((gemm_kernel<MB, L, E>), grid, block, 0, st, a)
  if (d->loader == LOAD_PLAIN && d->epilogue == EPI_PLAIN) { if (MBsel == 2) FZ_GEMM(2, LOAD_PLAIN, EPI_PLAIN); else FZ_GEMM(1, LOAD_PLAIN, EPI_PLAIN); }
  else if (d->loader == LOAD_PLAIN && d->epilogue == EPI_D2S) { if (MBsel == 2) FZ_GEMM(2, LOAD_PLAIN, EPI_D2S); else FZ_GEMM(1, LOAD_PLAIN, EPI_D2S); }
  else if (d->loader == LOAD_K3) { if (MBsel == 2) FZ_GEMM(2, LOAD_K3, EPI_PLAIN); else FZ_GEMM(1, LOAD_K3, EPI_PLAIN); }
  else { if (MBsel == 2) FZ_GEMM(2, LOAD_S2D, EPI_PLAIN); else FZ_GEMM(1, LOAD_S2D, EPI_PLAIN); }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}
