// nmf_cf.hip — FactMixer core on channels-first tensors: shifted-window matricize → NMF →
// inverse matricize in ONE kernel per window (hot shape: head_dim 8, patch 8x8x8).
//
// Replaces the chain SWMatricize.forward → NMF.forward → SWMatricize.inverse_forward
// (factorizer/factorizer.py:41-50; operations.py:417-434; matrix_factorization.py:514-546):
// the (W·B·h, G, 8, 512) matricized tensors are never materialised.  A wave gathers its 8x512
// matrix straight from t (B, C, D, H, W) with the window's cyclic shift, runs the per-wave NMF
// program of nmf_core.h with X in registers, and scatters u vᵀ back to the same voxels of the
// averaged output a:  window 0 stores (0 + z_0), window w>0 adds z_w, the last window divides by
// the number of windows — the reference's ((0.0 + z_0) + z_1 + …) / W order (operations.py:426-433).
// Windows are separate launches on one stream, so the accumulation is deterministic.
//
// Lane map: lane l = (p0 & 3 = l>>4, p1 = (l>>1)&7, half = l&1); for every channel dd and
// p0-group jp the lane moves one 16-byte vector = voxels p2 = 4·half..4·half+3 of patch row
// (p0 = 4·jp + (l>>4), p1).  Column index of local element (jp, e): n = (p0·8 + p1)·8 + 4·half + e.
#include "nmf_cf.h"

namespace fz {

// One tile (WPB patches along W) of one window of the forward.
template <int R, int SOLVER, int WPB, bool HALF, typename AT>
__device__ __forceinline__ void cf_fwd_tile_body(const AT* __restrict__ t, const float* __restrict__ u0,
                                                 const float* __restrict__ v0, AT* __restrict__ out, const CfGeom& q,
                                                 const CfTileId& id, int T, float eps, float* S) {
  using TL = CfTile<WPB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t base, V;
  unsigned off[2], off2[2];
  int lidx[2];
  cf_tile_decode<WPB>(q, id, tid, base, V, off, lidx, off2);
  const int own0 = cf_owner_lidx<WPB>(lane, wave, 0), own1 = cf_owner_lidx<WPB>(lane, wave, 1);

  float x[8][8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(t + base + dd * V, off[k], off2[k]);
      x[dd][k * 4 + 0] = v.x; x[dd][k * 4 + 1] = v.y; x[dd][k * 4 + 2] = v.z; x[dd][k * 4 + 3] = v.w;
    }
  cf_to_owner<WPB>(S, lidx, own0, own1, x);

  CfWave w{lane};
  float u[8][R], v[8][R];
  nmf_forward_wave<8, 8, R, SOLVER>(w, u0, v0, x, u, v, 8, T, eps);

  // owner → coalesced, then the (read-modify-)write of the running window average; all loads of
  // the running sum are issued at once (one exposed round trip)
  const float dv = (float)q.divisor;
  const bool dv_pow2 = cf_pow2(dv);
  // (the plane base back in a scalar register pair: after the wave program the compiler otherwise carries it in vector
  // registers and every epilogue access pays a 64-bit vector address)
  asm volatile("" : "+s"(base));
  float4 old[8][2];
  if (q.accumulate) {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd)
#pragma unroll
      for (int k = 0; k < 2; ++k) old[dd][k] = cf_ld4<HALF>(out + base + dd * V, off[k], off2[k]);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own0) =
          make_float4(x[2 * s + c][0], x[2 * s + c][1], x[2 * s + c][2], x[2 * s + c][3]);
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own1) =
          make_float4(x[2 * s + c][4], x[2 * s + c][5], x[2 * s + c][6], x[2 * s + c][7]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 z = *reinterpret_cast<const float4*>(S + c * 64 * TL::LW + lidx[k]);
        float4 o;
        if (q.accumulate) {
          o = old[2 * s + c][k];
          o.x += z.x; o.y += z.y; o.z += z.z; o.w += z.w;
        } else {
          o = make_float4(0.0f + z.x, 0.0f + z.y, 0.0f + z.z, 0.0f + z.w);
        }
        if (q.divisor > 1) o = cf_divide4(o, dv, dv_pow2);
        cf_st4<HALF>(out + base + (2 * s + c) * V, off[k], off2[k], o);
      }
    __syncthreads();
  }
}

template <int R, int SOLVER, int WPB, bool HALF, typename AT>
// (second launch-bounds argument = minimum WAVES PER SIMD in HIP, not workgroups per CU: 8 capped the one-patch variant
// at 64 VGPRs — 170 spilled registers)
// (rank 2 needs more than the 128 registers of four waves per SIMD: two — NO variant may spill to scratch, see nmf_pcf.hip)
__global__ __launch_bounds__(WPB * 64, (WPB == 8 || R >= 2) ? 2 : 4) void nmf_cf_fwd_tile_kernel(const AT* __restrict__ t,
                                                                   const float* __restrict__ u0,
                                                                   const float* __restrict__ v0,
                                                                   AT* __restrict__ out, CfGeom q, int T, float eps,
                                                                   int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_tile[];
  cf_fwd_tile_body<R, SOLVER, WPB, HALF, AT>(t, u0, v0, out, q, cf_tile_id<WPB>(q, cf_logical_block(xcd_remap)), T, eps, fz_lds_tile);
}

// backward: gY = gather_w(ga) / W ; gt (+)= [t > 0] ∘ scatter_w(gX)
template <int R, int SOLVER, typename AT>
// (launched with 64 .. 256 threads; everything but the hot HALS rank-1 form runs one wave per SIMD rather than spill)
__global__ __launch_bounds__(256, (R == 1 && SOLVER == 1) ? 2 : 1) void nmf_cf_bwd_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                            const float* __restrict__ v0,
                                                            const AT* __restrict__ ga, AT* __restrict__ gt,
                                                            CfGeom q, int64_t nmat, int T, int G, float eps,
                                                            int relu_gate, int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cf[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t mat = cf_logical_block(xcd_remap) * (blockDim.x >> 6) + wave;
  if (mat >= nmat) return;
  CfWave w{lane};
  CfAddr a;
  cf_decode(q, mat, lane, a);
  Hist<8, 8, R> h;
  h.carve(fz_lds_cf + wave * Hist<8, 8, R>::floats(G), G);
  float x[8][8], g[8][8];
  cf_load(t, a, x);
  cf_load(ga, a, g);
  cf_divide(g, q.gscale_div);
  nmf_backward_wave<8, 8, R, SOLVER>(w, u0, v0, x, g, h, 8, T, G, eps, nullptr, nullptr);
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      AT* p = gt + a.base + dd * a.V + a.off[jp];
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = g[dd][jp * 4 + e];
        r[e] = (!relu_gate || x[dd][jp * 4 + e] > 0.f) ? gv : 0.f;
      }
      if (q.accumulate) {
        const float4 o = ld4(p);
        r[0] += o.x; r[1] += o.y; r[2] += o.z; r[3] += o.w;
      }
      st4(p, make_float4(r[0], r[1], r[2], r[3]));
    }
}

// line-coalesced backward: same exchange for t and for the incoming gradient, ReLU gate applied on
// the owner side before the exchange back, read-modify-write of gt with the coalesced map
template <int R, int SOLVER, int WPB, bool HALF, typename AT>
__device__ __forceinline__ void cf_bwd_tile_body(const AT* __restrict__ t, const float* __restrict__ u0,
                                                 const float* __restrict__ v0, const AT* __restrict__ ga,
                                                 AT* __restrict__ gt, const CfGeom& q, const CfTileId& id, int T, int G, float eps,
                                                 int relu_gate, float* S) {
  using TL = CfTile<WPB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t base, V;
  unsigned off[2], off2[2];
  int lidx[2];
  cf_tile_decode<WPB>(q, id, tid, base, V, off, lidx, off2);
  const int own0 = cf_owner_lidx<WPB>(lane, wave, 0), own1 = cf_owner_lidx<WPB>(lane, wave, 1);
  Hist<8, 8, R> h;
  h.carve(S + TL::STAGE_FLOATS + wave * Hist<8, 8, R>::floats(G), G);

  float x[8][8], g[8][8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(t + base + dd * V, off[k], off2[k]);
      x[dd][k * 4 + 0] = v.x; x[dd][k * 4 + 1] = v.y; x[dd][k * 4 + 2] = v.z; x[dd][k * 4 + 3] = v.w;
    }
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 v = cf_ld4<HALF>(ga + base + dd * V, off[k], off2[k]);
      g[dd][k * 4 + 0] = v.x; g[dd][k * 4 + 1] = v.y; g[dd][k * 4 + 2] = v.z; g[dd][k * 4 + 3] = v.w;
    }
  cf_to_owner<WPB>(S, lidx, own0, own1, x);
  cf_to_owner<WPB>(S, lidx, own0, own1, g);
  cf_divide(g, q.gscale_div);
  CfWave w{lane};
  nmf_backward_wave<8, 8, R, SOLVER>(w, u0, v0, x, g, h, 8, T, G, eps, nullptr, nullptr);

  // gate first (frees x), then ALL loads of the running sum at once: one exposed round trip, not four
  if (relu_gate) {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd)
#pragma unroll
      for (int e = 0; e < 8; ++e) g[dd][e] = x[dd][e] > 0.f ? g[dd][e] : 0.f;
  }
  // (the plane base back in a scalar register pair: after the wave program the compiler otherwise carries it in vector
  // registers and every epilogue access pays a 64-bit vector address)
  asm volatile("" : "+s"(base));
  float4 old[8][2];
  if (q.accumulate) {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd)
#pragma unroll
      for (int k = 0; k < 2; ++k) old[dd][k] = cf_ld4<HALF>(gt + base + dd * V, off[k], off2[k]);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own0) =
          make_float4(g[2 * s + c][0], g[2 * s + c][1], g[2 * s + c][2], g[2 * s + c][3]);
      *reinterpret_cast<float4*>(S + c * 64 * TL::LW + own1) =
          make_float4(g[2 * s + c][4], g[2 * s + c][5], g[2 * s + c][6], g[2 * s + c][7]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float4 z = *reinterpret_cast<const float4*>(S + c * 64 * TL::LW + lidx[k]);
        if (q.accumulate) {
          const float4 o = old[2 * s + c][k];
          z.x += o.x; z.y += o.y; z.z += o.z; z.w += o.w;
        }
        cf_st4<HALF>(gt + base + (2 * s + c) * V, off[k], off2[k], z);
      }
    __syncthreads();
  }
}

template <int R, int SOLVER, int WPB, bool HALF, typename AT>
__global__ __launch_bounds__(WPB * 64, (R == 1 && SOLVER == 1 && (WPB == 4 || WPB == 1)) ? 2 : 1) void nmf_cf_bwd_tile_kernel(
    const AT* __restrict__ t, const float* __restrict__ u0, const float* __restrict__ v0,
    const AT* __restrict__ ga, AT* __restrict__ gt, CfGeom q, int T, int G, float eps, int relu_gate,
    int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float fz_lds_cf[];
  cf_bwd_tile_body<R, SOLVER, WPB, HALF, AT>(t, u0, v0, ga, gt, q, cf_tile_id<WPB>(q, cf_logical_block(xcd_remap)), T, G, eps,
                                             relu_gate, fz_lds_cf);
}


static int cf_geom(CfGeom& q, int B, int C, int D, int H, int W, const int* shift, int accumulate, int divisor) {
  if (B < 0 || C < 8 || (C % 8) || D < 8 || H < 8 || W < 8 || (D % 8) || (H % 8) || (W % 8))
    return fail(FZ_E_SHAPE, "fz_nmf_cf: needs C % 8 == 0 and spatial dims multiples of 8");
  if (!shift) return fail(FZ_E_ARG, "fz_nmf_cf: shift is null");
  if ((int64_t)D * H * W >= ((int64_t)1 << 30)) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: 2^30 or more voxels per channel plane");   // 32-bit lane offsets
  q.B = B; q.C = C; q.D = D; q.H = H; q.W = W; q.h = C / 8; q.G0 = D / 8; q.G1 = H / 8; q.G2 = W / 8;
  int s[3];
  const int S[3] = {D, H, W};
  for (int i = 0; i < 3; ++i) { s[i] = shift[i] % S[i]; if (s[i] < 0) s[i] += S[i]; }
  if (s[2] % 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: W-axis shift must be even");
  q.s0 = s[0]; q.s1 = s[1]; q.s2 = s[2];
  q.accumulate = accumulate; q.divisor = divisor; q.gscale_div = 1.0f;
  q.plane = (int64_t)D * H * W;
#ifdef FZ_PROBE_PLANE_PAD   // timing probe (tools/probes/gram_floor.sh): channel planes FZ_PROBE_PLANE_PAD elements further apart
  q.plane += FZ_PROBE_PLANE_PAD;
#endif
  return FZ_OK;
}

}  // namespace fz

using namespace fz;

extern "C" int fz_nmf_cf_supported(int C, int D, int H, int W, int d, int pd, int ph, int pw, int R, int T, int Tgrad) {
  if (d != 8 || pd != 8 || ph != 8 || pw != 8 || (C % 8) || (D % 8) || (H % 8) || (W % 8)) return 0;
  if (R < 1 || R > 2 || T < 0 || (int64_t)D * H * W >= ((int64_t)1 << 30)) return 0;
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int per_wave = ((G + 1) * R * 8 * 64 + (G + 1) * 8 * R + G * (8 * R + R * R)) * 4;
  return per_wave <= 160 * 1024 ? 1 : 0;
}

template <typename AT>
static int cf_fwd_launch(const AT* t, const float* u0, const float* v0, AT* out, int B, int C, int D,
                         int H, int W, const int* shift, int accumulate, int divisor, int R, int T,
                         int solver, float eps, fz_stream_t stream) {
  CfGeom q;
  int rc = cf_geom(q, B, C, D, H, W, shift, accumulate, divisor);
  if (rc != FZ_OK) return rc;
  if (!t || !u0 || !v0 || !out) return fail(FZ_E_ARG, "fz_nmf_cf_fwd: null pointer");
  if (R < 1 || R > 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_fwd: rank 1..2");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_cf_fwd: bad solver");
  if (B == 0) return FZ_OK;
  const int64_t nmat = (int64_t)B * q.h * q.G0 * q.G1 * q.G2;
  if (nmat >= (int64_t)1 << 31) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: more than 2^31 matrices");
  // measured on MI355X (round-1/2 probe `cf_probe`): a workgroup = one full row of patches along W
  // (up to 16 waves) consumes whole 128-B lines inside one CU: 0.856 -> 0.725 ms at stage 0
  int wpb = 16;
  if (wpb > q.G2) wpb = q.G2;
  if (wpb < 1) wpb = 1;
  const int xr = 1 | (tile_order() << 1);
  hipStream_t st = (hipStream_t)stream;
  const int tile = FZ_KNOB("FZ_CF_TILE").set ? FZ_KNOB("FZ_CF_TILE").val : 1;   // probe builds: 0 = the direct-gather kernels
  const bool half = (q.s2 % 4) != 0;  // W-axis shift ≡ 2 (mod 4): only the line-coalesced kernels handle it
  if (half || (tile && (q.G2 % 8) == 0)) {
    // line-coalesced kernel: WPB patches along W per workgroup.  8 patches per workgroup, two
    // workgroups per CU out of phase: 0.524 ms vs 0.554 (16) vs 0.747 (direct gather) at stage 0
    const int twpb = (q.G2 % 8) == 0 ? 8 : ((q.G2 % 4) == 0 ? 4 : 1);
    const unsigned nblk = (unsigned)(nmat / twpb);
#define FZ_CF_TILE(RR, SS, WW, HH)                                                                          \
  do {                                                                                                      \
    auto kern = nmf_cf_fwd_tile_kernel<RR, SS, WW, HH, AT>;                                                   \
    const int lds = CfTile<WW>::STAGE_FLOATS * (int)sizeof(float);                                          \
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WW), lds, st, t, u0, v0, out, q, T, eps, xr);            \
  } while (0)
#define FZ_CF_TILE_W(RR, SS)                                                                                \
  do {                                                                                                      \
    if (half) { if (twpb == 8) FZ_CF_TILE(RR, SS, 8, true); else if (twpb == 4) FZ_CF_TILE(RR, SS, 4, true); else FZ_CF_TILE(RR, SS, 1, true); } \
    else FZ_CF_TILE(RR, SS, 8, false);                                                                      \
  } while (0)
    if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_TILE_W(1, SOLVER_MU); else FZ_CF_TILE_W(1, SOLVER_HALS); }
    else { if (solver == FZ_SOLVER_MU) FZ_CF_TILE_W(2, SOLVER_MU); else FZ_CF_TILE_W(2, SOLVER_HALS); }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
  }
  dim3 grid((unsigned)((nmat + wpb - 1) / wpb)), block(64 * wpb);
#define FZ_CF_FWD(RR, SS) hipLaunchKernelGGL((nmf_cf_fwd_kernel<RR, SS, AT>), grid, block, 0, st, t, u0, v0, out, q, nmat, T, eps, xr)
  if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_FWD(1, SOLVER_MU); else FZ_CF_FWD(1, SOLVER_HALS); }
  else { if (solver == FZ_SOLVER_MU) FZ_CF_FWD(2, SOLVER_MU); else FZ_CF_FWD(2, SOLVER_HALS); }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_nmf_cf_fwd(const void* t, const float* u0, const float* v0, void* out, int B, int C, int D,
                             int H, int W, const int* shift, int accumulate, int divisor, int R, int T,
                             int solver, float eps, int act_dtype, fz_stream_t stream) {
  if (act_dtype == FZ_STORE_F32)
    return cf_fwd_launch<float>((const float*)t, u0, v0, (float*)out, B, C, D, H, W, shift, accumulate, divisor, R, T,
                                solver, eps, stream);
  if (act_dtype == FZ_STORE_BF16)
    return cf_fwd_launch<bf16>((const bf16*)t, u0, v0, (bf16*)out, B, C, D, H, W, shift, accumulate, divisor, R, T,
                               solver, eps, stream);
  return fail(FZ_E_ARG, "fz_nmf_cf_fwd: bad act_dtype");
}

template <typename AT>
static int cf_bwd_launch(const AT* t, const float* u0, const float* v0, const AT* ga, AT* gt, int B,
                         int C, int D, int H, int W, const int* shift, int accumulate, int nshift,
                         int relu_gate, int R, int T, int Tgrad, int solver, float eps, fz_stream_t stream) {
  CfGeom q;
  int rc = cf_geom(q, B, C, D, H, W, shift, accumulate, 1);
  if (rc != FZ_OK) return rc;
  if (!t || !u0 || !v0 || !ga || !gt) return fail(FZ_E_ARG, "fz_nmf_cf_bwd: null pointer");
  if (R < 1 || R > 2) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd: rank 1..2");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fail(FZ_E_ARG, "fz_nmf_cf_bwd: bad solver");
  if (B == 0) return FZ_OK;
  q.gscale_div = (float)(nshift > 1 ? nshift : 1);
  const int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  const int64_t nmat = (int64_t)B * q.h * q.G0 * q.G1 * q.G2;
  if (nmat >= (int64_t)1 << 31) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf: more than 2^31 matrices");
  const int per_wave = (R == 1 ? Hist<8, 8, 1>::floats(G) : Hist<8, 8, 2>::floats(G)) * (int)sizeof(float);
  if (per_wave > 160 * 1024) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd: history exceeds LDS");
  int wpb = 65536 / per_wave;
  if (wpb > 4) wpb = 4;
  if (wpb > 8) wpb = 8;
  if (wpb < 1) wpb = 1;
  // patch neighbours on the same XCD share its L2 (shifted windows straddle lines): 1.31 -> 1.17 ms
  const int xr = 1 | (tile_order() << 1);
  hipStream_t st = (hipStream_t)stream;
  const int tile = FZ_KNOB("FZ_CF_TILE_BWD").set ? FZ_KNOB("FZ_CF_TILE_BWD").val : 1;
  const bool half = (q.s2 % 4) != 0;
  if (half || (tile && (q.G2 % 4) == 0)) {
    const int twpb = (q.G2 % 4) == 0 ? 4 : 1;
    // HALS rank 1 behind a ReLU (t >= 0 by the relu_gate contract): the row-space reverse mode, no per-column history
    const bool gram_on = !(FZ_KNOB("FZ_CF_GRAM").set && FZ_KNOB("FZ_CF_GRAM").val == 0);   // probe builds: 0 = the general kernel
    // (measured, tools/probes/gram_floor.sh: both kernels sit on the tile's memory skeleton; the row-space one is 6-11 % faster
    //  everywhere except fp32 windows w > 0 of >= 2^15 matrices, where the general kernel's single late burst of
    //  running-sum reads is 3-5 % ahead: 461-464 against 478-487 us at the README's stage 0)
    const bool gram_wins = !(sizeof(AT) == 4 && accumulate && nmat >= 32768);
    if (gram_on && gram_wins && R == 1 && solver == FZ_SOLVER_HALS && relu_gate && G >= 1) {
      rc = cf_bwd_gram_launch<AT>(t, v0, ga, gt, q, nmat, T, G, eps, xr, st);
      if (rc != FZ_E_UNSUPPORTED) return rc;
    }
    const int tlds = (twpb == 4 ? CfTile<4>::STAGE_FLOATS : CfTile<1>::STAGE_FLOATS) * (int)sizeof(float) + per_wave * twpb;
    if (tlds > 160 * 1024) {
      if (half) return fail(FZ_E_UNSUPPORTED, "fz_nmf_cf_bwd: history exceeds LDS for a W-axis shift of 2 (mod 4)");
    } else {
      const unsigned nblk = (unsigned)(nmat / twpb);
#define FZ_CF_BWD_TILE(RR, SS, WW, HH)                                                                      \
  do {                                                                                                      \
    auto kern = nmf_cf_bwd_tile_kernel<RR, SS, WW, HH, AT>;                                                    \
    if (tlds > 65536)                                                                                       \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, tlds));                     \
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WW), tlds, st, t, u0, v0, ga, gt, q, T, G, eps, relu_gate, xr); \
  } while (0)
#define FZ_CF_BWD_TILE_W(RR, SS)                                                                            \
  do {                                                                                                      \
    if (half) { if (twpb == 4) FZ_CF_BWD_TILE(RR, SS, 4, true); else FZ_CF_BWD_TILE(RR, SS, 1, true); }     \
    else FZ_CF_BWD_TILE(RR, SS, 4, false);                                                                  \
  } while (0)
      if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_BWD_TILE_W(1, SOLVER_MU); else FZ_CF_BWD_TILE_W(1, SOLVER_HALS); }
      else { if (solver == FZ_SOLVER_MU) FZ_CF_BWD_TILE_W(2, SOLVER_MU); else FZ_CF_BWD_TILE_W(2, SOLVER_HALS); }
      FZ_LAUNCH_CHECK();
      return FZ_OK;
    }
  }
  const int lds = per_wave * wpb;
  dim3 grid((unsigned)((nmat + wpb - 1) / wpb)), block(64 * wpb);
#define FZ_CF_BWD(RR, SS)                                                                                 \
  do {                                                                                                    \
    auto kern = nmf_cf_bwd_kernel<RR, SS, AT>;                                                          \
    if (lds > 65536)                                                                                      \
      FZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                    \
    hipLaunchKernelGGL(kern, grid, block, lds, st, t, u0, v0, ga, gt, q, nmat, T, G, eps, relu_gate, xr);     \
  } while (0)
  if (R == 1) { if (solver == FZ_SOLVER_MU) FZ_CF_BWD(1, SOLVER_MU); else FZ_CF_BWD(1, SOLVER_HALS); }
  else { if (solver == FZ_SOLVER_MU) FZ_CF_BWD(2, SOLVER_MU); else FZ_CF_BWD(2, SOLVER_HALS); }
  FZ_LAUNCH_CHECK();
  return FZ_OK;
}

extern "C" int fz_nmf_cf_bwd(const void* t, const float* u0, const float* v0, const void* ga, void* gt, int B,
                             int C, int D, int H, int W, const int* shift, int accumulate, int nshift,
                             int relu_gate, int R, int T, int Tgrad, int solver, float eps, int act_dtype,
                             fz_stream_t stream) {
  if (act_dtype == FZ_STORE_F32)
    return cf_bwd_launch<float>((const float*)t, u0, v0, (const float*)ga, (float*)gt, B, C, D, H, W, shift, accumulate,
                                nshift, relu_gate, R, T, Tgrad, solver, eps, stream);
  if (act_dtype == FZ_STORE_BF16)
    return cf_bwd_launch<bf16>((const bf16*)t, u0, v0, (const bf16*)ga, (bf16*)gt, B, C, D, H, W, shift, accumulate,
                               nshift, relu_gate, R, T, Tgrad, solver, eps, stream);
  return fail(FZ_E_ARG, "fz_nmf_cf_bwd: bad act_dtype");
}
