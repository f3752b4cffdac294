// nmf_cf.h — declarations shared by the translation units of the fused FactMixer core (nmf_cf.hip, nmf_cf_gram.hip):
// geometry, the per-wave policy of the NMF programs, the line-coalesced tile map and its LDS exchange.
#pragma once
#include <cstdlib>

#include "fz_common.h"
#include "nmf_core.h"
#include "nmf_gram.h"

namespace fz {

struct CfGeom {
  int B, C, D, H, W;   // channels-first tensor
  int h;               // heads (C / 8)
  int G0, G1, G2;      // patch grid (D/8, H/8, W/8)
  int s0, s1, s2;      // this window's shift, normalised to [0, S)
  int accumulate;      // add to the existing output (windows > 0)
  int divisor;         // > 1: divide the result by it (last window, forward)
  float gscale_div;    // backward: gY = gather(ga) / gscale_div
  int64_t plane;       // distance between channel planes in elements (D·H·W for dense tensors)
};

struct CfWave {
  using F = float;
  int lane;
  __device__ __forceinline__ int col(int j) const {
    const int jp = j >> 2, e = j & 3;
    return (((jp * 4 + (lane >> 4)) * 8 + ((lane >> 1) & 7)) * 8) + (lane & 1) * 4 + e;
  }
  __device__ __forceinline__ float sum(float v) const { return wave_sum(v); }
  __device__ __forceinline__ void sum8(float (&v)[8]) const { wave_sum8(v, lane); }
  __device__ __forceinline__ void st_priv(float* base, int idx, float v) const { base[idx * 64 + lane] = v; }
  __device__ __forceinline__ float ld_priv(const float* base, int idx) const { return base[idx * 64 + lane]; }
  __device__ __forceinline__ void st_uni(float* base, int idx, float v) const {
    if (lane == 0) base[idx] = v;
  }
  __device__ __forceinline__ float ld_uni(const float* base, int idx) const { return base[idx]; }
  __device__ __forceinline__ float ld_uni_global(const float* p, int idx) const { return p[idx]; }
  __device__ __forceinline__ float ld_v0(const float* v0, int j, int r, int R) const { return v0[col(j) * R + r]; }
  __device__ __forceinline__ float keep_col(int, float v) const { return v; }
  __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  // factor rows distributed over the eight 8-lane groups (nmf_core.h DistRows)
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
  // row-space reverse mode (nmf_gram.h)
  __device__ __forceinline__ float mask_rows(float d, int mreal) const { return (lane >> 3) < mreal ? d : 0.f; }
  __device__ __forceinline__ float pick_call(int c, float tot, float mine) const { return (lane & 7) == c ? tot : mine; }
  // lane (g, i) holds K[g][(g + i) & 7]  ->  Kd[k] = K[g][k] for every lane of group g, through 256 B of the wave's LDS
  __device__ __forceinline__ void k_unrotate(float* scratch, float mine, float (&Kd)[8]) const {
    const int g = lane >> 3;
    scratch[g * 8 + ((g + lane) & 7)] = mine;
    fence();
    const float4 a = *reinterpret_cast<const float4*>(scratch + g * 8), b = *reinterpret_cast<const float4*>(scratch + g * 8 + 4);
    Kd[0] = a.x; Kd[1] = a.y; Kd[2] = a.z; Kd[3] = a.w; Kd[4] = b.x; Kd[5] = b.y; Kd[6] = b.z; Kd[7] = b.w;
    fence();
  }
};

// element offsets (inside one channel plane) of this lane's two vectors, and the plane base
struct CfAddr {
  int64_t base;     // (b*C + hh*8) * V
  int64_t off[2];   // jp = 0, 1
  int64_t V;
};

__device__ __forceinline__ bool cf_decode(const CfGeom& q, int64_t mat, int lane, CfAddr& a) {
  unsigned t = (unsigned)mat;  // the host rejects > 2^31 matrices
  const int g2 = (int)(t % (unsigned)q.G2); t /= (unsigned)q.G2;
  const int g1 = (int)(t % (unsigned)q.G1); t /= (unsigned)q.G1;
  const int g0 = (int)(t % (unsigned)q.G0); t /= (unsigned)q.G0;
  const int hh = (int)(t % (unsigned)q.h);
  const int b = (int)(t / (unsigned)q.h);
  const int p1 = (lane >> 1) & 7, half = lane & 1;
  int z1 = g1 * 8 + p1 - q.s1; if (z1 < 0) z1 += q.H;
  int z2 = g2 * 8 + half * 4 - q.s2; if (z2 < 0) z2 += q.W;
  a.V = q.plane;
  a.base = ((int64_t)b * q.C + (int64_t)hh * 8) * a.V;
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    int z0 = g0 * 8 + jp * 4 + (lane >> 4) - q.s0; if (z0 < 0) z0 += q.D;
    a.off[jp] = ((int64_t)z0 * q.H + z1) * q.W + z2;
  }
  return true;
}

// AT = storage type of the channels-first tensors t / out / ga / gt (float or bf16); the wave program
// and the window accumulation run in fp32 either way
template <typename AT>
__device__ __forceinline__ void cf_load(const AT* __restrict__ t, const CfAddr& a, float (&x)[8][8]) {
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const float4 v = ld4(t + a.base + dd * a.V + a.off[jp]);
      x[dd][jp * 4 + 0] = v.x; x[dd][jp * 4 + 1] = v.y; x[dd][jp * 4 + 2] = v.z; x[dd][jp * 4 + 3] = v.w;
    }
}

// x / dv for the window average.  A power-of-two divisor (2 or 4 windows — the usual case) is an
// exact scaling, so the multiply is bit-identical to the division and 10x cheaper (IEEE fp32
// division is ~11 VALU instructions); other divisors keep the true division.
__device__ __forceinline__ bool cf_pow2(float dv) { return (__float_as_uint(dv) & 0x007fffffu) == 0u && dv > 0.f; }

__device__ __forceinline__ void cf_divide(float (&g)[8][8], float dv) {
  if (dv == 1.0f) return;
  if (cf_pow2(dv)) {
    const float inv = 1.0f / dv;
#pragma unroll
    for (int dd = 0; dd < 8; ++dd)
#pragma unroll
      for (int j = 0; j < 8; ++j) g[dd][j] = g[dd][j] * inv;
  } else {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd)
#pragma unroll
      for (int j = 0; j < 8; ++j) g[dd][j] = g[dd][j] / dv;
  }
}

__device__ __forceinline__ float4 cf_divide4(float4 o, float dv, bool pow2) {
  if (pow2) {
    const float inv = 1.0f / dv;
    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
  } else {
    o.x /= dv; o.y /= dv; o.z /= dv; o.w /= dv;
  }
  return o;
}

// logical workgroup id: optionally remapped so that consecutive workgroups (patch neighbours
// along W, then H) run on the same XCD and share its L2 (blocks are dealt round-robin over 8 XCDs)
// (bit 1 of xcd_remap: walk the tiles in DESCENDING order — the tail of a tensor the previous launch wrote in ascending order
//  is what the 256 MiB Infinity Cache still holds when this launch starts)
__device__ __forceinline__ int64_t cf_logical_block(int xcd_remap) {
  const int64_t nb = gridDim.x;
  const int64_t bid = (xcd_remap & 2) ? nb - 1 - (int64_t)blockIdx.x : (int64_t)blockIdx.x;
  if (!(xcd_remap & 1) || nb < 16) return bid;
  const int64_t q = nb / 8, r = nb % 8, xcd = bid % 8, i = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <int R, int SOLVER, typename AT>
__global__ __launch_bounds__(1024) void nmf_cf_fwd_kernel(const AT* __restrict__ t, const float* __restrict__ u0,
                                                          const float* __restrict__ v0, AT* __restrict__ out,
                                                          CfGeom q, int64_t nmat, int T, float eps, int xcd_remap) {
  const int lane = threadIdx.x & 63;
  const int64_t mat = cf_logical_block(xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (mat >= nmat) return;
  CfWave w{lane};
  CfAddr a;
  cf_decode(q, mat, lane, a);
  float x[8][8], u[8][R], v[8][R];
  cf_load(t, a, x);
  nmf_forward_wave<8, 8, R, SOLVER>(w, u0, v0, x, u, v, 8, T, eps);
  const float dv = (float)q.divisor;
  const bool dv_pow2 = cf_pow2(dv);
#pragma unroll
  for (int dd = 0; dd < 8; ++dd)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      AT* p = out + a.base + dd * a.V + a.off[jp];
      float4 o;
      if (q.accumulate) {
        o = ld4(p);
        o.x += x[dd][jp * 4 + 0]; o.y += x[dd][jp * 4 + 1]; o.z += x[dd][jp * 4 + 2]; o.w += x[dd][jp * 4 + 3];
      } else {
        o = make_float4(0.0f + x[dd][jp * 4 + 0], 0.0f + x[dd][jp * 4 + 1], 0.0f + x[dd][jp * 4 + 2],
                        0.0f + x[dd][jp * 4 + 3]);
      }
      if (q.divisor > 1) o = cf_divide4(o, dv, dv_pow2);
      st4(p, o);
    }
}

// ---- line-coalesced variant -----------------------------------------------------------------------
// A workgroup owns WPB patches that are neighbours along W, i.e. for every (channel, p0, p1) one
// contiguous run of WPB·8 floats.  Global memory is touched only with the COALESCED map
//   thread → (row = (p0, p1), 16-byte chunk of the run)      [whole 128-B lines per request]
// and the patch-owner map of CfWave is reached through an LDS exchange, two channels per stage
// (the 64 data registers are reused in place).  The direct kernel above touches 32 lines per load
// instruction and uses 32 B of each; this one touches 1/4 as many, fully.
//
// LDS image of one channel: 64 rows (p0, p1) of CHUNKS 16-byte chunks.  Four access patterns touch it — coalesced-map
// writes and owner-map reads on the way in, owner-map writes and coalesced-map reads on the way out — and the hardware
// services them in different lane groups (MI355X_MICROARCH.md §LDS: ds_write_b128 in 8 contiguous lanes on 32 banks,
// ds_read_b128 in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... on 64 banks).  A padded row
// (WPB·8 + 8 floats, rounds 1-2) is conflict-free for three of them but 2-way for the coalesced-map reads (the group's
// lanes 20-27 sit in the next row, 8 banks on: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.10-0.14, profiles/r03_pmc_sq.md).
// WPB = 8 and 4 use an unpadded row with the chunk index XOR-swizzled by the row instead, conflict-free for all four:
//   WPB = 8 (one row = the 64 banks):  chunk ^ 2·(row & 7)        — owner reads: the 8 rows of a group get 8 distinct
//                                       chunk pairs; coalesced reads: rows R, R+1 keep complementary chunk classes
//   WPB = 4 (two rows = the 64 banks): chunk ^ 2·v(row & 7), v = (0, 2, 1, 3, 2, 0, 3, 1) — distinct over rows {0,2,4,6},
//                                       {1,3,5,7} (owner reads), {0..3}, {4..7} (owner writes), bit 1 equal over row
//                                       pairs (r, r+2) (coalesced reads)
template <int WPB>
struct CfTile {
  static constexpr bool SWZ = WPB == 8 || WPB == 4;
  static constexpr int LW = SWZ ? WPB * 8 : WPB * 8 + 8;
  static constexpr int NT = WPB * 64;
  static constexpr int CHUNKS = WPB * 2;  // 16-byte chunks per row
  static constexpr int STAGE_FLOATS = 2 * 64 * LW;
  // float index of 16-byte chunk `chunk` of row `row`
  static __device__ __forceinline__ int at(int row, int chunk) {
    if (WPB == 8) return row * LW + ((chunk ^ (2 * (row & 7))) << 2);
    if (WPB == 4) return row * LW + ((chunk ^ (2 * (((row >> 1) & 1) | ((((row & 1) ^ (row >> 2)) & 1) << 1)))) << 2);
    return row * LW + chunk * 4;
  }
};

// HALF (W-axis shift ≡ 2 mod 4, e.g. the production windows [None, 2, 4, 6]): a 16-byte chunk of the
// shifted run starts 8 bytes into an aligned quad of the tensor and may straddle the cyclic wrap, so
// it moves as two 8-byte halves with separately wrapped addresses (off2 = offset of voxels +2, +3).
// (p is wave-uniform — a channel plane of the workgroup's (sample, head) — and o a 32-bit element offset: the access
// compiles to `global_load_dwordx4 v, v_off, s[p:p+1]`, no 64-bit address arithmetic per lane and access)
template <typename AT>
__device__ __forceinline__ const AT* cf_at(const AT* p, unsigned o) {
  return reinterpret_cast<const AT*>(reinterpret_cast<const char*>(p) + o * (unsigned)sizeof(AT));
}
template <typename AT>
__device__ __forceinline__ AT* cf_at(AT* p, unsigned o) {
  return reinterpret_cast<AT*>(reinterpret_cast<char*>(p) + o * (unsigned)sizeof(AT));
}
template <bool HALF, typename AT>
__device__ __forceinline__ float4 cf_ld4(const AT* p, unsigned o, unsigned o2) {
  if (HALF) {
    float a[2], b[2];
    aload<2>(cf_at(p, o), a);
    aload<2>(cf_at(p, o2), b);
    return make_float4(a[0], a[1], b[0], b[1]);
  }
  return ld4(cf_at(p, o));
}
template <bool HALF, typename AT>
__device__ __forceinline__ void cf_st4(AT* p, unsigned o, unsigned o2, float4 v) {
  if (HALF) {
    const float a[2] = {v.x, v.y}, b[2] = {v.z, v.w};
    astore<2>(cf_at(p, o), a);
    astore<2>(cf_at(p, o2), b);
  } else {
    st4(cf_at(p, o), v);
  }
}

struct CfTileId { int b, hh, g0, g1, gq; };   // sample, head, patch plane, patch row, tile (WPB patches) along W

template <int WPB>
__device__ __forceinline__ CfTileId cf_tile_id(const CfGeom& q, int64_t blk) {
  // 32-bit index arithmetic (the host rejects > 2^31 matrices): 64-bit div/mod is ~100 instructions each
  const unsigned ngrp = (unsigned)(q.G2 / WPB);
  unsigned t = (unsigned)blk;
  CfTileId id;
  id.gq = (int)(t % ngrp); t /= ngrp;
  id.g1 = (int)(t % (unsigned)q.G1); t /= (unsigned)q.G1;
  id.g0 = (int)(t % (unsigned)q.G0); t /= (unsigned)q.G0;
  id.hh = (int)(t % (unsigned)q.h);
  id.b = (int)(t / (unsigned)q.h);
  return id;
}

template <int WPB>
__device__ __forceinline__ void cf_tile_decode(const CfGeom& q, const CfTileId& id, int tid, int64_t& base, int64_t& V,
                                               unsigned (&off)[2], int (&lidx)[2], unsigned (&off2)[2]) {
  using TL = CfTile<WPB>;
  V = q.plane;
  base = ((int64_t)id.b * q.C + (int64_t)id.hh * 8) * V;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * TL::NT;
    const int row = idx / TL::CHUNKS, chunk = idx % TL::CHUNKS;
    int z0 = id.g0 * 8 + (row >> 3) - q.s0; if (z0 < 0) z0 += q.D;
    int z1 = id.g1 * 8 + (row & 7) - q.s1; if (z1 < 0) z1 += q.H;
    int z2 = id.gq * WPB * 8 + chunk * 4 - q.s2; if (z2 < 0) z2 += q.W;
    off[k] = (unsigned)((z0 * q.H + z1) * q.W + z2);   // in-plane element offset < 2^30 (host-checked): 32 bits, so that every
    int z2b = z2 + 2; if (z2b >= q.W) z2b -= q.W;
    off2[k] = (unsigned)((z0 * q.H + z1) * q.W + z2b); // access is `uniform plane pointer (SGPR pair) + one 32-bit lane offset`
    lidx[k] = TL::at(row, chunk);
  }
}

// owner-side LDS index of local vector jp of this lane (patch = wave)
template <int WPB>
__device__ __forceinline__ int cf_owner_lidx(int lane, int wave, int jp) {
  return CfTile<WPB>::at((jp * 4 + (lane >> 4)) * 8 + ((lane >> 1) & 7), wave * 2 + (lane & 1));
}

// exchange the 64 data registers from the coalesced map to the patch-owner map, in place
template <int WPB>
__device__ __forceinline__ void cf_to_owner(float* S, const int (&lidx)[2], int own0, int own1, float (&x)[8][8]) {
  using TL = CfTile<WPB>;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        *reinterpret_cast<float4*>(S + c * 64 * TL::LW + lidx[k]) =
            make_float4(x[2 * s + c][k * 4 + 0], x[2 * s + c][k * 4 + 1], x[2 * s + c][k * 4 + 2], x[2 * s + c][k * 4 + 3]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float4 a0 = *reinterpret_cast<const float4*>(S + c * 64 * TL::LW + own0);
      const float4 a1 = *reinterpret_cast<const float4*>(S + c * 64 * TL::LW + own1);
      x[2 * s + c][0] = a0.x; x[2 * s + c][1] = a0.y; x[2 * s + c][2] = a0.z; x[2 * s + c][3] = a0.w;
      x[2 * s + c][4] = a1.x; x[2 * s + c][5] = a1.y; x[2 * s + c][6] = a1.z; x[2 * s + c][7] = a1.w;
    }
    __syncthreads();
  }
}


// row-space reverse mode for HALS rank 1 behind a ReLU (nmf_cf_gram.hip); returns FZ_E_UNSUPPORTED when the launch does
// not fit its LDS budget (the caller then takes the general kernels)
template <typename AT>
int cf_bwd_gram_launch(const AT* t, const float* v0, const AT* ga, AT* gt, const CfGeom& q, int64_t nmat, int T, int G,
                       float eps, int xcd_remap, hipStream_t st);

}  // namespace fz
