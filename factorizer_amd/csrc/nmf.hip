// nmf.hip — C-ABI entry points fz_nmf_fwd / fz_nmf_bwd (include/factorizer_hip.h); the
// kernels live in nmf_kernels.inc, one translation unit per rank.
#include "fz_common.h"

namespace fz {
struct NmfArgs {
  const void *x;            // matrices: storage type of the launcher
  const float *u0, *v0;
  const void *gy;
  const float *gu, *gv;
  void *y;
  float *uo, *vo;
  void *gx;
  int64_t nmat;
  int M, N, T, G, solver;
  float eps;
  hipStream_t stream;
};
int nmf_launch_fwd_r1(const NmfArgs&);
int nmf_launch_fwd_r2(const NmfArgs&);
int nmf_launch_fwd_r3(const NmfArgs&);
int nmf_launch_fwd_r4(const NmfArgs&);
int nmf_launch_bwd_r1(const NmfArgs&);
int nmf_launch_bwd_r2(const NmfArgs&);
int nmf_launch_bwd_r3(const NmfArgs&);
int nmf_launch_bwd_r4(const NmfArgs&);
int nmf_launch_fwd_r1_bf16(const NmfArgs&);
int nmf_launch_fwd_r2_bf16(const NmfArgs&);
int nmf_launch_fwd_r3_bf16(const NmfArgs&);
int nmf_launch_fwd_r4_bf16(const NmfArgs&);
int nmf_launch_bwd_r1_bf16(const NmfArgs&);
int nmf_launch_bwd_r2_bf16(const NmfArgs&);
int nmf_launch_bwd_r3_bf16(const NmfArgs&);
int nmf_launch_bwd_r4_bf16(const NmfArgs&);

static int hist_floats(int M, int N, int R, int G) {
  int MP, NPL;
  // same order as launch_shape (nmf_kernels.inc): the smallest register footprint that holds the matrix
  if (M <= 8 && N <= 64) { MP = 8; NPL = 1; }
  else if (M <= 8 && N <= 128) { MP = 8; NPL = 2; }
  else if (M <= 8 && N <= 192) { MP = 8; NPL = 3; }
  else if (M <= 8 && N <= 256) { MP = 8; NPL = 4; }
  else if (M <= 8 && N <= 512) { MP = 8; NPL = 8; }
  else if (M <= 16 && N <= 64) { MP = 16; NPL = 1; }
  else if (M <= 16 && N <= 256) { MP = 16; NPL = 4; }
  else if (M <= 32 && N <= 64) { MP = 32; NPL = 1; }
  else if (M <= 32 && N <= 128) { MP = 32; NPL = 2; }
  else return -1;
  return (G + 1) * R * NPL * 64 + (G + 1) * MP * R + G * (MP * R + R * R);
}
}  // namespace fz

static int check_common(int64_t nmat, int M, int N, int R, int T, int solver) {
  if (nmat < 0 || M < 1 || N < 1 || T < 0) return fz::fail(FZ_E_SHAPE, "fz_nmf: bad sizes");
  if (solver != FZ_SOLVER_MU && solver != FZ_SOLVER_HALS) return fz::fail(FZ_E_ARG, "fz_nmf: bad solver");
  if (R < 1 || R > 4) return fz::fail(FZ_E_UNSUPPORTED, "fz_nmf: rank outside 1..4");
  if (fz::hist_floats(M, N, R, 0) < 0) return fz::fail(FZ_E_UNSUPPORTED, "fz_nmf: (M,N) outside the native kernel families");
  return FZ_OK;
}

extern "C" int fz_nmf_supported(int M, int N, int R, int T, int Tgrad) {
  if (R < 1 || R > 4 || M < 1 || N < 1 || T < 0) return 0;
  int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  int f = fz::hist_floats(M, N, R, G);
  return (f > 0 && f * 4 <= 160 * 1024) ? 1 : 0;
}

extern "C" int fz_nmf_fwd(const void* x, const float* u0, const float* v0, void* y, float* u_out,
                          float* v_out, int64_t nmat, int M, int N, int R, int T, int solver, float eps,
                          int act_dtype, fz_stream_t stream) {
  int rc = check_common(nmat, M, N, R, T, solver);
  if (rc != FZ_OK) return rc;
  if (!x || !u0 || !v0 || !y) return fz::fail(FZ_E_ARG, "fz_nmf_fwd: null pointer");
  if (nmat == 0) return FZ_OK;
  fz::NmfArgs a{x, u0, v0, nullptr, nullptr, nullptr, y, u_out, v_out, nullptr, nmat, M, N, T, 0, solver, eps,
                (hipStream_t)stream};
  if (act_dtype == FZ_STORE_BF16) {
    switch (R) {
      case 1: return fz::nmf_launch_fwd_r1_bf16(a);
      case 2: return fz::nmf_launch_fwd_r2_bf16(a);
      case 3: return fz::nmf_launch_fwd_r3_bf16(a);
      default: return fz::nmf_launch_fwd_r4_bf16(a);
    }
  }
  if (act_dtype != FZ_STORE_F32) return fz::fail(FZ_E_ARG, "fz_nmf_fwd: bad act_dtype");
  switch (R) {
    case 1: return fz::nmf_launch_fwd_r1(a);
    case 2: return fz::nmf_launch_fwd_r2(a);
    case 3: return fz::nmf_launch_fwd_r3(a);
    default: return fz::nmf_launch_fwd_r4(a);
  }
}

extern "C" int fz_nmf_bwd(const void* x, const float* u0, const float* v0, const void* gy, const float* gu,
                          const float* gv, void* gx, int64_t nmat, int M, int N, int R, int T, int Tgrad,
                          int solver, float eps, int act_dtype, fz_stream_t stream) {
  int rc = check_common(nmat, M, N, R, T, solver);
  if (rc != FZ_OK) return rc;
  if (!x || !u0 || !v0 || !gx || (!gy && !gu && !gv)) return fz::fail(FZ_E_ARG, "fz_nmf_bwd: null pointer");
  if (nmat == 0) return FZ_OK;
  int G = Tgrad < 0 ? 0 : (Tgrad > T ? T : Tgrad);
  fz::NmfArgs a{x, u0, v0, gy, gu, gv, nullptr, nullptr, nullptr, gx, nmat, M, N, T, G, solver, eps,
                (hipStream_t)stream};
  if (act_dtype == FZ_STORE_BF16) {
    switch (R) {
      case 1: return fz::nmf_launch_bwd_r1_bf16(a);
      case 2: return fz::nmf_launch_bwd_r2_bf16(a);
      case 3: return fz::nmf_launch_bwd_r3_bf16(a);
      default: return fz::nmf_launch_bwd_r4_bf16(a);
    }
  }
  if (act_dtype != FZ_STORE_F32) return fz::fail(FZ_E_ARG, "fz_nmf_bwd: bad act_dtype");
  switch (R) {
    case 1: return fz::nmf_launch_bwd_r1(a);
    case 2: return fz::nmf_launch_bwd_r2(a);
    case 3: return fz::nmf_launch_bwd_r3(a);
    default: return fz::nmf_launch_bwd_r4(a);
  }
}
