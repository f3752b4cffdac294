/* factorizer_hip.h — C ABI of libfactorizer_hip.so (MI355X / gfx950).
 *
 * The reference (pashtari/factorizer) has no FFI of its own: the boundary its hot path sits
 * behind is the Python nn.Module API.  The entry points below are what the
 * host-side mirrors of those modules (factorizer_amd/*.py) bind through ctypes; each comment
 * names the reference code the entry point replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; pointers are DEVICE pointers unless marked host.
 *   - return 0 on success, <0 on error (FZ_E_*); never throw, never abort.
 *   - never allocate or free caller memory; workspaces are passed in.
 *   - asynchronous on the passed hipStream_t (void*); no host synchronisation inside.
 *   - re-entrant and thread-safe: no mutable globals besides a thread-local error string.
 *   - tensors are dense, contiguous, row-major ("channels-first": B,C,D,H,W).
 */
#ifndef FACTORIZER_HIP_H
#define FACTORIZER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FZ_OK 0
#define FZ_E_SHAPE (-1)       /* inconsistent / non-divisible shapes            */
#define FZ_E_UNSUPPORTED (-2) /* valid request outside the compiled kernel set  */
#define FZ_E_HIP (-3)         /* a HIP runtime call failed                      */
#define FZ_E_ARG (-4)         /* null pointer / bad enum                        */

#define FZ_SOLVER_MU 0   /* matrix_factorization.py:232-247 MultiplicativeUpdate        */
#define FZ_SOLVER_HALS 1 /* matrix_factorization.py:194-229 CoordinateDescent + ReLU    */

typedef void* fz_stream_t; /* hipStream_t */

/* Library version (major*10000 + minor*100 + patch). */
int fz_version(void);
/* Message for the last error returned on this thread ("" if none). */
const char* fz_last_error_string(void);
/* Number of kernel launches issued through this library by this process (test hook that
 * proves the native path ran). */
int64_t fz_launch_count(void);

/* ---- shifted-window matricize --------------------------------------------------------
 * Replaces SWMatricize.forward = per window torch.roll + einops rearrange, then torch.cat
 * (factorization/operations.py:266-272, 321-325, 417-421).
 *   x: (B, C, D, H, W)   y: (nshift*B*(C/d), G, d, P),  G=(D/pd)(H/ph)(W/pw), P=pd*ph*pw
 *   y[w*B*h + b*h + hh, g, dd, p] = x[b, hh*d+dd, (g_i*p_i + p_i - s_w,i) mod S_i] * (1/div)
 *   shifts: HOST pointer, nshift*3 ints (0 for an unshifted window).
 *   elem_bytes: 4 (fp32) or 2 (bf16/fp16, moved as opaque 16-bit words; relu/div must be 0/1).
 *   relu: if nonzero apply max(.,0) while moving (fp32 only) — FactMixer.act, factorizer.py:44.
 *   div: if >1 divide by it (fp32 only) — used as the backward of fz_swm_inv.
 */
int fz_swm_fwd(const void* x, void* y, int B, int C, int D, int H, int W, int d, int pd,
               int ph, int pw, int nshift, const int* shifts, int elem_bytes, int relu,
               int div, fz_stream_t stream);

/* Replaces SWMatricize.inverse_forward (operations.py:274-280, 423-434):
 *   x = (((0.0 + z_0) + z_1) + ...) / nshift, z_w = inverse window of chunk w of y (fp32).
 *   average: 1 → divide by nshift (the module's forward); 0 → plain sum (backward of fwd).
 *   gate: optional (may be NULL) tensor shaped like y; when given, element e of y counts
 *         only where gate[e] > 0 (fused ReLU backward for the relu=1 forward).
 */
int fz_swm_inv(const void* y, void* x, int B, int C, int D, int H, int W, int d, int pd,
               int ph, int pw, int nshift, const int* shifts, int average, const void* gate,
               fz_stream_t stream);

/* ---- batched NMF ---------------------------------------------------------------------
 * Replaces MatrixFactorization.forward = decompose (init → T × [update U, update V]) then
 * reconstruct u @ v.mT (matrix_factorization.py:514-546) for solver "mu" (:241-247) or
 * "hals" (:210-229), RandomInit broadcast buffers u0 (M,R), v0 (N,R) (:52-58).
 *   x, y: (nmat, M, N) fp32;  u_out (nmat,M,R) / v_out (nmat,N,R) optional (NULL to skip).
 * Supported natively: M <= 32, N <= 64*floor(64/Mpad) (8x512, 16x256, 32x128 families),
 * 1 <= R <= 4; anything else returns FZ_E_UNSUPPORTED (the Python layer then uses its
 * composed path).
 */
int fz_nmf_fwd(const float* x, const float* u0, const float* v0, float* y, float* u_out,
               float* v_out, int64_t nmat, int M, int N, int R, int T, int solver, float eps,
               fz_stream_t stream);

/* Backward of fz_nmf_fwd w.r.t. x (what autograd does through the unrolled iterations,
 * matrix_factorization.py:522-533; formulas: SURVEY.md Appendix A).  The forward is
 * recomputed inside the kernel; only the last Tgrad iterations carry gradient
 * (num_grad_steps, matrix_factorization.py:476,506-512).
 *   gy: (nmat,M,N) grad of y, may be NULL if gu/gv given;  gu (nmat,M,R), gv (nmat,N,R):
 *   optional grads of the decompose() outputs (NULL to skip);  gx: (nmat,M,N) out.
 */
int fz_nmf_bwd(const float* x, const float* u0, const float* v0, const float* gy,
               const float* gu, const float* gv, float* gx, int64_t nmat, int M, int N, int R,
               int T, int Tgrad, int solver, float eps, fz_stream_t stream);

/* 1 if (M,N,R,T,Tgrad) is covered by the native kernels (fwd and bwd), else 0. */
int fz_nmf_supported(int M, int N, int R, int T, int Tgrad);

#ifdef __cplusplus
}
#endif
#endif /* FACTORIZER_HIP_H */
