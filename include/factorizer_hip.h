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
 *   - re-entrant and thread-safe.  Process-wide state is exactly: the launch counter (atomic), the per-stream queues of
 *     deferred finish jobs (mutex-guarded; fz_finish_* below), the FZ_GEMM_BX default read once.  Per-thread state: the last
 *     error string, the `products` default (fz_set_products), the tile order (fz_set_tile_order), the defer flag
 *     (fz_finish_defer).  Nothing else outlives a call.
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

/* Storage type of ACTIVATION tensors (inputs, outputs, their gradients, saved pre-activations).
 * FZ_STORE_BF16 is the mixed-precision mode of BASELINE configs[4]: activations live in HBM as bf16,
 * every kernel converts to fp32 on load and rounds to nearest-even on store; parameters, LayerNorm
 * statistics, weight gradients and all arithmetic — MFMA accumulation, the NMF factors / Gram
 * matrices / eps (matrix_factorization.py:200,236) — stay fp32.  Pointers documented as "activation"
 * below are `void*`: float* or (bf16) uint16_t* according to the call's act_dtype. */
#define FZ_STORE_F32 0
#define FZ_STORE_BF16 1

#define FZ_SOLVER_MU 0   /* matrix_factorization.py:232-247 MultiplicativeUpdate        */
#define FZ_SOLVER_HALS 1 /* matrix_factorization.py:194-229 CoordinateDescent + ReLU    */

typedef void* fz_stream_t; /* hipStream_t */

/* Library version (major*10000 + minor*100 + patch). */
int fz_version(void);
/* ABI revision: bumped whenever an entry point's signature or a descriptor's layout changes (a caller built against another
 * revision would pass a stream where an int is expected, or leave new trailing descriptor fields uninitialised).  A binding
 * compares fz_abi_version() with the FZ_ABI_VERSION of the header it was written against BEFORE the first call and refuses
 * to run on a mismatch (factorizer_amd/_native.py does).  History: 3 = round 3; 4 = round 4 (`products` / `tune` descriptor
 * fields, `products` argument of fz_conv3_*); 5 = round 5 (the two-window entry points fz_nmf_cf_fwd2 / _bwd2 removed, this
 * function added); 6 = round 6 (fz_finish_defer is per THREAD, one finish queue per stream, fz_finish_flush_all added, fz_gemm_dw_desc.ldw). */
#define FZ_ABI_VERSION 6
int fz_abi_version(void);
/* Walking order of the fused-core launches (fz_nmf_cf_fwd / fz_nmf_cf_bwd) this THREAD issues from now on: 0 = ascending over
 * the patch tiles (the default), 1 = descending; < 0 only queries.  Returns the previous setting.  Results do not depend on it.
 * Why it exists: the windows of one SWMatricize are separate launches that read the SAME tensors; a window that walks them in
 * the direction opposite to the previous window starts at the part the 256 MiB Infinity Cache still holds — measured on
 * fz_nmf_cf_bwd: 3 % less time at the README model's stage 0 (537 MB tensors), 10 % at stage 1 (134 MB),
 * profiles/r05_tile_order.md.  (Reading a tensor backwards right after the launch that WROTE it gains nothing measurable: the
 * whole-step A/B with every streaming kernel alternating was -0.13 ms of 17.4, all of it from the window pairs.) */
int fz_set_tile_order(int descending);
/* Deferred finishes.  Every weight-gradient entry point (fz_wgrad, fz_wgrad_group, fz_gemm_dw, fz_mlp_chain mode 2, fz_ln_bwd with
 * affine gradients, fz_reduce_rows, fz_rowsum, fz_chunk_reduce, fz_upcat_wgrads) ends with a tiny launch that adds per-workgroup
 * partial rows in a fixed order.  Those launches produce PARAMETER gradients — nothing in a backward pass reads them — but each
 * one is a 5-10 us serial slot of the stream (54 per README training step: 0.33 ms).  fz_finish_defer(1) makes the calls that
 * follow ON THIS THREAD queue them instead (the descriptors only; nothing is copied), fz_finish_defer(0) stops queueing; both
 * return the previous setting, fz_finish_defer(-1) only queries.  There is one queue per stream the deferred calls were given
 * (a stream belongs to one device).  fz_finish_flush(stream) runs what is queued for THAT stream, on it, as one or two grids
 * and returns how many jobs that was (0: nothing queued for it; < 0: an error code).  fz_finish_flush_all(waiter) runs every
 * queue on its own stream and makes `waiter` wait (event) for each queue whose stream is another one: the caller that is about
 * to read the gradients on `waiter` is ordered behind all of them; returns the total.  fz_finish_pending() counts queued jobs
 * over all queues.  While finishes are queued the caller keeps alive, and does not touch, every buffer handed to those calls
 * (workspaces, gradient outputs); a call that ACCUMULATES into its output drains its stream's queue and runs at once.  The
 * sums and their order are the same deferred or not: bit-identical.  Reference counterpart: none — autograd launches one
 * reduction kernel per gradient as it goes. */
int fz_finish_defer(int on);
int fz_finish_pending(void);
int fz_finish_flush(fz_stream_t stream);
int fz_finish_flush_all(fz_stream_t waiter);
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
 *   elem_bytes: 4 (fp32) or 2 (16-bit words: moved as opaque bits — bf16 or fp16 alike — unless relu/div
 *               ask for arithmetic, in which case they are taken to be bf16).
 *   relu: if nonzero apply max(.,0) while moving — FactMixer.act, factorizer.py:44.
 *   div: if >1 divide by it — used as the backward of fz_swm_inv.
 */
int fz_swm_fwd(const void* x, void* y, int B, int C, int D, int H, int W, int d, int pd,
               int ph, int pw, int nshift, const int* shifts, int elem_bytes, int relu,
               int div, fz_stream_t stream);

/* Replaces SWMatricize.inverse_forward (operations.py:274-280, 423-434):
 *   x = (((0.0 + z_0) + z_1) + ...) / nshift, z_w = inverse window of chunk w of y (fp32; with
 *   act_dtype FZ_STORE_BF16 the sum and the division are fp32 and the result is rounded once).
 *   average: 1 → divide by nshift (the module's forward); 0 → plain sum (backward of fwd).
 *   gate: optional (may be NULL) tensor shaped like y; when given, element e of y counts
 *         only where gate[e] > 0 (fused ReLU backward for the relu=1 forward).
 */
int fz_swm_inv(const void* y, void* x, int B, int C, int D, int H, int W, int d, int pd,
               int ph, int pw, int nshift, const int* shifts, int average, const void* gate,
               int act_dtype, fz_stream_t stream);

/* ---- batched NMF ---------------------------------------------------------------------
 * Replaces MatrixFactorization.forward = decompose (init → T × [update U, update V]) then
 * reconstruct u @ v.mT (matrix_factorization.py:514-546) for solver "mu" (:241-247) or
 * "hals" (:210-229), RandomInit broadcast buffers u0 (M,R), v0 (N,R) (:52-58).
 *   x, y: (nmat, M, N) activations (act_dtype: fp32, or bf16 storage with the whole factorisation —
 *   u, v, Gram matrices, eps — in fp32);  u_out (nmat,M,R) / v_out (nmat,N,R) fp32, optional (NULL to skip).
 * Supported natively: M <= 32, N <= 64*floor(64/Mpad) (8x512, 16x256, 32x128 families),
 * 1 <= R <= 4; anything else returns FZ_E_UNSUPPORTED (the Python layer then uses its
 * composed path).
 */
int fz_nmf_fwd(const void* x, const float* u0, const float* v0, void* y, float* u_out,
               float* v_out, int64_t nmat, int M, int N, int R, int T, int solver, float eps,
               int act_dtype, fz_stream_t stream);

/* Backward of fz_nmf_fwd w.r.t. x (what autograd does through the unrolled iterations,
 * matrix_factorization.py:522-533; formulas: SURVEY.md Appendix A).  The forward is
 * recomputed inside the kernel; only the last Tgrad iterations carry gradient
 * (num_grad_steps, matrix_factorization.py:476,506-512).
 *   gy: (nmat,M,N) grad of y, may be NULL if gu/gv given;  gu (nmat,M,R), gv (nmat,N,R):
 *   optional grads of the decompose() outputs (NULL to skip);  gx: (nmat,M,N) out.
 */
int fz_nmf_bwd(const void* x, const float* u0, const float* v0, const void* gy,
               const float* gu, const float* gv, void* gx, int64_t nmat, int M, int N, int R,
               int T, int Tgrad, int solver, float eps, int act_dtype, fz_stream_t stream);

/* 1 if (M,N,R,T,Tgrad) is covered by the native kernels (fwd and bwd), else 0. */
int fz_nmf_supported(int M, int N, int R, int T, int Tgrad);

/* ---- split-N NMF: matrices too wide for one wavefront ----------------------------
 * The reference's default FactMixer reshape Matricize(num_heads=1, grid_size=1) (factorizer/factorizer.py:17;
 * tests/test_factorizer.py:25: one 16 x 262 144 matrix) and `num_heads=` forms with M != 8
 * (tests/test_factorizer.py:123).  Same semantics and argument meaning as fz_nmf_fwd / fz_nmf_bwd
 * (matrix_factorization.py:514-546; backward: SURVEY.md Appendix A), fp32, 1 <= M <= 64, any N, 1 <= R <= 4;
 * the columns are split over workgroups, cross-workgroup sums are two-stage and fixed-order (no float atomics:
 * bitwise reproducible).  One call issues fz_gnmf_launches(T, Tgrad, backward) kernels (2T+1 forward).
 * workspace: fz_gnmf_workspace_bytes(nmat, M, N, R, T, backward) bytes of device memory, caller-owned. */
int fz_gnmf_supported(int M, int64_t N, int R, int T, int Tgrad);
int64_t fz_gnmf_workspace_bytes(int64_t nmat, int M, int64_t N, int R, int T, int backward);
int fz_gnmf_launches(int T, int Tgrad, int backward);
int fz_gnmf_fwd(const float* x, const float* u0, const float* v0, float* y, float* u_out, float* v_out,
                int64_t nmat, int M, int64_t N, int R, int T, int solver, float eps, void* workspace,
                fz_stream_t stream);
int fz_gnmf_bwd(const float* x, const float* u0, const float* v0, const float* gy, const float* gu,
                const float* gv, float* gx, int64_t nmat, int M, int64_t N, int R, int T, int Tgrad, int solver,
                float eps, void* workspace, fz_stream_t stream);

/* ---- FactMixer core on channels-first tensors (hot shape: head_dim 8, patch 8x8x8) -----------
 * One launch per shift window performs SWMatricize.forward → NMF.forward →
 * SWMatricize.inverse_forward (factorizer.py:41-50; operations.py:417-434;
 * matrix_factorization.py:514-546) without materialising the matricized tensors:
 *   fwd: out = ((0 + z_0) + z_1 + ...) / nshift, z_w = scatter_w(NMF(gather_w(t)));
 *        call once per window in order with accumulate = (w > 0), divisor = nshift on the last
 *        window (1 otherwise).
 *   bwd: gt (+)= [t > 0 if relu_gate] * scatter_w(dNMF(gather_w(t); gather_w(ga) / nshift));
 *        call once per window with accumulate = (w > 0).  relu_gate = 1 is the CONTRACT "t = relu(z) >= 0 and gt is the
 *        gradient with respect to z" (factorizer.py:44): for HALS rank 1 the backward then runs in the 8-dimensional row space
 *        (csrc/nmf_gram.h — for a non-negative matrix no ReLU of the iteration ever clips, and v can be eliminated); a t with
 *        negative entries must be passed with relu_gate = 0.
 * t, out, ga, gt: (B, C, D, H, W) activations (act_dtype: fp32, or bf16 storage — the running window sum is
 * then rounded to bf16 between windows, the factorisation itself stays fp32); shift: HOST pointer to 3 ints
 * (W-axis shift even).
 */
int fz_nmf_cf_supported(int C, int D, int H, int W, int d, int pd, int ph, int pw, int R, int T, int Tgrad);
int fz_nmf_cf_fwd(const void* t, const float* u0, const float* v0, void* out, int B, int C, int D,
                  int H, int W, const int* shift, int accumulate, int divisor, int R, int T,
                  int solver, float eps, int act_dtype, fz_stream_t stream);
int fz_nmf_cf_bwd(const void* t, const float* u0, const float* v0, const void* ga, void* gt, int B,
                  int C, int D, int H, int W, const int* shift, int accumulate, int nshift,
                  int relu_gate, int R, int T, int Tgrad, int solver, float eps, int act_dtype,
                  fz_stream_t stream);

/* The same fused core for ANY patch (pd, ph, pw) with head_dim 8 and at most 256 voxels per patch (csrc/nmf_pcf.hip:
 * BASELINE configs[4] uses patch (5,6,5) because 160x192x160 is not divisible by 8; the p = 4 test configurations):
 * same call protocol and semantics as fz_nmf_cf_fwd / fz_nmf_cf_bwd, any shift parity. */
int fz_nmf_pcf_supported(int C, int D, int H, int W, int d, int pd, int ph, int pw, int R, int T, int Tgrad);
int fz_nmf_pcf_fwd(const void* t, const float* u0, const float* v0, void* out, int B, int C, int D, int H, int W,
                   int pd, int ph, int pw, const int* shift, int accumulate, int divisor, int R, int T, int solver,
                   float eps, int act_dtype, fz_stream_t stream);
int fz_nmf_pcf_bwd(const void* t, const float* u0, const float* v0, const void* ga, void* gt, int B, int C, int D,
                   int H, int W, int pd, int ph, int pw, const int* shift, int accumulate, int nshift, int relu_gate,
                   int R, int T, int Tgrad, int solver, float eps, int act_dtype, fz_stream_t stream);
/* Windows w > 0 of the generic-patch backward: accumulate = 1 is a scattered read-modify-write of gt.  Where
 * fz_nmf_pcf_bwd_prefers_separate() returns 1 (129..160 voxels per patch, bf16 storage: measured) it is cheaper to
 * give the window its own buffer (accumulate = 0) and add it with fz_act_add (dst += src, n elements of act_dtype, both
 * 16-byte aligned) — what the reference's modular chain does anyway: every window's inverse_forward term is a tensor of its
 * own before the sum (factorizer/layers/reshape/operations.py:423-434, autograd of it). */
int fz_nmf_pcf_bwd_prefers_separate(int pd, int ph, int pw, int act_dtype);
int fz_act_add(void* dst, const void* src, int64_t n, int act_dtype, fz_stream_t stream);

/* ---- channels-first GEMM family (1x1 layers, k2s2 conv / transposed conv, input grads) ----
 * Out[m, n] = epilogue( sum_k A[m,k] * prologue(In)[k,n] ), n = voxel.  One descriptor drives
 * every dense layer of the block and of the U-shape:
 *   Linear (Conv1d k=1 on flatten(2))      layers/linear.py:53-58
 *   LayerNorm over channels as a prologue  layers/norm.py:29-34
 *   MLP Linear-GELU-Linear                 layers/mlp.py:54-60
 *   window average of inverse_forward      factorization/operations.py:426-433 (src_mode 1)
 *   adapter on cat([skip, up])             unet.py:128 + factorizer.py:116 (two sources, no cat)
 *   Conv3d(k=2,s=2) downsample             unet.py:53   (loader = FZ_LOAD_S2D)
 *   ConvTranspose3d(k=2,s=2) upsample      unet.py:123  (epilogue = FZ_EPI_D2S)
 *   head Conv3d(k=1)                       unet.py:253
 * and, with w_t (use the weight transposed), the input gradient of each of them.
 */
#define FZ_LOAD_PLAIN 0 /* In[k][n] = x[b, k, n]                                             */
#define FZ_LOAD_S2D 1   /* In[(c,td,th,tw)][coarse n] = x[b, c, 2d+td, 2h+th, 2w+tw]         */
#define FZ_LOAD_K3 2    /* In[(c,kd,kh,kw)][n] = x[b, c, d+kd-1, h+kh-1, w+kw-1], zero padded  */
#define FZ_EPI_PLAIN 0  /* y[b, m, n]                                                         */
#define FZ_EPI_D2S 1    /* rows m = (o, td,th,tw): y[b, o, 2d+td, 2h+th, 2w+tw]               */
#define FZ_EPI_LNBWD 2  /* M == 32: result = dL/d(LayerNorm out); y = LayerNorm backward of it  */
#define FZ_ACT_NONE 0
#define FZ_ACT_RELU 1
#define FZ_ACT_GELU 2 /* exact erf GELU, layers/mlp.py:56 */

/* How a layer forms its fp32 products: field `products` of fz_gemm_desc / fz_mlp_desc / fz_wgrad_desc and the last int of
 * fz_conv3_fwd / fz_conv3_wgrad_partials.  A descriptor left zero-initialised follows the process default. */
#define FZ_PRODUCTS_DEFAULT 0    /* fz_gemm_bx_enable()'s setting (environment FZ_GEMM_BX read once, else split-bf16) */
#define FZ_PRODUCTS_SPLIT_BF16 1 /* six exact bf16 products per fp32 product on the bf16 matrix cores (csrc/gemm_bx.h)  */
#define FZ_PRODUCTS_FP32_MFMA 2  /* v_mfma_f32_32x32x2_f32 / _16x16x4_f32                                               */

typedef struct fz_gemm_desc {
  const void* x[4];    /* activation: input source tensors (B, C_i, Vin)                                */
  int nsrc;            /* number of sources                                                    */
  int src_mode;        /* 0: channel concat of x[0] (c0 ch) and x[1]; 1: average of nsrc sources */
  int c0;              /* channels of x[0] in concat mode (0 = all)                            */
  int Cin;             /* input channels                                                       */
  int64_t Vin;         /* voxels per sample of the input                                       */
  int Di, Hi, Wi;      /* input D, H, W (FZ_LOAD_S2D: fine grid; FZ_LOAD_K3)                   */
  const float* w;      /* weights                                                              */
  int w_t, ldw;        /* A[m][k] = w_t ? w[k*ldw + m] : w[m*ldw + k]                          */
  int M, K;            /* output rows, reduction length (even)                                 */
  const float* bias;   /* [M] (plain) / [M/8] (d2s) or NULL                                    */
  int ln;              /* LayerNorm prologue over Cin                                          */
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  float* stats_out;    /* (B, 2, Vin): mean, rstd of the LN prologue, or NULL                  */
  int bact;            /* activation applied to the input operand                              */
  const void* bmul;    /* activation: input operand *= act'(bmul) (same shape as the input) or NULL        */
  int bmul_kind;
  int eact;            /* activation applied to the result                                     */
  const void* res;     /* activation: residual added to the result (same shape as y; with FZ_EPI_D2S: the fine-
                          resolution tensor, e.g. the skip-connection gradient) or NULL            */
  const void* emul;    /* activation: result *= act'(emul) (same shape as y) or NULL                       */
  int emul_kind;
  void* y;             /* activation */
  int64_t Ncol;        /* columns per sample: Vin (plain) or coarse voxel count (s2d)          */
  int Ho, Wo;          /* coarse H, W (s2d columns / d2s input grid)                           */
  int B;
  int loader, epilogue;
  /* FZ_EPI_LNBWD (M = 32 with K = 32 or 64; or M = K = 64 with the added gradient): LayerNorm input x (B,M,V),
   * saved stats (B,2,V), gamma (M), gradient added to the result (optional for M = 32), and a partial buffer of
   * fz_gemm_lnbwd_partials(desc) x 2*M floats that receives per-workgroup (dgamma | dbeta) sums (reduce over
   * rows in order). */
  const void* lnb_x;      /* activation */
  const float* lnb_stats;
  const float* lnb_g;
  const void* lnb_gadd;   /* activation */
  float* lnb_part;
  int act_dtype;          /* FZ_STORE_F32 / FZ_STORE_BF16: element type of every "activation" pointer */
  int products;           /* FZ_PRODUCTS_* */
  int tune;               /* 0 = the library chooses the tile.  Timing probes (tools/probes/gemm_bx_bench.py; results do not depend
                             on it): split-bf16 family only — 100 + 10*nacc + mb = streaming form with that tile (nacc 2|4, mb 1|2),
                             200 + 10*nacc + mb = K-split form (nacc 1|2, mb 1|2), 300 = streaming form, library's tile */
} fz_gemm_desc;

/* number of 64-float partial rows fz_gemm writes to lnb_part for this descriptor */
int64_t fz_gemm_lnbwd_partials(const fz_gemm_desc* desc);
/* out[e] = sum over rows of part[row][e], fixed order (rows x n floats); tmp: 64 x n floats of
 * scratch for the two-stage path (rows > 512), may be NULL */
int fz_reduce_rows(const float* part, int64_t rows, int n, float* out, float* tmp, fz_stream_t stream);

int fz_gemm(const fz_gemm_desc* desc, fz_stream_t stream);
/* fz_gemm runs layers with a reduction length >= 64 on the bf16 matrix cores with every fp32 operand split into three
 * bf16 values and six exact products per fp32 product (csrc/gemm_bx.hip: fp32 accuracy at 6/16 of the fp32-MFMA time).
 * on = 0 keeps them on v_mfma_f32_32x32x2_f32, on = 1 enables, on < 0 only queries; returns the previous setting.
 * Default: environment FZ_GEMM_BX (read once), else enabled.  This is only the DEFAULT for descriptors whose `products`
 * field is FZ_PRODUCTS_DEFAULT: a caller that wants two models of one process to differ sets the field. */
int fz_gemm_bx_enable(int on);

/* ---- MLP chain ((C, H) = (32, 64), (32, 128) or (64, 128)): two GEMMs, hidden tensor stays in the accumulators
 * (modes 0 and 1; part rows are 2*C floats: dgamma | dbeta)
 * Replaces per FactorizerBlock (factorizer.py:76, layers/mlp.py:54-63, layers/norm.py:29-34):
 *   mode 0  out = in + W2·gelu(W1·LN(in) + b1) + b2 ; z1 = W1·LN(in) + b1 and stats (mean, rstd)
 *           are written for the backward
 *   mode 1  gz1 = (W2ᵀ·in) ∘ gelu'(z1)  (written for the weight gradients) ;
 *           out = LayerNormBackward(W1ᵀ·gz1; x1, stats, gamma) + in ; part receives
 *           fz_mlp_partials(B, V) rows of 64 floats (dgamma | dbeta partial sums, reduce with
 *           fz_reduce_rows)
 *   mode 2  (C = 32; H = 64 in one launch, H = 128 in one launch per 64-row half of the hidden tensor) mode 1 WITH
 *           the two weight gradients and bias gradients of the MLP, gz1 never reaching
 *           HBM: gw2 (C, H) = sum_v in[.,v] gelu(z1[.,v])^T, gb2 (C) = sum_v in, gb1 (H) = sum_v gz1,
 *           gw1 (H, C) = sum_v gz1 (ln_g * xhat + ln_b)^T.  wpart: fz_mlp_wgrad_workspace_bytes(B, V) bytes of
 *           caller workspace (one row of partial sums per resident workgroup, added in row order).
 */
typedef struct fz_mlp_desc {
  int mode;
  const void* in;     /* activation; mode 0: x1 (B, C, V) ; mode 1: g2 = dL/d(out of the block) (B, C, V)  */
  const float* w1;    /* (H, C)                                                              */
  const float* w2;    /* (C, H)                                                              */
  const float* b1;    /* (H) or NULL, mode 0                                                 */
  const float* b2;    /* (C) or NULL, mode 0                                                 */
  const float* ln_g;  /* (C)                                                                 */
  const float* ln_b;  /* (C), mode 0                                                         */
  float ln_eps;
  float* stats;       /* (B, 2, V): written in mode 0, read in mode 1                        */
  void* z1;           /* activation (B, H, V): written in mode 0, read in mode 1                        */
  void* gz1;          /* activation (B, H, V): written in mode 1                                        */
  const void* x1;     /* activation (B, C, V): the LayerNorm input, mode 1                              */
  void* out;          /* activation (B, C, V)                                                */
  float* part;        /* mode 1: fz_mlp_partials(B, V) x 64 floats                           */
  int B, C, H;
  int64_t V;
  int act_dtype;      /* FZ_STORE_F32 / FZ_STORE_BF16                                        */
  void* wpart;        /* mode 2: workspace, fz_mlp_wgrad_workspace_bytes(B, V)                */
  float* gw1;         /* mode 2: (H, C)                                                      */
  float* gb1;         /* mode 2: (H)                                                         */
  float* gw2;         /* mode 2: (C, H)                                                      */
  float* gb2;         /* mode 2: (C)                                                         */
  float* gln;         /* mode 2: (2*C) dgamma | dbeta of the LayerNorm (part is not used)    */
  float* glp;         /* mode 2, H = 128: (B, C, V) fp32 scratch (the first half's part of W1^T gz1) */
  int products;       /* FZ_PRODUCTS_* */
  /* [r5] mode 0, (C, H) = (32, 64) or (64, 128: an even number of 256-voxel tiles per sample), split-bf16 products
   * (fz_mlp_pre_supported answers for a shape): the block's out-projection in front of the chain,
   * x1 = pre_w . pre_in + pre_b + pre_res (factorizer.py:53,75) formed on the accumulators, written to pre_out and
   * normalised in registers — `in` is ignored, x1 is never read back by another launch (6 instead of 7 tensor passes for
   * steps 3 + 4 of the block at C = 32; at C = 64 the epilogue re-reads the lane's own x1 from L2).  pre_in == NULL: the
   * plain chain on `in`. */
  const void* pre_in;   /* activation (B, C, V): the core's output a, or NULL                 */
  const float* pre_w;   /* (C, C) out_proj weight                                              */
  const float* pre_b;   /* (C) or NULL                                                         */
  const void* pre_res;  /* activation (B, C, V): the block input x                             */
  void* pre_out;        /* activation (B, C, V): x1                                            */
  /* [r5] with pre_in and C = 32: the network's head Linear(C -> post_m <= 4, k1) (unet.py:253,274) applied to `out` while it is in
   * registers — the launch that would read the block output back (fz_head_fwd) disappears.  post_out == NULL: no head. */
  const float* post_w;  /* (post_m, C)                                                         */
  const float* post_b;  /* (post_m) or NULL                                                    */
  void* post_out;       /* activation (B, post_m, V) or NULL                                   */
  int post_m;
} fz_mlp_desc;

/* ---- input gradient AND weight gradient of a 32 -> 32 1x1 layer in one pass (in_proj behind LayerNorm, out_proj;
 * factorizer.py:38,53 + norm.py:29-34 as autograd sees them):
 *   y = W^T g                                  (ln: then LayerNormBackward(.; q, stats, ln_g) + gadd; gln receives
 *                                               dgamma | dbeta)
 *   gw (32, 32) = sum_v g[m,v] * in[k,v]       (ln: in = ln_g * xhat(q) + ln_b, the LayerNorm output)
 *   gb (32)     = sum_v g[m,v]                 (optional)
 * wpart: fz_gemm_dw_workspace_bytes(B, V) bytes of caller workspace; rows are added in index order. */
typedef struct fz_gemm_dw_desc {
  const void* g;       /* activation (B, 32, V): gradient of the layer output                 */
  const void* q;       /* activation (B, 32, V): layer input (ln: the LayerNorm input)        */
  const float* w;      /* (32, 32) forward weight                                             */
  int ln;
  const float* stats;  /* ln: (B, 2, V) mean, rstd                                            */
  const float* ln_g;   /* ln: (32)                                                            */
  const float* ln_b;   /* ln: (32)                                                            */
  const void* gadd;    /* ln: activation (B, 32, V) added to y, or NULL                       */
  void* y;             /* activation (B, 32, V)                                               */
  float* gln;          /* ln: (64) dgamma | dbeta                                             */
  void* wpart;
  float* gw;           /* (32, 32), row stride ldgw                                           */
  float* gb;           /* (32) or NULL                                                        */
  int B, C;
  int64_t V;
  int act_dtype;
  int ldgw;            /* floats between rows of gw; 0 = 32 (a column block of a wider weight gradient: its width) */
  int ldw;             /* floats between rows of w;  0 = 32 (a column block of a wider weight, read in place)  [ABI 6] */
} fz_gemm_dw_desc;
int fz_gemm_dw_rows(int B, int64_t V);
int64_t fz_gemm_dw_workspace_bytes(int B, int64_t V);
int fz_gemm_dw(const fz_gemm_dw_desc* desc, fz_stream_t stream);

/* ---- backward of a Linear(32 -> M), M <= 4, in one pass (the network's head: factorizer/unet.py:253) ----
 * gx = W^T gy (activation, (B, 32, V)); weight and bias gradient as fz_head_bwd_rows() partial rows of 132 floats
 * (gW[m][c] at m*32 + c, gb[m] at 128 + m) for fz_chunk_reduce(part, rows, 132, out132, 0, stream). */
int fz_head_bwd_rows(void);
int64_t fz_head_bwd_workspace_bytes(void);
int fz_head_bwd(const void* gy, const void* x, const float* w, void* gx, float* part, int B, int M, int C, int64_t V,
                int act_dtype, fz_stream_t stream);
/* forward of the same layer: y = W x + b (w row-major [M][32], bias [M] or null), (B, 32, V) -> (B, M, V); fz_gemm takes this
 * path by itself for a plain Linear(32 -> M <= 4) without LayerNorm / activation / residual. */
int fz_head_fwd(const void* x, const float* w, const float* bias, void* y, int B, int M, int C, int64_t V, int act_dtype,
                fz_stream_t stream);

/* ---- decoder level forward in one pass ---------------------------------------------------
 * out = adapter(cat([skip, ConvTranspose3d(k2, s2)(deep)], 1))  (factorizer/unet.py:125-127 with the stage adapter of
 * factorizer.py:116) without forming the up-sampled tensor: the caller passes the composed weights
 * wbt[tap][m][k] = sum_c W_b[m][c] W_t[k][c][tap] (tap = kd*4 + kh*2 + kw), the skip half W_a (row stride lda) and
 * bias' = b_ad + W_b b_t (or NULL).  C = 32 skip / output channels, Cd = 64 deep channels, (D, H, W) = coarse extent. */
/* the weight compositions of that node (a few Mflop; device pointers, fp32):
 *   compose: wc[k][m][t] = sum_c w_t[k][c][t] w_b[m][c] (and/or wbt[t][m][k], bias'[m] = b_ad[m] + sum_c w_b[m][c] b_t[c]; outputs may be NULL)
 *   wgrads : from gt[k][m][t] = sum_n deep[k][n] g[m][fine(n,t)]:  gw_t = w_b^T gt,  gw_b = gt . w_t + gb_ad (x) b_t (row stride ldg),
 *            gb_t = w_b^T gb_ad                                                                      */
int fz_upcat_compose(const float* w_t, const float* w_b, int ldb, const float* b_t, const float* b_ad, float* wc, float* wbt, float* bias,
                     int Cd, int O, int M, fz_stream_t stream);
int fz_upcat_wgrads(const float* gt, const float* w_t, const float* w_b, int ldb, const float* gb_ad, const float* b_t, float* gw_t,
                    float* gw_b, int ldg, float* gb_t, int Cd, int O, int M, fz_stream_t stream);
int fz_upcat_supported(int C, int Cd, int D, int H, int W);
int fz_upcat(const void* skip, const void* deep, const float* wa, int lda, const float* wbt, const float* bias, void* out,
             int B, int C, int Cd, int D, int H, int W, int act_dtype, fz_stream_t stream);
/* [r5] The first layer of the FactorizerBlock that CONSUMES a 32-channel tensor, t = relu(in_proj(LayerNorm1(x)))
 * (factorizer.py:38,44,75; norm.py:29-34; in_proj has no bias), applied by the launch that PRODUCES x while the tile is in
 * registers: the launch that would read x back disappears (x itself is still written: the block's backward and its residual
 * read it).  t: activation (B, 32, V) of the launch's storage type; stats: (B, 2, V) fp32 mean | rstd of every voxel, as
 * fz_gemm's stats_out.  Producers that take it: fz_upcat2 (a decoder level's output), fz_conv3_fwd2 (the stem). */
typedef struct fz_block_prologue {
  const float* ln_g;  /* (32) */
  const float* ln_b;  /* (32) */
  float ln_eps;
  const float* w;     /* (32, 32) in_proj weight */
  void* t;
  float* stats;
} fz_block_prologue;
int fz_upcat2(const void* skip, const void* deep, const float* wa, int lda, const float* wbt, const float* bias, void* out,
              int B, int C, int Cd, int D, int H, int W, int act_dtype, const fz_block_prologue* pro, fz_stream_t stream);

int fz_mlp_supported(int C, int H, int64_t V);
int fz_mlp_pre_supported(int C, int H, int64_t V, int products);
int64_t fz_mlp_partials(int B, int64_t V);
int fz_mlp_wgrad_rows(int B, int64_t V);
int64_t fz_mlp_wgrad_workspace_bytes(int B, int64_t V);
int fz_mlp_chain(const fz_mlp_desc* desc, fz_stream_t stream);

/* ---- weight gradients of the GEMM family ------------------------------------------------
 * GW[m,k] = sum_{b,n} P[b,m,n] * Q(In)[b,k,n] — what autograd computes for the weights of
 * Conv1d(k=1) (layers/linear.py:44-58), Conv3d k2s2 / ConvTranspose3d k2s2 (unet.py:53,123)
 * and the k3 stem (factorizer.py:145-149).  Deterministic two-stage reduction; the caller
 * passes a workspace of fz_wgrad_workspace_bytes().
 */
#define FZ_QL_PLAIN 0 /* Q[k][n] = in[b,k,n]                                     */
#define FZ_QL_S2D 1   /* Q[(c,td,th,tw)][coarse n] = in[b,c,2d+td,2h+th,2w+tw]   */
#define FZ_QL_K3 2    /* Q[(c,kd,kh,kw)][n] = in[b,c,d+kd-1,h+kh-1,w+kw-1] (zero pad) */

typedef struct fz_wgrad_desc {
  const void* p;      /* activation (B, M, N) output-side gradient                                    */
  int M;
  const void* pmul;   /* activation, optional: P *= act'(pmul)                                      */
  int pmul_kind;      /* FZ_ACT_RELU / FZ_ACT_GELU                                           */
  const void* q[4];   /* activation: input sources                                                 */
  int nsrc, src_mode, c0;
  int Cin;            /* input channels                                                      */
  int K;              /* Q rows: Cin / 8*Cin / 27*Cin                                        */
  int64_t Vq;         /* voxels per sample of the input                                      */
  int D, H, W;        /* input grid (s2d: fine grid; k3)                                     */
  int64_t N;          /* reduction columns per sample                                        */
  int Ho, Wo;         /* coarse grid (s2d)                                                   */
  const float* stats; /* (B,2,Vq) mean/rstd: Q = (in-mean)*rstd, or NULL                     */
  int qact;           /* activation applied to Q                                             */
  const float* ln_g;  /* with stats: gw[m][k] = ln_g[k]*acc + ln_b[k]*rowsum(P)[m]           */
  const float* ln_b;
  float* gw;          /* (M, K) out                                                          */
  float* gbias;       /* (M) row sums of P, or NULL                                          */
  int accumulate;     /* add into gw/gbias instead of overwriting                            */
  int B;
  int loader;
  int act_dtype;      /* FZ_STORE_F32 / FZ_STORE_BF16; with bf16 the products run on bf16 MFMAs (fp32 accumulation) */
  int products;       /* FZ_PRODUCTS_* (fp32 activations) */
} fz_wgrad_desc;

int64_t fz_wgrad_workspace_bytes(const fz_wgrad_desc* desc);
int fz_wgrad(const fz_wgrad_desc* desc, void* workspace, fz_stream_t stream);
/* Up to 4 weight-gradient problems in ONE partial-sum grid (same results as n fz_wgrad calls, bit for bit): the four dense
 * layers of a FactorizerBlock at C >= 64 — fc2, fc1 behind LayerNorm (mlp.py:54-63), out_proj, in_proj behind LayerNorm
 * (factorizer.py:38,53) — whose single launches each leave half of every CU's workgroup slots empty.  Problems that do not
 * take the 64 x 64 register-operand kernel are launched one by one, in order.  workspaces[i] >= fz_wgrad_workspace_bytes(descs[i]). */
int fz_wgrad_group(const fz_wgrad_desc* const* descs, void* const* workspaces, int n, fz_stream_t stream);

/* ---- channels-first LayerNorm (layers/norm.py:29-34), standalone ---------------------------
 * fwd: y = (x-mean)*rstd*gamma + beta over C per voxel; stats (B,2,V) = (mean, rstd) optional.
 * bwd: gx = rstd*(gl*gamma - mean_c(gl*gamma) - n*mean_c(gl*gamma*n)) [+ gadd]; the parameter
 * gradients come from the same pass (C <= 64) or from fz_wgrad (diag of P=gl, Q=normalised x).
 */
int fz_ln_fwd(const void* x /* activation */, const float* gamma, const float* beta, void* y /* activation */,
              float* stats, int B, int C, int64_t V, float eps, int act_dtype, fz_stream_t stream);
/* gparams (optional): [gamma grad (C) | beta grad (C)] computed in the same pass; needs a
 * workspace of fz_ln_bwd_workspace_bytes2(B, C, V). */
int64_t fz_ln_bwd_workspace_bytes(int C);                        /* C <= 64 */
int64_t fz_ln_bwd_workspace_bytes2(int B, int C, int64_t V);     /* any C   */
int fz_ln_bwd(const void* gl, const void* x, const float* stats, const float* gamma,
              const void* gadd, void* gx /* gl, x, gadd, gx: activations */, float* gparams, void* workspace,
              int B, int C, int64_t V, int act_dtype, fz_stream_t stream);

/* ---- stem: Conv3d(kernel 3, padding 1) (factorizer/factorizer.py:145-149 → unet.py:231,261) ----
 * fwd: y = conv3d(x, w) [+ bias]; needs even C_in, W % 4 == 0.
 * wgrad: per-workgroup partial sums part[nchunk][M][27*C_in], part_bias[nchunk][M]
 * (nchunk = fz_conv3_wgrad_chunks), reduced in a fixed order by fz_chunk_reduce; needs
 * W % 32 == 0 and 27*C_in <= 128. */
int fz_conv3_fwd(const void* x /* activation */, const float* w, const float* bias, void* y /* activation */,
                 int B, int Cin, int M, int D, int H, int W, int act_dtype, int products, fz_stream_t stream);
/* [r5] the same with the consuming block's first layer applied to the output tile (fz_block_prologue, below): C_in = 4,
 * M = 32, split-bf16 products (fz_conv3_prologue_supported); pro == NULL: fz_conv3_fwd. */
struct fz_block_prologue;
int fz_conv3_prologue_supported(int Cin, int M, int W, int products);
int fz_conv3_fwd2(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int M, int D, int H, int W,
                  int act_dtype, int products, const struct fz_block_prologue* pro, fz_stream_t stream);
int fz_conv3_wgrad_chunks(int B, int D, int H, int W);
int fz_conv3_wgrad_partials(const void* gy, const void* x /* activations */, float* part, float* part_bias, int B,
                            int Cin, int M, int D, int H, int W, int act_dtype, int products, fz_stream_t stream);
int fz_chunk_reduce(const float* part, int nchunk, int64_t n, float* out, int accumulate, fz_stream_t stream);
/* the same over rows `ld` floats apart (ld >= n): a column block of wider partial rows (the head's 132-float rows: weight block
 * and bias block into their own gradient tensors) */
int fz_chunk_reduce_ld(const float* part, int nchunk, int64_t n, int64_t ld, float* out, int accumulate, fz_stream_t stream);

/* out[c] = sum over batch and voxels of x[b,c,v] (bias gradient of ConvTranspose3d, unet.py:123);
 * part: workspace of B * fz_rowsum_chunks(V) * C floats. */
int fz_rowsum_chunks(int64_t V);
int fz_rowsum(const void* x /* activation */, float* part, float* out, int B, int C, int64_t V, int act_dtype,
              fz_stream_t stream);

/* ---- grouped "same" cross-correlation of the Deconver family ----------------------------
 * The one operator of the reference's blind-deconvolution updates (factorizer/factorization/deconvolution.py:
 * 21-40 `conv`, 136-156 `update_s`): out[b, g*Co+o, v] = sum_i sum_t in[b, g*Ci+i, v+t-p] * w[b or 0, g, o, i, t],
 * zero padding p = k/2 — used as H, as H^T (channel-transposed, flipped filters) and for input gradients.
 *   in (B, G*Ci, D, H, W); w (Bw, G, Co, Ci, kd, kh, kw), Bw = B if w_batched else 1; out (B, G*Co, D, H, W); fp32.
 *   mul_a, mul_b: both NULL -> out = corr + add_eps; both given (shape of out) -> the fused multiplicative update
 *   out = mul_a * mul_b / (corr + add_eps)  (s * (H^T x + eps) / (H^T H s + eps)).
 * Supported: any odd kernel extent <= 7 per axis (2-D layers as D = 1, kd = 1) and any channel counts.  Compile-time
 * instantiations serve Co <= 16 per group with cubic / depth-1 square 3-5-7 kernels (the reference's defaults); everything else —
 * anisotropic kernels such as (5, 3, 3) of the reference's tests/test_deconver.py, more channels per group — runs the run-time-extent
 * kernels (eight output channels per workgroup) [r6]. */
int fz_gcorr_supported(int Ci, int Co, int kd, int kh, int kw);
int fz_gcorr(const float* in, const float* w, float* out, const float* mul_a, const float* mul_b, int B, int G,
             int Ci, int Co, int D, int H, int W, int kd, int kh, int kw, int w_batched, float add_eps,
             fz_stream_t stream);
/* Filter gradient of fz_gcorr (what autograd derives through F.conv{2,3}d in the reference, deconvolution.py:21-40):
 * gw[bw, g, o, i, t] = sum_b sum_v gout[b, g*Co+o, v] * in[b, g*Ci+i, v+t-p]  (b = bw for per-sample filters).
 * Deterministic two-stage reduction; ws: fz_gcorr_wgrad_workspace_bytes(...) bytes of caller workspace. */
int64_t fz_gcorr_wgrad_workspace_bytes(int B, int G, int Ci, int Co, int D, int H, int W, int kd, int kh, int kw);
int fz_gcorr_wgrad(const float* in, const float* gout, float* gw, void* ws, int B, int G, int Ci, int Co, int D, int H,
                   int W, int kd, int kh, int kw, int w_batched, fz_stream_t stream);

/* ---- fused soft-Dice + BCE-with-logits loss (training step; the form of the bundle's
 * DiceCELoss(sigmoid=True, squared_pred=True), model_zoo/factorizer_brats23/configs/train.yaml:67-70).
 * sums: part (planes, fz_dice_bce_chunks(V), 4) = {sum p*t, sum p^2, sum t^2, sum bce} per (b,c) plane.
 * grad: gz = gscale * (cd * dDice/dz + cb * dBCE/dz) with coef (planes,2) = {2*inter+s, den+s}. */
int fz_dice_bce_chunks(int64_t V);
int fz_dice_bce_sums(const float* z, const float* t, float* part, int planes, int64_t V, fz_stream_t stream);
int fz_dice_bce_grad(const float* z, const float* t, const float* coef, float* gz, int planes, int64_t V,
                     float cd, float cb, const float* gscale, fz_stream_t stream);

/* DiceCELoss(sigmoid=True, squared_pred=True) as MONAI 1.4 evaluates it for C > 1 channels (the bundle
 * pins monai==1.4.0, model_zoo/factorizer_brats23/docs/requirements.txt:11; train.yaml:67-70): soft Dice on
 * sigmoid(z) per (b,c) plane + nn.CrossEntropyLoss over the channel softmax with the float target as class
 * probabilities.  z, t: (B, C, V), 2 <= C <= 8.
 * sums: part (B, fz_dice_bce_chunks(V), 3C+1) = per channel {sum p*t, sum p^2, sum t^2}, then sum CE.
 * grad: gz = gscale * (cd * dDice/dz + cb * (softmax_c * sum_c' t_c' - t_c)), coef (B*C, 2) = {2*inter+s, den+s}. */
int fz_dice_ce_sums(const float* z, const float* t, float* part, int B, int C, int64_t V, fz_stream_t stream);
/* loss (1 float) = mean_{b,c}(1 - num/den) + sum(CE)/(B V) and coef (B*C, 2) = {num, den} from the partial sums of
 * fz_dice_ce_sums, chunks added in a fixed order; B <= 8 (larger batches: reduce `part` with framework ops) */
int fz_dice_ce_finish(const float* part, int B, int C, int64_t V, float smooth, float* loss, float* coef, fz_stream_t stream);
int fz_dice_ce_grad(const float* z, const float* t, const float* coef, float* gz, int B, int C, int64_t V,
                    float cd, float cb, const float* gscale, fz_stream_t stream);

/* ---- sliding-window inference stitching -------------------------------------
 * What MONAI's SlidingWindowInfererAdapt(roi 128^3, sw_batch 2, overlap 0.5, mode "gaussian") does
 * around the network (model_zoo/factorizer_brats23/configs/inference.yaml:96-102, train.yaml:206-212):
 * per window  gather: win (C, rd, rh, rw) <- vol (C, D, H, W)[:, z0:, y0:, x0:]
 *             accumulate: out (C, D, H, W)[window] += g * prob (C, rd, rh, rw) ; cnt (D, H, W)[window] += g
 *             with g[z,y,x] = max(gz[z]*gy[y]*gx[x], wmin)  (separable Gaussian importance map)
 * and once    finalize: out /= cnt.
 * One sample per call; rw a multiple of 4 (16-byte vectors on the volume side when W and x0 are
 * too).  Windows must be accumulated by separate
 * (stream-ordered) calls: they overlap. */
int fz_sw_gather(const float* vol, float* win, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0,
                 int x0, fz_stream_t stream);
int fz_sw_accumulate(const float* prob, float* out, float* cnt, const float* gz, const float* gy, const float* gx,
                     float wmin, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0,
                     fz_stream_t stream);
int fz_sw_finalize(float* out, const float* cnt, int C, int64_t V, fz_stream_t stream);

/* ---- AdamW over one flat buffer (torch.optim.AdamW of the training recipe,
 * model_zoo/factorizer_brats23/configs/train.yaml:72-76).  step >= 1 is the 1-based update count;
 * grad_scale multiplies the gradient first (1/world after a summed all-reduce). */
int fz_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  fz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FACTORIZER_HIP_H */
