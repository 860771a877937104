/*
 * semivl_hip.h — C-ABI of libsemivl_hip.so: the MI355X (gfx950) kernels behind the SemiVL training hot path.
 *
 * The reference (google-research/semivl) has no FFI layer: its boundary is Python
 * (`model/builder.py:56-159`, `model/vlm.py:90-127`, `utils/train_utils.py:19-49`, `semivl.py:52-58,223-345`).
 * This header is the operator boundary a maintainer would bind from Python (ctypes stub in INTEGRATION.md);
 * `semivl_amd/` is the host-side mirror of the reference's Python surface built on top of it.
 *
 * Conventions (all entry points) -- what is TRUE of the library, checked by tests/test_abi.py and
 * tests/test_ops_gpu.py::test_two_streams_*:
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller, including every
 *     scratch buffer (svl_*_ws_* size queries): the library never allocates or frees device memory;
 *   - it never synchronises the device or a stream.  Work is enqueued on `stream` of the caller's CURRENT device.
 *     Two entry-point families (svl_gemm_f32 on ragged token counts, svl_attention_{fwd,bwd}) additionally run a thin,
 *     disjoint-output launch on a HELPER stream that is event-forked from and event-joined back into `stream` inside
 *     the same call (capture-legal: events only), so from the caller's point of view everything is ordered on `stream`;
 *   - host-side state: one helper stream + two events per (device, caller stream) pair, created lazily under a mutex on
 *     first use and kept until svl_stream_release(stream) / svl_shutdown(); per-device kernel attributes set once per
 *     device; and the process-wide configuration switches svl_set_gemm_emulation / svl_set_conv_tiled (relaxed
 *     atomics seeded from the environment, read once per call, selecting between kernels that compute the same
 *     function).  Nothing else is mutable: the library is re-entrant across host threads, devices and streams -- calls
 *     that use different streams never share a helper stream, an event or a buffer;
 *   - tensors are dense fp32 unless stated; token tensors are [rows, C] row-major ("NHWC"/[B,T,C]);
 *   - returns 0 (SVL_OK) or a negative svl_status; svl_last_error() gives the message (thread-local).
 */
#ifndef SEMIVL_HIP_H_
#define SEMIVL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* svl_stream_t; /* hipStream_t */

typedef enum svl_status {
  SVL_OK = 0,
  SVL_ERR_INVALID_ARG = -1,
  SVL_ERR_LAUNCH = -2,
  SVL_ERR_UNSUPPORTED = -3
} svl_status;

int svl_version(void); /* 600: round-6 ABI (+ svl_permute4_f32, svl_bound2_f32, svl_attention tail kernels replace the row kernels); 502: + svl_ce_up_fused_f32, svl_softmax_max_up_f32, svl_ce_up_num_blocks (pixel losses on head-resolution logits); 501: + svl_attention_{fwd,bwd}_h2, svl_attention_h2_ws_bytes (fused attention on fp16 x 2 pre-packed operands); 500: round-5 ABI (fp16 x 2 planes: svl_split_planes_f16x2, svl_planes_bytes_fmt, fmt / scale fields of svl_pgemm_desc); 401: + svl_conv3x3_weight_planes, svl_gemm_desc::conv_w_planes, w_planes of svl_conv3x3_gn_f32; 400: round-4 ABI (gn_in / svl_conv3x3_gn_f32 / svl_groupnorm_apply / _scale_shift, svl_permute_rows_f32,
                           svl_stream_prepare, svl_last_gemm_path; gn_in arguments of the tiled weight gradient and the Conv2d(C -> 1)
                           entries, `accumulate` of svl_avgpool_cat_bwd); 300: round-3 ABI (packed-planes operands; planes outputs of LayerNorm / attention; loss-mode arguments of
                           the pixel-loss entries; 200 = round 2: helper-stream contexts, caller-owned scratch everywhere) */
/* Destroys the helper stream/events this library created for `stream` on the current device (no-op if none), or for
 * every stream.  Call after the stream has been synchronised; not required before process exit. */
int svl_stream_release(svl_stream_t stream);
/* Creates the helper stream/events of `stream` NOW instead of on first use (no-op if they exist).  The HIP runtime maps a
 * process's streams onto its hardware queues in creation order: a caller that wants a reproducible map (the data-parallel
 * reducer creates the step's streams before its communication stream) creates them up front.  No reference counterpart. */
int svl_stream_prepare(svl_stream_t stream);
/* The helper stream svl_stream_prepare(stream) created (a hipStream_t): for probing hardware-queue sharing only. */
int svl_stream_helper(svl_stream_t stream, void** helper);
int svl_shutdown(void);
int svl_num_stream_contexts(void); /* live (device, stream) helper contexts -- introspection for tests */
/* Copies the calling thread's last error message (NUL-terminated) into buf; returns its length. */
int svl_last_error(char* buf, size_t len);
/* Measurement aid (bench.py's `roofline.clock_mhz`; no counterpart in the reference): n_waves <= 1024 single-wave
 * workgroups compare the shader-clock counter with the constant 100 MHz counter for `ticks_100mhz` (<= 1e9 = 10 s) and
 * leave out[2 i] = shader-clock cycles, out[2 i + 1] = 100 MHz ticks (device memory, 16 n_waves bytes).  Launched on a
 * stream of its own next to a kernel under measurement: cycles / ticks x 100 = the MHz sustained under that kernel. */
int svl_clock_probe(unsigned long long* out, int n_waves, unsigned long long ticks_100mhz, svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM core (fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32: exact f32 fma chain).
 * C[z][m][n] = epilogue( alpha * sum_k A[z](m,k) * B[z](n,k) )
 * Replaces every dense contraction of the path: F.linear / nn.MultiheadAttention in_proj+out_proj
 * (maskclip_vit.py:110-118,141 via mmcv), FFN (maskclip_vit.py:94-100,142), PatchEmbed conv
 * (maskclip_vit.py:266-276,495), proj 1x1 (maskclip_vit.py:338,554), the cosine-sim einsum
 * (vlg_head.py:217), F.conv2d text classifier (vlm.py:99), all VLGHead convs (vlg_head.py:70-137,169-190),
 * q k^T / p v bmm inside attention, and all their dgrad / wgrad counterparts (autograd in the reference).
 * ---------------------------------------------------------------------------------------------- */
enum {
  SVL_A_KCONTIG = 0, /* A(m,k) = A[m*lda + k]                       (activations x, dY, Q)           */
  SVL_A_MCONTIG = 1, /* A(m,k) = A[k*lda + m]                       (dY^T for wgrad, P^T)            */
  SVL_A_CONV = 2,    /* A(m,k): m = NHWC pixel, k = (tap, ci): implicit im2col (see svl_conv_geom)   */
  SVL_A_PATCH = 3    /* A(m,k): m = (img, py, px), k = (c, i, j) of an NCHW image, patch P x P       */
};
enum {
  SVL_B_KCONTIG = 0, /* B(n,k) = B[n*ldb + k]   (torch Linear weight [out,in], K rows, packed conv w) */
  SVL_B_NCONTIG = 1, /* B(n,k) = B[k*ldb + n]   (dgrad: W as [K=out][N=in]; V in P·V; X in wgrad)     */
  SVL_B_CONVW = 2    /* B(n,k): n = (tap, ci), k = NHWC pixel: im2col^T for conv wgrad                */
};
enum {
  SVL_ACT_NONE = 0,
  SVL_ACT_GELU = 1,      /* erf GELU */
  SVL_ACT_RELU = 2,
  /* backward forms (SVL_OUT_STRIDED): `resid` is NOT added but read as the saved pre-activation z[m][n] */
  SVL_ACT_MUL_DGELU = 3, /* v *= GELU'(z)      -- dgrad of Linear -> GELU fused with the activation's derivative */
  SVL_ACT_MUL_DRELU = 4  /* v  = z > 0 ? v : 0 */
};
enum {
  SVL_OUT_STRIDED = 0, /* C + zo*bs_outer + zi*bs_inner + m*ld_m + n*ld_n                              */
  SVL_OUT_CONVT2X = 1, /* ConvTranspose2d k2 s2: m=(img,h,w), n=(a,b,co) -> pixel (img,2h+a,2w+b), co  */
  SVL_OUT_PATCH = 2    /* patch tokens: row m=(img,p) -> img*(P+1)+1+p ; resid row = 1+p (pos_embed)   */
};

typedef struct svl_operand {
  const float* ptr;
  int64_t ld;        /* leading dimension (elements)                                   */
  int64_t bs_outer;  /* batch stride for zo = z / batch_inner                          */
  int64_t bs_inner;  /* batch stride for zi = z % batch_inner                          */
} svl_operand;

/* Stride-1, same-size NHWC convolution geometry used by SVL_A_CONV / SVL_B_CONVW.
 * Logical input channel ci in [0, C1+C2): ci < C1 reads src1 (the operand ptr, pixel stride ld),
 * otherwise src2 (pixel stride ld2) of image (img / rep) — the `repeat`+`cat` of vlg_head.py:131-134
 * without materialising it. Tap (ti,tj) reads pixel (oh*stride + sign*(ti*dil - pad), ow*stride + sign*(tj*dil - pad));
 * sign=+1 for forward/wgrad, -1 for dgrad. Out-of-range pixels read 0. */
typedef struct svl_conv_geom {
  int H, W;          /* INPUT spatial size (bounds + addressing)                                  */
  int Ho, Wo;        /* OUTPUT spatial size used to decompose the pixel index (0 -> same as H, W) */
  int stride;        /* output -> input pixel stride (0/1 -> 1); >1 only with sign=+1 (ConvT k2s2 backward = conv k2 s2) */
  int C1, C2;
  int rep;
  int KH, KW, dil, pad, sign;
  const float* src2;
  int64_t ld2;
  int patch;  /* SVL_A_PATCH: patch size P; image is [img, C1, H, W] NCHW */
} svl_conv_geom;

typedef struct svl_gemm_desc {
  int a_mode, b_mode;
  int M, N, K;
  int batch;        /* number of z slices (>=1)                                              */
  int batch_inner;  /* z -> (zo, zi) = (z / batch_inner, z % batch_inner); >=1               */
  int ksplit;       /* >0: split-K — z indexes K ranges [z*ksplit, min(K,(z+1)*ksplit)), operand batch strides ignored */
  svl_operand A, B;
  svl_conv_geom conv;
  /* output */
  float* C;
  int out_mode;
  int64_t ldc_m, ldc_n, c_bs_outer, c_bs_inner;
  int ct_H, ct_W, ct_Cout; /* SVL_OUT_CONVT2X: input spatial size and Cout; SVL_OUT_PATCH: ct_H = patches per image */
  /* epilogue: v = alpha*acc; v += bias[n % bias_mod]; v = act(v); v += resid; if(accumulate) v += C */
  float alpha;
  const float* bias;
  int bias_mod;     /* 0: bias[n]; >0: bias[n % bias_mod] */
  int act;
  float* preact;    /* optional: value BEFORE act is also stored here (same addressing as C, SVL_OUT_STRIDED only) */
  const float* resid; /* addressed like C for SVL_OUT_STRIDED (own strides below); see SVL_OUT_PATCH */
  int64_t ldr_m, ldr_n, r_bs_outer, r_bs_inner;
  int accumulate;
  const void* conv_w_planes; /* optional, SVL_A_CONV with a 3x3 / stride 1 geometry: the weights B once more as
                              * svl_conv3x3_weight_planes(B, N, C1 + C2) wrote them (two fp16 planes + per-output-channel
                              * exponents).  Used in emulation mode 6 by (a) the spatially tiled kernel (pad 1, dilation 1,
                              * N = 32 / 64: three fp16 products per term instead of six bf16 ones) and (b) the whole-image
                              * kernel of the DILATED convolutions (pad = dilation > 1 on 32 x 32 maps, one source, N % 64 == 0,
                              * no bias / activation: the ASPP branches of vlg_head.py:38-50 and their input gradients);
                              * ignored otherwise. */
  void* emu_ws;              /* optional (round 5): 8 bytes of device scratch private to this call.  With it, emulation mode 6
                              * serves a large launch of the in-register split kernel (implicit-GEMM convolutions, their weight
                              * gradients, dense GEMMs outside the packed-planes path; >= 4 GFLOP) on fp16 x 2 terms -- three
                              * products instead of six -- with ONE power-of-two scale per operand tensor, found by a maximum
                              * pass over exactly the elements the launch reads (a per-row scale does not factor out of a
                              * convolution's taps).  Same error level vs fp64 (tests/test_ops_gpu.py); NULL: bf16 x 3 terms. */
} svl_gemm_desc;

int svl_gemm_f32(const svl_gemm_desc* d, svl_stream_t stream);

/* Process-wide (relaxed-atomic, read once per call) arithmetic mode of the LARGE dense GEMMs (M >= 256, N >= 96, K >= 64, dense operand modes):
 *   0  v_mfma_f32_32x32x2_f32, exact fp32 fma chain (default);
 *   6  fp32-accurate emulation on the bf16 matrix pipe: every operand element is split into 3 bf16 terms and the 6
 *      leading cross products are accumulated in fp32 (error <= the fp32 path's, 2.7x the MFMA rate).  Mode 6 also
 *      covers svl_attention_{fwd,bwd} (all five products of the fused attention; S is recomputed bit-identically in
 *      the backward), the forward / input gradient of the spatially tiled 3x3 convolutions, and the K = 64 / 128 row
 *      streams with the ConvTranspose2d(k 2, s 2) pixel-shuffle store (SVL_OUT_CONVT2X; csrc/gemm_shortk.hip);
 *   3  2-term split, 3 products (~16 mantissa bits, 5.3x the MFMA rate); dense GEMMs only.
 * Initial value: environment variable SVL_GEMM_EMU (0 if unset).  Inputs, outputs and accumulation stay fp32.
 * A/B switches (read once per process): SVL_ATTN_NO_EMU, SVL_CONV_TILED_NO_EMU keep those kernels on the fp32 pipe. */
int svl_set_gemm_emulation(int mode);
int svl_get_gemm_emulation(void);
/* 1 (default): narrow (N = 32 / 64) 3x3 stride-1 convolutions run on the spatially tiled kernel; 0: implicit GEMM only.
 * Initial value: 0 if the environment variable SVL_CONV_NO_TILED is set. */
int svl_set_conv_tiled(int on);
/* Measurement aid (bench.py prices a launch against the pipe that served it; no reference counterpart): which kernel family
 * the calling thread's LAST svl_gemm_f32 call dispatched to -- 0 exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 bf16 split
 * products (v_mfma_f32_32x32x16_bf16; emulation mode 3 / 6: dense, implicit-GEMM forward / input gradient / weight gradient,
 * tiled 3x3), 2 the short-K row stream (fp32 MFMA, HBM-bound), 3 an elementwise kernel (no MFMA), 4 the in-register split
 * kernel on fp16 x 2 terms (three products; svl_gemm_desc::emu_ws). */
int svl_last_gemm_path(void);

/* ------------------------------------------------------------------------------------------------
 * fp32-accurate GEMM with PRE-SPLIT, fragment-packed operands (the fast form of emulation mode 6; csrc/gemm_planes.hip).
 * An fp32 matrix X [rows, K] (K % 16 == 0) is held as three bf16 "planes" x = x0 + x1 + x2, packed in the register image
 * of the MFMA operand fragments: for k-group kg = k / 16, row block rb = row / 32 and plane pl one 1 KiB chunk
 *     planes[((kg * rows_padded / 32 + rb) * 3 + pl) * 1024 + (h * 32 + row % 32) * 16 .. + 16)
 * holds, for h = 0 / 1, the 8 bf16 values of row `row` at k = kg * 16 + 4 h + {0, 1, 2, 3, 8, 9, 10, 11}.  rows_padded =
 * svl_planes_rows(rows) (a multiple of 256: tiles read whole 256-row bands; the padding is never stored from);
 * svl_planes_bytes(rows, K) bytes, 1 KiB aligned.  Weights are packed once per parameter version, activations by their
 * producer (svl_layernorm_* planes outputs, the previous GEMM's epilogue, svl_split_planes_bf16x3 as the generic pass);
 * the GEMM then runs LDS-DMA + ds_read_b128 + v_mfma_f32_32x32x16_bf16 only (6 cross products, fp32 accumulate).
 * Replaces F.linear / its input gradient on the ViT linears (maskclip_vit.py:110-144 via mmcv MultiheadAttention / FFN)
 * when svl_set_gemm_emulation(6) is active.
 *   svl_split_planes_bf16x3   element (r, k) read at x[r * ld + k * k_stride] (k_stride 1: row-major; ld 1 + k_stride =
 *                             leading dim: the transpose of a row-major matrix), written to rows [row_off, row_off+rows)
 *                             of a plane buffer with planes_rows (% 256 == 0) rows; row_off % 32 == 0; the rest of the
 *                             last 32-row block is zero-filled
 *   svl_gemm_planes_f32       C[m, n] = epi(sum_k A[m, k] B[n, k]) for rows m in [m_off, m_off + M) (m_off % 32 == 0):
 *                             bias[n], act (svl_act: GELU / RELU; MUL_DGELU / MUL_DRELU multiply by the activation
 *                             derivative at resid = saved pre-activation), preact (pre-activation copy out), resid
 *                             (added), accumulate (into C); outputs: C fp32 row-major (ldc) and / or planes_out = the
 *                             result as packed planes [M, N] (p_rows % 256 == 0 rows, N % 16 == 0) for the next GEMM.
 *                             Either may be NULL, not both. */
typedef struct svl_pgemm_desc {
  const void* A;        /* packed planes of [a_rows, K], a_rows % 256 == 0 */
  const void* B;        /* packed planes of [b_rows, K], b_rows % 256 == 0, b_rows >= N */
  int64_t a_rows, b_rows;
  int m_off, M, N, K;
  float* C;
  int64_t ldc;
  void* planes_out;
  int64_t p_rows;
  const float* bias;
  int act;
  float* preact;
  const float* resid;
  int64_t ldr;
  int accumulate;
  /* round 5: the fp16 x 2 operand format (below).  All zero / NULL = the bf16 x 3 form above. */
  int fmt;                /* format of A and B: 0 = bf16 x 3 planes (six products), 1 = fp16 x 2 planes + row scales (three) */
  int p_fmt;              /* format of planes_out: 0 / 1; 1 needs fmt = 1, a_rnorm, b_bound, p_sexp and no residual add */
  const int32_t* a_sexp;  /* fmt 1: scale exponents of A's rows [a_rows] / of B's rows [b_rows]; NULL = all zero */
  const int32_t* b_sexp;
  const float* a_rnorm;   /* p_fmt 1: upper bounds of the L2 norms of A's rows [a_rows] ...                          */
  const float* b_bound;   /* ... and a device float[2] = {max row L2 norm of B, max |bias|}: row m of the result is   */
  int32_t* p_sexp;        /* bounded by a_rnorm[m] b_bound[0] + b_bound[1]; its scale exponent is written to p_sexp[m] */
} svl_pgemm_desc;
int64_t svl_planes_rows(int64_t rows);             /* rows rounded up to the 256-row band a tile reads */
int64_t svl_planes_bytes(int64_t rows, int K);
int svl_split_planes_bf16x3(const float* x, int64_t ld, int64_t k_stride, int64_t rows, int K, void* planes,
                            int64_t planes_rows, int64_t row_off, svl_stream_t stream);
/* The fp16 x 2 operand format (round 5; csrc/gemm_planes_h2.hip): x(r, k) = 2^e(r) (h0 + h1), h0 = fp16(x 2^-e(r)), h1 =
 * fp16(x 2^-e(r) - h0), one scale exponent e(r) per ROW placing the row's largest magnitude in [2^14, 2^15); chunk layout as
 * above with two planes per (k-group, row block): svl_planes_bytes_fmt(rows, K, 1) = 4 bytes per element.  Two
 * round-to-nearest fp16 terms carry 23 significand bits, the three leading cross products (v_mfma_f32_32x32x16_f16, fp32
 * accumulate) leave out a1 b1 <= 2^-22 |a b|: half the matrix work of the bf16 x 3 form at an error vs fp64 at or below the
 * plain fp32 MFMA chain's for K >= 48 (tests/test_ops_gpu.py::test_gemm_planes_path).  The GEMM multiplies its
 * accumulators by 2^(e_A(m) + e_B(n)) (exact).  svl_split_planes_f16x2 is the generic pack pass (arguments as
 * svl_split_planes_bf16x3); it also writes sexp[row_off + r] and, if rnorm != NULL, an upper bound of each row's L2 norm --
 * what a GEMM needs to scale a planes OUTPUT in this format (svl_pgemm_desc::a_rnorm / b_bound). */
int64_t svl_planes_bytes_fmt(int64_t rows, int K, int fmt);
int svl_split_planes_f16x2(const float* x, int64_t ld, int64_t k_stride, int64_t rows, int K, void* planes,
                           int64_t planes_rows, int64_t row_off, int32_t* sexp, float* rnorm, svl_stream_t stream);
int svl_gemm_planes_f32(const svl_pgemm_desc* d, svl_stream_t stream);
/* Weight gradient of a narrow (Co = 32 / 64) 3x3 / stride 1 / pad 1 convolution over NHWC activations with an optional
 * second concat source (read at image img / rep): slabs[g][co][tap * (C1 + C2) + ci] for g < groups (forward-pack
 * layout per slab); the caller sums the slabs (svl_reduce_slabs_f32).  Replaces the conv2d weight-gradient of
 * vlg_head.py:121-127 (Up.conv) for the layers whose implicit-GEMM form is im2col-address bound.  C1, C2 % 4 == 0, (C1 + C2) % 32 == 0.
 * Under svl_set_gemm_emulation(6) the bf16 x 6 kernel serves Co = 64, Co = 32 with (C1 + C2) % 64 == 0, and additionally
 * Co = 128 (an error in mode 0); svl_conv3x3_wgrad_tiled_groups sizes `groups` for the kernel the current mode selects
 * (any groups >= 1 is valid for either). */
int svl_conv3x3_wgrad_tiled_groups(int imgs, int H, int W, int Ct, int Co);
/* gn_in (may be NULL; both entry points that take it): a [imgs][2][C1] table of svl_groupnorm_scale_shift -- src1 then holds
 * the PRE-normalisation output of the previous convolution and the operand relu(fma(src1, scale, shift)) = GroupNorm + ReLU
 * (vlg_head.py:122-123) is formed while the tile is staged: the normalised tensor is never written (round 4). */
int svl_conv3x3_wgrad_tiled(const float* dy, int64_t lddy, int Co, const float* src1, int64_t ld1, int C1,
                            const float* src2, int64_t ld2, int C2, int rep, int imgs, int H, int W, float* slabs,
                            int groups, const float* gn_in, svl_stream_t stream);

/* out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s*count + i]  — deterministic split-K combine. */
int svl_reduce_slabs_f32(float* out, const float* slabs, int nslab, int64_t count, int accumulate,
                         svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pixel-loss family (semivl.py:232,252,267-323; utils/train_utils.py:19-49; semivl.py:52-58).
 * logits are NCHW [B, N, H, W] fp32; label-like maps are int64 [B, H, W] (reference dtype).
 * ---------------------------------------------------------------------------------------------- */

/* conf[b,h,w] = max_c softmax(logits)[b,c,h,w]; label = argmax (ties -> lowest index). semivl.py:232,252 */
int svl_softmax_max_f32(const float* logits, int B, int N, int64_t HW, float* conf, int64_t* label,
                        svl_stream_t stream);

/* out = where(box == 1, b, a) elementwise over [B, HW] maps broadcast over C channels
 * (cutmix_img_: C=3 fp32 in place when out==a; cutmix_mask: C=1). utils/train_utils.py:19-27 */
int svl_cutmix_f32(float* out, const float* a, const float* b, const float* box, int B, int C, int64_t HW,
                   svl_stream_t stream);
int svl_cutmix_i64(int64_t* out, const int64_t* a, const int64_t* b, const float* box, int B, int64_t HW,
                   svl_stream_t stream);

/* Fused per-branch loss, forward + backward in one pass over the logits.
 *   ce_t(p)  = -log_softmax(logits)[target[p]]          (target == ignore_index -> 0, if use_ignore_t)
 *   ce_m(p)  = -log_softmax(logits)[mc_target[p]]       (mc_target == 255 -> 0)
 * Per block (deterministic, no atomics) partials[blk] = {
 *   sum_p w_t(p) * ce_t(p)      with w_t = 1 (supervised, conf==NULL) or
 *                                [conf >= conf_thresh && ign != 255] (pixelwise, train_utils.py:36-38),
 *   sum_p ce_m(p)               (mc term, semivl.py:52-58; 0 when mc_target == NULL),
 *   sum_p conf(p) * [ign != 255]   (pixelavg bookkeeping, train_utils.py:43-46),
 *   #(target != ignore) (supervised) or #(ign != 255) }
 * svl_ce_finalize() reduces the partials in double, in a fixed order, to sums[4].
 * If dlogits != NULL: dlogits = gscale_t * w_t * (softmax - onehot(target)) + gscale_m * [mc valid] * (softmax - onehot(mc))
 * where gscale_* are read from DEVICE memory (gscale[0], gscale[1]) so no host sync is needed.
 */
typedef struct svl_ce_desc {
  const float* logits;   /* [B, N, HW] */
  int B, N;
  int64_t HW;
  const int64_t* target; /* [B, HW] */
  int use_ignore_t;      /* 1: target==255 ignored (criterion_l, ignore_index=255); 0: criterion_u */
  const float* conf;     /* [B, HW] or NULL */
  const int64_t* ign;    /* [B, HW] or NULL */
  float conf_thresh;
  int all_pixels;        /* 1: conf_mode 'pixelavg' — w_t = 1 on every pixel (train_utils.py:43-46 sums the whole CE map) */
  const int64_t* mc_target; /* [B, HW] or NULL */
  float* partials;       /* workspace [svl_ce_num_blocks(B,N,HW)][4]: per-block {sum w*ce_t, sum ce_m, sum conf*valid, #valid} */
  float* dlogits;        /* [B, N, HW] or NULL */
  const float* gscale;   /* [2] device scalars, required when dlogits != NULL */
  const float* img_weight; /* [B] device or NULL: per-image factor on w_t -- conf_mode 'pixelratio' (train_utils.py:39-42:
                              the whole CE map of image b times its share of confident valid pixels; with all_pixels = 1) */
} svl_ce_desc;
int64_t svl_ce_num_blocks(int B, int N, int64_t HW); /* -1 if N is unsupported (N > 256) */
int svl_ce_fused_f32(const svl_ce_desc* d, svl_stream_t stream);
int svl_ce_finalize(const float* partials, int64_t nblocks, double* sums /* [4] device */, svl_stream_t stream);

/* Round 5: the same two pixel-loss entry points on logits that exist ONLY at the head's resolution.  The reference
 * resizes the head's [B, N, h, w] map to the crop (vlg_head.py:247, builder.py:93-97: bilinear, align_corners as
 * configured) and runs semivl.py:232,252 / 267-323 on the [B, N, H, W] result; here the resize (ATen upsample_bilinear2d
 * index math) is evaluated inside the kernels and the gradient is returned at the LOW resolution -- what
 * F.interpolate's backward would hand to the head.  Full-resolution logits / dlogits are never written.
 * Geometry: upsampling only (H >= h, W >= w), ratio <= 4.5 with at most 40 resized rows / columns per 8 head-resolution cells, N <= 160; svl_ce_up_num_blocks < 0 when unsupported
 * (callers then resize with svl_bilinear_planes_fwd and use svl_ce_fused_f32).  Maps (target, conf, ign, mc_target) are
 * full resolution [B, H, W]; partials: [svl_ce_up_num_blocks][4], reduced by svl_ce_finalize.  Deterministic. */
typedef struct svl_ce_up_desc {
  const float* logits;   /* [B, N, h, w] */
  int B, N;
  int h, w, H, W;
  int align_corners;
  const int64_t* target; /* [B, H, W] */
  int use_ignore_t;
  const float* conf;     /* [B, H, W] or NULL */
  const int64_t* ign;    /* [B, H, W] or NULL */
  float conf_thresh;
  int all_pixels;
  const int64_t* mc_target; /* [B, H, W] or NULL */
  float* partials;
  float* dlogits;        /* [B, N, h, w] or NULL: d(loss)/d(logits) at the head's resolution */
  const float* gscale;   /* [2] device scalars, required when dlogits != NULL */
  const float* img_weight; /* [B] device or NULL */
} svl_ce_up_desc;
int64_t svl_ce_up_num_blocks(int B, int N, int h, int w, int H, int W, int align_corners);
int svl_ce_up_fused_f32(const svl_ce_up_desc* d, svl_stream_t stream);
/* conf / label [B, H, W] = softmax(dim 1).max(dim 1) of the resized logits (semivl.py:232,252); first maximum wins. */
int svl_softmax_max_up_f32(const float* logits, int B, int N, int h, int w, int H, int W, int align_corners, float* conf,
                           int64_t* label, svl_stream_t stream);

/* Loss assembly without host syncs (semivl.py:267-323).
 * counts: int64[4] device = #valid of {mask_x != 255, ignore_mask_mixed1, ignore_mask_mixed2, ignore_mask} != 255.
 * gscale out: float[4][2] device = {g_t, g_m} for the branches {x, s1, s2, fp}: the factor each per-pixel CE term
 * carries in d(loss)/d(logits).  numel_u = B*H*W of one unlabeled branch; lam = current mcc lambda.
 * mc_counts: int64[3] device or NULL -- the guidance loss's normalisers for the branches {s1, s2, fp} when
 * mcc_loss_reduce is not 'mean_all' (semivl.py:52-58,156-162): 'mean_valid' = #(ignore mask != 255) (= counts[1..3]),
 * 'mean' = #(guidance label != 255) (nn.CrossEntropyLoss(ignore_index=255) mean); NULL = numel_u ('mean_all'). */
int svl_semivl_gscale(const int64_t* counts, double numel_u, float lam, const double* factors, const int64_t* mc_counts,
                      float* gscale, svl_stream_t stream);
/* conf_mode 'pixelavg' (train_utils.py:43-46): factor[0] = sum over images b of mean_{valid pixels}(conf_b).  The
 * unsupervised branch loss is (sum over ALL pixels of CE) * factor / #valid.  `factors` above/below: double[3] device
 * for the branches {s1, s2, fp}, or NULL for 'pixelwise'. */
int64_t svl_conf_avg_ws_doubles(int B); /* size of `workspace` below, in doubles */
int svl_conf_avg_factor(const float* conf, const int64_t* ign, int B, int64_t HW, double* factor, double* workspace,
                        svl_stream_t stream);
/* conf_mode 'pixelratio' (train_utils.py:39-42): ratio[b] = #(conf_b >= thresh & valid) / #valid of image b, the fp32
 * quotient of the two counts (torch's int / int); fed to svl_ce_fused_f32 as img_weight with all_pixels = 1.
 * workspace: svl_conf_avg_ws_doubles(B) doubles. */
int svl_conf_ratio_f32(const float* conf, const int64_t* ign, int B, int64_t HW, float thresh, float* ratio,
                       double* workspace, svl_stream_t stream);
/* sums: double[4 branches][4] device (from svl_ce_finalize); out float[8] device =
 * {loss, loss_x, loss_s1, loss_s2, loss_fp, loss_mc_s1, loss_mc_s2, loss_mc_fp}. */
int svl_semivl_loss(const double* sums, double numel_u, float lam, const double* factors, const int64_t* mc_counts,
                    float* out, svl_stream_t stream);
/* out[b, c, p] = softmax over classes c of NCHW logits (probability accumulation of the sliding-window eval modes,
 * supervised.py:61,113). */
int svl_softmax_planes_f32(const float* logits, int B, int N, int64_t HW, float* out, svl_stream_t stream);

/* counts-only pre-pass so the normalisers of semivl.py:38/57 exist before the fused pass:
 * counts[0] += #(map != 255) over [n] */
int svl_count_valid_i64(const int64_t* map, int64_t n, int64_t* count, svl_stream_t stream);

/* MaskCLIP guidance tail (vlm.py:100-109): dense [B, N, h, w] class scores ->
 * bilinear(align_corners=False) to [H, W] -> softmax(100 x) -> max/argmax -> (<thr -> 255); then
 * ign == 255 -> 255 (semivl.py:239-240). out int64 [B, H, W]. */
int svl_maskclip_labels(const float* dense, int B, int N, int h, int w, int H, int W, float scale,
                        float thresh, const int64_t* ign, int64_t* out, svl_stream_t stream);

/* per class max over its concept channels (text_embeddings.py:188-193).
 * concept_offsets [N+1] device int32, concepts of class c are channels [off[c], off[c+1]). */
int svl_concept_max_f32(const float* pred, int B, int NC, int64_t HW, const int* concept_offsets, int N,
                        float* out, svl_stream_t stream);

/* Evaluation: intersectionAndUnion of third_party/unimatch/util/utils.py:91-103 as integer histograms.
 * hist int64 [3K] (caller zeroes; accumulated): [0,K) intersection, [K,2K) prediction area, [2K,3K) target area;
 * pixels with target == ignore_index are dropped from all three. */
int svl_iou_hist_i64(const int64_t* pred, const int64_t* target, int64_t n, int K, int ignore_index, int64_t* hist,
                     svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation / elementwise / attention helpers (maskclip_vit.py:120-144, vlg_head.py:39-137).
 * ---------------------------------------------------------------------------------------------- */

/* LayerNorm over the last dim C of x [rows, C]; stats [rows, 2] = (mean, rstd). */
int svl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                      float* y, float* stats, svl_stream_t stream);
/* The same with the result ALSO (y != NULL) or ONLY (y == NULL) emitted as packed bf16x3 planes, the A operand of the
 * following svl_gemm_planes_f32 (C % 16 == 0; planes_rows % 256 == 0 rows in the plane buffer, rows start at 0):
 * LN1 -> qkv and LN2 -> FFN-1 of every ViT block (maskclip_vit.py:120-144). */
int svl_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                             float* y, float* stats, void* planes, int64_t planes_rows, svl_stream_t stream);
/* The same with the result as fp16 x 2 planes (svl_pgemm_desc.fmt = 1): sexp [planes_rows] receives the rows' scale exponents,
 * rnorm [planes_rows] (optional) upper bounds of their L2 norms; the planes are bit-identical to svl_split_planes_f16x2 over
 * the fp32 result (which `y`, optional here, still receives). */
int svl_layernorm_fwd_planes_f16x2(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                                   float* y, float* stats, void* planes, int64_t planes_rows, int32_t* sexp, float* rnorm,
                                   svl_stream_t stream);
/* dx = LN backward (+ dx_add if non-NULL, fused residual-grad add). If dgamma_part != NULL also writes
 * per-block partial sums dgamma_part/dbeta_part [nparts, C] (nparts = svl_layernorm_bwd_parts(rows)). */
int svl_layernorm_bwd_parts(int64_t rows);
int svl_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, int64_t rows,
                      int C, const float* dx_add, float* dx, float* dgamma_part, float* dbeta_part,
                      svl_stream_t stream);
/* Row softmax in place over the first `cols` entries of rows with stride ld: p = softmax(scale * s).
 * Entries [cols, ld) are zeroed. */
int svl_softmax_rows_fwd(float* s, int64_t rows, int cols, int64_t ld, float scale, svl_stream_t stream);
/* ds = scale * p * (dp - sum_j dp_j p_j), in place on dp. */
int svl_softmax_rows_bwd(float* dp, const float* p, int64_t rows, int cols, int64_t ld, float scale,
                         svl_stream_t stream);

/* y = x / max(||x||_2, eps) over rows of [rows, C]; inv_norm [rows] saved. (maskclip_vit.py:555, vlg_head.py:215) */
int svl_l2norm_fwd(const float* x, int64_t rows, int C, float eps, float* y, float* inv_norm, svl_stream_t stream);
int svl_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int64_t rows, int C, float* dx,
                   svl_stream_t stream);

/* out[c] (+)= sum_r x[r*ld + c]  (bias gradients). Two deterministic stages through ws
 * (svl_colsum_ws_floats(rows, C) floats). */
int64_t svl_colsum_ws_floats(int64_t rows, int C);
int svl_colsum_f32(const float* x, int64_t rows, int C, int64_t ld, float* out, int accumulate, float* ws,
                   svl_stream_t stream);

/* Elementwise: mode 0: out = a + b; 1: out = a * gelu'(b) (a = dY, b = pre-activation); 2: out = a * (b > 0) (relu bwd,
 * b = post-activation); 3: out = a * b; 4: out = a (copy); 5: out = gelu(a); 6: out = relu(a); 7: out = a / b. n elements. */
int svl_eltwise_f32(int mode, const float* a, const float* b, float* out, int64_t n, svl_stream_t stream);
/* out[r, c] = x[r, c] * mask[(r / rows_per_img) * C + c] * scale  — F.dropout2d on token layout (builder.py:79-85). */
int svl_chanmask_f32(const float* x, const float* mask, float scale, int64_t rows, int rows_per_img, int C,
                     float* out, svl_stream_t stream);
int svl_fill_f32(float* p, float v, int64_t n, svl_stream_t stream);
/* p[i] = 1 with probability keep_prob else 0: the per-(sample, channel) draws of F.dropout2d (builder.py:79-85; torch's
 * own generator stream is not reproduced -- the reference's masks are random too).  Counter-based: element i is a hash of
 * (seed, offset + i); the caller advances `offset` by n between calls. */
int svl_bernoulli_f32(float* p, int64_t n, float keep_prob, uint64_t seed, uint64_t offset, svl_stream_t stream);
/* y = ((x * k[0][c] + k[1][c]) - k[2][c]) / k[3][c] on NCHW planes (planes = B * C, k4 = float[4][C] device): the loader
 * (ImageNet) -> CLIP statistics re-normalisation of VLM.renormalize_img_for_clip (model/vlm.py:69-78), same op order. */
int svl_affine_planes_f32(const float* x, int64_t planes, int C, int64_t HW, const float* k4, float* y, svl_stream_t stream);
/* Strided row copy / gather / scatter / broadcast of `rows` rows of C floats:
 *   dst[(i / dgrp)*dst_go + (i % dgrp)*dst_ld + c] (=|+=) src[(i / sgrp)*src_go + (i % sgrp)*src_ld + c]
 * (token slicing x[:, 1:], cls-row scatter, torch.cat into channel slices, batch broadcast, strided grad adds). */
int svl_copy2d_f32(const float* src, int64_t sgrp, int64_t src_go, int64_t src_ld, float* dst, int64_t dgrp,
                   int64_t dst_go, int64_t dst_ld, int64_t rows, int C, int accumulate, svl_stream_t stream);
/* dst (contiguous [n0, n1, n2, n3]) = src read through the element strides (s0 .. s3): the weight-sized permutes between the
 * nn.Conv2d / nn.ConvTranspose2d parameter layouts ([Co, Ci, kh, kw], vlg_head.py:84-137) and the kernels' packs. */
int svl_permute4_f32(const float* src, float* dst, int64_t n0, int64_t n1, int64_t n2, int64_t n3, int64_t s0, int64_t s1,
                     int64_t s2, int64_t s3, svl_stream_t stream);
/* out2 = {max_r rnorm[r], max_i |bias[i]|} (bias may be null: 0): svl_pgemm_desc::b_bound of an fp16 x 2 planes output from
 * the row norms svl_split_planes_f16x2 leaves (F.linear's bias, maskclip_vit.py:110-144). */
int svl_bound2_f32(const float* rnorm, int64_t rows, const float* bias, int64_t nbias, float* out2, svl_stream_t stream);
/* Row permutation [outer, A, B, C] -> [outer, B, A, C] (C % 4 == 0): einops '(b n) (h w) c -> (b h w) n c' of the
 * SemanticTransformer (vlg_head.py:44-62) when its class sequences run on svl_attention_fwd / _bwd (N >= 64). */
int svl_permute_rows_f32(const float* src, int64_t outer, int A, int B, int C, float* dst, svl_stream_t stream);

/* GroupNorm (+ optional ReLU) on NHWC class-images: x [imgs, HW, C] (pixel stride ldx), groups of C/G channels,
 * stats [imgs, G, 2] = (mean, rstd); y pixel stride ldy (lets the result land in a concat slice). vlg_head.py:74-137 */
int svl_groupnorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int imgs,
                      int64_t HW, int C, int G, int relu, float* y, int64_t ldy, float* stats, svl_stream_t stream);
/* Conv2d(3x3, stride 1, pad 1, no bias; NHWC, optional second concat source read at image img / rep) of the Up blocks
 * FUSED with the statistics of the GroupNorm that follows it (vlg_head.py:120-127, groups of 16 channels): out [imgs H W, N]
 * (pixel stride ldo) and stats [imgs, N / 16, 2] = (mean, rstd), the latter from per-tile partial sums left by the
 * convolution's epilogue (ws: svl_conv3x3_gn_ws_doubles(...) doubles of device scratch) -- the separate statistics pass
 * over the result (4 B per element of HBM reads) disappears.  w = the forward pack [N, 9 (C1 + C2)].  Returns
 * SVL_ERR_UNSUPPORTED without launching anything when the spatially tiled kernel does not take the shape (N not 32 / 64,
 * channels not a multiple of 16, fewer than 16384 pixels): run svl_gemm_f32 + svl_groupnorm_fwd instead. */
int64_t svl_conv3x3_gn_ws_doubles(int imgs, int H, int W, int N);
int svl_conv3x3_gn_f32(const float* src1, int64_t ld1, int C1, const float* src2, int64_t ld2, int C2, int rep,
                       const float* w, int imgs, int H, int W, int N, float* out, int64_t ldo, float eps, double* ws,
                       float* stats, const float* gn_in, const void* w_planes, svl_stream_t stream);
/* The weights of a 3x3 convolution (forward pack [N, 9 Ct], or the input-gradient pack [Cin, 9 Cout]; N % 32 == 0,
 * Ct % 16 == 0) split ONCE into two fp16 planes scaled by a power of two per OUTPUT channel (w = 2^e[n] (h0 + h1), e[n] from
 * the channel's largest |w|), in the LDS image of the tiled kernel (per slab of 16 input channels: [plane][tap N + n][16]),
 * followed by the N int32 exponents -- svl_conv3x3_weight_planes_bytes(N, Ct) = 4 bytes per weight + 4 N.  Optional operand
 * `w_planes` of svl_conv3x3_gn_f32 / svl_conv3x3_dgrad_gnb_f32 and `conv_w_planes` of svl_gemm_desc (N a multiple of 32; the
 * dilated whole-image kernel takes N % 64 == 0): with it the tiled kernel
 * runs three fp16 MFMA products per term with a running per-tile exponent on the pixel operand (error below the exact fp32
 * kernel's on the step's shapes); null = the kernel splits the fp32 weights into three bf16 planes in every block (six
 * products, ~25 % slower).  Build it once per weight version (the reference has no counterpart: cuDNN re-lays its filters out
 * internally).  Replaces nothing of the reference by itself: vlg_head.py:120-127's Conv2d weights. */
int64_t svl_conv3x3_weight_planes_bytes(int N, int Ct);
int svl_conv3x3_weight_planes(const float* w, int N, int Ct, void* planes, svl_stream_t stream);
/* scsh [imgs][2][C]: the per-(image, channel) affine form of GroupNorm, y = fma(x, scsh[img][0][c], scsh[img][1][c]), from
 * statistics [imgs, G, 2] -- the table the tiled convolutions apply (+ ReLU) to a pre-normalisation operand (gn_in). */
int svl_groupnorm_scale_shift(const float* stats, const float* gamma, const float* beta, int imgs, int C, int G, float* scsh,
                              svl_stream_t stream);
/* The apply pass of svl_groupnorm_fwd alone, on given statistics: bit-identical y.  Backward uses it to re-materialise a
 * normalised activation from the kept pre-normalisation tensor instead of keeping both (no reference counterpart: autograd
 * keeps every intermediate, vlg_head.py:84-137). */
int svl_groupnorm_apply(const float* x, int64_t ldx, const float* gamma, const float* beta, int imgs, int64_t HW, int C, int G,
                        int relu, const float* stats, float* y, int64_t ldy, svl_stream_t stream);
/* dy has pixel stride lddy.  ReLU mask: from y (post-activation, stride ldy) when given; with y == NULL it is
 * re-derived from x, stats, gamma and beta with the forward's own fma (bit for bit the sign the forward saw) -- one tensor
 * pass less in each of the two kernels.  beta is only read in that case.
 * chan_sums [imgs, 2, C] out: per image (sum_p dy', sum_p dy' * xhat); dbeta / dgamma are their column sums
 * over images (svl_colsum_f32 with ld = 2C). */
int svl_groupnorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                      const float* stats, const float* gamma, const float* beta, int imgs, int64_t HW, int C, int G,
                      int relu, float* dx, int64_t lddx, float* chan_sums, svl_stream_t stream);
/* Round 6: the GroupNorm-backward statistics from the epilogue of the kernel that PRODUCES dy.  svl_conv3x3_dgrad_gnb_f32 = the
 * input gradient of a narrow 3x3 convolution (dgrad pack w [N, 9 C1], mirrored taps; N = 32 / 64) whose result out [pix, N] is
 * the dy of a GroupNorm + ReLU over gnb_x [pix, N] (same pixel stride ldo; vlg_head.py:120-127: conv -> GN -> ReLU -> conv):
 * besides out it leaves chan_sums [imgs][2][N] = (sum dy', sum dy' xhat) per (image, channel) -- what svl_groupnorm_bwd's first
 * pass computes from a read of dy and x.  gnb_table = that GroupNorm's [imgs][2][N] (scale, shift) (svl_groupnorm_scale_shift),
 * gnb_stats its [imgs][N / 16][2] (mean, rstd); ws = svl_conv3x3_gnb_ws_doubles(...) doubles of scratch.  Returns
 * SVL_ERR_UNSUPPORTED without launching when the split tiled kernel does not take the shape or the arithmetic mode is not 6
 * (callers then run svl_gemm_f32 + svl_groupnorm_bwd).  svl_groupnorm_bwd_apply = svl_groupnorm_bwd's second pass on given sums. */
int64_t svl_conv3x3_gnb_ws_doubles(int imgs, int H, int W, int N);
int svl_conv3x3_dgrad_gnb_f32(const float* dy, int64_t lddy, int C1, const float* w, int imgs, int H, int W, int N, float* out,
                              int64_t ldo, int accumulate, const float* gnb_x, const float* gnb_table, const float* gnb_stats,
                              double* ws, float* chan_sums, const void* w_planes, svl_stream_t stream);
int svl_groupnorm_bwd_apply(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* stats, const float* gamma,
                            const float* beta, int imgs, int64_t HW, int C, int G, int relu, const float* chan_sums, float* dx,
                            int64_t lddx, svl_stream_t stream);


/* Fused (flash-style) multi-head self-attention of the ViT blocks, head dim 64, softmax scale 64^-0.5, fp32 MFMA
 * (or the bf16 x 6 split emulation with fp32 accumulation under svl_set_gemm_emulation(6): same error level vs fp64)
 * (nn.MultiheadAttention inside mmcv's wrapper, maskclip_vit.py:77-84,141).  qkv [B*T, 3E] = in-proj output
 * (q | k | v, E = 64*H); out [B*T, E]; lse [B*H*T] (log-sum-exp per query, saved for backward; may be NULL).
 * Backward: dqkv [B*T, 3E] fully written; dsum_ws is a [B*H*T] float workspace.  Deterministic.
 * Optional packed bf16x3 planes of the results (the A-operand format of svl_gemm_planes_f32, emulation mode 6 only --
 * SVL_ERR_UNSUPPORTED otherwise): out_planes = the attention output [B*T, E] (then `out` may be NULL: gradient-free
 * passes only feed it to the out-projection GEMM); dq_planes = dqkv [B*T, 3E] (in_proj's input-gradient operand).
 * planes_rows = the buffers' padded row count (svl_planes_rows(B*T)).  NULL = not wanted. */
int svl_attention_fwd(const float* qkv, int B, int T, int H, float* out, float* lse, void* out_planes,
                      int64_t planes_rows, svl_stream_t stream);
int svl_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int H,
                      float* dsum_ws, float* dqkv, void* dqkv_planes, int64_t planes_rows, svl_stream_t stream);
/* The same two operations on fp16 x 2 PRE-PACKED operands (round 5; csrc/attn_h2.hip): a pack pass writes the (image, head)
 * slices of q / k / v / dout as two fp16 planes with one power-of-two scale per slice, in the fragment layouts the MFMA
 * kernels read (LDS-DMA, no operand split inside the loops); three products h1 b0 + a0 h1' + a0 b0 per term on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation replace the six of the bf16 x 3 form -- error vs fp64 at the level of the
 * fp32 kernels' (tests/test_ops_gpu.py).  `ws` = caller-provided scratch of svl_attention_h2_ws_bytes(B, T, H, backward)
 * bytes, 1 KiB aligned, private to the call's stream until the call's work has finished.  Results, argument meaning and the
 * optional bf16 x 3 planes outputs as above (the planes outputs need no emulation mode here).  Match
 * maskclip_vit.py:77-84,141 (q pre-scaled by 64^-0.5 -- exact --, fp32 softmax). */
int64_t svl_attention_h2_ws_bytes(int B, int T, int H, int backward);
int svl_attention_fwd_h2(const float* qkv, int B, int T, int H, float* out, float* lse, void* out_planes,
                         int64_t planes_rows, void* ws, int64_t ws_bytes, svl_stream_t stream);
int svl_attention_bwd_h2(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T, int H,
                         float* dsum_ws, float* dqkv, void* dqkv_planes, int64_t planes_rows, void* ws, int64_t ws_bytes,
                         svl_stream_t stream);

/* Small-sequence multi-head attention for the SemanticTransformer (vlg_head.py:39-67; seq = num classes).
 * qkv rows: token (g, s) at row  (g / inner) * outer_stride + (g % inner) * inner_stride + s * seq_stride,
 * each row = [q(E) | k(E) | v(E)], E = heads*D, D = 64. out has the same row mapping, width E. */
typedef struct svl_seqattn_desc {
  int groups, inner, seq, heads;
  int64_t outer_stride, inner_stride, seq_stride; /* in rows */
  const float* qkv;  /* [rows, 3E] */
  float* out;        /* [rows, E] */
  float* probs;      /* [groups, heads, seq, seq] saved for backward */
  /* backward */
  const float* dout; /* [rows, E] */
  float* dqkv;       /* [rows, 3E] */
  float* dscores;    /* workspace [groups, heads, seq, seq] */
} svl_seqattn_desc;
int svl_seqattn_fwd(const svl_seqattn_desc* d, svl_stream_t stream);
int svl_seqattn_bwd(const svl_seqattn_desc* d, svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Thin (HBM-bound) convolutions of the VLG head, NHWC, stride 1, same size.
 * ---------------------------------------------------------------------------------------------- */
/* Conv2d(C -> 1): y[pix] = bias[0] + sum_{tap,ci} x[pix + off(tap)][ci] * w[tap*C + ci]  (vlg_head.py:190,239; w is the
 * forward pack [1, KH*KW*C]). */
/* gn_in (NULL or the [imgs][2][C] table of svl_groupnorm_scale_shift; both entry points): x is the PRE-normalisation output
 * of the last Up convolution and relu(fma(x, scale, shift)) = GroupNorm + ReLU (vlg_head.py:126-127) is formed on the way
 * in -- the forward through an LDS-tiled kernel (3x3, pad 1, C = 16 / 32 / 64: one HBM read per input element). */
int svl_conv_cout1_fwd(const float* x, int64_t ldx, int imgs, int H, int W, int C, int KH, int KW, int dil, int pad,
                       const float* w, const float* bias, const float* gn_in, float* y, svl_stream_t stream);
/* Weight gradient of Conv2d(C -> 1, 3x3): per-block partial sums slabs[nblocks][9*C] (nblocks =
 * svl_conv_cout1_wgrad_blocks); combine with svl_reduce_slabs_f32. */
int svl_conv_cout1_wgrad_blocks(int imgs, int H, int W);
int svl_conv_cout1_wgrad(const float* dy, const float* x, int64_t ldx, int imgs, int H, int W, int C, int dil, int pad,
                         const float* gn_in, float* slabs, svl_stream_t stream);
/* out[p] = sum_tap T[p - sign*off(tap)][tap] with T [pix, KH*KW]: the shifted-tap sum that completes a Cin=1 input
 * gradient (conv1 7x7, vlg_head.py:169,221) after the GEMM T = dY . W. */
int svl_tap_gather(const float* T, int imgs, int H, int W, int KH, int KW, int dil, int pad, int sign, float* out,
                   svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Resampling on NHWC / NCHW maps.
 * ---------------------------------------------------------------------------------------------- */
/* Bilinear resize, channels-last: x [imgs, h, w, C] (pixel stride ldx) -> y [imgs*rep, H, W, :] written at channel
 * offset 0 with pixel stride ldy; output image i reads input image i / rep (`repeat`, vlg_head.py:133).
 * accumulate: y += . */
int svl_bilinear_nhwc_fwd(const float* x, int64_t ldx, int imgs, int h, int w, int C, int align_corners, int rep,
                          int H, int W, float* y, int64_t ldy, int accumulate, svl_stream_t stream);
/* dx[imgs,h,w,C] (=|+=) sum over rep and taps of dy. Deterministic gather form. */
/* out[g][row][c] = sum_{r < rep} src[g * rep + r][row][c]  (src rows `ld` apart, out packed [groups][rows][C]): the
 * class-repeated skip gradient of vlg_head.py:131-135 summed over its classes before the bilinear backward. */
int svl_sum_rep_f32(const float* src, int64_t ld, int64_t groups, int rep, int64_t rows, int C, float* out,
                    svl_stream_t stream);
int svl_bilinear_nhwc_bwd(const float* dy, int64_t lddy, int imgs, int h, int w, int C, int align_corners, int rep,
                          int H, int W, float* dx, int64_t lddx, int accumulate, svl_stream_t stream);
/* Bilinear resize of planes: x [planes, h, w] -> y [planes, H, W] (NCHW logits, vlg_head.py:247, builder.py:93-97). */
int svl_bilinear_planes_fwd(const float* x, int64_t planes, int h, int w, int align_corners, int H, int W, float* y,
                            svl_stream_t stream);
int svl_bilinear_planes_bwd(const float* dy, int64_t planes, int h, int w, int align_corners, int H, int W,
                            float* dx, svl_stream_t stream);
/* AvgPool PH x PW (floor) on NHWC + concat of a per-class text vector: x [imgs, H, W, C] ->
 * y [imgs, H/PH, W/PW, C + Ct], y[..., C:] = text[(img % nclass), :] (vlg_head.py:43-53; PH = H, PW = W is the global
 * average pool of ASPPPooling, vlg_head.py:70-81, on any map shape). text may be NULL (Ct=0). */
int svl_avgpool_cat_fwd(const float* x, int imgs, int H, int W, int C, int PH, int PW, const float* text, int Ct,
                        int nclass, float* y, svl_stream_t stream);
/* dx [imgs,H,W,C] (=|+=) avgpool backward of dy[..., :C] (ld = C+Ct); pixels outside the floor region get 0 (accumulate:
 * the pooled gradient is ADDED to dx -- the residual sum of SemanticTransformer / ASPP pooling without an extra pass). */
int svl_avgpool_cat_bwd(const float* dy, int imgs, int H, int W, int C, int PH, int PW, int Ct, float* dx,
                        int accumulate, svl_stream_t stream);
/* dtext [nclass, Ct] = sum over images of class n (img % nclass == n) and pooled pixels of dy[..., C + ct]; part: imgs * Ct
 * floats of scratch (per-image partial sums, added up per class in image order). */
int svl_avgpool_cat_bwd_text(const float* dy, int imgs, int64_t HWp, int C, int Ct, int nclass, float* part, float* dtext,
                             svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser (semivl.py:123-125,328,339-345: torch AdamW, one param group per tensor, poly LR).
 * Flat arena of `nseg` segments; seg_* arrays are DEVICE arrays of length nseg.
 * p,g,m,v are flat fp32 arenas; segment s covers [seg_off[s], seg_off[s+1]).
 * Update (torch.optim.AdamW, amsgrad=False):  p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr / (1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  gscale multiplies g first (1/world_size).
 * If ema != NULL: ema = ema_decay*ema + (1-ema_decay)*p_new (extension).
 * ---------------------------------------------------------------------------------------------- */
int svl_adamw_step(float* p, const float* g, float* m, float* v, const int64_t* seg_off, const float* seg_lr,
                   const float* seg_wd, int nseg, int64_t total, float beta1, float beta2, float eps, int step,
                   float gscale, float* ema, float ema_decay, svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * BatchNorm2d (batch statistics, SyncBN-ready) and MaxPool2d(3, 2, 1) on channels-last [rows, C] activations: the ops of
 * the Cityscapes recipe's convolutional side encoder (`conv_encoder` = mmseg ResNetV1c deep stem + layer1 with
 * norm_cfg SyncBN; reference model/vlm.py:50-53,120-121, configs/_base_/models/vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb.py:50-60).
 * Replaces torch.nn.SyncBatchNorm forward/backward (sum / sum-of-squares and sum dy / sum dy*xhat are produced as ONE
 * [2][C] double vector each, which is what the data-parallel ranks all-reduce) and F.max_pool2d + its backward.
 * C % 4 == 0, 16-byte aligned rows. */
int64_t svl_bn_ws_doubles(int64_t rows, int C);
/* sums[0][c] = sum_r x, sums[1][c] = sum_r x^2 */
int svl_bn_stats(const float* x, int64_t ldx, int64_t rows, int C, double* sums, double* ws, svl_stream_t stream);
/* mean, invstd from (all-reduced) sums over `count` rows; running stats updated like torch (unbiased variance) if given */
int svl_bn_finalize(const double* sums, double count, float eps, float momentum, float* running_mean, float* running_var,
                    int C, float* mean, float* invstd, svl_stream_t stream);
int svl_bn_eval_invstd(const float* running_var, float eps, int C, float* invstd, svl_stream_t stream);
/* y = [relu]((x - mean) * invstd * gamma + beta [+ resid]) */
int svl_bn_apply(const float* x, int64_t ldx, int64_t rows, int C, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, const float* resid, int64_t ldr, int relu, float* y, int64_t ldy,
                 svl_stream_t stream);
/* sums[0][c] = sum dy', sums[1][c] = sum dy' * xhat, dy' = dy masked by the fused ReLU: y > 0 when y is given; with
 * y == NULL and remask_gamma / remask_beta given, the forward's own output expression re-evaluated from x (bit-identical
 * sign; BatchNorm + ReLU WITHOUT a residual input only) -- one tensor pass less; all three NULL = no ReLU. */
int svl_bn_bwd_reduce(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                      const float* mean, const float* invstd, const float* remask_gamma, const float* remask_beta,
                      int64_t rows, int C, double* sums, double* ws, svl_stream_t stream);
/* dx = gamma * invstd * (dy' - sums0/count - xhat * sums1/count); dres (optional) = dy' for the residual branch;
 * dgamma = sums1, dbeta = sums0 (of this rank's rows, i.e. BEFORE the all-reduce) */
int svl_bn_bwd_apply(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                     const float* mean, const float* invstd, const float* gamma, const float* remask_beta,
                     const double* sums, double count, int64_t rows, int C, float* dx, int64_t lddx, float* dres,
                     int64_t lddr, svl_stream_t stream);
/* NHWC max pooling, kernel 3, stride 2, padding 1 (Ho = (H-1)/2+1); idx = winning tap 0..8 per output element */
int svl_maxpool3x3s2_fwd(const float* x, int imgs, int H, int W, int C, float* y, unsigned char* idx,
                         svl_stream_t stream);
int svl_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, int imgs, int H, int W, int C, float* dx,
                         svl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * GPU-side input pipeline: the reference loader's per-sample PIL chain
 * (third_party/unimatch/dataset/semi.py:61-127, transform.py:9-84) on uint8 HWC device images.  Statistical parity
 * (the random streams differ); the deterministic arithmetic follows Pillow.
 *   svl_aug_resample_u8   resize to rh x rw -- mode 0 / 1: Pillow BILINEAR (antialiased) / NEAREST (transform.py:43-57, the
 *                         img_scale=None branch); mode 2 / 3: OpenCV INTER_LINEAR / INTER_NEAREST semantics (mmseg `Resize`
 *                         = mmcv.imrescale, semi.py:53-71: the img_scale branch and the val transform; mmcv/cv2 are
 *                         un-vendored -> parity unpinned, +-1 level vs cv2's fixed-point arithmetic) -- then pad
 *                         right/bottom with `fill` (transform.py:9-14), crop OH x OW at (x0, y0) (:16-20), hflip (:25-29)
 *   svl_aug_to_float      ToTensor + Normalize (transform.py:32-40) -> float CHW; mean3/std3 are HOST pointers
 *   svl_aug_mask_i64      uint8 mask -> int64 with one value remapped (semi.py:122-123: 254 -> 255 / ignore_mask)
 *   svl_aug_photometric_u8  in place: 0 brightness, 1 contrast, 2 saturation (ImageEnhance blends, semi.py:99-100 via
 *                         torchvision ColorJitter), 3 hue shift, 4 grayscale (semi.py:101); scratch = 1 device uint64
 *   svl_aug_gaussian_blur_u8  transform.py:60-64 (true separable Gaussian; Pillow approximates it with box filters) */
int svl_aug_resample_u8(const unsigned char* src, int H, int W, int C, int rh, int rw, int x0, int y0, int OH, int OW,
                        int flip, int mode, int fill, unsigned char* dst, svl_stream_t stream);
int svl_aug_to_float(const unsigned char* src, int npix, const float* mean3, const float* std3, float* dst,
                     svl_stream_t stream);
int svl_aug_mask_i64(const unsigned char* src, int npix, int from, int to, int64_t* dst, svl_stream_t stream);
int svl_aug_photometric_u8(unsigned char* img, int npix, int op, float factor, unsigned long long* scratch,
                           svl_stream_t stream);
int svl_aug_gaussian_blur_u8(const unsigned char* src, int H, int W, float sigma, unsigned char* tmp, unsigned char* dst,
                             svl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMIVL_HIP_H_ */
