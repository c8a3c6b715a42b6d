/*
 * omg_hip.h — C ABI of libomg_hip.so: the MI355X (gfx950) kernels behind OMG's
 * two-stage SDXL denoising hot path (boundary row B5).
 *
 * The reference (kongzhecn/OMG) is pure Python with no native code; its FFI for
 * this path is "torch ops called from diffusers modules".  Each entry point
 * below names the reference call site (file:line under /root/reference, or the
 * diffusers==0.25.0 symbol that call site dispatches to) that it replaces.
 *
 * Conventions (all entry points)
 *   - return 0 on success, a negative OMG_E* code on error; never throw,
 *     never allocate, never synchronise, never touch the host heap: safe under
 *     hipGraph capture.
 *   - every pointer is a DEVICE pointer owned by the caller; sizes are in
 *     elements of the named dtype unless suffixed _bytes.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 *   - `dtype` is OMG_F16 or OMG_BF16: the storage type of activations and
 *     weights.  All contractions accumulate in fp32 on MFMA.
 *   - activations inside the UNet are NHWC ("pixels × channels", i.e. the
 *     (B, H*W, C) token layout the transformer blocks want); the NCHW latent
 *     layout of the reference exists only at conv_in / conv_out.
 */
#ifndef OMG_HIP_H
#define OMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMG_ABI_VERSION 6

enum { OMG_F16 = 0, OMG_BF16 = 1, OMG_F32 = 2 /* only where a signature says so */ };

enum {
  OMG_OK = 0,
  OMG_EINVAL = -1,   /* bad shape / alignment / null pointer          */
  OMG_EDTYPE = -2,   /* unsupported dtype                              */
  OMG_ELAUNCH = -3   /* hipLaunchKernel reported an error              */
};

/* activation applied in the GEMM / conv epilogue */
enum {
  OMG_ACT_NONE = 0,
  OMG_ACT_SILU = 1,  /* x*sigmoid(x)   — TimestepEmbedding.act (diffusers embeddings.py) */
  OMG_ACT_GEGLU = 2  /* out[:, j] = h[:, j] * gelu(h[:, N/2 + j]); W rows pre-packed by omg_pack_geglu_rows */
};

int omg_abi_version(void);
/* message of the last failed call made by the CALLING thread (thread-local storage; valid until that thread's next failing call) */
const char* omg_last_error(void);

/* ------------------------------------------------------------------------
 * omg_gemm — C[M,N] = epi( A[M,K] · W[N,K]^T  (+ A2[M,K2] · W2[N,K2]^T) )
 *
 * Replaces every nn.Linear on the path: attn.to_q/to_k/to_v/to_out[0]
 * (src/pipelines/lora_pipeline.py:98-124), Transformer2DModel.proj_in/out,
 * FeedForward GEGLU + out Linear, ResnetBlock2D.time_emb_proj, TimestepEmbedding
 * (diffusers 0.25.0).  The second K-segment is the PEFT LoRA branch
 * `base(x) + s·B(A(x))` (concept_models.set_adapters, lora_pipeline.py:588-591)
 * folded into the same MFMA accumulator: A2 = x·A_c^T (rank-r), W2 = s·B_c.
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  int32_t M, N, K;            /* K % 8 == 0, N % 8 == 0                              */
  const void* A;  int64_t lda;
  const void* W;  int64_t ldw;
  /* optional LoRA segment (K2 == 0 disables) */
  const void* A2; int64_t lda2;
  const void* W2; int64_t ldw2;
  int32_t K2;
  int32_t a2_col_block;       /* >0: A2 column offset = (n / a2_col_block) * K2 (fused q|k|v) */
  /* per-sample ("group") selection.  M = groups * rows_per_group.            */
  int32_t groups;             /* >=1                                                  */
  int32_t rows_per_group;
  const int32_t* group_adapter; /* [groups] adapter id or -1 (no LoRA); NULL = id 0  */
  int64_t w_adapter_stride;   /* W_eff  = W  + id * stride (LoRA-down GEMM); 0 = shared */
  int64_t w2_adapter_stride;  /* W2_eff = W2 + id * stride                             */
  /* epilogue */
  const void* bias;           /* [N] or NULL                                          */
  const void* group_bias; int64_t ldgb; /* [groups, N] per-sample bias (temb) or NULL  */
  const void* residual; int64_t ldr;    /* [M, N_out] or NULL                          */
  int32_t act;                /* OMG_ACT_*                                            */
  float out_scale;            /* applied before the residual add                      */
  void* C; int64_t ldc;       /* [M, N_out]; N_out = N/2 for GEGLU else N             */
} omg_gemm_args;

int omg_gemm(const omg_gemm_args* a, void* stream);

/* ------------------------------------------------------------------------
 * omg_quant_mx8 / omg_gemm_mx8 — the same Linear layers with OCP MX fp8
 * operands (BASELINE.json north_star "MFMA bf16/fp8", configs[4]): elements
 * e4m3, one E8M0 scale (2^(s-127)) per 32 consecutive K elements, contraction
 * on v_mfma_scale_f32_32x32x64_f8f6f4 (the 5 PFLOP/s path of gfx950), fp32
 * accumulation, fp16 / bf16 output with the epilogues of omg_gemm.
 *
 * Scale layout (both operands): uint32 S[K/128][ld]; S[t][r] packs the four
 * scale bytes of row r for K blocks 4t .. 4t+3 (byte b = block 4t + b).
 *
 * omg_quant_mx8: X[M,K] (fp16 / bf16, row stride ldx elements) -> Q[M,K]
 * bytes (row stride ldq) + S.  Scale = smallest power of two with
 * amax / scale <= 448 (no element saturates).  K % 128 == 0, s_ld >= M.
 * Weights are quantised once with the same call (rows = output features).
 * ---------------------------------------------------------------------- */
int omg_quant_mx8(int32_t dtype, const void* x, int64_t ldx, int32_t M, int32_t K,
                  void* q, int64_t ldq, void* scales, int32_t s_ld, void* stream);

typedef struct {
  int32_t dtype;              /* type of C, bias, residual: OMG_F16 | OMG_BF16        */
  int32_t M, N, K;            /* K % 128 == 0, N % 8 == 0                             */
  const void* A; int64_t lda; /* e4m3 bytes [M, K], lda % 16 == 0                     */
  const void* a_scale; int32_t sa_ld;   /* uint32 [K/128][sa_ld], sa_ld >= M          */
  const void* W; int64_t ldw; /* e4m3 bytes [N, K] (per adapter), ldw % 16 == 0       */
  const void* w_scale; int32_t sw_ld;   /* uint32 [K/128][sw_ld]                      */
  int32_t groups, rows_per_group;       /* as omg_gemm                                */
  const int32_t* group_adapter;         /* [groups] weight slot per sample or NULL    */
  int64_t w_adapter_stride;   /* bytes (= elements) between weight slots; 0 = shared  */
  int64_t sw_adapter_stride;  /* scale columns between weight slots (normally N)      */
  const void* bias;           /* [N] or NULL                                          */
  const void* residual; int64_t ldr;
  int32_t act;                /* OMG_ACT_NONE | OMG_ACT_SILU | OMG_ACT_GEGLU          */
  float out_scale;
  void* C; int64_t ldc;
  /* c_scale != NULL (GEGLU only, N % 256 == 0): the result is written as the NEXT omg_gemm_mx8's A operand instead of 16 bits —
   * C = e4m3 bytes [M][N/2] (ldc in bytes, % 16), c_scale = uint32 [N/2/128][sc_ld] — bit-identical to omg_quant_mx8 of the
   * 16-bit result (FeedForward: GEGLU -> Linear, diffusers attention.py). */
  void* c_scale; int32_t sc_ld;
} omg_gemm_mx8_args;

int omg_gemm_mx8(const omg_gemm_mx8_args* a, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of an MX-fp8 NHWC feature map (omg_groupnorm_mx8's output) with MX-fp8 weights
 * (omg_quant_mx8 of the packed [Cout][ky][kx][Cin] weight, K = 9 Cin) on the block-scaled MFMA: the resnet convolutions
 * (conv1 with the time-embedding projection as per-sample bias, conv2 with the skip tensor as residual) of BASELINE configs[4].
 * Output, bias, group_bias and residual are fp16 / bf16 (`dtype`). */
typedef struct {
  int32_t dtype;
  int32_t B, H, W, Cin, Cout;        /* Cin % 128 == 0, Cout % 8 == 0                         */
  const void* X; const void* x_scale;      /* e4m3 [B*H*W][Cin]; uint32 [Cin/128][B*H*W]      */
  const void* Wq; const void* w_scale;     /* e4m3 [Cout][9*Cin]; uint32 [9*Cin/128][sw_ld]   */
  int32_t sw_ld, act;
  const void* bias;                  /* [Cout] or NULL                                        */
  const void* group_bias; int64_t ldgb;    /* [B][ldgb] per-sample bias or NULL               */
  const void* residual;              /* [B*H*W][Cout] or NULL                                 */
  float out_scale;
  void* Y;                           /* [B*H*W][Cout]                                         */
} omg_conv2d_mx8_args;
int omg_conv2d_mx8(const omg_conv2d_mx8_args* a, void* stream);

/* ------------------------------------------------------------------------
 * omg_conv2d — NHWC implicit-GEMM convolution (3x3 pad 1, or 1x1), stride 1|2,
 * optional fused nearest-2x upsample of the input and fused channel-concat of
 * two inputs (the UNet skip connection), same epilogues as omg_gemm.
 *
 * Replaces ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D.conv,
 * Upsample2D (F.interpolate nearest + conv) and torch.cat([h, res], dim=1) in
 * CrossAttnUpBlock2D/UpBlock2D — all inside the `self.unet(...)` call at
 * src/pipelines/lora_pipeline.py:546-566 (diffusers 0.25.0 unet_2d_blocks.py).
 * W is [Cout][ky][kx][C1+C2] (packed by the host from the diffusers OIHW key).
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  int32_t B, Hin, Win;        /* input spatial size (before upsample)                 */
  int32_t C1, C2;             /* channels of X1 and (optional) X2; both % 64 == 0     */
  int32_t Hout, Wout, Cout;   /* Cout % 8 == 0                                        */
  int32_t ksize;              /* 1 or 3                                               */
  int32_t stride;             /* 1 or 2                                               */
  int32_t upsample;           /* 1: conv runs on nearest-2x-upsampled input           */
  const void* X1; const void* X2;
  const void* W;              /* [Cout][ksize*ksize*(C1+C2)]                           */
  const void* bias;           /* [Cout] or NULL                                       */
  const void* group_bias; int64_t ldgb; /* [B, Cout] per-sample bias (time emb) or NULL */
  const void* residual;       /* NHWC [B,Hout,Wout,Cout] or NULL                      */
  float out_scale;
  void* Y;                    /* NHWC [B,Hout,Wout,Cout]                              */
  int32_t act;                /* OMG_ACT_NONE | OMG_ACT_SILU (ControlNetConditioningEmbedding)  */
} omg_conv2d_args;

int omg_conv2d(const omg_conv2d_args* a, void* stream);

/* ------------------------------------------------------------------------
 * omg_attn_fwd — softmax(Q K^T * scale) V, head_dim 64, never materialising
 * the probabilities, with prompt-to-prompt "probability borrowing".
 *
 * Replaces, in one kernel: attn.get_attention_scores (baddbmm + softmax),
 * the controller's in-place edit of the conditional half of the probabilities,
 * and torch.bmm(probs, value) — src/pipelines/lora_pipeline.py:114-116 with
 * src/prompt_attention/p2p_attention.py:28-40,124-138.  With mapper = I and
 * alpha = 1 (always, in OMG flows) the controller's
 * `probs[edit] := probs[base]` is "sample b attends with the Q,K of sample
 * qk_src[b] but its own V".  Also replaces F.scaled_dot_product_attention /
 * xformers in the concept UNet (src/ip_adapter/attention_processor.py:273,383,399)
 * — `accumulate` implements IPAttnProcessor2_0's `text + scale*ip` (:409).
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  int32_t B, heads, Nq, Nkv;  /* head_dim is 64                                       */
  const void* Q; int64_t ldq; int64_t q_bstride;   /* row stride / batch stride (elements) */
  const void* K; int64_t ldk; int64_t k_bstride;
  const void* Vt;             /* [B, heads, 64, Nkv_pad] from omg_transpose_v (its key order); columns >= Nkv MUST be zero (it writes
                                 them so): the kernels that read it mask no score, the padded keys cancel against those zeros.  May be
                                 NULL when V (below) is given and Nkv > 128 */
  int32_t Nkv_pad;            /* % 64 == 0                                            */
  const int32_t* qk_src;      /* device [B]: batch index supplying Q,K; NULL = identity */
  float scale;
  int32_t accumulate;         /* 0: O = out_scale*attn ; 1: O += out_scale*attn       */
  float out_scale;
  void* O; int64_t ldo; int64_t o_bstride;
  /* ABI 6: V ROW-MAJOR — [B, Nkv, (head, 64)] with row stride ldv and batch stride v_bstride (elements, both % 8 == 0), i.e. the V
   * columns of the fused QKV projection's output exactly as omg_gemm wrote them.  When given and Nkv > 128 (self-attention) the kernel
   * stages it like K and transposes on the LDS read (ds_read_b64_tr_b16): no omg_transpose_v pass, Vt / Nkv_pad may be NULL / 0.
   * With Nkv <= 128 pass Vt (the resident-K/V kernels want the V^T image); V alone is rejected there.  NULL = ABI 5 behaviour.
   * The row-major path addresses one (sample, head) slice of K / of V with 32-bit offsets: Nkv * ldk * 2 and Nkv * ldv * 2 bytes must stay below 2 GB
   * (OMG_EINVAL otherwise; 16 384 keys of a 3 840-wide fused projection are 126 MB). */
  const void* V; int64_t ldv; int64_t v_bstride;
} omg_attn_args;

int omg_attn_fwd(const omg_attn_args* a, void* stream);

/* V[B, Nkv, (head, 64)] (row stride ldv) -> Vt[B, heads, 64, Nkv_pad], zero padded.  mfma_key_order = 1 (what omg_attn_fwd
 * consumes): inside every group of 16 keys the order is [0-3, 8-11, 4-7, 12-15] — the operand order of the P·V MFMA, one
 * 16-byte LDS read per lane; 0: natural key order (a plain batched transpose, used by the VAE / text-encoder attention). */
int omg_transpose_v(int dtype, const void* V, int64_t ldv, int64_t v_bstride,
                    int B, int heads, int Nkv, int Nkv_pad, void* Vt, int mfma_key_order, void* stream);

/* ------------------------------------------------------------------------
 * Normalisation.  GroupNorm (NHWC, fp32 statistics, deterministic two-stage
 * reduction) replaces ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2D
 * .norm and UNet conv_norm_out + conv_act; LayerNorm replaces
 * BasicTransformerBlock.norm1/2/3 (diffusers 0.25.0 attention.py).
 * ---------------------------------------------------------------------- */
/* workspace: floats, at least omg_groupnorm_ws_floats(B, groups, HW) */
int64_t omg_groupnorm_ws_floats(int B, int groups, int HW);
/* X may be the channel-concat of X1 (C1) and X2 (C2) — pass X2=NULL,C2=0 otherwise */
int omg_groupnorm(int dtype, const void* X1, int C1, const void* X2, int C2,
                  int B, int HW, int groups, float eps,
                  const void* gamma, const void* beta, int silu,
                  float* workspace, void* Y, void* stream);
int omg_layernorm(int dtype, const void* X, int64_t ldx, int M, int C, float eps,
                  const void* gamma, const void* beta, void* Y, int64_t ldy, void* stream);
/* LayerNorm whose consumer is omg_gemm_mx8: the normalised row (rounded to `dtype` exactly as omg_layernorm stores it) is
 * written as MX-fp8 bytes Q[M, C] + stage-major scale dwords (layout of omg_quant_mx8).  C % 128 == 0. */
int omg_layernorm_mx8(int dtype, const void* X, int64_t ldx, int M, int C, float eps,
                      const void* gamma, const void* beta, void* Q, int64_t ldq,
                      void* scales, int s_ld, void* stream);

/* GroupNorm(+SiLU)(+concat) whose consumer is omg_conv2d_mx8 (the resnets' norm1 -> conv1, norm2 -> conv2, diffusers 0.25.0
 * resnet.py ResnetBlock2D.forward): y, rounded to `dtype` exactly as omg_groupnorm stores it, is written as MX-fp8 bytes
 * Q[B*HW][Cq] and per-pixel scales S[Cq/128][B*HW] dwords (byte j of a dword: the E8M0 scale of channels 128 k + 32 j .. + 31);
 * C = C1 + C2 must be a multiple of 32, Cq = C rounded up to a multiple of 128, and the Cq - C pad channels are written as
 * zeros with scale byte 0 (SDXL's 320- and 960-channel maps: Cq = 384, 1024; the convolution weight is padded alike). */
int omg_groupnorm_mx8(int dtype, const void* X1, int C1, const void* X2, int C2,
                      int B, int HW, int groups, float eps,
                      const void* gamma, const void* beta, int silu,
                      float* workspace, void* Q, void* scales, void* stream);

/* ------------------------------------------------------------------------
 * fp32 path of the VAE decode.  The reference upcasts the VAE before decoding ("it overflows in float16", upcast_vae at
 * src/pipelines/lora_pipeline.py:639-652); with torch 2's attention processor diffusers 0.25 then runs post_quant_conv, conv_in
 * and the mid block in fp16 and the UP BLOCKS, conv_norm_out and conv_out in fp32.  omg_conv2d_f32: NHWC fp32 convolution
 * (3x3 pad 1 or 1x1, stride 1, optional fused nearest-2x upsample of the input), fp32 weights [Cout][ky][kx][Cin], fp32 bias and
 * residual, on the f32-input MFMA (exact fp32 products, fp32 accumulation).  omg_groupnorm / omg_conv_out accept
 * dtype = OMG_F32 (fp32 storage, fp32 gamma / beta / weights); omg_cast_f32 is the decoder's `sample.to(upscale_dtype)`.
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t B, Hin, Win, Cin;   /* Cin % 32 == 0                                        */
  int32_t Hout, Wout, Cout;   /* Hout = Hin * (upsample ? 2 : 1); Cout % 4 == 0       */
  int32_t ksize, upsample;    /* X may exceed 4 GB: the kernel addresses it from the first image a 128-pixel tile touches; ONE image (Hin * Win * Cin * 4 bytes)
                                 must stay below ~2 GB and the packed weight below 2 GB (OMG_EINVAL otherwise) */
  const void* X; const void* W; const void* bias; const void* residual;   /* fp32; bias / residual may be NULL */
  void* Y;                    /* [B, Hout, Wout, Cout] fp32                           */
} omg_conv2d_f32_args;

int omg_conv2d_f32(const omg_conv2d_f32_args* a, void* stream);
int omg_cast_f32(int dtype, const void* X, float* Y, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * Boundary convolutions (NCHW latents <-> NHWC features).
 * conv_in : UNet2DConditionModel.conv_in  (4 -> C0, 3x3), input NCHW fp32|T; executed as a 64-column
 *           im2col + the MFMA GEMM (K = 36 padded to one 64-wide slice).
 * conv_out: UNet2DConditionModel.conv_out (C0 -> 4, 3x3) on the already
 *           normalised+SiLU'd NHWC features, output NCHW in fp32.
 * ---------------------------------------------------------------------- */
int omg_conv_in(int dtype, const void* X_nchw, int x_is_f32, int B, int Cin, int H, int W,
                const void* Wt /*[Cout][64]: (ky,kx,ci) order, zero padded to 64 columns*/, const void* bias, int Cout,
                void* workspace /* B*H*W*64 elements of dtype: im2col patches */,
                void* Y_nhwc, void* stream);
int omg_conv_out(int dtype, const void* X_nhwc, int B, int H, int W, int Cin,
                 const void* Wt /*[Cout][3][3][Cin]*/, const void* bias, int Cout,
                 float* Y_nchw, void* stream);

/* ------------------------------------------------------------------------
 * Small elementwise pieces of the time/text conditioning path
 * (diffusers embeddings.py get_timestep_embedding, flip_sin_to_cos=True,
 *  downscale_freq_shift=0; and `self.nonlinearity(temb)` in ResnetBlock2D).
 * ---------------------------------------------------------------------- */
int omg_timestep_embedding(int dtype, const float* t, int n, int dim, void* out, int64_t ldo, void* stream);
int omg_silu(int dtype, const void* x, void* y, int64_t n, void* stream);
/* y[i] += a[i]  (ControlNet residuals added to the UNet skip tensors: `sample += residual`,
 * diffusers unet_2d_condition.py, reached from lora_pipeline.py:546-556) */
int omg_add_inplace(int dtype, void* y, const void* a, int64_t n, void* stream);
/* ------------------------------------------------------------------------
 * VAE decode (row N1): the two pieces AutoencoderKL.decode needs beyond the UNet's kernels
 * (diffusers 0.25.0 models/autoencoder_kl.py `post_quant_conv`, models/attention_processor.py
 * `Attention` with ONE head of dim C over all H*W tokens; reached from lora_pipeline.py:635-661).
 *   omg_softmax_rows : X[r, 0:cols] = softmax(X[r, 0:cols] * scale) in place, fp32 math, rows of `ld` elements
 *                      (the (HW x HW) score matrix of the mid-block attention, produced and consumed by omg_gemm).
 *   omg_channel_mix  : NCHW fp32, Y[b, o, p] = bias[o] + sum_c Wm[o, c] * X[b, c, p]   (1x1 conv, Cin, Cout <= 8)
 * ---------------------------------------------------------------------- */
int omg_softmax_rows(int dtype, void* X, int64_t rows, int64_t cols, int64_t ld, float scale, void* stream);
int omg_channel_mix(const float* X, const float* Wm, const float* bias, int B, int Cin, int Cout, int64_t HW,
                    float* Y, void* stream);
/* dst[r, col0 : col0+cols] = src[r, 0:cols]  (row-wise copy with strides, elements) */
int omg_copy2d(int dtype, const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------
 * omg_fuse_cfg_step — region-masked noise fusion + classifier-free guidance +
 * scheduler update + next model input, one launch, zero host syncs.
 *
 * Replaces src/pipelines/lora_pipeline.py:568-607 (fusion, incl. get_region_mask
 * :674-681 and the nearest mask resize), :610-612 (CFG), :615 (scheduler.step)
 * and :491-492 (cat + scale_model_input of the next iteration); identical block
 * in src/pipelines/instantid_pipeline.py:618-707.
 *
 * noise_pred : fp32 [4,C,H,W] = [unc0, unc1, cond0, cond1]
 * region_pred: K pointers to fp32 [2,C,H,W] = [unc, cond] (NULL entry = concept without a mask)
 * masks      : K pointers to fp32 [Hm,Wm] full-resolution {0,1} masks (NULL = None)
 * coef       : device table [n_steps][4] = {cx, ce, cin_next, unused}; row = *step_idx
 *              latents' = cx*latents + ce*eps ; model_input' = cin_next * latents'
 * step_idx   : device int; read, and incremented when `advance` != 0
 * ---------------------------------------------------------------------- */
#define OMG_MAX_CONCEPTS 8
typedef struct {
  int32_t C, H, W;            /* latent channels (4) and size                         */
  int32_t Hm, Wm;             /* mask size (e.g. 1024 x 1024)                          */
  int32_t n_concepts;
  int32_t fuse;               /* 0: plain CFG step; 1: masked fusion (i > 15, stage 2) */
  float guidance_scale;
  const float* noise_pred;
  const float* region_pred[OMG_MAX_CONCEPTS];
  const float* masks[OMG_MAX_CONCEPTS];
  const float* coef;
  int32_t* step_idx;
  int32_t advance;
  float* latents;             /* fp32 [2,C,H,W], updated in place                     */
  int32_t out_dtype;          /* dtype of model_input_next                            */
  void* model_input_next;     /* [4,C,H,W] = cat([latents']*2) * cin_next  (may be NULL) */
  float* fused_noise_out;     /* optional fp32 [2,C,H,W]: the fused (unc1, cond1) — parity tap */
} omg_step_args;

int omg_fuse_cfg_step(const omg_step_args* a, void* stream);

/* out[0:n] = table[*step_idx * n : (*step_idx + 1) * n]: per-step conditioning (time/text embedding rows, hoisted out
 * of the loop by the host) selected with the DEVICE step counter, so a captured step graph has no host argument. */
int omg_gather_step(int dtype, const void* table, const int32_t* step_idx, void* out, int64_t n_per_step, void* stream);

/* model_input[4,C,H,W] (dtype) = cin * cat([latents]*2)  — first iteration of the loop */
int omg_scale_model_input(int dtype, const float* latents, const float* coef_cin, int n_per_sample, void* out, void* stream);

/* ------------------------------------------------------------------------
 * Protocol-mode pieces (materialised probabilities) so that ANY controller
 * object — including the reference's own AttentionReplace with a non-identity
 * mapper — can be driven through RegionControlNet_AttnProcessor's exact
 * sequence (lora_pipeline.py:114-116): scores -> softmax -> controller -> bmm.
 * ---------------------------------------------------------------------- */
/* P[bh, q, kv] = softmax_kv(scale * Q[bh,q,:] . K[bh,kv,:])   (Q/K as in omg_attn_fwd) */
int omg_attn_probs(const omg_attn_args* a, void* P /*[B*heads, Nq, Nkv] dtype*/, void* stream);
/* O[b, q, h*64+d] = sum_kv P[bh,q,kv] * V[b,kv,h*64+d] */
int omg_attn_apply_probs(int dtype, const void* P, const void* V, int64_t ldv, int64_t v_bstride,
                         int B, int heads, int Nq, int Nkv, void* O, int64_t ldo, int64_t o_bstride, void* stream);

/* ------------------------------------------------------------------------
 * EfficientViT LiteMLA (the segmentation hand-off between the two stages: /root/reference
 * src/efficientvit/models/nn/ops.py:335-455).  The qkv / grouped / proj 1x1 convolutions are omg_gemm launches; these are the rest.
 * omg_dwconv2d: depthwise ksize x ksize convolution of the multi-scale aggregation (ops.py:372-380), NHWC rows of ldx / ldy
 *   elements (so that it can read a column slice of the fused qkv buffer), weights [ksize*ksize][C] tap-major, stride 1, same padding.
 * omg_relu_linear_att: relu_linear_att (ops.py:405-441) in fp32 as the reference computes it.  QKV [B*HW][ld]: group g holds its
 *   q | k | v (dim each) at columns 3 dim g; OUT [B*HW][ldo]: group g at columns dim g.  dim in {8, 16, 32}.
 * ---------------------------------------------------------------------- */
int omg_dwconv2d(int dtype, const void* X, int64_t ldx, int B, int H, int W, int C, int ksize,
                 const void* Wt, const void* bias, void* Y, int64_t ldy, void* stream);
int64_t omg_relu_linear_att_ws_floats(int B, int groups, int dim, int HW);
int omg_relu_linear_att(int dtype, const void* QKV, int64_t ld, int B, int HW, int groups, int dim, float eps,
                        float* workspace, void* OUT, int64_t ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMG_HIP_H */
