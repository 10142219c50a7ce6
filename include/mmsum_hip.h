/* libmmsum_hip -- C ABI of the MI355X (gfx950) kernels behind the MultimodalSum training hot path.
 *
 * The reference (nc-ai/MultimodalSum) has no FFI/plugin interface: every device op on its hot
 * path is an implicit PyTorch/ATen/cuDNN/apex launch.  Each entry point below replaces the
 * launches of one reference call site (cited as /root/reference/<file>:<line>); the Python host
 * (multimodalsum_amd/) binds them with ctypes and keeps the reference's nn.Module signatures.
 *
 * Conventions: plain pointers to DEVICE memory owned by the caller (PyTorch), explicit sizes and
 * leading dimensions in ELEMENTS, `dtype` = MMSUM_F32 | MMSUM_BF16 for activations/weights
 * (statistics, gradients of parameters and optimiser state are always f32), `stream` = a
 * hipStream_t.  Every function only enqueues work: it never allocates, synchronises or throws,
 * keeps no mutable global state (the only statics are once-initialised kernel attributes and the
 * CU count of the device), and returns MMSUM_OK or a negative error code.
 *
 * `live_rows` (device int32, may be NULL) on the row-streaming entry points: the number of rows that
 * are live in this call; rows at and past it are neither read nor written.  The value is read on the
 * DEVICE when the kernel runs, so a HIP graph captured once for the row capacity (the host-side `M` /
 * `R` / `nrows` argument) serves batches with any number of valid tokens: the padding-free text encoder
 * and the cross-attention K/V projections work on compacted rows whose count changes every step.
 */
#ifndef MMSUM_HIP_H
#define MMSUM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only the entry points declared here are exported. */
#pragma GCC visibility push(default)

#define MMSUM_ABI_VERSION 10

enum { MMSUM_F32 = 0, MMSUM_BF16 = 1 };
enum { MMSUM_OK = 0, MMSUM_ERR_BAD_SHAPE = -1, MMSUM_ERR_BAD_DTYPE = -2, MMSUM_ERR_BAD_ALIGN = -3,
       MMSUM_ERR_WORKSPACE = -4, MMSUM_ERR_HIP = -5 };

/* mmsum_gemm flags */
#define MMSUM_GEMM_A_T     0x001  /* A(m,k) = A[k*lda+m] (else A[m*lda+k]) */
#define MMSUM_GEMM_B_T     0x002  /* B(n,k) = B[k*ldb+n] (else B[n*ldb+k]) */
#define MMSUM_GEMM_BIAS    0x004  /* + bias[n] (f32) */
#define MMSUM_EPI_NONE     0
#define MMSUM_EPI_GELU     1      /* aux <- pre-activation, C <- gelu(erf form) */
#define MMSUM_EPI_GELU_BWD 2      /* C <- v * gelu'(aux) */
#define MMSUM_EPI_RELU     3
#define MMSUM_EPI_RELU_BWD 4      /* C <- v * (aux > 0), aux = forward output */
#define MMSUM_GEMM_EPI(e)  ((e) << 3)
#define MMSUM_GEMM_ACCUM   0x040  /* C += */
#define MMSUM_GEMM_OUT_F32 0x080  /* C is f32 regardless of dtype */
#define MMSUM_GEMM_SLABS   0x100  /* split-K without atomics: slice s writes its partial to C + s*M*ldc (f32) */
#define MMSUM_GEMM_COLSUM  0x200  /* `bias` is an OUTPUT (f32 atomics: not bit-reproducible run to run).  bf16 NT fast path, plain
                                    or GELU_BWD epilogue, bf16 store: bias[n] += sum_m C[m][n] of the stored result (the bias
                                    gradient of the layer that produced the GEMM's input gradient).  Weight-gradient product
                                    (A_T | B_T, bf16, SLABS, 256x256 tiles): bias[m] += alpha * sum_k A(m,k), the bias gradient
                                    of the same Linear, summed from the operand tiles the kernel stages anyway */

#define MMSUM_GEMM_COLSUM2 0x400  /* with COLSUM on the NT path: `bias` holds 2 N floats and bias[N + n] += sum_m C[m][n]^2 as well -- the
                                    BatchNorm statistics of a convolution's output taken in the convolution's own epilogue */

#define MMSUM_GEMM_A_F32   0x800  /* dtype BF16, weight-streaming kernel only (M <= 64, plain epilogue, with OUT_F32): A is f32 (lda in f32
                                    elements); the kernel multiplies its bf16 hi and lo parts separately, so the product carries 16
                                    significant bits of A.  The decode step's LM head: final LayerNorm output and logits stay f32 */

int mmsum_abi_version(void);
/* First 16 hex digits of the SHA-256 over the library's sources (csrc/Makefile: HASH_SRCS): lets the host tell a stale
 * build from a fresh one. */
const char* mmsum_build_id(void);

/* C[m][n] = epi(alpha * sum_k A(m,k) B(n,k) + bias[n]) (+C).  Replaces every nn.Linear / F.linear
 * on the path and their autograd (modeling_multimodalsum.py:302,304,783-792,885,738-739,2281;
 * table_encoder.py:62,65,71-73; img_encoder.py:40; 1x1 and im2col'ed convolutions of ResNet101).
 * A2/ksplit: for k >= ksplit the A operand continues in A2 (K split over two tensors: the
 * torch.cat([text, table]) of :738-739 without the concat).  splitk > 1 needs OUT_F32 and either ACCUM
 * (f32 atomics into C) or SLABS (C = workspace of splitk partial slabs, summed by mmsum_slab_reduce:
 * deterministic, and cheaper than atomics).
 * live_rows: natural A (no A_T): live rows of A and C (M); weight-gradient layout (A_T | B_T): live reduction length K.
 * alpha_dev (device f32, may be NULL): alpha is multiplied by *alpha_dev when the kernel runs -- the upstream gradient of
 * the loss (loss.backward(g), loss / accumulation_steps) enters the LM-head backward products this way, with no host read.
 * workspace (device, 16-byte aligned, may be NULL) / workspace_bytes: scratch the library may use to cut the reduction of a SMALL
 * bf16 NT product (tile list <= half the CUs: the per-GPU batch 1 of multimodal_train.py:420 gives 1,152 decoder rows = 72 tiles)
 * into slices on otherwise idle CUs, met inside the launch (MMSUM_GEMM_WORKSPACE_BYTES covers every such product).  It must be
 * ZERO before its first use; the kernel leaves it reusable, so one workspace serves every product of a stream in turn -- but not
 * two streams at once.  The result does not depend on the order the slices arrive in.  NULL: such products run unsplit. */
#define MMSUM_GEMM_WORKSPACE_BYTES (4096L + 256L * 128 * 128 * 4)
int mmsum_gemm(int dtype, const void* A, long lda, const void* A2, long lda2, int ksplit, const void* B, long ldb,
               void* C, long ldc, const float* bias, void* aux, long ldaux, int M, int N, int K, float alpha,
               const float* alpha_dev, int flags, int splitk, const int* live_rows, void* workspace, long workspace_bytes, void* stream);

/* What mmsum_gemm would launch for these arguments (pure: no device work, pointers are only checked for alignment / NULL --
 * live_rows and alpha_dev too: kernel selection depends on whether they are given, e.g. the weight-streaming kernel takes
 * neither):
 * plan[0] = kernel family, plan[1] x plan[2] = block tile, plan[3] = workgroups launched (fewer than tiles * splitk:
 * persistent workgroups walk the tile list; more than tiles * splitk: the reduction is cut over a lent workspace of
 * workspace_bytes, see mmsum_gemm).  Tests use it to assert that a shape reaches the kernel they mean to cover. */
enum { MMSUM_PLAN_GENERIC = 0,   /* gemm_kernel: register-staged 128x128, f32 or bf16, any layout */
       MMSUM_PLAN_NT_RING = 1,   /* LDS-DMA kernels for bf16 K-contiguous operands: gemm_nt_w4_kernel (256x256 tiles, four waves) or
                                    gemm_nt_ring_kernel (smaller tiles) */
       MMSUM_PLAN_TN_RING = 2,   /* the same pair for reduction-major operands (weight gradients): gemm_tn_w4_kernel / gemm_tn_ring_kernel */
       MMSUM_PLAN_SKINNY = 3 };  /* gemm_skinny_kernel: M <= 128 weight-streaming (decode steps) */
int mmsum_gemm_plan(int dtype, const void* A, long lda, const void* A2, long lda2, int ksplit, const void* B, long ldb,
                    const void* C, long ldc, const float* bias, const void* aux, long ldaux, int M, int N, int K, int flags,
                    int splitk, const int* live_rows, const float* alpha_dev, long workspace_bytes, int* plan);

/* Two INDEPENDENT bf16 products of one shape in one launch (the decode step's alpha and beta projections, :738-739: each alone is 64
 * workgroups on a per-launch floor): C_i[M <= 64, N <= 4096] = A_i[M, ksplit] | A2_i[M, K - ksplit] . B_i[N, K]^T + bias_i, i = 0, 1.
 * K a multiple of 256, ksplit a multiple of 32 (0 = no second tensor).  Leading dimensions in elements. */
typedef struct { const void* A; const void* A2; const void* B; void* C; const float* bias; long lda, lda2, ldb, ldc; } mmsum_gemm_operands;
int mmsum_gemm_pair(const mmsum_gemm_operands* ops, int M, int N, int K, int ksplit, void* stream);

/* The decode step's products (generation.py; modeling_multimodalsum.py:783-792,885,738-739,302-304,2281 at one token per hypothesis):
 *   out[M <= 96, N] = gelu?(x[M,K] . W[N,K]^T + bias) (+ residual),  bf16 x / W (x f32 with MMSUM_GEMM_A_F32), bf16 or f32 (OUT_F32) out.
 * The REDUCTION is split over one-wave workgroups ((N / 16) x splitk of them: every CU pulls weights, every load of a slice in
 * flight at once); a column tile's slices meet through f32 slabs + an arrival ticket in `workspace` and the last arriver adds
 * them and runs the epilogue (agent-scope release / acquire: correct for any placement).  x2 / ksplit as in mmsum_gemm (K split
 * over two tensors), K and ksplit multiples of 256.  flags: MMSUM_GEMM_EPI(MMSUM_EPI_GELU), MMSUM_GEMM_OUT_F32, MMSUM_GEMM_A_F32.
 * workspace: mmsum_dec_gemm_workspace(M, N, K) bytes, ZERO before its first use; the kernel leaves the ticket words zero, so one
 * workspace serves every product of a stream in turn. */
long mmsum_dec_gemm_workspace(int M, int N, int K);
int mmsum_dec_gemm(const void* x, long ldx, const void* x2, long ldx2, int ksplit, const void* W, long ldw, const float* bias,
                   const void* residual, long ldres, void* out, long ldo, int M, int N, int K, int flags, void* workspace, void* stream);

/* 3x3 convolution, stride 1, padding 1, bf16, as an IMPLICIT GEMM (ResNet101's conv2 of every bottleneck but the two strided ones,
 * torchvision resnet101 as used by img_encoder.py:21-24,31-35): y[(n,y,x)][co] = sum_{ky,kx,c} xp[n][y+ky][x+kx][c] w[co][(3 ky + kx) C + c].
 *   xp : the input in the PADDED NHWC layout [n, H+2, W+2, C] with ZERO borders (mmsum_bn_apply writes it: pad_H / pad_W);
 *   w  : [Cout, ldw >= 9 C] in (ky, kx, c) column order (mmsum_conv_weight_permute);  y : [n*H*W, Cout], compact rows.
 *   stats (f32 [2 Cout], may be NULL) += column sums of y and of y^2 as stored (the BatchNorm statistics, as MMSUM_GEMM_COLSUM2).
 * The LDS-DMA pieces of the NT kernels read C-contiguous runs of one tap straight from xp: no im2col matrix.  C a power of two >= 64.
 *   live_rows (device int32, may be NULL): output rows (n_run * H * W, mmsum_image_plan) -- rows at and past it are neither computed nor summed. */
int mmsum_conv3x3_gemm(const void* xp, const void* w, long ldw, void* y, long ldy, float* stats, int n, int H, int W, int C,
                       int Cout, const int* live_rows, void* stream);
/* Weight gradient of that convolution, no im2col matrix either: out[co][(3 ky + kx) C + c] = sum over pixels dy[pixel][co] x[pixel + (ky-1, kx-1)][c]
 * (the layout mmsum_conv_weight_permute(to_matrix = 0) turns into the [Cout, C, 3, 3] gradient).  dyp [n, H+2, W+2, Cout] and xp [n, H+2, W+2, C]
 * both PADDED with zero borders (mmsum_bn_bwd_apply's dx_pad / mmsum_bn_apply's pad): the four-wave reduction-major kernel runs over all
 * padded positions and shifts its xp rows by the tap of the tile.  C a power of two >= 256.  out: f32 [Cout, ldo >= 9 C], or with
 * splitk > 1 splitk such slabs (slab s at out + s * Cout * ldo; mmsum_slab_reduce adds them).
 * live_positions (device int32, may be NULL): the reduction's length, n_run * (H+2) * (W+2) - 2 * (W + 3) for the first n_run images. */
int mmsum_conv3x3_wgrad(const void* dyp, const void* xp, float* out, long ldo, int n, int H, int W, int C, int Cout, int splitk,
                        const int* live_positions, void* stream);

/* out[r][c] (+)= sum_s ws[s][r][c] over nslabs f32 slabs of [rows, cols] (split-K reduction). */
int mmsum_slab_reduce(const float* ws, int nslabs, int rows, int cols, float* out, long ldo, int accumulate, void* stream);

/* out[c] (+)= sum_r X[r][c]  (bias gradients; BatchNorm reductions).  partial: f32 workspace of
 * mmsum_colsum_workspace(C) bytes. */
long mmsum_colsum_workspace(int C);
int mmsum_colsum(int dtype, const void* X, long ld, int R, int C, float* out, int accumulate, void* workspace,
                 const int* live_rows, void* stream);

/* Dropout (F.dropout, modeling_multimodalsum.py:294,305,371,458,474,486,596): masks are a hash of (seed, element index);
 * the reference draws them from the torch generator, so bit parity of masks is not defined and parity runs use p = 0.
 * `salt` (device uint64, may be NULL) on every dropout entry point: the kernel uses seed + *salt * golden-ratio constant.
 * A captured HIP graph replays the seed ARGUMENTS it was captured with; bumping the salt with mmsum_bump_u64 (a one-thread
 * kernel, itself capturable) as the first node of the forward graph gives every replay fresh masks while the backward
 * graphs of the same step regenerate the same ones. */
int mmsum_bump_u64(void* dev_u64, unsigned long long inc, void* stream);

/* K1/K7: y = dropout(LN(E[ids] + P[t+pos_offset] + rating_diff[seq]*rvec))
 * (modeling_multimodalsum.py:368-372, 581-597).  ids [nseq*T] int64; rating_diff/rvec may be NULL. */
int mmsum_embed_ln_fwd(int dtype, const int64_t* ids, const void* E, const void* P, const float* rating_diff,
                       const void* rvec, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                       int nseq, int T, int D, int pos_offset, float eps, float p_drop, uint64_t seed, const void* salt,
                       void* stream);
/* backward of the above: scatter-adds into dE (skipping pad_id rows), dP, drvec, dgamma, dbeta (all f32). */
int mmsum_embed_ln_bwd(int dtype, const void* dy, const int64_t* ids, const void* E, const void* P,
                       const float* rating_diff, const void* rvec, const void* gamma, const float* mean,
                       const float* rstd, float* dE, float* dP, float* drvec, float* dgamma, float* dbeta, int nseq,
                       int T, int D, int pos_offset, int pad_id, float p_drop, uint64_t seed, const void* salt, void* stream);

/* Padding-free text encoder (MI355X-side restructuring, results identical): encoder rows that are padding never reach a
 * result (their keys are masked, :836-837; their outputs are masked again in the decoder's cross-attention, :858-866), so
 * the fused step runs the encoder's GEMM / LayerNorm work on the valid rows only and moves rows between the padded
 * layout the attention kernel reads and the compact one with this gather:
 *   dst[i, :] = map[i] >= 0 ? src[map[i], :] : 0   (i < nrows; row_bytes multiple of 16; pitches in bytes). */
int mmsum_rows_gather(const void* src, long src_pitch, int src_rows, void* dst, long dst_pitch, const int64_t* map, int nrows,
                      int row_bytes, const int* live_rows, void* stream);

/* K4/K6/K21: y = LN(res + dropout(x))  (modeling_multimodalsum.py:294-297,305-308,458-461,474-477,486-489;
 * apex FusedLayerNorm :972-980). */
/* y_f32 (may be NULL): the same result as un-rounded f32 [R, D] beside y (bf16 mode: the decode step's last LayerNorm feeds the LM
 * head in f32). */
int mmsum_add_ln_fwd(int dtype, const void* x, const void* res, const void* gamma, const void* beta, void* y,
                     float* mean, float* rstd, int R, int D, float eps, float p_drop, uint64_t seed, const void* salt,
                     const int* live_rows, float* y_f32, void* stream);
/* dres <- dz (or += if accumulate_dres), dx <- dz * dropmask/(1-p); dgamma/dbeta += (f32);
 * dxsum (f32 [D], may be NULL) += column sums of dx = the bias gradient of the Linear that produced x (:302,304,885). */
int mmsum_add_ln_bwd(int dtype, const void* dy, const void* x, const void* res, const void* gamma, const float* mean,
                     const float* rstd, void* dx, void* dres, int accumulate_dres, float* dgamma, float* dbeta, int R,
                     int D, float p_drop, uint64_t seed, const void* salt, float* dxsum, const int* live_rows, void* stream);

/* Entity attention (K3, K8, K11; modeling_multimodalsum.py:752-875).  One description covers
 * encoder self-attention, causal decoder self-attention and the per-entity cross-attention with
 * entity mean:
 *   query block qb (T rows, row = qb*T + t, T <= 128) belongs to business b = qb / qpb; it attends,
 *   entity by entity, to entities n = 0..N-1 of b (rows mem_row0 + ((b*N+n)*S + s), S <= 224, N <= 32),
 *   skipping n == qb % qpb when `exclude_self` (leave-one-out) and entities with null[b*N+n] != 0;
 *   out = mean over the attended entities of softmax_s(scale * q.k + mask) v   (0 if none).
 *   pad [B*N*S] uint8 (1 = masked key) or NULL; causal: key s > query t (+ causal_q0) masked (self-attention).
 * head_dim is 64; q/k/v/out are head-merged [rows, H*64] with leading dimensions ldq/ldk/ldv/ldo. */
typedef struct {
    const void* q; const void* k; const void* v; void* out;
    long ldq, ldk, ldv, ldo;
    const uint8_t* pad; const uint8_t* null_entity;
    int n_qblocks, T, qpb, N, S, H;
    int exclude_self, causal;
    float scale;
    /* Optional row maps (int32, device memory; bf16 only).  With a map the padded layouts above are LOGICAL: the matrices hold
     * only the rows that exist (the padding-free encoder's [live rows, 3D] q/k/v; the compacted memory's K/V), in any order.
     *   q_rows [n_qblocks*T]: physical row of q / out (and of dout / dq in the backward pass) of logical query row qb*T + t, or -1;
     *   kv_rows [B*N*S]: physical row of k / v (dk / dv) of logical key row (b*N+n)*S + s, or -1.
     * A missing query row reads as zeros and is never written.  A missing key row must be a masked key (pad != 0); its dk / dv
     * rows do not exist.  Physical row * row pitch must stay below 2 GiB.  NULL = identity (the padded layout itself). */
    const int* q_rows; const int* kv_rows;
    /* causal only: key position of the query block's first row (a multiple of 32, 0 = the usual self-attention): key s is masked
     * for query row t when s > causal_q0 + t.  A causal sequence of 129 .. 224 positions runs as its first 128 queries (causal_q0 = 0
     * over the first 128 keys) + the remaining ones as a second query block with causal_q0 = 128 over all keys. */
    int causal_q0;
} mmsum_attn_desc;
int mmsum_entity_null(const uint8_t* pad, uint8_t* null_entity, int n_entities, int S, void* stream);
int mmsum_attn_fwd(int dtype, const mmsum_attn_desc* d, void* stream);
/* Backward: dq [n_qblocks*T, H*64] (accumulated if accumulate_dq), dk/dv [entity rows, H*64]
 * overwritten for every entity row of the modality.  stats: f32 workspace of
 * mmsum_attn_bwd_workspace() bytes (per-entity log-sum-exp and delta handed from the dQ kernel
 * to the dK/dV kernel). */
long mmsum_attn_bwd_workspace(const mmsum_attn_desc* d);
int mmsum_attn_bwd(int dtype, const mmsum_attn_desc* d, const void* dout, long lddo, void* dq, long lddq,
                   int accumulate_dq, void* dk, long lddk, void* dv, long lddv, void* stats, void* stream);

/* The decode step's cross-attention over the CACHED K / V of every modality in one launch (generation.py; reference: the cached branch
 * of SelfAttention.get_head_output, modeling_multimodalsum.py:794-815,819-869, at one query per hypothesis).  dtype: MMSUM_BF16 (the
 * timed mode) or MMSUM_F32 (the parity mode: q / k / v / out f32, pitches in elements).  q [B * qpb, H*64] bf16:
 * the qpb hypotheses of business b are rows b*qpb ..; modality m: k / v [B * N * S rows, pitch ldkv] bf16 (entity (b, n) at row
 * (b*N + n)*S), pad [B*N*S] uint8 (1 = masked key, filled with -2^16 like the reference) or NULL, null_entity [B*N] uint8 or NULL.
 * out [nmod * B*qpb, H*64] bf16: row m * B*qpb + r = the entity MEAN of modality m for hypothesis r (null entities dropped, zeros
 * when all are null).  One workgroup per (entity, head); the mean crosses workgroups through `workspace`
 * (mmsum_decode_cross_attn_workspace bytes, ZERO before its first use; the kernel leaves its ticket words zero). */
typedef struct { const void* k; const void* v; const uint8_t* pad; const uint8_t* null_entity; int N, S; } mmsum_xattn_memory;
long mmsum_decode_cross_attn_workspace(int n_entities, int H, int qpb, int B, int nmod);
int mmsum_decode_cross_attn(int dtype, const void* q, long ldq, const mmsum_xattn_memory* mods, int nmod, long ldkv, void* out, long ldo,
                            int B, int qpb, int H, float scale, void* workspace, void* stream);

/* K13 elementwise part (modeling_multimodalsum.py:732-744): given pre-activations pa, pb,
 * out = yt + relu(tanh(pa))*[!no_table[b]]*ytab + relu(tanh(pb))*[!no_img[b]]*yimg; row r -> b = r / rows_per_b. */
int mmsum_gate_fwd(int dtype, const void* pa, const void* pb, const void* yt, const void* ytab, const void* yimg,
                   const uint8_t* no_table, const uint8_t* no_img, void* out, int R, int D, int rows_per_b,
                   void* stream);
/* The decode step's form of the two lines above and the LayerNorm behind them in one launch (generation: nothing saved):
 * y = LN(res + yt + relu(tanh(pa)) [table] ytab + relu(tanh(pb)) [image] yimg)  (:732-744 then :474-477); D in {256, 512, 768, 1024}. */
int mmsum_gate_add_ln_fwd(int dtype, const void* pa, const void* pb, const void* yt, const void* ytab, const void* yimg,
                          const uint8_t* no_table, const uint8_t* no_img, const void* res, const void* gamma, const void* beta,
                          void* y, int R, int D, int rows_per_b, float eps, void* stream);
/* Backward of the gate.  sum_dpa / sum_dpb (f32 [D], both or both NULL): += column sums of dpa and of dpb -- the bias
 * gradients of alpha_proj and beta_proj (:738-739), taken while the rows are in registers instead of by two more passes over
 * them (f32 atomics: not bit-reproducible run to run). */
int mmsum_gate_bwd(int dtype, const void* dout, const void* pa, const void* pb, const void* ytab, const void* yimg,
                   const uint8_t* no_table, const uint8_t* no_img, void* dpa, void* dpb, void* dyt, void* dytab,
                   void* dyimg, int R, int D, int rows_per_b, float* sum_dpa, float* sum_dpb, void* stream);

/* K16: label-smoothing loss (/root/reference/src/utils.py:32-38), fused forward + backward:
 * row_loss[r] = -sum_v true_dist*log_softmax(logits[r,:V]); logits[r,:] <- gscale*(softmax - true_dist)
 * in place (columns V..ld-1 are zeroed).  smoothing == 0 gives nn.CrossEntropyLoss rows. */
int mmsum_ls_loss(int dtype, void* logits, long ld, const int64_t* target, float* row_loss, int R, int V,
                  float smoothing, float gscale, int write_grad, void* stream);
/* out[s] = scale * sum(x[s*seg : (s+1)*seg]) -- deterministic (per-pass losses, mean loss). */
int mmsum_segment_sum(const float* x, float* out, int nseg, int seg, float scale, void* stream);

/* K22: out[0] = sum g^2 over n f32 elements (clip_grad_norm_, multimodal_train.py:361-362).
 * workspace: mmsum_l2_workspace() bytes. */
long mmsum_l2_workspace(void);
int mmsum_l2norm_sq(const float* g, long n, float* out, int accumulate, void* workspace, void* stream);
/* K23: HF AdamW (transformer/optimization.py:240-265) over a flat f32 arena slice.
 * hyper (device, f32[4]) = {step_size = lr*sqrt(bc2)/bc1, lr*weight_decay, max_grad_norm (<=0: no clip), unused};
 * norm_sq (device) = total squared grad norm; grads are scaled by min(1, max_norm/(sqrt(norm_sq)+1e-6))
 * on the fly (the stored gradient is left unscaled).  shadow (bf16, may be NULL) <- bf16(p). */
int mmsum_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, const float* hyper,
                const float* norm_sq, float beta1, float beta2, float eps, void* stream);
/* dst <- cast(src) over n elements; dtype_dst/dtype_src in {F32,BF16}. */
int mmsum_cast(int dtype_dst, void* dst, int dtype_src, const void* src, long n, void* stream);
/* dst[i] = scale * src[i] (f32), used to apply the clip coefficient to stored gradients when the
 * caller (torch.nn.utils.clip_grad_norm_ drop-in) needs them scaled in place. */
int mmsum_scale_by_clip(float* g, long n, const float* norm_sq, float max_norm, void* stream);

/* bf16 transposes feeding the NT GEMM: dst[c][r] = src[r][c]; dst columns [rows, rows_pad) are zero
 * filled (reduction padding for wgrad); colsum (may be NULL): colsum[c] += sum_r src[r][c] (the bias gradient comes
 * for free while dy is being transposed for its weight-gradient product).  Batched form: desc[i] = {src_off, dst_off, rows, cols, ld_src,
 * ld_dst} in elements (device memory), one matrix per entry (all 2-D weights after an optimiser step). */
int mmsum_transpose_bf16(const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols, int rows_pad, float* colsum,
                         void* stream);
int mmsum_transpose_bf16_batched(const void* src_base, void* dst_base, const long* desc, int n, int max_tiles, void* stream);

/* ---- ResNet101 stages (img_encoder.py:21-24,31-41; torchvision 0.6.1 resnet101) -------------
 * Activations are NHWC.  A KxK convolution is im2col (this kernel) + mmsum_gemm; 1x1 stride-1
 * convolutions are plain GEMMs on the NHWC matrix.  Column order of the im2col matrix is
 * (kh, kw, c), c fastest; Kpad >= kh*kw*C columns (zero-filled tail). */
int mmsum_im2col(int dtype, const void* x, void* col, int N, int H, int W, int C, int KH, int KW, int stride,
                 int pad, int Ho, int Wo, int Kpad, const int* images, void* stream);
/* dx[n,h,w,c] = sum of dcol entries that im2col copied from x[n,h,w,c] (gather form, no atomics). */
int mmsum_col2im(int dtype, const void* dcol, void* dx, int N, int H, int W, int C, int KH, int KW, int stride,
                 int pad, int Ho, int Wo, int Kpad, const int* images, void* stream);
/* f32 [Cout, Cin, KH, KW] <-> dtype [Cout, Kpad] with (kh,kw,c) column order.
 * to_matrix=1: weight -> matrix (cast, zero tail); to_matrix=0: f32 matrix gradient -> weight-layout
 * gradient (+= if accumulate); to_matrix=2: weight -> the INPUT-GRADIENT matrix dtype [Cin, Kpad >= KH*KW*Cout],
 * column ((KH-1-kh) KW + (KW-1-kw)) Cout + co = w[co][ci][kh][kw] (rotated by 180 degrees, channel roles exchanged): the
 * input gradient of a stride-1 convolution is mmsum_conv3x3_gemm of the padded output gradient with this matrix. */
int mmsum_conv_weight_permute(int dtype, void* matrix, float* weight, int Cout, int Cin, int KH, int KW, int Kpad,
                              int to_matrix, int accumulate, void* stream);
/* BatchNorm2d (train mode, batch statistics over R = N*H*W rows of an [R, C] NHWC matrix).
 * stats: sums[2*C] f32 = {mean, biased variance} produced by mmsum_bn_reduce (pivot-shifted sums, no
 * catastrophic cancellation); bn_apply normalises with
 * them (y = relu?(gamma*(x-mean)*rstd + beta (+ residual))) and updates running stats (momentum,
 * unbiased variance) when running_mean != NULL. */
long mmsum_bn_workspace(int C);
/* `images` + `rows_per_image` of the BatchNorm entry points and `images` of the im2col / pooling / layout kernels = the LIVE-IMAGE WINDOW of
 * the fused step's image branch (device int32 plan[0..2] of mmsum_image_plan; NULL = every image): only the first images[0] images' rows are
 * read and written; image images[1] (>= 0) is the representative of the batch's images[2] empty slots -- the statistics count its rows
 * images[2] times (R stays the row count of ALL images), and its gradient rows travel through the backward pass multiplied by images[2]. */
int mmsum_bn_reduce(int dtype, const void* x, int R, int C, float* sums, void* workspace, const int* images, int rows_per_image, void* stream);
/* raw (the plain column sums a convolution's GEMM epilogue left over the rows that ran) += (images[2] - 1) * {sum y, sum y^2} over the
 * representative's rows of y [R, C]: the statistics of all R rows.  No work when the batch has no representative. */
int mmsum_bn_rep_fix(int dtype, const void* y, float* raw, int R, int C, const int* images, int rows_per_image, void* stream);
/* Which slots of a batch of n images [n, elems_per_image] f32 (mask [n] uint8, 1 = a real image) the image branch runs (reference:
 * /root/reference/src/data_utils.py:54-65 pads every business to the batch's image count with all-zero images, img_mask False; the reference
 * pushes them through ResNet101 and its batch statistics like the real ones, /root/reference/src/multimodal_train.py:186-190).  A slot that is
 * masked AND all zero is EMPTY; run order = the non-empty slots in batch order, then ONE representative of the empty ones:
 *   plan [4 + n_rpi] int32 = {images that run, index of the representative among them (-1: none), number of empty slots (1 if none),
 *                             non-empty slots, max(0, images that run * rows_per_image[k] + row_adjust[k]) ... (device row counts for the
 *                             live_rows arguments of the GEMM entry points)}
 *   src [n] int32           = slot whose image runs as image r (mmsum_nchw_to_nhwc gathers through it)
 *   slot_rows [n*positions] = for slot row (slot, p): the row of the run-order result [n*positions, .] it takes (mmsum_rows_gather map)
 *   run_rows  [n*positions] = for run-order row (r, p): the slot row whose output gradient it receives, -1 (zero) for the representative
 * workspace: mmsum_image_plan_workspace(n) bytes.  n <= 8192, n_rpi <= 8. */
long mmsum_image_plan_workspace(int n);
int mmsum_image_plan(const float* img, long elems_per_image, const uint8_t* mask, int n, int positions, const int* rows_per_image,
                     const int* row_adjust, int n_rpi, int* plan, int* src, int64_t* slot_rows, int64_t* run_rows, void* workspace, void* stream);
/* BatchNorm batch statistics from plain column sums: raw = {sum_r x[r][c], sum_r x[r][c]^2} (2 C floats, as the convolution's GEMM
 * epilogue leaves them with MMSUM_GEMM_COLSUM | MMSUM_GEMM_COLSUM2) -> sums = {mean, biased variance}; running_mean / running_var
 * (may be NULL) receive the momentum update with the unbiased variance (torchvision BatchNorm2d in train mode,
 * /root/reference/src/img_encoder.py:21-41).  Pass running_mean = NULL to mmsum_bn_apply afterwards: the update is done here. */
int mmsum_bn_stats_from_sums(const float* raw, int R, int C, float* sums, float* running_mean, float* running_var, float momentum,
                             void* stream);
/* pad_H, pad_W (0, 0 = no): y is written in the zero-bordered PADDED layout [n, pad_H + 2, pad_W + 2, C] of an [n, pad_H, pad_W] image
 * (R = n pad_H pad_W): pixel (n, y, x) lands at row n (pad_H+2)(pad_W+2) + (y+1)(pad_W+2) + x + 1 -- the operand layout of
 * mmsum_conv3x3_gemm.  The borders are NOT written: the caller zeroes the buffer once.  x and residual stay compact.
 * raw (f32 [2 C], may be NULL; training only): the statistics as PLAIN column sums {sum x, sum x^2} over the R rows, as the convolution's
 * GEMM epilogue leaves them (MMSUM_GEMM_COLSUM | MMSUM_GEMM_COLSUM2): the kernel derives {mean, biased variance} itself, WRITES them to
 * `sums` for the backward pass and updates the running statistics -- no statistics launch (mmsum_bn_stats_from_sums does the same alone). */
int mmsum_bn_apply(int dtype, const void* x, float* sums, const float* raw, const float* gamma, const float* beta,
                   const void* residual, void* y, float* running_mean, float* running_var, int R, int C, float eps,
                   float momentum, int relu, int training, int pad_H, int pad_W, const int* images, int rows_per_image, void* stream);
/* BN backward, two launches: bn_reduce over (dy', dy'*xhat) via mmsum_bn_bwd_reduce, then bn_bwd_apply.
 * dy' = dy * (y > 0) when relu (y = forward output; pad_H / pad_W: y is in the padded layout mmsum_bn_apply wrote).
 * dsums[2*C] = {sum dy', sum dy'*xhat}. */
int mmsum_bn_bwd_reduce(int dtype, const void* dy, const void* y, const void* x, const float* sums, int R, int C,
                        float eps, int relu, float* dsums, void* workspace, int pad_H, int pad_W, const int* images, int rows_per_image,
                        void* stream);
/* dx_pad_H, dx_pad_W (0, 0 = no): dx is written in the padded layout as well (borders not written) -- the operand of
 * mmsum_conv3x3_wgrad and of the input-gradient convolution. */
int mmsum_bn_bwd_apply(int dtype, const void* dy, const void* y, const void* x, const float* sums, const float* dsums,
                       const float* gamma, void* dx, void* dresidual, float* dgamma, float* dbeta, int R, int C,
                       float eps, int relu, int pad_H, int pad_W, int dx_pad_H, int dx_pad_W, const int* images, int rows_per_image,
                       void* stream);
int mmsum_maxpool3x3s2(int dtype, const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, const int* images, void* stream);
/* NCHW f32 image -> NHWC dtype.  src (device int32 [N], may be NULL): image n of y is image src[n] of x (mmsum_image_plan's run order). */
int mmsum_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, const int* images, const int* src, void* stream);

/* ---- Table encoder (table_encoder.py:14-83) ---------------------------------------------------
 * Builds all_embeddings [B,47,2D] = [field-name masked sum | field value] and the mask [B,47];
 * embedding reads are gradient-free in the reference (torch.no_grad). */
int mmsum_table_gather(int dtype, const void* E, const int64_t* field, const int64_t* name, const int64_t* category,
                       const int64_t* str_cat, const int64_t* str_bool, const int64_t* rating, const int64_t* hours,
                       const void* w_rating, const void* w_hours, void* out, uint8_t* mask, int B, int D, int pad_id,
                       void* stream);
/* d w_rating [D,4] += sum_b rating[b,k] * dvalue[b,39,d]; d w_hours [D,4] += sum_{b,j} hours[b,j,k]*dvalue[b,40+j,d]
 * where dvalue = dall[:, :, D:2D] (f32 accumulation). */
int mmsum_table_gather_bwd(int dtype, const void* dall, const int64_t* rating, const int64_t* hours, float* dw_rating,
                           float* dw_hours, int B, int D, void* stream);

/* ---- Amazon table encoder (table_encoder.py:86-167) ----------------------------------------------
 * all_embeddings [B,133,2D] = [field-name embedding (field [6], the last repeated 128x) | value] with values
 * price = Linear(11->D), rating = Linear(4->D), brand / name = masked token sums, category [B,3,8,12] = token sums
 * averaged over valid rows then over valid groups, description = 128 raw token embeddings; mask [B,133] (:160-166). */
int mmsum_amazon_table_gather(int dtype, const void* E, const int64_t* field, const int64_t* price, const int64_t* rating,
                              const int64_t* brand, const int64_t* name, const int64_t* category, const int64_t* description,
                              const void* w_price, const void* w_rating, void* out, uint8_t* mask, int B, int D, int pad_id,
                              void* stream);
/* d w_price [D,11] += sum_b price[b,k] * dvalue[b,0,d]; d w_rating [D,4] += sum_b rating[b,k] * dvalue[b,1,d]. */
int mmsum_amazon_table_gather_bwd(int dtype, const void* dall, const int64_t* price, const int64_t* rating, float* dw_price,
                                  float* dw_rating, int B, int D, void* stream);

/* ---- Beam-search decode step (modeling_multimodalsum.py:2857-3010; generation_utils.py:57-98,848-868) -----------------
 * mmsum_beam_topk: the tail of one step on the [rows, V] logits (rows = businesses * num_beams, hypotheses of a business
 * consecutive): forced token (adjust_logits_during_generation :3084-3089: BOS at cur_len 1, EOS at max_length - 1; -1 =
 * none), log_softmax (:2874), then -- AFTER the normalisation, as the reference does -- ban_token (EOS while cur_len <
 * min_length; -1 = none) and the no-repeat-n-gram bans (`banned` [rows, nban] int32: a row's list is filled from the front and its first -1 ends it; may be NULL), + beam_scores
 * [rows], and the top 2*num_beams of each business's num_beams * V candidates (:2925): out_scores / out_ids
 * [rows / num_beams, 2 * num_beams], best first, id = beam * V + token, ties by lower id.  The banned positions of `logits`
 * are overwritten with -inf.  workspace: mmsum_beam_topk_workspace() bytes (per-chunk statistics and candidates: the logits are
 * read once by rows x 8 chunk blocks).  num_beams <= 8, V <= 65,536.
 * Repetition penalty (enforce_repetition_penalty_, generation_utils.py:47-55; ABI 7): `penalized` [rows, npen] int32 = a row's DISTINCT
 * previous tokens (first -1 ends the list; NULL or penalty == 1: none); a listed score s becomes s * penalty when negative, s / penalty
 * otherwise, BEFORE the bans (postprocess_next_token_scores' order) -- on the log-probabilities (penalty_on_logits = 0: beam search,
 * :2874-2890; costs a second pass over the logits) or on the raw logits (penalty_on_logits = 1: _generate_no_beam_search post-processes
 * the logits themselves, :2749-2783; with num_beams = 1 and beam_scores 0 the first candidate of a row is then the reference's argmax).
 * ncand (ABI 10; 0 = 2 * num_beams): candidates returned per business, <= 64 -- sampling (_generate_no_beam_search with do_sample, :1831-1839)
 * asks for the top_k best of a row (num_beams = 1, force_token -1: the reference skips the forced tokens when it samples) and draws among
 * them on the host; out_scores / out_ids are then [rows / num_beams, ncand].
 * mmsum_decode_self_attn: single-query self-attention of every hypothesis over its K/V cache rows (:776-815), reached
 * through an ancestor table: key s (< len) of row r is row ancestors[r * Tmax + s] * Tmax + s of k_cache / v_cache
 * ([rows * Tmax, H*64]).  A beam reorder (_reorder_cache :3104-3115) is then a gather of the table, not of the caches.
 * k_new / v_new [rows, H*64] (optional, both or neither): this step's key / value projections (position len - 1, where
 * ancestors[r, len - 1] must be r): the kernel stores them into the caches and uses them for that position, which replaces the
 * cache append of the reference (:804-815). */
long mmsum_beam_topk_workspace(int rows, int num_beams, int ncand);
int mmsum_beam_topk(int dtype, void* logits, long ld, int V, const float* beam_scores, const int* banned, int nban, int force_token,
                    int ban_token, int rows, int num_beams, void* workspace, float* out_scores, long long* out_ids,
                    const int* penalized, int npen, float penalty, int penalty_on_logits, int ncand, void* stream);
int mmsum_decode_self_attn(int dtype, const void* q, long ldq, void* k_cache, void* v_cache, long ld_cache, const int* ancestors,
                           void* out, long ldo, int rows, int H, int len, int Tmax, float scale, const void* k_new, const void* v_new,
                           long ld_new, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
