/*
 * ctmi355 — C ABI of the MI355X (gfx950) kernels behind the CleanTransformer SFT hot path.
 *
 * The reference (firechecking/CleanTransformer @ 2024-10-16) is pure Python on stock PyTorch: it has
 * no FFI.  Every entry point below therefore replaces the *aten call sites* of one reference
 * function; the citation names that function (file:line relative to the reference checkout).
 * The reference-side binding a maintainer would add is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All pointers are DEVICE pointers unless marked host.
 *   - `dtype` selects the storage type of activations / matrices: CTMI_F32 or CTMI_BF16.
 *     Statistics, biases, LayerNorm affine parameters, optimizer state and ALL parameter
 *     gradients are fp32 in both modes.  Accumulation is always fp32.
 *   - `stream` is a hipStream_t (NULL = default stream).  Every call is asynchronous; nothing
 *     inside synchronises the device or allocates persistent memory.  Workspaces are passed in.
 *   - return value: 0 on success, negative ctmi_status otherwise; ctmi_last_error() gives the text
 *     (thread-local).  Nothing throws or exits across the ABI.
 *   - re-entrant; safe to call from PyTorch's autograd worker threads.
 */
#ifndef CTMI355_H
#define CTMI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTMI_ABI_VERSION 15

enum ctmi_dtype { CTMI_F32 = 0, CTMI_BF16 = 1, CTMI_F16 = 2 };
/* CTMI_F16 (ABI v14): IEEE half storage with fp32 accumulation and statistics — torch.autocast(dtype=float16) of examples/ft_bloom_DDP.py:107-128.
 * Every kernel family of the bf16 path has an fp16 twin (v_mfma_f32_16x16x32_f16 / 32x32x16_f16); CTMI_BF16 is the metric's dtype. */
enum ctmi_status { CTMI_OK = 0, CTMI_ERR_ARG = -1, CTMI_ERR_LAUNCH = -2, CTMI_ERR_UNSUPPORTED = -3 };

int ctmi_abi_version(void);
const char* ctmi_last_error(void);

/* ---- LayerNorm  (transformer.py:71-89 LayerNorm._mean/forward; used at modeling_bloom.py:143,155,191,205)
 * y = w * (x-mean)/sqrt(var_biased+eps) + b over the last `cols` elements of each of `rows` rows.
 * mean/rstd (fp32 [rows]) are saved for the backward. */
int ctmi_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                       int64_t rows, int64_t cols, float eps, int dtype, void* stream);
/* dx = LN'(dy) (+ dres if non-NULL: the residual-branch gradient, fused);  dw/db (fp32 [cols]) are
 * overwritten (accumulate=0) or added to (accumulate=1).  ws: fp32 workspace of ctmi_layernorm_bwd_ws(rows, cols) floats. */
int64_t ctmi_layernorm_bwd_ws(int64_t rows, int64_t cols);
int ctmi_layernorm_bwd(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                       const void* dres, void* dx, float* dw, float* db, int accumulate, float* ws,
                       int64_t rows, int64_t cols, int dtype, void* stream);

/* ---- dense GEMM family  (torch.nn.Linear call sites: modeling_bloom.py:79,121,256,267,220 and their autograd)
 *   C[M,N] = epilogue( alpha * sum_k opA(m,k) * opB(k,n) )
 *   a_kmajor=0: A is [M,K] row-major (lda);  a_kmajor=1: A is stored [K,M] row-major (lda)
 *   b_kmajor=0: B is [N,K] row-major (ldb)   b_kmajor=1: B is stored [K,N] row-major (ldb)
 *     forward  y = x W^T      : a_kmajor=0,b_kmajor=0 (A=x[T,in],  B=W[out,in])
 *     dgrad    dx = dy W      : a_kmajor=0,b_kmajor=1 (A=dy[T,out], B=W[out,in] read as [K=out, N=in])
 *     wgrad    dW = dy^T x    : a_kmajor=1,b_kmajor=1 (A=dy[T,out] as [K=T,M=out], B=x[T,in] as [K=T,N=in])
 *   epilogue (applied in this order):
 *     + bias[n] (fp32, NULL = none)
 *     CTMI_EPI_GELU : aux_out[m,n] = v (pre-activation, storage dtype);  v = bloom tanh-GELU(v)   (modeling_bloom.py:335-344)
 *     CTMI_EPI_DGELU: v *= gelu'(aux_in[m,n])                                                     (modeling_bloom.py:348-363)
 *     CTMI_EPI_GELUG: aux_out[m,n] = gelu'(v) (storage dtype);  v = bloom tanh-GELU(v) — the forward leaves the DERIVATIVE
 *                     for the backward (modeling_bloom.py:348-363 evaluated where its logistic factor already exists), and
 *     CTMI_EPI_MUL  : v *= aux_in[m,n] is then the whole activation backward (one multiply in the data-gradient epilogue)
 *     CTMI_EPI_RELU : v = max(v,0)          CTMI_EPI_DRELU: v = aux_in[m,n] > 0 ? v : 0            (transformer.py:98-102 FFN)
 *     + residual[m,n] (storage dtype, NULL = none; ldc stride)                                     (modeling_bloom.py:122,269)
 *     beta=1: + C_old
 *   out_f32=1 writes C as fp32 regardless of dtype (parameter gradients).
 *   workspace (optional, may be NULL): scratch for deterministic split-K of small-tile-count problems (weight
 *   gradients): fp32 slabs [splits][M][N]; more workspace = more splits (up to 8). */
enum ctmi_epilogue { CTMI_EPI_NONE = 0, CTMI_EPI_GELU = 1, CTMI_EPI_DGELU = 2, CTMI_EPI_RELU = 3, CTMI_EPI_DRELU = 4,
                     CTMI_EPI_GELUG = 5, CTMI_EPI_MUL = 6 };
int ctmi_gemm(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
              void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
              float alpha, int beta, const float* bias, const void* residual, int epilogue,
              const void* aux_in, void* aux_out, int out_f32, int dtype,
              void* workspace, int64_t workspace_bytes, void* stream);

/* ---- grouped weight gradients (ABI v13).  The four parameter-gradient products of one block's backward — autograd of the Linears at
 * modeling_bloom.py:79 (query_key_value), :121 (dense), :256 (dense_h_to_4h), :267 (dense_4h_to_h) — in ONE persistent launch:
 *   dw[n_out, n_in] (fp32, overwritten) = dy[T, n_out]^T x[T, n_in]        dy, x: row-major, dense, bf16 (or fp16)
 *   db[n_out]       (fp32, overwritten, optional) = column sums of dy      (the bias gradient; computed by the matrix cores against a vector of ones)
 *   in_out = 1: dw is written [n_in, n_out] (a Conv1D weight's own layout, modeling_gpt.py:32-46); db (round 6: allowed there too) is then the sum of the
 *               launch's B operand, taken by the tiles of the first tile row.  db must be 16-byte aligned.
 * Output tiles are 128 x 256; whole rounds of the 256 CUs accumulate over all T rows, the tiles of the last partial round are cut in two along T:
 * each half stores its partial tile into `workspace` and a small second launch adds the two in a fixed order (deterministic; without enough
 * workspace — 2 x (rows x columns + max(rows, columns)) floats per gradient — nothing is cut).  No split-K slabs of whole gradients, no separate column-sum pass.
 * Needs dtype = CTMI_BF16 or CTMI_F16, n <= 4, T % 32 == 0, every gradient's rows a multiple of 128 and columns of 256, 16-byte aligned pointers: anything
 * else returns CTMI_ERR_UNSUPPORTED (callers then use ctmi_gemm(a_kmajor = b_kmajor = 1) per product).
 * CTMI_WGRAD_GROUP = 0 in the environment disables it (ctmi_bloom_block_bwd then launches the four products separately, as in ABI <= 12). */
typedef struct ctmi_wgrad_problem {
    const void* dy; const void* x;      /* device, [T, n_out] and [T, n_in] */
    float* dw; float* db;               /* device; db may be NULL */
    int64_t n_out, n_in;
    int in_out; int pad_;
} ctmi_wgrad_problem;
int ctmi_wgrad_grouped(const ctmi_wgrad_problem* problems /* host */, int count, int64_t T, int dtype,
                       void* workspace /* device, may be NULL */, int64_t workspace_bytes, void* stream);

/* GEMM launch policy (process-wide).  shared = 0: the GPU is ours — persistent launches sized to the 256 CUs, one-workgroup-
 * per-CU ping-pong tiles.  shared = 1: another long-running kernel holds CUs under our GEMMs (the RCCL all-reduce of a
 * data-parallel job, trainer DDP at examples/ft_bloom_DDP.py:99): no persistent launches, 2-3 workgroups per CU for the
 * layer GEMMs.  shared = 2 ("flow", ABI v15): one workgroup per tile like 1, but the tile choice of shared = 0 (the ping-pong tiles) — the
 * dispatcher flows the workgroups over the CUs the collectives leave free; costs +0.x ms per step at world 1 where shared = 1 costs +2.8
 * (profiles/r06_ddp_policy_world1.json).  reserve_cus: CUs left out of persistent launches.  Initial values: CTMI_GEMM_SHARED / CTMI_GEMM_RESERVE_CUS. */
int ctmi_set_launch_policy(int shared, int reserve_cus);
int ctmi_get_launch_policy(int* shared /* host, may be NULL */, int* reserve_cus /* host, may be NULL */);

/* ---- per-class device time of a step without an external profiler (bench.py's breakdown; SURVEY 8(d) measurement).  Between
 * ctmi_profile_begin() and ctmi_profile_end() every entry point of this library brackets its launches with HIP events on the stream
 * it launches on; ctmi_profile_end() waits for them and returns, per class, the summed bracket time in milliseconds and the number of
 * brackets.  Brackets of concurrent streams overlap in wall time (sum of classes >= elapsed time then): measure with one stream. */
enum ctmi_prof_class {
    CTMI_PROF_GEMM_FWD = 0,   /* layer Linear forwards        x W^T          (modeling_bloom.py:79,121,256,267) */
    CTMI_PROF_GEMM_DGRAD = 1, /* layer data gradients         dy W           */
    CTMI_PROF_GEMM_WGRAD = 2, /* layer weight gradients       dy^T x  (+ split-K reduce) */
    CTMI_PROF_LM_HEAD = 3,    /* the three [T,H] x [V,H] products of the tied head (modeling_bloom.py:220-221) */
    CTMI_PROF_ATTN_FWD = 4, CTMI_PROF_ATTN_BWD = 5,
    CTMI_PROF_LAYERNORM = 6,  /* forward + backward */
    CTMI_PROF_LOSS = 7,       /* cross entropy forward + backward */
    CTMI_PROF_OPTIMIZER = 8,
    CTMI_PROF_REDUCE = 9,     /* bias / LayerNorm-parameter gradient reductions */
    CTMI_PROF_OTHER = 10,     /* embedding, mask digest, casts, ... */
    CTMI_PROF_NCLASS = 11
};
int ctmi_profile_begin(void);
int ctmi_profile_end(float* ms /* host [CTMI_PROF_NCLASS] */, int* brackets /* host [CTMI_PROF_NCLASS] */);

/* ---- data-parallel collectives straight on RCCL  (examples/ft_bloom_DDP.py:183 `init_process_group("nccl")`, :99
 *      `DDP(model, device_ids=[local_rank])`: the gradient all-reduce torch's DDP issues through ProcessGroupNCCL; ABI v10)
 * One process per GPU.  Rank 0 makes a 128-byte id (ctmi_ddp_unique_id) and hands it to the other ranks by any host channel (the
 * Python wrapper uses the torch.distributed group the reference already initialises); every rank then creates its communicator with
 * the SAME id.  max_channels > 0 caps the communicator's RCCL channels — one channel is one workgroup, i.e. one CU held for the length
 * of a collective — and is meant to equal the reserve_cus of ctmi_set_launch_policy(0, R): the GEMMs then never wait for a CU a
 * collective holds, and the collectives always find R free CUs.  A communicator owns one stream: every collective is ordered behind
 * what compute_stream holds at the call and runs asynchronously; ctmi_ddp_wait makes compute_stream wait for everything issued on the
 * communicator so far (the host never blocks).  Buffers must stay alive until a ctmi_ddp_wait that follows their collective has been
 * enqueued on the stream that frees / reuses them.  librccl is loaded on first use (no link-time dependency). */
int ctmi_ddp_unique_id(void* id128 /* host, 128 bytes out */);
int ctmi_ddp_create(const void* id128 /* host */, int rank, int world, int max_channels, void** comm_out);
int ctmi_ddp_destroy(void* comm);
int ctmi_ddp_all_reduce(void* comm, void* buf /* device, in place, SUM */, int64_t count, int dtype /* CTMI_F32 | CTMI_BF16 */, void* compute_stream);
int ctmi_ddp_all_gather(void* comm, const void* send /* device */, void* recv /* device, world * bytes_per_rank */, int64_t bytes_per_rank, void* compute_stream);
int ctmi_ddp_broadcast(void* comm, void* buf /* device, in place */, int64_t bytes, int root, void* compute_stream);
int ctmi_ddp_wait(void* comm, void* compute_stream);

/* column sum: out[n] (+)= sum_m x[m,n]  — bias gradients (autograd of the Linear biases). */
int ctmi_colsum(const void* x, int64_t ld, float* out, int accumulate, float* ws, int64_t M, int64_t N, int dtype, void* stream);
int64_t ctmi_colsum_ws(int64_t M, int64_t N);

/* ---- attention  (modeling_bloom.py:84-116 BloomAttentionLayer core; transformer.py:40-57 AttentionLayer core)
 * Fused  softmax( scale*q.k + slope[h]*kpos[b,k] + add_mask  ; masked -> finfo.min ) v , flash-style,
 * never materialising [S,S].  q/k/v/o are addressed as  base + b*bs + h*hs + row*rs + d  (element strides),
 * so the head-interleaved fused-QKV buffer (modeling_bloom.py:81-82) and a [B,nh,S,hd] KV cache are both native.
 *   slopes   fp32 [nh] or NULL   (ALiBi, modeling_bloom.py:309-331)
 *   kpos     fp32 [B,Sk] or NULL (ALiBi key positions (cumsum(mask)-1)*mask, modeling_bloom.py:328)
 *   kvalid   int32 [B,Sk] or NULL(1 = attend; the ~attention_mask half of _attn_mask, modeling_bloom.py:178-179)
 *   first_valid int32 [B] (index of the first attendable key, Sk if none; required with kvalid)
 *   causal   1: key j visible to query i iff j <= i + (Sk-Sq)  (tril half of _attn_mask, modeling_bloom.py:180-183)
 *   add_mask fp32 additive mask or NULL, element strides am_b, am_h, am_q, am_k (0 = broadcast) (transformer.py:43-45)
 * Masked scores take the value finfo(float).min exactly as the reference's masked_fill does, so a query row
 * whose visible keys are all masked gets a UNIFORM distribution over all Sk keys, not NaN.
 * stat_m / stat_l (fp32 [B,nh,Sq]) = row max and row sum of exp(score-max); saved for the backward. */
typedef struct ctmi_attn_desc {
    int64_t B, nh, Sq, Sk, hd;
    int64_t q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
    int64_t am_b, am_h, am_q, am_k;
    float scale;
    int causal;
    float future_fill;   /* score of a causally-future pair whose key is attendable.  0 = finfo(float).min: the bool-mask masked_fill
                            of modeling_bloom.py:108-110.  GPT-2 replaces future scores by -1e4 instead (w*b - 1e4*(1-b),
                            modeling_gpt.py:88-89): identical unless a query's whole causal window is padding (left padding) — then
                            the -1e4 of the future keys is ABOVE the finfo.min of the padded visible ones and the reference attends to
                            the future; pass -1e4 to reproduce that. */
    uint32_t dropout_seed;   /* attention-probability dropout (torch.nn.Dropout on the softmax output: modeling_bloom.py:111,               */
    float dropout_p;         /* modeling_gpt.py:96, transformer.py:46-47): element (b,h,q,k) is kept iff keep_hash(counter, seed) >= p*2^32   */
    int32_t reserved_;       /* with counter = ((b*nh+h)*Sq+q)*Sk+k mod 2^32 (ctmi_dropout_keep_hash) and scaled by 1/(1-p); the backward       */
                             /* regenerates the mask from the same seed.  p = 0: off.                                                        */
} ctmi_attn_desc;
int ctmi_attn_fwd(const void* q, const void* k, const void* v, void* o, float* stat_m, float* stat_l,
                  const float* slopes, const float* kpos, const int32_t* kvalid, const int32_t* first_valid,
                  const float* add_mask, const ctmi_attn_desc* desc /* host */, int dtype, void* stream);
/* dq/dk/dv use the q/k/v strides of desc; do uses the o strides.  delta: fp32 workspace [B,nh,Sq]. */
int ctmi_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                  const float* stat_m, const float* stat_l, void* dq, void* dk, void* dv, float* delta,
                  const float* slopes, const float* kpos, const int32_t* kvalid, const int32_t* first_valid,
                  const float* add_mask, const ctmi_attn_desc* desc /* host */, int dtype, void* stream);
/* Which kernels serve the bf16 TRAINING shapes of ctmi_attn_fwd / ctmi_attn_bwd (causal, Sq == Sk a multiple of 64 and <= 4096,
 * head_dim 64 or 128, no additive mask, no probability dropout): bit 0 = forward, bit 1 = backward through the 256-row /
 * 32x32-MFMA kernels of csrc/attention_w32.hip (default 3); 0 = the general 64-row kernels for everything.  Same contract and
 * results up to bf16 rounding either way (the statistics stat_m / stat_l are interchangeable between the two families); the
 * switch exists for A/B measurements and parity tests.  mask < 0: query only.  Returns the previous value.
 * (modeling_bloom.py:84-116 is served by both.) */
int ctmi_attn_set_path(int mask);
/* attention_mask [B,S] (int64 0/1) -> kpos fp32, kvalid int32, first_valid int32[B]  (modeling_bloom.py:328, 178-179) */
int ctmi_mask_prep(const int64_t* attention_mask, float* kpos, int32_t* kvalid, int32_t* first_valid,
                   int64_t B, int64_t S, void* stream);

/* ---- embedding  (modeling_bloom.py:190 word_embeddings; its autograd = scatter-add into the tied [V,H] grad)
 * ids are int64; out-of-range ids raise CTMI_ERR_ARG lazily via *err_flag (int32 device word, may be NULL). */
int ctmi_embed_fwd(const void* table, const int64_t* ids, void* out, int64_t n_tokens, int64_t H, int64_t V,
                   int dtype, int32_t* err_flag, void* stream);
int ctmi_embed_bwd(const void* dout, const int64_t* ids, float* dtable /* fp32 [V,H], += scale * row */, int64_t n_tokens,
                   int64_t H, int64_t V, int dtype, float scale /* 1 locally; 1/world for rows gathered from DDP peers */,
                   void* stream);

/* ---- cross entropy  (torch.nn.CrossEntropyLoss at modeling_bloom.py:224-230; repo loss.py:29-49)
 * logits [N, C] (ld = row stride).  Row r = (b, s) with b = r / seq, s = r % seq takes its target from
 * labels[b*seq + s + shift]; rows with s + shift >= seq, or whose target == ignore_index, carry no loss
 * (shift=1, seq=S reproduces shift_logits/shift_labels of modeling_bloom.py:225-226 without the
 * .contiguous() copy; shift=0, seq=N is a plain CE).
 * fwd: row_lse fp32 [N], row_loss fp32 [N];  loss_out[0] = sum(row_loss)/denom, loss_out[1] = 1/denom
 *      denom_mode 0: number of loss-carrying rows (torch 'mean'); 1: N_rows_total given by `denom_rows`
 *      (loss.py:47-48 divides by input.shape[0]); 2: 1 ('sum').
 * bwd: dlogits[r,c] = (softmax - onehot) * gout[0] * loss_out[1]   (0 for rows without loss). */
int ctmi_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* row_lse, float* row_loss,
                float* loss_out, int64_t N, int64_t C, int64_t seq, int64_t shift, int64_t ignore_index,
                int denom_mode, int64_t denom_rows, int dtype, void* stream);
int ctmi_ce_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* row_lse, const float* loss_out,
                const float* gout /* device scalar or NULL = 1 */, void* dlogits, int64_t ldd,
                int64_t N, int64_t C, int64_t seq, int64_t shift, int64_t ignore_index, int dtype, void* stream);

/* Training path: loss and dlogits in ONE pass over the logits (the loss forward of modeling_bloom.py:224-230 and the first
 * step of loss.backward()): same arguments and results as ctmi_ce_fwd, plus dlogits written for the upstream gradient the caller
 * EXPECTS, grad_factor * (grad_factor_dev ? grad_factor_dev[0] : 1) — 1 for a plain loss.backward(); 1/accumulation_steps for
 * trainer.py:468-504's `loss / ga`; the loss scale (a device scalar) for ft_bloom_DDP.py:123-127's scaler.scale(loss) — applied in
 * fp32 before the single rounding to the storage type.  Rows must be 16-byte aligned (ld, ldd multiples of 16 bytes). */
int ctmi_ce_fwd_bwd(const void* logits, int64_t ld, const int64_t* labels, float* row_lse, float* row_loss,
                    float* loss_out, void* dlogits, int64_t ldd, int64_t N, int64_t C, int64_t seq, int64_t shift,
                    int64_t ignore_index, int denom_mode, int64_t denom_rows, float grad_factor, const float* grad_factor_dev /* device, may be NULL */,
                    int dtype, void* stream);
/* The backward of that node: x[rows, cols] (ld) *= s_dev[0] / (applied * applied_dev[0]) where s_dev[0] is the ACTUAL upstream
 * gradient and (applied, applied_dev) the pair ctmi_ce_fwd_bwd was given; skipped on the device when the two are equal (the
 * expected case: no second pass over [T,V]).  ctmi_scale_if_passes() = how many calls did rescale (process lifetime; tests). */
int ctmi_scale_if(void* x, int64_t ld, int64_t rows, int64_t cols, const float* s_dev, float applied, const float* applied_dev /* device, may be NULL */,
                  int dtype, void* stream);
int64_t ctmi_scale_if_passes(void);

/* probability targets, the second branch of loss.py:43-46: loss = -sum t * log_softmax(x) (÷ denom_rows for 'mean': denom_mode 1;
 * 'sum': denom_mode 2).  target fp32 [N, C] (ldt).  row_tsum keeps sum_c t[n,c] for the backward:
 * dlogits = (softmax * row_tsum - target) * gout[0] * loss_out[1]. */
int ctmi_ce_soft_fwd(const void* logits, int64_t ld, const float* target, int64_t ldt, float* row_lse, float* row_tsum,
                     float* row_loss, float* loss_out, int64_t N, int64_t C, int denom_mode, int64_t denom_rows, int dtype, void* stream);
int ctmi_ce_soft_bwd(const void* logits, int64_t ld, const float* target, int64_t ldt, const float* row_lse, const float* row_tsum,
                     const float* loss_out, const float* gout, void* dlogits, int64_t ldd, int64_t N, int64_t C, int dtype, void* stream);

/* ---- dropout on activations  (torch.nn.functional.dropout / torch.nn.Dropout at modeling_bloom.py:122,270, modeling_gpt.py:74,100,
 *      136,190, transformer.py:111-119)
 * y[i] = (keep(i) ? x[i] / (1-p) : 0) (+ residual[i]),  keep(i) = keep_hash(i, seed) >= p * 2^32 (i = flat element index mod 2^32).
 * The same call on a gradient (residual = NULL, same seed) is the backward: no mask tensor is stored.  y may alias x. */
int ctmi_dropout(const void* x, const void* residual, void* y, int64_t n, float p, uint32_t seed, int dtype, void* stream);
/* the mask function itself, on the host: keep(counter) = ctmi_dropout_keep_hash(counter, seed) >= ctmi_dropout_threshold(p)
 * (what the kernels evaluate per element; exported so that callers / tests can restate a mask without a device).
 * keep_hash(counter, seed) = hash(hash(counter ^ seed) + seed * 0x9E3779B1 + 0x7F4A7C15) with hash = ctmi_dropout_hash, the
 * "lowbias32" finaliser (ABI v9: the seed keys BOTH rounds, so two seeds give two functions, not two windows of one sequence). */
uint32_t ctmi_dropout_hash(uint32_t x);
uint32_t ctmi_dropout_keep_hash(uint32_t counter, uint32_t seed);
uint32_t ctmi_dropout_threshold(float p);

/* ---- optimizers  (optimizer.py:53-97 AdamW [L2 form]; torch.optim.AdamW as called at ft_bloom.py:70 [decoupled];
 *                   optimizer.py:12-50 SGD)
 * Multi-tensor: `count` tensors described by host arrays of device pointers; one launch per <=CTMI_MT_MAX tensors.
 *   shadow[i] (may be NULL): bf16 copy of the updated parameter written in the same pass (compute-dtype weights).
 *   decoupled=0: g += wd*p (and, if mutate_grad, written back like the reference does), Adam update with
 *                p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
 *   decoupled=1: p *= 1-lr*wd;  p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 *   grad_scale: multiplies g on read (1/world for deferred DDP averaging, or a clip coefficient). */
#define CTMI_MT_MAX 24
/* (ABI v14) the `shadow` copies are bf16 unless CTMI_OPT_SHADOW_F16 is OR-ed into `mutate_grad` (AdamW) / `first_step` (SGD): IEEE half then */
#define CTMI_OPT_SHADOW_F16 2
/* (ABI v15) ctmi_adamw_step launches one workgroup per 16 Ki-element chunk, up to 64 tensors per launch (5 launches for the 294 parameters of
 * Bloom-560M, every workgroup with equal work); CTMI_OPT_LEGACY_GRID OR-ed into `mutate_grad` takes the (stride loop, tensor) grid of ABI <= 14
 * (24 tensors per launch) — same arithmetic, bit-identical results (no FMA contraction in either), kept for A/B measurements and the parity test */
#define CTMI_OPT_LEGACY_GRID 4
int ctmi_adamw_step(float* const* p /*host*/, float* const* g /*host*/, float* const* m /*host*/, float* const* v /*host*/,
                    void* const* shadow /*host, entries may be NULL*/, const int64_t* n /*host*/, int count,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    int decoupled, int mutate_grad, float grad_scale, void* stream);
/* (ABI v15) The same update with its hyper-parameters in DEVICE memory — the form a captured hipGraph replays (cleantransformer_amd/graph.py: the whole
 * SFT step, forward + backward + optimizer, as one graph launch; a launch's arguments are frozen at capture, so the per-step numbers — bias corrections
 * 1 - beta^t, a scheduler's lr — cannot ride in them).  ctmi_adamw_set_hyper writes the 48-byte record of step `step` (computed exactly like
 * ctmi_adamw_step does on the host: results are bit-identical) with one single-thread launch carrying the values in ITS arguments — it is issued eagerly
 * before every replay; ctmi_adamw_step_dev is ctmi_adamw_step reading that record (chunk-balanced launches; `mutate_grad` flags ride in the record). */
int ctmi_adamw_set_hyper(void* hyper_dev /* device, 48 bytes, 4-byte aligned */, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, int decoupled, int mutate_grad, float grad_scale, void* stream);
int ctmi_adamw_step_dev(float* const* p /*host*/, float* const* g /*host*/, float* const* m /*host*/, float* const* v /*host*/,
                        void* const* shadow /*host, entries may be NULL*/, const int64_t* n /*host*/, int count,
                        const void* hyper_dev /* device: written by ctmi_adamw_set_hyper */, void* stream);
int ctmi_sgd_step(float* const* p, float* const* g, float* const* buf /* momentum buffers or NULL */,
                  void* const* shadow, const int64_t* n, int count, float lr, float momentum, float dampening,
                  float weight_decay, int first_step, void* stream);

/* ---- loss scaling: the kernels under cleantransformer_amd.amp.GradScaler (torch.cuda.amp.GradScaler as ft_bloom_DDP.py:107-128
 *      drives it).  state = device float[3] {scale, growth tracker, found_inf}. */
/* g[i] *= 1/scale in place (multi-tensor); found_inf = 1 if any result is not finite */
int ctmi_amp_unscale(float* const* g /*host*/, const int64_t* n /*host*/, int count, float* state /*device*/, void* stream);
/* scale/tracker update after a step (backoff on found_inf, growth every `interval` clean steps); clears found_inf */
int ctmi_amp_update(float* state /*device*/, float growth, float backoff, int interval, void* stream);

/* ---- small utilities on flat buffers */
int ctmi_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* dst[c][r] = cast(src[r][c]): the [out,in] compute copy of a GPT-2 Conv1D weight kept as [in,out] (modeling_gpt.py:32-46) */
int ctmi_transpose_cast(const float* src, void* dst, int dst_dtype, int64_t rows, int64_t cols, void* stream);
/* out[0] (+)= sum(x^2), fp64 accumulation (device double) — global grad-norm (trainer.py:491-498 clip_grad_norm_) */
int ctmi_sumsq(const float* x, int64_t n, double* out, int accumulate, void* stream);
int ctmi_scale(float* x, int64_t n, float s, const float* s_dev /* optional device scalar multiplier */, void* stream);
/* dst = s * src (fp32; dst may alias src) — DDP bucket fill with torch-DDP's pre-division fused (torch Reducer: divide, then all-reduce) */
int ctmi_scale_copy(const float* src, float* dst, int64_t n, float s, void* stream);
/* argmax over the last dim of x[rows, cols] -> int64 (first maximal index, as torch.argmax; generation_util.py:86) */
int ctmi_argmax(const void* x, int64_t ld, int64_t* out, int64_t rows, int64_t cols, int dtype, void* stream);

/* ---- decode beyond argmax: beam search and samplers (generation_util.py:121-290, logits_processor.py) */
/* stats[row] = {max, log(sum exp(x - max))} over x[rows, cols]: the two terms of torch.log_softmax (generation_util.py:200) */
int ctmi_row_lse(const void* x, int64_t ld, float* stats /* [rows,2] */, int64_t rows, int64_t cols, int dtype, void* stream);
/* per group g of `group` consecutive rows: the k best of score(i,v) = ((x[g*group+i][v] - max) - logsum) + add[g*group+i]*add_mul
   (stats, add optional) as (value, flat index i*cols+v), descending, equal values by ascending index — replaces
   scores.view(bsz,-1).topk(2*beam) (generation_util.py:199-217) and the k-th-value threshold of TopKLogitsWrapper
   (logits_processor.py:48-50). */
int ctmi_group_topk(const void* x, int64_t ld, const float* stats, const float* add, float add_mul, float* out_val /* [groups,k] */,
                    int64_t* out_idx /* [groups,k] */, int64_t groups, int group, int64_t cols, int k, int dtype, void* stream);
/* out = x / divisor, then entries < thr[row*thr_stride] (thr optional) become `fill` — TemperatureLogitsWrapper and the
   masked_fill of TopKLogitsWrapper (logits_processor.py:35-56); fp32, out may alias x */
int ctmi_scores_filter(const float* x, int64_t ld, float divisor, const float* thr, int64_t thr_stride, float fill, float* out,
                       int64_t ldo, int64_t rows, int64_t cols, void* stream);

/* ---- partial-row reductions: dst[c] (+)= alpha * sum_{p < nparts} src[p * part_stride + c], c < n — up to
 * CTMI_REDUCE_MAX_JOBS independent vectors per launch, fixed summation order (deterministic).  Used by the block backward
 * below for the LayerNorm affine gradients and the Linear bias gradients (autograd of transformer.py:71-89 and of the biases
 * at modeling_bloom.py:79,121,256,267). */
#define CTMI_REDUCE_MAX_JOBS 16
typedef struct ctmi_reduce_job {
    const float* src; float* dst; int64_t n; int64_t part_stride; int32_t nparts; int32_t accumulate; float alpha; int32_t pad_;
} ctmi_reduce_job;
int ctmi_reduce_jobs(const ctmi_reduce_job* jobs /* host */, int count, void* stream);

/* ---- one Bloom block, forward and backward, as ONE call each  (modeling_bloom.py:142-159 BloomBlock.forward =
 * :76-124 BloomAttentionLayer + :255-271 BloomMLP, and its autograd).  The host makes one call per block instead of 7 / 17
 * kernel-level calls; the launch sequence (LayerNorm -> QKV GEMM -> attention -> dense GEMM(+residual) -> LayerNorm ->
 * h->4h GEMM(+GELU) -> 4h->h GEMM(+residual), and the hand-derived reverse) runs inside the library.
 * All pointers are device pointers.  Weight matrices are in the compute dtype ([out,in] row-major, as torch.nn.Linear
 * stores them), biases / LayerNorm parameters fp32.  Activations saved for the backward live in ONE caller-allocated slab
 * whose layout ctmi_bloom_block_layout() defines (byte offsets, 256-byte aligned). */
enum ctmi_block_slot {                  /* index into the offsets[] array filled by ctmi_bloom_block_layout */
    CTMI_BLK_LN1 = 0, CTMI_BLK_MEAN1, CTMI_BLK_RSTD1, CTMI_BLK_QKV, CTMI_BLK_ATT, CTMI_BLK_STAT_M, CTMI_BLK_STAT_L,
    CTMI_BLK_H1, CTMI_BLK_MEAN2, CTMI_BLK_RSTD2, CTMI_BLK_LN2, CTMI_BLK_U, CTMI_BLK_G, CTMI_BLK_OUT, CTMI_BLK_NSLOTS
};
/* flags: the GPT-2 spelling of the same block (modeling_gpt.py:52-101, 122-149).  QKV_BLOCKED: the fused activation is q | k | v
 * ([T, 3, nh, hd], c_attn of modeling_gpt.py:69-72) instead of Bloom's head-interleaved [T, nh, 3, hd]; WGRAD_IN_OUT: the four weight
 * gradients are written [in, out] (Conv1D stores its weight that way, modeling_gpt.py:32-46); W_IN_OUT: the four weights passed in
 * are [in, out] too (the parameter's own layout — the forward GEMMs read them K-major, the data-gradient GEMMs as the row-major
 * operand, so no transposed compute copy is made); without it they are [out, in] as for Bloom. */
enum ctmi_block_flags { CTMI_BLK_QKV_BLOCKED = 1, CTMI_BLK_WGRAD_IN_OUT = 2, CTMI_BLK_W_IN_OUT = 4 };
typedef struct ctmi_bloom_block {
    int64_t B, S, H, nh;
    float eps; int32_t post_ln_res;     /* apply_residual_connection_post_layernorm (modeling_bloom.py:145-148,157) */
    int32_t dtype;
    int32_t flags;                      /* CTMI_BLK_* below: the same pre-LN block as GPT-2 lays it out (modeling_gpt.py:144-149) */
    float attn_scale;                   /* 0: 1/sqrt(head_dim); GPT-1/2 pass 1 or 1/sqrt(hd) (modeling_gpt.py:83-84 `scale`) */
    float future_fill;                  /* ctmi_attn_desc.future_fill: 0 = finfo.min (Bloom), -1e4 for GPT-2 */
    const float *ln1_w, *ln1_b; const void* wqkv; const float* bqkv; const void* wd; const float* bd;
    const float *ln2_w, *ln2_b; const void* w1; const float* b1; const void* w2; const float* b2;
    const float* slopes;                /* [nh] ALiBi slopes */
    const float* kpos; const int32_t* kvalid; const int32_t* first_valid;   /* ctmi_mask_prep outputs */
    const void* x;                      /* block input [B*S, H] */
    void* slab;                         /* saved activations + block output, ctmi_bloom_block_layout bytes */
} ctmi_bloom_block;
/* fills offsets[CTMI_BLK_NSLOTS] (bytes from the slab base) and returns the slab size in bytes */
int64_t ctmi_bloom_block_layout(int64_t B, int64_t S, int64_t H, int64_t nh, int dtype, int64_t* offsets /* host */);
int ctmi_bloom_block_fwd(const ctmi_bloom_block* blk /* host */, void* stream);

typedef struct ctmi_bloom_block_grads {
    const void* dout;                   /* gradient of the block output [B*S, H], compute dtype */
    void* dx;                           /* gradient of the block input  [B*S, H], compute dtype */
    float *dln1_w, *dln1_b, *dwqkv, *dbqkv, *dwd, *dbd, *dln2_w, *dln2_b, *dw1, *db1, *dw2, *db2;   /* fp32, overwritten */
    void* ws; int64_t ws_bytes;         /* scratch of >= ctmi_bloom_block_bwd_ws() bytes (intermediate gradients, partial rows) */
    void* splitk_ws; int64_t splitk_ws_bytes;   /* optional split-K slabs for the weight-gradient GEMMs (see ctmi_gemm) */
    void* side_stream;                  /* optional second hipStream_t: weight/bias gradients run there, concurrently with the
                                           data-gradient chain; joined back into `stream` before the call returns (host-side
                                           enqueue: the call itself never blocks) */
    void* side_splitk_ws; int64_t side_splitk_ws_bytes;
    int defer_join;                     /* (ABI v12) 0: the side stream is joined into `stream` before the call returns.  1 (needs side_stream): NO
                                           join — the parameter gradients and the bias / LayerNorm-affine reductions complete on side_stream; the
                                           caller makes `stream` wait for side_stream before anything reads them (one join at the end of the
                                           backward pass), and must not reuse `ws`, the slab or `dout` on `stream` before side_stream has drained
                                           this call's work (alternate two workspaces) */
} ctmi_bloom_block_grads;
int64_t ctmi_bloom_block_bwd_ws(int64_t B, int64_t S, int64_t H, int64_t nh, int dtype);
/* (ABI v13) 1 if ctmi_bloom_block_bwd takes the grouped weight-gradient launch (ctmi_wgrad_grouped) at this geometry / dtype / flags: the caller then
 * issues the whole backward on one stream (side_stream = NULL) — the grouped launch fills the GPU on its own; 0: the four products are separate
 * launches that gain from a side stream. */
int ctmi_bloom_block_wgrad_grouped(int64_t B, int64_t S, int64_t H, int dtype, int flags);
int ctmi_bloom_block_bwd(const ctmi_bloom_block* blk /* host */, const ctmi_bloom_block_grads* gr /* host */, void* stream);

/* ---- hardware probe (diagnostics: dumps MFMA / LDS-transpose lane layouts into out[]; used by tests only) */
int ctmi_probe(int which, const float* in /* device */, float* out /* device, 256 floats */, void* stream);
/* shader-clock probe (measurement; no reference counterpart): 2048 workgroups x 4 waves each issue 4*mfma_iters bf16 MFMAs; wave 0 of
 * workgroup 0 writes out[0] = elapsed shader cycles (s_memtime), out[1] = elapsed ticks of the constant 100 MHz counter (s_memrealtime):
 * shader clock under matrix load = 100 MHz * out[0] / out[1].  bench.py launches it right before and right after the timed region. */
int ctmi_clock_probe(int mfma_iters, unsigned long long* out /* device, 2 x u64 */, void* stream);
/* dynamic-LDS opt-in probe (tests only; no reference counterpart): requests `bytes` of dynamic LDS for a trivial kernel through the helper every
 * product launch uses (hipFuncSetAttribute, return CHECKED) and launches it; a request beyond the 160 KiB of a CU returns CTMI_ERR_LAUNCH. */
int ctmi_probe_dyn_lds(int64_t bytes, unsigned* out /* device, 1 x u32, may be NULL */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTMI355_H */
