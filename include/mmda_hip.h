/*
 * mmda_hip.h - C ABI of libmmda_hip.so: the MI355X (gfx950) hot path of MISA training.
 *
 * The reference (SoyeonHH/MMDA) is 100 % Python and has no FFI: every FLOP is a stock PyTorch module call
 * (SURVEY.md section 8b).  Each entry point below therefore names the reference *call site* whose arithmetic it
 * replaces.  The Python host in mmda_amd/ binds these with ctypes (mmda_amd/_lib.py); INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless the name ends in _host
 *   - all matrices row-major fp32; sequences time-major (T,B,d) like the reference (data_loader.py:70-72)
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *   - return 0 on success, negative MMDA_E* on error (no exceptions cross the ABI); the caller owns all buffers
 *   - `mode`: MMDA_F32 = exact fp32 (f32 MFMA, parity 1e-4), MMDA_BF16 = bf16 MFMA operands, fp32 accumulate/state
 */
#ifndef MMDA_HIP_H
#define MMDA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMDA_F32 0
#define MMDA_BF16 1

/* recurrent cell (reference config.py:147 --rnncell, models.py:39: nn.LSTM if 'lstm' else nn.GRU) */
#define MMDA_CELL_LSTM 0
#define MMDA_CELL_GRU 1

#define MMDA_OK 0
#define MMDA_EINVAL (-1)   /* bad argument / unsupported shape */
#define MMDA_ELAUNCH (-2)  /* hipLaunch / runtime error (hipGetLastError text via mmda_last_error) */

/* activation ids (reference config.py:25-27 activation_dict; prelu/rrelu unsupported) */
#define MMDA_ACT_NONE 0
#define MMDA_ACT_RELU 1
#define MMDA_ACT_SIGMOID 2
#define MMDA_ACT_LEAKYRELU 3
#define MMDA_ACT_TANH 4
#define MMDA_ACT_ELU 5
#define MMDA_ACT_HARDTANH 6
#define MMDA_ACT_HARDSHRINK 7
#define MMDA_ACT_PRELU 8          /* nn.PReLU(): x > 0 ? x : a x with ONE learned slope a (mmda_act_params.slope) */
#define MMDA_ACT_RRELU 9          /* nn.RReLU(): x >= 0 ? x : s x; training: s ~ U(lo, hi) per element, else s = (lo + hi) / 2 */
/* Parameters of the two parametrised activations of the reference's activation_dict (config.py:25-27); ignored by the others.
 * slope / dslope: device pointers to the PReLU slope and (backward) its gradient, accumulated with one atomic per row.
 * rand != 0 (training): the RReLU slope of element idx is lo + (hi - lo) * u(seed, site, idx), regenerated in the backward pass. */
typedef struct mmda_act_params {
  const float* slope; float* dslope;
  float lo, hi; int rand; uint64_t seed; int site;
} mmda_act_params;

const char* mmda_last_error(void);
int mmda_abi_version(void);
/* The GEMM entry points split long reductions over K and combine the slices DETERMINISTICALLY: every slice writes its partial tile
 * into a slab, a reduce launch on the same stream sums the slabs in slice order (no float atomics: two identical calls give
 * identical bits; reference train.py:46-51 asks for reproducible runs).  The slabs live in a scratch buffer the library keeps per
 * stream -- the one exception to "the caller owns every buffer": it is grown on demand (hipMalloc, i.e. during warm-up) and
 * released by this call (or at process exit).  Call it only when no GEMM of this library is in flight. */
int mmda_scratch_release(void);

/* ---------------------------------------------------------------------------------------------- GEMM
 * C[b] = epilogue( opA(A[b] (+A2[b])) * opB(B[b]) + bias[b] + bias2[b] (+ C[b] if accumulate) )
 *   opA: transA=0 -> A is (M,K) row-major with leading dim lda; transA=1 -> A is (K,M) row-major
 *   opB: transB=1 -> B is (N,K) row-major (an nn.Linear weight: y = x W^T); transB=0 -> B is (K,N)
 *   gather: optional int64 row ids: row m of A is A + gather[m]*lda (embedding lookup fused into the GEMM)
 *   epilogue: act (MMDA_ACT_*), then optional inverted dropout (p, seed, site) on the result,
 *             then optional gate: C *= (gate[m,n] > 0 ? gate_scale : 0)   (relu/dropout backward)
 * Replaces: every nn.Linear / aten::mm / addmm on the path (models.py:48-55 W_ih, :63-153, :160-161) and their
 * autograd transposes. */
typedef struct mmda_gemm_args {
  int mode, transA, transB, M, N, K, batch;
  const float* A;  int lda;  int64_t strideA;
  const float* A2;                              /* optional, same layout as A */
  const int64_t* gather;                        /* optional, M entries (transA must be 0) */
  const float* B;  int ldb;  int64_t strideB;
  float* C;        int ldc;  int64_t strideC;
  const float* bias; const float* bias2; int64_t strideBias;
  int accumulate, act;
  float drop_p; uint64_t drop_seed; int drop_site;
  const float* gate; int ldgate; float gate_scale;
  float alpha;                                  /* scales the product before bias/accumulate; 0 is read as 1 */
  float* bias_grad; float* bias_grad2;          /* optional (transA=1 weight-gradient GEMMs): bias_grad[m] += sum_k A[k,m], i.e. the
                                                   column sums of dY come out of the same MFMA pass as a virtual all-ones column
                                                   of B; bias_grad2 receives the same sums (b_ih and b_hh share a gradient) */
} mmda_gemm_args;
int mmda_gemm(const mmda_gemm_args* args, void* stream);
/* n independent GEMMs (any mix of shapes / layouts / modes) in one launch; results as n mmda_gemm calls in any order */
int mmda_gemm_grouped(const mmda_gemm_args* args, int n, void* stream);

/* ---------------------------------------------------------------------------------------------- bf16-operand GEMM
 * The LSTM-sized GEMMs of the bf16 mode read bf16 operands that are K-MAJOR (rows of A and of B are contiguous along k):
 *   C[M,N] (+)= alpha * A[M,K] * B[N,K]^T + bias + bias2,  fp32 accumulate and output.
 * Leading dimensions are in bf16 elements and must be multiples of 8 (16-byte rows); K may be ragged (k >= K reads as 0 as
 * long as the copies are zero-padded up to a multiple of 8, which mmda_convert_bf16 guarantees).  bias_grad[m] += sum_k A[m,k]
 * (a virtual all-ones row of B).  Operand copies come from mmda_convert_bf16 (plain and/or transposed).  Up to 16 problems go
 * out as one grouped launch; few-tile/long-K problems are split along K with float atomics (C zeroed first if !accumulate). */
typedef struct mmda_gemm_bf16_args {
  int M, N, K;
  const void* A; int lda;            /* bf16 (M, lda) */
  const void* B; int ldb;            /* bf16 (N, ldb) */
  float* C; int ldc;
  const float* bias; const float* bias2;
  float* bias_grad; float* bias_grad2;
  int accumulate; float alpha;       /* alpha 0 is read as 1 */
  /* gate interleave (the LSTM's gate-minor layout, see mmda_lstm_desc.gate_minor): index j of the interleaved axis stands for
   * torch's index orig(j) = (j / 4H) * 4H + (j % 4) * H + (j % 4H) / 4.   perm_n_H = H: bias/bias2 are read at orig(n)
   * (forward: B rows were interleaved by mmda_convert_bf16).  perm_m_H = H: C rows and bias_grad entries are written at
   * orig(m) (weight gradients: A rows are interleaved).  0 = no interleave. */
  int perm_n_H, perm_m_H;
  /* tn = 1: BOTH operands are M/N-major -- A is bf16 (K, lda) with row k holding A[., k] along m, B is bf16 (K, ldb) likewise:
   * C[M,N] (+)= alpha * A^T B, the weight-gradient form dW = dG^T X on the tensors as the backward pass leaves them (no transposed
   * copies: the kernel stages k-rows into LDS and takes its MFMA fragments with the transposing LDS read of gfx950,
   * ds_read_b64_tr_b16).  lda / ldb multiples of 4 (8-byte rows), bases 4-byte aligned; columns m >= M / n >= N of a row may hold
   * anything finite (they only reach outputs that are not stored).  bias_grad then is a virtual all-ones COLUMN n == N of B. */
  int tn;
} mmda_gemm_bf16_args;
int mmda_gemm_bf16_grouped(const mmda_gemm_bf16_args* args, int n, void* stream);
/* fp32 (rows, cols) matrix with leading dim ld -> bf16 copies: `plain` (rows, ldp) and/or `transposed` (cols, ldt); either may be
 * NULL.  ldp >= round_up(cols,8), ldt >= round_up(rows,8); the padding columns are written as zero.  `gather` (optional int64
 * row ids) reads row ids[r] of `src` instead of row r (embedding lookup).  Up to 16 jobs per launch. */
typedef struct mmda_convert_job {
  const float* src; int ld; int rows, cols;
  const int64_t* gather;
  void* plain; int ldp;
  void* transposed; int ldt;
  int row_perm_H;                    /* H > 0: output row r reads source row orig(r) (gate interleave, see mmda_gemm_bf16_args) */
  int src_bf16;                      /* 1: `src` points at bf16 elements (ld in elements): re-layout (transpose / pad) without conversion */
} mmda_convert_job;
int mmda_convert_bf16(const mmda_convert_job* jobs, int n, void* stream);
int mmda_debug_gemm_dma_mode(int mode);      /* tools/ only: bit 0 = the LDS-DMA kernel skips its MFMA work, bit 1 = its DMA (wrong results) */

/* ---------------------------------------------------------------------------------------------- block-scaled fp8 GEMM (MX)
 * The two feed-forward products of the fusion transformer layer (linear1 128 -> 2048, linear2 2048 -> 128; reference
 * models.py:160-161) on v_mfma_scale_f32_16x16x128_f8f6f4: OCP MX operands -- e4m3 elements, one E8M0 scale per 32 consecutive k
 * (shared exponent floor(log2 amax) - 8, elements clamped to +-448, round to nearest even) -- fp32 accumulate.  BASELINE.json
 * configs[4] ("mixed fp8 fusion GEMMs"); off by default (mmda_misa_set_fusion_fp8).
 * mmda_mx8_quant: fp32 (rows, K) row-major, K a multiple of 128 (16-byte loads when the rows are 16-byte aligned) -> q: rows * K element bytes in the MFMA's
 * operand order ([row][k-step of 128][lane group 0..3][32 bytes]) and s: rows * K / 32 scale bytes ([row][block]).  <= 8 jobs / launch.
 * mmda_gemm_mx8: C (M, N) = act(A (M, K) . B (N, K)^T + bias) * dropout, both operands quantised as above; N a multiple of 16. */
typedef struct mmda_mx8_quant_job {
  const float* src; int ld; int rows, K;
  unsigned char* q; unsigned char* s;
} mmda_mx8_quant_job;
typedef struct mmda_mx8_args {
  int M, N, K;
  const unsigned char* Aq; const unsigned char* As;
  const unsigned char* Bq; const unsigned char* Bs;
  float* C; int ldc;
  const float* bias; int act;
  float drop_p; uint64_t drop_seed; int drop_site;     /* element index m * N + n, as in the f32 path */
} mmda_mx8_args;
int64_t mmda_mx8_quant_bytes(int rows, int K);
int mmda_mx8_quant(const mmda_mx8_quant_job* jobs, int n, void* stream);
int mmda_gemm_mx8(const mmda_mx8_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------- row-skinny f32 GEMM
 * The fusion block's GEMMs have M = B or 6B rows (models.py:63-153,243-249 and their input gradients): one workgroup per
 * 32 x 16 output tile whose eight waves split K, exact f32 MFMA, operands straight from global memory, fixed-order reduction
 * (bitwise reproducible), fused epilogue.  Up to 8 independent problems per launch.
 *   raw   = alpha * ( (A (+A2)) * op(B)  +  A_2nd * op(B_2nd) )          op(B) = B^T (transB=1: B is (N, ldb), y = x W^T)
 *                                                                               or B   (transB=0: B is (K, ldb), dx = dy W)
 *   C[m,n]  = dsig( gate( dropout( act( raw + bias[n] (+ C[m,n] if accumulate) ) ) ) )
 *   C2[m,n] = dsig2( raw (+ C2[m,n] if accumulate) )                        (optional second destination of the same product)
 * gate: v *= gate[m,n] > 0 ? gate_scale : 0 (relu/dropout mask of a stored activation); dsig: v *= s (1 - s), s = dsig[m,n].
 * float4 operand loads are used when the bases are 16-B aligned, ld % 4 == 0 and K % 4 == 0; otherwise scalar loads. */
typedef struct mmda_skinny_args {
  int M, N, K, transB;
  const float* A; const float* A2; int lda;
  const float* B; int ldb;
  int K2; const float* A_2nd; int lda_2nd; const float* B_2nd; int ldb_2nd;   /* optional second product (K2 = 0: none) */
  float* C; int ldc;
  float* C2; int ldc2;
  const float* bias;
  int accumulate; float alpha; int act;                                        /* alpha 0 is read as 1 */
  float drop_p; uint64_t drop_seed; int drop_site;                             /* element index m * N + n */
  const float* gate; int ldgate; float gate_scale;
  const float* dsig; const float* dsig2; int lddsig;
} mmda_skinny_args;
int mmda_gemm_skinny(const mmda_skinny_args* args, int n, void* stream);

/* fp32 transposes, up to 20 per launch: dst[c * ldd + r] = src[r * ld + c] for r < rows, c < cols. */
typedef struct mmda_transpose_job { const float* src; int rows, cols, ld; float* dst; int ldd; } mmda_transpose_job;
int mmda_transpose_f32(const mmda_transpose_job* jobs, int n, void* stream);

/* column sums: out[n] += sum_m X[m*ld + n] (and out2[n] += the same, if out2 != NULL)   (bias gradients; atomics) */
int mmda_colsum(const float* X, int ld, int M, int N, float* out, float* out2, void* stream);

/* ---------------------------------------------------------------------------------------------- embedding
 * models.py:47,201 nn.Embedding forward; backward = dense scatter-add (sparse=False). */
int mmda_embed_gather(const float* W, const int64_t* ids, int rows, int dim, float* out, void* stream);
int mmda_embed_scatter_add(float* dW, const int64_t* ids, int rows, int dim, const float* dX, void* stream);
/* Deterministic form of the scatter-add for data-parallel ranks (and any caller that needs run-to-run identical sums):
 *   dW[id] = sum over the list positions p with ids[p] == id of rows[p], added in LIST ORDER (fixed two-level order: runs of <= 64
 * sorted positions, then the runs of a segment); rows of dW whose id occurs are OVERWRITTEN, all others are left alone; ids < 0 are
 * skipped.  ids (n) int64 and rows (n, D) fp32 on the device.  `work`: mmda_embed_segment_sum_work_bytes(n, D) bytes, 256-B aligned.
 * Ranks all-gather (ids, rows) and each runs this on the identical gathered list: replicas stay bit-identical, which the atomic
 * scatter-add cannot give (<= T*B rows per rank on the wire instead of V rows). */
int64_t mmda_embed_segment_sum_work_bytes(int n, int D);
int mmda_embed_segment_sum(float* dW, const int64_t* ids, int n, int D, const float* rows, void* work, int64_t work_bytes, void* stream);
/* In-place sum all-reduce of n_floats fp32 values on an RCCL communicator (ncclComm_t passed as void*) -- the data-parallel exchange
 * the reference never had (its only trace: the commented nn.DataParallel, solver.py:88-91) for hosts that own a communicator.
 * ncclAllReduce is resolved at run time (process symbols first, then librccl.so): no link-time dependency. */
int mmda_allreduce(void* buf, size_t n_floats, void* nccl_comm, void* stream);

/* ---------------------------------------------------------------------------------------------- LayerNorm
 * y = LN(act(x) + res * dropmask) * gamma + beta over the last dim n (eps 1e-5).   models.py:155-157,172 and the
 * LayerNorms inside project_* (models.py:65-80) and the fusion layer (torch TransformerEncoderLayer norm1/norm2).
 *   permute_sb > 0: rows are (s,b) with s-major, b < permute_sb... output row (s*Bp + b) is written at
 *   out + b*(S*n) + s*n  (i.e. (S,B,n) -> (B,S,n)), used to emit h = cat(h[0..5],dim=1) (models.py:245).
 *   stash: mean,rstd (rows each). */
typedef struct mmda_ln_args {
  int rows, n;
  const float* x; const float* res;             /* res optional */
  const float* gamma; const float* beta;
  float* y; float* mean; float* rstd;
  int act;                                      /* applied to x before the residual add */
  float drop_p; uint64_t drop_seed; int drop_site;   /* dropout on res */
  int permute_S, permute_B;                     /* 0,0 = no permutation */
  float eps;
  mmda_act_params actp;                         /* act = MMDA_ACT_PRELU / MMDA_ACT_RRELU only */
  void* y_bf16; int ld_bf16;                    /* optional second output (then y may be NULL): y as bf16 (rows, ld_bf16), columns n..ld_bf16-1 zero -- the
                                                 * K-major operand copy the next layer's input GEMM reads (no conversion launch between) */
} mmda_ln_args;
int mmda_layernorm_fwd(const mmda_ln_args* a, void* stream);
/* backward: dx_pre = LN'(dy); outputs: d_x = dx_pre * act'(x) (written or accumulated), d_res = dx_pre*dropmask,
 * dgamma += , dbeta += .  dy may be given in permuted (B,S,n) layout (same permute args as forward). */
typedef struct mmda_ln_bwd_args {
  int rows, n;
  const float* dy; const float* x; const float* res; const float* gamma;
  const float* mean; const float* rstd;
  float* d_x; int accumulate_dx; float* d_res;  /* either may be NULL */
  float* dgamma; float* dbeta;                  /* accumulated with atomics */
  int act; float drop_p; uint64_t drop_seed; int drop_site;
  int permute_S, permute_B;
  mmda_act_params actp;                         /* act = MMDA_ACT_PRELU / MMDA_ACT_RRELU only (dslope accumulated when d_x is written) */
} mmda_ln_bwd_args;
int mmda_layernorm_bwd(const mmda_ln_bwd_args* a, void* stream);
/* Several independent LayerNorms in one launch (the three modalities'): results as n single calls. */
int mmda_layernorm_fwd_multi(const mmda_ln_args* args, int n, void* stream);
int mmda_layernorm_bwd_multi(const mmda_ln_bwd_args* args, int n, void* stream);
/* dgamma / dbeta alone (same argument struct; d_x / d_res are ignored).  With dgamma = dbeta = NULL mmda_layernorm_bwd computes
 * only the input gradients, so the parameter gradients of the large inter-layer LayerNorms (rows = T*B) can run on another
 * stream, off the path into the next recurrent kernel. */
int mmda_layernorm_param_grads(const mmda_ln_bwd_args* args, int n, void* stream);

/* ---------------------------------------------------------------------------------------------- biLSTM
 * Recurrent part of nn.LSTM(bidirectional=True) on a packed sequence (models.py:48-55 via extract_features
 * :163-180), PyTorch gate order i,f,g,o, variable lengths (a sample stops updating at t >= len_b; the reverse
 * direction starts at len_b-1), final h written straight into the utterance layout of models.py:203.
 *
 * Weight packing: W_hh (4H,H) of each direction must first be packed into MFMA fragment order with
 * mmda_lstm_pack_whh (once per optimizer step).  Packed sizes from mmda_lstm_packed_bytes. */
int64_t mmda_lstm_packed_bytes(int mode, int H, int which);   /* which: 0 forward, 1 backward, 2 cluster-backward (bf16 only) */
int mmda_lstm_pack_whh(int mode, int H, const float* whh, void* packed_fwd, void* packed_bwd, void* stream);
/* the cluster-backward packing of one matrix (bf16): [(ht*nHT + nt)*2 + ks2][lane][8] */
int mmda_lstm_pack_whh_cluster(int H, const float* whh, void* packed_c, void* stream);
/* n (<= 16) matrices in one launch */
int mmda_lstm_pack_whh_multi(int mode, int n, const int* H, const float* const* whh, void* const* packed_fwd,
                             void* const* packed_bwd /* entries may be NULL: that packing is skipped (as may packed_fwd's, not both) */,
                             void* const* packed_c /* NULL or per-matrix (bf16) */, void* stream);
/* The same packings (bf16 mode) and up to 16 conversion jobs (mmda_convert_bf16) in ONE launch: both depend only on the step's inputs
 * and weights; as two launches on two streams they cost a fork and a cross-stream wait in front of the first recurrent kernel. */
int mmda_lstm_pack_whh_and_convert(int n, const int* H, const float* const* whh, void* const* packed_fwd, void* const* packed_bwd,
                                   void* const* packed_c, const mmda_convert_job* jobs, int njobs, void* stream);
/* ... and up to 20 fp32 transposes (mmda_transpose_f32) in the same launch: the K-major copies of the fusion block's weights that
 * the backward pass reads depend on the weights only, and a launch of their own costs a stream 7 - 15 us for 6 MB of traffic. */
int mmda_lstm_pack_convert_transpose(int n, const int* H, const float* const* whh, void* const* packed_fwd, void* const* packed_bwd,
                                     void* const* packed_c, const mmda_convert_job* jobs, int njobs, const mmda_transpose_job* tjobs,
                                     int ntjobs, void* stream);

typedef struct mmda_lstm_desc {
  int H;
  float* gates;        /* (T,B,2,4H)  in: x W_ih^T + b_ih + b_hh per direction; out: activated i,f,g,o (stash) */
  float* cstash;       /* (T,B,2,H)   cell state after each step (stash).  With gate_minor = 1 the resident kernels keep it
                          batch-minor-by-4, (T, ceil(B/4), 2, H, 4): allocate T * round_up(B,4) * 2 * H floats */
  float* hseq;         /* (T,B,2H)    layer output [fwd H | rev H], zero at padded positions */
  const void* wpack[2];/* packed W_hh per direction (forward packing for fwd, backward packing for bwd) */
  const void* wpack_c[2]; /* backward only, optional: "cluster-backward" packing (mmda_lstm_packed_bytes(mode,H,2)) used by the
                          resident-weights kernel; NULL -> the streaming backward kernel runs */
  float* utt;          /* (B,4H)      final-h destination / its gradient source in backward */
  int layer;           /* 0 or 1: column block (dir*2+layer)*H of utt */
  const float* d_hseq; /* backward only: (T,B,2H) gradient w.r.t. hseq, or NULL */
  void* xchg;          /* optional cluster-exchange buffer of mmda_lstm_xchg_bytes(H,B) bytes, ZEROED once by the caller when
                          allocated.  Non-NULL (all descriptors) + MMDA_BF16 selects the LDS-resident-weights kernels; NULL
                          selects the streaming kernels. */
  uint32_t epoch_base; /* cluster kernels: monotonic epoch counter of descs[0] is used for the launch; the caller advances it by
                          at least T+1 between launches that share an xchg buffer (flags are never reset) */
  int gate_minor;      /* 0: `gates` columns are [dir][gate][unit] (torch's weight_ih row order).  1: [dir][unit][gate], i.e. the four
                          gates of one hidden unit are 16 contiguous bytes: the resident-weights kernels then move them with one
                          16-byte access instead of four 4-byte ones (the address unit, shared by the CU's four waves, is what
                          their per-step stash traffic is bound by).  Only the resident-weights kernels accept 1. */
  int forward_only;    /* forward: 1 = no backward pass will follow (evaluation): the activated gates and the cell states need not be
                          stashed; `gates` / `cstash` contents are then unspecified after the call.  hseq and utt are written as usual.
                          Honoured by the wave-autonomous kernel, ignored (stash written) by the others. */
  int cell;            /* MMDA_CELL_LSTM (0) or MMDA_CELL_GRU.  GRU: the caller pads torch's three gate blocks into the four slots,
                          W_ih / b_ih rows [r; z; n; 0] and W_hh / b_hh rows [r; z; 0; n] (so `gates` slot 2 = x W_in^T + b_in and slot 3 =
                          b_hn), see mmda_gru_pad_params.  Stash after forward: gates = [r, z, n, h W_hn^T + b_hn], cstash = h_t;
                          after backward gates = d[pre_r, pre_z, pre_n, h W_hn^T + b_hn].  Runs on the streaming kernels and on the
                          wave-autonomous resident-weights kernels (H <= 320; gate_minor allowed there), not on the barrier-form
                          resident kernels; all descriptors of one launch share the cell. */
  void* dg_bf16;       /* backward, optional: (T*B, 8H) bf16, the gate gradients rounded to bf16 in the column order of `gates` (the A
                          operand of dX = dG W_ih on the bf16 GEMM; mmda_convert_bf16 with src_bf16 makes its transpose for the
                          weight-gradient GEMMs).  Written only by the wave-autonomous resident kernel with gate_minor = 1; NULL
                          otherwise / ignored by the other kernels, so a caller passes it only when
                          mmda_lstm_resident_applicable() said those kernels will run. */
  int dg_bf16_only;    /* with dg_bf16: 1 = the fp32 gate gradients are NOT written (`gates` is unspecified after the call): the
                          per-step stores are what the backward kernel's memory pipeline is busy with */
} mmda_lstm_desc;
int64_t mmda_lstm_xchg_bytes(int H, int B);
/* 1 if mmda_lstm_fwd/bwd would run these descriptors on the resident-weights kernels (so gate_minor = 1 may be used), else 0 */
int mmda_lstm_resident_applicable(int mode, int n, const mmda_lstm_desc* descs, int B, int T, int backward);
/* 1 if mmda_lstm_bwd on these descriptors will write mmda_lstm_desc.dg_bf16 (wave-autonomous resident kernel, gate_minor = 1) */
int mmda_lstm_bwd_emits_dg_bf16(int mode, int n, const mmda_lstm_desc* descs, int B, int T);
/* diagnostics only: 8 x uint64 per workgroup, phase cycle sums of the resident-weights forward kernel (NULL disables) */
int mmda_debug_set_lstm_stamps(void* device_buffer);
/* up to 4 independent biLSTMs (modalities) in ONE launch; all share B, T and lengths (device int32, B entries) */
int mmda_lstm_fwd(int mode, int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream);
/* backward: reads gates/cstash (forward stash), utt = d(utterance), d_hseq; overwrites `gates` with d(pre-activation)
 * (zero at padded positions) for the time-batched weight/input gradient GEMMs. */
int mmda_lstm_bwd(int mode, int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream);

/* GRU parameters <-> the four-slot layout the recurrent kernels and the time-batched GEMMs work on (mmda_lstm_desc.cell).
 * torch side (per direction d): w_ih[d] (3H, D), w_hh[d] (3H, H), b_ih[d] (3H), b_hh[d] (3H), gate blocks r, z, n (nn.GRU).
 * padded side: pw_ih (8H, D) rows [dir][r; z; n; 0], pw_hh[d] (4H, H) rows [r; z; 0; n], pb_ih / pb_hh (8H) alike.
 * mmda_gru_pad_params : padded <- torch.
 * mmda_gru_unpad_grads: the same pointers name GRADIENTS: torch-side += padded, padded <- 0.  The padded bias gradient is ONE
 *   vector, pb_ih (8H) = column sums of the four-slot gate gradients (what the weight-gradient GEMMs emit); it feeds both b_ih
 *   (slots r, z, n) and b_hh (slots r, z and the fourth); pb_hh is not read.  Rows of the padded weight gradients that have no
 *   torch gate (W_ih slot 4, W_hh slot 3) hold by-products of the padded GEMMs and are dropped. */
#define MMDA_GRU_PAD_MAX 6
typedef struct mmda_gru_pad_job {
  int H, D;
  float* w_ih[2]; float* w_hh[2]; float* b_ih[2]; float* b_hh[2];
  float* pw_ih; float* pw_hh[2]; float* pb_ih; float* pb_hh;
} mmda_gru_pad_job;
int mmda_gru_pad_params(const mmda_gru_pad_job* jobs, int n, void* stream);
int mmda_gru_unpad_grads(const mmda_gru_pad_job* jobs, int n, void* stream);

/* ---------------------------------------------------------------------------------------------- fusion attention
 * Self-attention core of nn.TransformerEncoderLayer(d_model=E, nhead) on (S,B,E) with S small (6): softmax(QK^T/sqrt(hd))V
 * per (sample, head), attention-prob dropout.  qkv: (S*B, 3E) rows (s,b); ctx: (S*B, E); probs: (B,nhead,S,S). */
int mmda_attn_fwd(const float* qkv, int S, int B, int E, int nhead, float* ctx, float* probs,
                  float drop_p, uint64_t seed, int site, void* stream);
int mmda_attn_bwd(const float* qkv, const float* probs, const float* dctx, int S, int B, int E, int nhead,
                  float* dqkv, float drop_p, uint64_t seed, int site, void* stream);

/* ---------------------------------------------------------------------------------------------- elementwise
 * y = a + b */
int mmda_add(const float* a, const float* b, float* y, int64_t n, void* stream);
/* d *= y*(1-y)  (sigmoid backward, in place) */
int mmda_sigmoid_bwd_inplace(float* d, const float* y, int64_t n, void* stream);
/* h = dropout(act(z));  dz = dh * dropmask * act'(z)   (discriminator hidden layer, models.py:124-126) */
int mmda_act_dropout_fwd(const float* z, float* h, int64_t n, int act, float drop_p, uint64_t seed, int site, void* stream);
int mmda_act_dropout_bwd(const float* dh, const float* z, float* dz, int64_t n, int act, float drop_p, uint64_t seed, int site,
                         void* stream);
/* the same for the parametrised activations (MMDA_ACT_PRELU / MMDA_ACT_RRELU): parameters in *ap (host struct, copied) */
int mmda_act_dropout_fwd_p(const float* z, float* h, int64_t n, int act, const mmda_act_params* ap, float drop_p, uint64_t seed, int site,
                           void* stream);
int mmda_act_dropout_bwd_p(const float* dh, const float* z, float* dz, int64_t n, int act, const mmda_act_params* ap, float drop_p,
                           uint64_t seed, int site, void* stream);

/* ---------------------------------------------------------------------------------------------- heads + losses
 * logits12 (B,12) = h [W_conf;W_cls]^T + b.  tcp = sigmoid(logits[:, :6]); scores = sigmoid(dropout(logits[:,6:]));
 * labels = scores > threshold.   models.py:138-153,247-249, functions.py:112-115 */
int mmda_heads_fwd(const float* logits, int B, int ncls, float threshold, float* tcp, float* scores, float* labels,
                   float drop_p, uint64_t seed, int site, void* stream);
/* dlogits from dscores/dtcp (either may be NULL = zero) */
int mmda_heads_bwd(const float* tcp, const float* scores, const float* dtcp, const float* dscores, int B, int ncls,
                   float* dlogits, float drop_p, uint64_t seed, int site, void* stream);

/* Every loss entry point computes the loss value AND d(loss)/d(inputs) in one pass ("gradient in forward").
 * `scale` multiplies the gradients (loss weight); *loss gets the UNscaled value added (zero it first).
 * Gradients are ACCUMULATED into the d_* buffers.
 * cls: solver.py:373-385   sum_c mean_b BCE (log clamp -100) */
int mmda_loss_cls(const float* scores, const float* emo, int B, int ncls, float scale, float* loss, float* dscores, void* stream);
/* conf: solver.py:451-462 */
int mmda_loss_conf(const float* scores, const float* tcp, const float* emo, int B, int ncls, float scale,
                   float* loss, float* dscores, float* dtcp, void* stream);
/* diff: solver.py:422-441 + functions.py:54-78.  x: 6 tensors (B,D) at x + k*stride, order
 * [private_t, private_v, private_a, shared_t, shared_v, shared_a]; the six reference pairs are built in. */
int mmda_loss_diff(const float* x, int64_t stride, int B, int D, float scale, float* loss, float* dx, float* work, void* stream);
int64_t mmda_loss_diff_work_floats(int B, int D);
/* general form: nt (2..6) tensors at x + k*stride, np (1..6) index pairs (host array of 2*np ints).  The utils.DiffLoss
 * module call DiffLoss()(a, b) (functions.py:54-78) is nt=2, pairs={0,1}. */
int mmda_loss_diff_pairs(const float* x, int64_t stride, int nt, int np, const int* pairs_host, int B, int D, float scale,
                         float* loss, float* dx, float* work, void* stream);
/* cmd: solver.py:409-420 + functions.py:88-109 over shared_t, shared_v, shared_a = x + k*stride (k=0..2), 5 moments */
int mmda_loss_cmd(const float* x, int64_t stride, int B, int D, float scale, float* loss, float* dx, void* stream);
/* general form: nt (2..3) tensors, np pairs, n_moments (1..5); *loss += value_scale * sum over pairs; gradients are
 * scaled by scale*value_scale.  utils.CMD()(x1, x2, n) (functions.py:88-109) is nt=2, pairs={0,1}, value_scale=1. */
int mmda_loss_cmd_pairs(const float* x, int64_t stride, int nt, int np, const int* pairs_host, int n_moments, int B, int D,
                        float scale, float value_scale, float* loss, float* dx, void* stream);
/* recon: solver.py:443-449.  recon/orig: 3 tensors (B,D) each at +k*stride */
int mmda_loss_recon(const float* recon, const float* orig, int64_t stride, int B, int D, float scale, float* loss,
                    float* drecon, float* dorig, void* stream);
/* domain: solver.py:388-407.  dom: (3,B,3) logits stacked t,v,a; labels 0/1/2 */
/* cls + conf (ConfidNet, ncls == 6 only, with_conf) + recon over n_recon contiguous elements in ONE launch, plus the weighted
 * total L[5] = L[0] + diff_w L[1] + sim_w L[2] + recon_w L[3] (+ conf_w L[4] if use_conf) written by the last block to finish.
 * L: 8 floats {cls, diff, sim, recon, conf, total, -, ticket}; L[0], L[3], L[4], L[7] must be zero on entry, L[1], L[2] final.
 * Gradients (optional) are ADDED to d_scores / d_tcp (atomics: cls and conf share d_scores) and d_recon / d_orig.
 * Replaces: solver.py:165-181 (the criterion, get_conf_loss, get_recon_loss calls and the weighted sum). */
int mmda_loss_misc(const float* scores, const float* tcp, const float* emo, int B, int ncls, float* d_scores, float* d_tcp,
                   int with_conf, int conf_grads, float conf_scale, const float* recon, const float* orig, int64_t n_recon,
                   float recon_scale, float* d_recon, float* d_orig, float* L, float diff_w, float sim_w, float recon_w,
                   float conf_w, int use_conf, void* stream);
/* Evaluation counts on the device (reference utils/eval.py:14-31 get_accuracy and the tp/fp/fn that sklearn's
 * f1/precision/recall in get_metrics :33-65 are functions of), ACCUMULATED into `state` = 3*C + 2 doubles (zeroed by the caller
 * before the first batch): tp[C], fp[C], fn[C], sum_i |y_i & p_i| / max(|y_i | p_i|, 1), number of samples.  pred / truth are
 * (N, C) fp32, an entry counts as set when > 0.  C <= 16.  One read-back per evaluation pass instead of one per batch. */
int mmda_eval_accumulate(const float* pred, const float* truth, int N, int C, double* state, void* stream);
int mmda_loss_domain(const float* dom, int B, float scale, float* loss, float* ddom, void* stream);

/* ---------------------------------------------------------------------------------------------- optimizer
 * clip_grad_value_(clip) then Adam (solver.py:185-186, :97-99; betas 0.9/0.999, eps 1e-8, no weight decay), fused over
 * a flat bucket.  step = 1-based step count.  grad_scale multiplies g first (1/world for DP averaging). */
int mmda_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float clip, float grad_scale, int step, void* stream);
int mmda_clamp(float* g, int64_t n, float clip, void* stream);
/* The same update over the rows of a (rows, dim) table whose mask byte equals `want`; mmda_mark_rows clears a mask of `rows` bytes and
 * sets the bytes of the ids that occur (ids outside [0, rows) are ignored).  The fused training step uses the pair to update the
 * embedding rows a batch does not touch (zero gradient, known from the ids alone) beside the last recurrence -- torch.optim.Adam's dense
 * update (reference solver.py:97-99, nn.Embedding without sparse=True) moves every row with momentum, touched or not -- and only the
 * touched rows after the scatter. */
int mmda_clamp_adam_rows(float* p, const float* g, float* m, float* v, int rows, int dim, const unsigned char* mask, int want, float lr,
                         float beta1, float beta2, float eps, float clip, float grad_scale, int step, void* stream);
int mmda_mark_rows(unsigned char* mask, int rows, const int64_t* ids, int n, void* stream);
/* clip_grad_value_(clip) + torch.optim.RMSprop with torch's defaults besides lr (alpha 0.99, eps 1e-8, no momentum, not centered):
 * the other entry of the reference's optimizer_dict (config.py:24).  grad_scale as in mmda_clamp_adam. */
int mmda_clamp_rmsprop(float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha, float eps, float clip,
                       float grad_scale, void* stream);

/* ============================================================================================== whole-model API
 * The reference's per-batch loop body (solver.py:139-186) as five calls.  `mmda_misa` is the native runtime object
 * behind mmda_amd.models.MISA: it owns NO device memory - the host binds one flat fp32 parameter bucket (+ grad, Adam
 * m/v buckets of the same size) and one workspace; the library only lays tensors out inside them.
 *
 * Flat bucket layout: all non-embedding parameters first ("dense" part, the RCCL all-reduce bucket), embed.weight last.
 * mmda_misa_param_info enumerates (state_dict key, offset, rows, cols) - keys are exactly the reference's
 * state_dict keys, so reference checkpoints load by name. */
typedef struct mmda_misa mmda_misa;
typedef struct mmda_misa_config {
  int vocab, d_t, d_v, d_a, hidden, ncls;       /* config.embedding_size / visual_size / acoustic_size / hidden_size / num_classes */
  int act;                                      /* MMDA_ACT_* for project_* and the discriminator (config.activation) */
  int use_cmd_sim, use_confidNet;               /* config.use_cmd_sim / use_confidNet */
  float dropout;                                /* config.dropout: classifier + discriminator dropout */
  float fusion_dropout;                         /* nn.TransformerEncoderLayer default 0.1 (not a reference flag) */
  float threshold, reverse_grad_weight;         /* config.threshold, config.reverse_grad_weight */
  float diff_weight, sim_weight, recon_weight, conf_weight;   /* config.py:134-138 */
  int mode;                                     /* MMDA_F32 / MMDA_BF16 for the encoder + fusion GEMMs and the recurrences */
  int rnncell;                                  /* MMDA_CELL_LSTM / MMDA_CELL_GRU (config.rnncell, models.py:39) */
} mmda_misa_config;

int mmda_misa_create(const mmda_misa_config* cfg, mmda_misa** out);
void mmda_misa_destroy(mmda_misa* m);
int mmda_misa_num_params(const mmda_misa* m);
int mmda_misa_param_info(const mmda_misa* m, int i, const char** name, int64_t* offset, int* rows, int* cols);
int64_t mmda_misa_flat_floats(const mmda_misa* m);    /* whole bucket incl. embedding (multiple of 4) */
int64_t mmda_misa_dense_floats(const mmda_misa* m);   /* non-embedding prefix */
int mmda_misa_bind(mmda_misa* m, float* params, float* grads, float* adam_m, float* adam_v);
int64_t mmda_misa_workspace_floats(const mmda_misa* m, int B, int T);
int mmda_misa_set_workspace(mmda_misa* m, float* ws, int64_t floats, int B, int T);
/* The same with the (rare) clears enqueued on `stream`.  The exchange buffers of the recurrences sit at the front of the workspace at
 * offsets that depend on B alone and are cleared only when the buffer or B changes, so a new T per batch (the reference's collate pads
 * to the batch maximum, data_loader.py:70-72) costs no device work; abort words found before a clear stay visible through
 * mmda_misa_cluster_status.  Before handing over a NEW buffer read mmda_misa_cluster_status yourself: the old one is not touched. */
int mmda_misa_set_workspace_async(mmda_misa* m, float* ws, int64_t floats, int B, int T, void* stream);
/* offset (in floats, inside the workspace) of a named activation / activation-gradient, -1 if unknown. Names:
 *   scores labels tcp logits hfused x6 orig recon dom utt_t utt_v utt_a losses
 *   d_scores d_tcp d_x6 d_orig d_recon d_dom   xchg_t xchg_v xchg_a (exchange buffers; word 0 of each = its sticky abort word)
 * x6 = [private_t, private_v, private_a, shared_t, shared_v, shared_a] each (B,hidden); orig/recon = (3,B,hidden);
 * losses = float[8]: cls, diff, sim, recon, conf, total */
int64_t mmda_misa_tensor_offset(const mmda_misa* m, const char* name);
/* set the mode / loss switches after creation (bench toggles) */
int mmda_misa_set_mode(mmda_misa* m, int mode);
/* bf16 recurrences: 1 (default) = W_hh resident in LDS across a cluster of workgroups, 0 = streamed from L2 every step */
int mmda_misa_set_recurrence(mmda_misa* m, int resident_weights);
/* bf16 mode only: 1 (default) = the LSTM-sized GEMMs read bf16 operand copies made by mmda_convert_bf16 (gemm_bf16.hip);
 * 0 = they stage the fp32 tensors through the generic kernel (same rounding of the operands, different summation order). */
int mmda_misa_set_gemm_operands(mmda_misa* m, int bf16_copies);
/* 1 = forward passes are evaluation passes (no backward follows): the recurrences skip their stash stores and the transposed
 * bf16 copies that only the weight-gradient GEMMs read are not made.  Default 0.  mmda_misa_backward after a forward in this mode
 * returns MMDA_EINVAL. */
int mmda_misa_set_inference(mmda_misa* m, int forward_only);
/* 1 = the forward feed-forward products of the fusion transformer layer (linear1 / linear2) run on block-scaled fp8 operands
 * (mmda_gemm_mx8); their backward stays on the exact f32 path with the stored activations (straight-through).  Default 0. */
int mmda_misa_set_fusion_fp8(mmda_misa* m, int on);
/* Data parallel: the gradient bucket is laid out in the order the backward pass completes it (fusion block, LayerNorms,
 * layer-2 recurrent layers, layer-1 recurrent layers, embedding).  After mmda_misa_backward / mmda_misa_train_step has been
 * ISSUED, the first mmda_misa_early_grad_floats() floats of the bucket are final as soon as an event recorded inside that call
 * fires -- beside the layer-1 backward recurrence, long before the call's last kernel; mmda_misa_wait_early_grads makes `stream`
 * wait for that event, so a collective enqueued on it overlaps the rest of the backward pass.  0 floats = nothing is early. */
int64_t mmda_misa_early_grad_floats(const mmda_misa* m);
int mmda_misa_wait_early_grads(mmda_misa* m, void* stream);
/* 1 (default) = weight-gradient GEMMs run on an internal side stream underneath the recurrent kernels (joined before
 * mmda_misa_backward returns control of `stream`); 0 = everything on `stream` */
int mmda_misa_set_overlap(mmda_misa* m, int side_stream);
/* *aborted_host: bit 0 = a cluster exchange of a recurrence ever timed out, bit 1 = a kernel of the fused training step ever timed
 * out waiting on the device for the other stream's chain (flag joins: e.g. under a tool that serialises dispatches -- run those with
 * MMDA_FLAG_JOIN=0 MMDA_SORT_EARLY=0); results after either are invalid.  Synchronous D2H, off the step path */
int mmda_misa_cluster_status(const mmda_misa* m, int* aborted_host);

/* models.py:282-285 forward.  t_ids (T,B) int64, v (T,B,d_v), a (T,B,d_a) device; lengths (B) int32 device.
 * training != 0 enables dropout with the given seed.  Packs W_hh first (weights may have changed). */
int mmda_misa_forward(mmda_misa* m, const int64_t* t_ids, const float* v, const float* a, const int32_t* lengths,
                      int training, uint64_t seed, void* stream);
/* solver.py:163-181: the six losses into `losses`; with_grads != 0 also zeroes and seeds the d_* activation gradients
 * with the weighted loss gradients (loss.backward() seeds). emo (B,ncls) fp32. */
int mmda_misa_losses(mmda_misa* m, const float* emo, int with_grads, void* stream);
/* Data-parallel "global statistics" mode (no reference counterpart: the reference is single-device, where DiffLoss
 * functions.py:64-76, CMD :89-108 and the confidence loss solver.py:451-462 see the whole batch).  on != 0: the next mmda_misa_losses
 * (with_grads) does NOT clear the activation gradients and does not compute DiffLoss / CMD / conf -- the caller has, behind
 * mmda_misa_zero_act_grads, on the all-gathered tensors of every rank with mmda_loss_diff / mmda_loss_cmd / mmda_loss_conf, added
 * the sums into losses[1], [2], [4] and this rank's gradient rows (times the world size: the exchange averages) into d_x6 / d_scores /
 * d_tcp (mmda_misa_tensor_offset).  cls, recon and the weighted total are added as usual. */
int mmda_misa_set_external_batch_losses(mmda_misa* m, int on);
/* solver.py:183 loss.backward(): from d_scores/d_tcp/d_x6/d_orig/d_recon/d_dom to every parameter gradient
 * (ACCUMULATED into the bound grad bucket: zero it first with mmda_misa_zero_grad).  External seeds (the autograd
 * compat path) are whatever the caller left in the d_* buffers. */
int mmda_misa_backward(mmda_misa* m, const int64_t* t_ids, const float* v, const float* a, const int32_t* lengths,
                       void* stream);
int mmda_misa_zero_grad(mmda_misa* m, void* stream);
int mmda_misa_zero_act_grads(mmda_misa* m, void* stream);
/* solver.py:185-186: clip_grad_value_(clip) + Adam over the whole bucket; grad_scale = 1/world after an all-reduce */
int mmda_misa_adam_step(mmda_misa* m, float lr, float clip, float grad_scale, int step, void* stream);
/* zero_grad + forward + losses + backward (+ adam if do_adam) = one reference loop iteration */
int mmda_misa_train_step(mmda_misa* m, const int64_t* t_ids, const float* v, const float* a, const int32_t* lengths,
                         const float* emo, int training, uint64_t seed, int do_adam, float lr, float clip, int step,
                         void* stream);

/* Per-kernel timing of the recurrent launches with HIP events recorded on the SAME stream the kernels run on
 * (bench.py's roofline leg).  begin(max_steps) arms it; every train_step/forward/backward then brackets its four
 * recurrent launches (0: fwd layer1, 1: fwd layer2, 2: bwd layer2, 3: bwd layer1); collect() synchronises the events and
 * returns the mean milliseconds per launch of each and the number of timed steps; end() disarms and frees the events. */
/* record the recurrent kernels' event pairs only on every stride-th step (default 1; set before mmda_misa_timing_begin, whose
 * max_steps then counts RECORDED steps): eight event records per step cost ~35 us of a 1.4 ms step */
int mmda_misa_timing_stride(mmda_misa* m, int stride);
/* on != 0: a sampled step brackets ONE of the four recurrent launches (they take turns): two event records (~9 us) instead of eight;
 * mmda_misa_timing_collect then returns per-launch means over the steps that sampled that launch and, in *steps, the fewest samples behind a mean */
int mmda_misa_timing_rotate(mmda_misa* m, int on);
int mmda_misa_timing_begin(mmda_misa* m, int max_steps);
int mmda_misa_timing_collect(mmda_misa* m, float mean_ms[4], int* steps);
int mmda_misa_timing_end(mmda_misa* m);

#ifdef __cplusplus
}
#endif
#endif
