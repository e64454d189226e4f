// Internal (not part of the C ABI): the row-local stretches of the fusion block's backward pass as ONE launch each.
// Every operation between the classifier and the utterance vectors acts on the rows of one sample at a time (linear layers, LayerNorms,
// the six-token attention, sigmoids, dropout) -- only the weight gradients and the losses mix samples, and those run elsewhere.  A
// workgroup therefore takes the rows of `nb` samples through a whole stretch by itself, its stages separated by workgroup barriers
// (~1 us each: store drain + L2 round trip) instead of kernel boundaries (5 - 12 us each for these latency-bound sizes).
#pragma once
#include "../../include/mmda_hip.h"
#include <stdint.h>

// stretch C: heads backward -> d_hfused = d_logits W_head -> LayerNorm 2 backward (d_x1, d_f2 + its gamma/beta gradients)
struct FusedBwdC {
  int B, hs, ncls, nb;
  const float* tcp; const float* scores; const float* d_tcp; const float* d_scores; float* d_logits;
  float p_cls; uint64_t seed; int site_cls;
  const float* head_w;            // (6 + ncls, 6 hs) row-major
  float* d_hfused;                // (B, 6 hs)
  mmda_ln_bwd_args ln2;           // rows = 6 B (token-major), permute_S / permute_B set
  float* pg_parts;                // (B, 5, 2, hs): per-sample partial gamma / beta gradients of the block's five LayerNorms (slot 0: ln2)
  // optional: rec_part (3, B, hs) = d_recon[i] W_rec[i], made by a second set of workgroups of this launch for stretch A (all three
  // given or none)
  const float* d_recon; const float* rec_wT; float* rec_part;
};

// stretch A: LayerNorm 1 backward -> d_ctx = d_attn_out W_out -> attention backward -> d_x6 = (d_x6 + d_qkv W_in + d_recon W_rec) s(1-s)
//            -> d_orig += d_private W_priv + d_shared W_shared -> the three projection LayerNorms' backward (d_z)
struct FusedBwdA {
  int B, hs, nhead, nb;
  mmda_ln_bwd_args ln1;           // rows = 6 B; d_x = d_x6 (accumulated), d_res = d_attn_out
  const float* ffn_parts; int n_parts; float* d_x1;                       // optional: d_x1 += sum of the partials (slice order), first
  const float* d_attn_out; const float* out_wT; float* d_ctx;
  const float* qkv; const float* probs; float* d_qkv; float p_tf; uint64_t seed; int site_attn;
  const float* in_wT;             // (hs, 3 hs): K-major copy of in_proj_weight
  const float* d_recon;           // (3, B, hs)
  const float* rec_part;          // optional (3, B, hs): d_recon[i] W_rec[i] already multiplied (see FusedBwdC); then rec_wT is not read
  const float* rec_wT;            // 3 x (hs, hs) K-major
  const float* x6;                // (6, B, hs) sigmoid outputs
  float* d_x6;                    // (6, B, hs)
  const float* priv_wT;           // 3 x (hs, hs) K-major
  const float* sh_wT;             // (hs, hs) K-major
  float* d_orig;                  // (3, B, hs)
  mmda_ln_bwd_args lnp[3];        // rows = B each
  float* pg_parts;                // (B, 5, 2, hs): slot 1: ln1, slots 2..4: the projection LayerNorms
  unsigned long long* dbg;        // diagnostics: cycle counter at each stage boundary of workgroup 0 (NULL in production)
  // optional flag join (common.h: flag_wait): d_x6 holds the gradients of the batch-statistic losses, written by another stream
  const unsigned* wait_flag; unsigned wait_value; unsigned* wait_err;
};

// forward stretch A: recon = (private + shared) W_rec^T + b and qkv = x6 W_in^T + b (both read x6 only) -> six-token attention ->
//                    attn_out = ctx W_out^T + b -> x1 = LayerNorm 1 (x6 + dropout(attn_out))
struct FusedFwdA {
  int B, hs, nhead, nb;
  const float* x6;                // (6, B, hs)
  const float* rec_w; const float* rec_b; float* recon;           // 3 x (hs, hs) row-major [out][in], 3 x hs, (3, B, hs)
  const float* in_w; const float* in_b; float* qkv;               // (3 hs, hs), 3 hs, (6 B, 3 hs)
  float* ctx; float* probs; float p_tf; uint64_t seed; int site_attn;
  const float* out_w; const float* out_b; float* attn_out;        // (hs, hs), hs, (6 B, hs)
  mmda_ln_args ln1;               // rows = 6 B: x = x6, res = attn_out
  // optional (training step): the reconstruction loss's gradient seeds, written where recon is produced instead of by the loss launch
  // behind the forward pass -- d_recon = 2 (recon - orig) recon_g, d_orig = -d_recon (stores: the buffers need not be cleared)
  const float* orig; float* d_recon; float* d_orig; float recon_inv_n, recon_scale;
  int split_recon;                // set by mmda_fused_fwd_a: the reconstruction runs in workgroups of its own (twice the grid)
};
// forward stretch C: hfused = LayerNorm 2 (x1 + dropout(f2)) permuted to (B, 6 hs) -> logits = hfused W_head^T + b -> heads
struct FusedFwdC {
  int B, hs, ncls, nb;
  mmda_ln_args ln2;               // rows = 6 B, permute_S / permute_B set, y = hfused
  const float* ffn_parts; int n_parts; const float* b2; float* f2;      // optional: f2 = sum of the partials (slice order) + b2, first
  const float* hfused; const float* head_w; const float* head_b; float* logits;      // (B, 6 hs), (6 + ncls, 6 hs), 6 + ncls, (B, 6 + ncls)
  float threshold; float* tcp; float* scores; float* labels; float p_cls; uint64_t seed; int site_cls;
  // optional (training step): the classification loss's gradient seed d_scores = (s - y) / max(s (1 - s), 1e-12) / B, stored here
  const float* emo; float* d_scores;
};
// The feed-forward pair of the transformer layer (linear1 hs -> F with relu + dropout, linear2 F -> hs) as ONE launch per direction,
// split over the HIDDEN units: workgroup (row block, slice j) takes S of the F hidden units of 192 rows through both products -- it needs
// S rows of W1 and S columns of W2 only, so the 2 MB of weights are spread over F / S workgroups instead of streaming through every one
// -- and leaves a partial (rows, hs) product of the second GEMM in parts[j]; the consumer (the fused stretch behind it) adds the F / S
// partials in slice order.
struct FusedFfnFwd {
  int M, hs, F, S;                // rows (6 B), 128, 2048, hidden units per workgroup
  const float* x1; const float* w1; const float* b1; float* f1;   // (M, hs), (F, hs), F, (M, F): f1 = dropout(relu(x1 W1^T + b1))
  float p; uint64_t seed; int site;                                // dropout on f1, element index m * F + n
  const float* w2;                // (hs, F)
  float* parts;                   // (F / S, M, hs): partial f2 (no bias)
};
struct FusedFfnBwd {
  int M, hs, F, S;
  const float* d_f2; const float* f1; float gate_scale;           // (M, hs), (M, F): d_f1 = (d_f2 W2) * [f1 > 0] * gate_scale
  const float* l2_wT;             // (F, hs): K-major copy of W2
  float* d_f1;                    // (M, F)
  const float* l1_wT;             // (hs, F): K-major copy of W1
  float* parts;                   // (F / S, M, hs): partial d_x1 = d_f1 W1
};
int mmda_fused_ffn_fwd(const FusedFfnFwd* a, void* stream);
int mmda_fused_ffn_bwd(const FusedFfnBwd* a, void* stream);
int mmda_fused_fwd_a(const FusedFwdA* a, void* stream);
int mmda_fused_fwd_c(const FusedFwdC* a, void* stream);
int mmda_fused_bwd_c(const FusedBwdC* a, void* stream);
int mmda_fused_bwd_a(const FusedBwdA* a, void* stream);
// The LayerNorm parameter gradients of the two backward stretches, deterministic: the stretches leave per-sample partial sums in
// pg_parts (B, 5, 2, hs) -- slots ln2, ln1, proj_t, proj_v, proj_a -- and this launch adds them, in sample order, into dgamma[k] /
// dbeta[k] (no float atomics: identical bits on every run).  Runs wherever the weight-gradient GEMMs of the block run (side stream).
constexpr int FUSED_PG_SLOTS = 5;
int mmda_fused_pg_finish(const float* pg_parts, int B, int hs, float* const* dgamma, float* const* dbeta, void* stream);
extern "C" int mmda_debug_set_fused_stamps(void* device_buffer);     // tools/ only (16 x u64)
