// Native runtime for one MISA training iteration (reference loop body solver.py:139-186 over models.py:163-285).
// Host-side C++ only: lays the parameters out in one flat bucket, carves the workspace, and issues the HIP kernels of
// gemm.hip / lstm.hip / norm.hip / attn.hip / losses.hip / optim.hip in dependency order on one stream.  No device
// memory is owned here; no torch types; no host<->device synchronisation anywhere in a step.
#include "common.h"
#include "fused_rows.h"
// internal entry points of norm.hip (not part of the C ABI)
bool mmda_ln_bwd_parts_applies(const mmda_ln_bwd_args* a, int n);
int64_t mmda_ln_parts_floats(const mmda_ln_bwd_args* a, int n);
int mmda_ln_bwd_parts(const mmda_ln_bwd_args* a, int n, float* parts, void* stream);
int mmda_ln_parts_finish(const mmda_ln_bwd_args* a, int n, float* parts, void* stream);
int mmda_embed_scatter_add_masked(float* dW, const int64_t* ids, int rows, int dim, const float* dX, const int* lengths, int B, void* stream);
bool mmda_embed_scatter_sorts(int rows);
// ... and of dist.hip: the sort-based scatter in two halves (the sorted id list early, the sums behind the gradient rows)
int mmda_embed_sort_ids(const int64_t* ids, int n, const int* lengths, int B, int table_rows, unsigned* sorted, void* stream);
int mmda_embed_scatter_presorted(float* dW, const unsigned* sorted, int n, int D, int table_rows, const float* rows, void* stream);

#include <map>
#include <string>
#include <vector>
#include <stdlib.h>

namespace {

struct ParamInfo { std::string name; int64_t off; int rows, cols; };

struct Rnn {           // one bidirectional LSTM layer
  int D, H;
  int64_t w_ih, w_hh[2], b_ih, b_hh;     // w_ih: (8H,D) = [fwd;rev]; b_*: (8H) = [fwd;rev]
  int64_t pack_f[2], pack_b[2], pack_c[2];   // workspace float offsets of the packed W_hh (per direction)
  // bf16 operand copies for the bf16-mode GEMMs (workspace float offsets; leading dimensions in bf16 elements):
  int64_t wb, wbT;                           // W_ih (8H, ldD) and W_ih^T (D, ldG)
  int64_t xb, xbT;                           // layer input (R, ldD) and its transpose (D, ldR)
  int64_t dgb, dgbT;                         // gate gradients (R, ldG) and transpose (8H, ldR)
  int64_t hbT;                               // hseq^T (2H, ldR)
  int64_t hbp[2]; int ldH;                   // tn weight-gradient GEMMs: hseq of one direction as bf16 (R, ldH), ldH = round_up(H, 8)
  int ldD, ldG;
  // GRU (cfg.rnncell): workspace offsets of the parameters / gradients in the four-slot layout (mmda_gru_pad_job); -1 for LSTM
  int64_t pw_ih = -1, pw_hh[2] = {-1, -1}, pb_ih = -1, pb_hh = -1, gw_ih = -1, gw_hh[2] = {-1, -1}, gb = -1;
};

struct Mod {           // one modality: two stacked biLSTMs with a LayerNorm between, then a projection
  int D, H;
  Rnn rnn[2];
  int64_t ln_w, ln_b;                    // {t,v,a}layer_norm
  int64_t pw, pb, plw, plb;              // project_*: Linear + LayerNorm
  // workspace
  int64_t x, gates[2], c[2], hseq[2], normed, ln_mean, ln_rstd, utt, d_utt, d_hseq1, d_normed, d_x, xchg, xchg_floats;
};

enum { SITE_ATTN = 1, SITE_DROP1 = 2, SITE_FFN = 3, SITE_DROP2 = 4, SITE_CLS = 5, SITE_DISC = 6,
       SITE_RRELU = 7 /* .. 9: the three projections' random slopes */, SITE_RRELU_DISC = 10 };
constexpr int FFN = 2048, NHEAD = 2, S6 = 6;

}  // namespace

struct mmda_misa {
  mmda_misa_config cfg;
  std::vector<ParamInfo> params;
  std::map<std::string, int> index;
  int64_t dense = 0, flat = 0;
  int64_t rnn2_begin = 0, rnn1_begin = 0;    // bucket offsets where the layer-2 / layer-1 recurrent parameters start
  hipEvent_t ev_early = nullptr;             // recorded by backward() when the gradients of the bucket prefix are final
  int64_t early_floats = 0; int early_valid = 0;
  Mod mod[3];
  // fusion parameter offsets
  int64_t priv_w, priv_b, sh_w, sh_b, rec_w, rec_b, d1_w = -1, d1_b = -1, d2_w = -1, d2_b = -1, sp_w, sp_b;
  int64_t head_w, head_b, embed;
  int64_t in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b;
  int64_t prelu_a = -1;                      // config.activation = prelu: the ONE learned slope (nn.PReLU() shared by every use, models.py:30)
  float *P = nullptr, *G = nullptr, *M1 = nullptr, *V1 = nullptr;
  // workspace
  float* ws = nullptr; int64_t ws_floats = 0; int B = 0, T = 0;
  std::map<std::string, int64_t> tens;
  int64_t zero_begin = 0, zero_end = 0;      // activation-gradient region that is zeroed per step
  int64_t zero_cls = 0, zero_recon = 0;      // ... its tail: d_scores from zero_cls, d_orig + d_recon from zero_recon (see loss seeds)
  int64_t gpad_begin = 0, gpad_end = 0;      // GRU: four-slot weight gradients (zeroed at set_workspace, re-zeroed by the unpad kernel)
  int64_t z, pmean, prstd, orig, x6, rsum, recon, dom_z, dom_h, dom, qkv, probs, ctx, attn_out, ln1_mean, ln1_rstd, x1, f1, f2,
      ln2_mean, ln2_rstd, hfused, logits, tcp, scores, labels, losses, diff_work, touched, ffn_parts, pg_parts, ln_parts;
  // K-major (transposed) fp32 copies of the fusion block's weights for its input-gradient GEMMs (made once per step)
  int64_t head_wT, l2_wT, l1_wT, out_wT, in_wT, rec_wT, priv_wT, sh_wT, d1_wT = -1, d2_wT = -1, pwT[3];
  int wT_valid = 0;
  int64_t d_scores, d_tcp, d_x6, d_orig, d_recon, d_dom, d_logits, d_hfused, d_x1, d_f2, d_f1, d_attn_out, d_ctx, d_qkv, d_z,
      d_dom_h, d_dom_z;
  // block-scaled fp8 operands of the feed-forward products (fusion_fp8): element bytes and scale bytes, as float offsets
  // fused train step without a gradient exchange: clamp+Adam of the bucket prefix whose gradients are final beside the layer-1 backward
  // recurrence runs there, on the side stream (set by mmda_misa_train_step around its backward pass)
  int adam_early_on = 0; float ae_lr = 0.f, ae_clip = 0.f; int ae_step = 0; int64_t adam_early_done = 0;
  int tn_wgrad = 0;                // bit l: layer l + 1's weight-gradient GEMMs read dG / inputs / hseq as they lie (tn form): no transposed copies
  int embed_early_done = 0;        // this step's early optimizer pass also covered the embedding rows the batch does not touch
  int wT_pending = 0;              // the K-major fusion-weight copies of this step are still to be made (on the next fork)
  int fusion_fp8 = 0;
  int64_t x1q, x1s, w1q, w1s, f1q, f1s, w2q, w2s;
  // state of the last forward (dropout replay in backward)
  int training = 0; uint64_t seed = 0;
  // optional per-launch timing of the four recurrent kernels (bench.py roofline leg)
  unsigned epoch = 1;              // monotonic cluster-exchange epoch (never reset; see lstm_cluster.hip)
  hipStream_t side = nullptr;      // second stream for weight-gradient GEMMs (created lazily; no device memory)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pack = nullptr;
  int pack_b_valid = 0;            // the streaming backward packing of W_hh was made by the last forward
  int side_pending = 0, use_side = 1;
  int use_cluster = 1, packed_c_valid = 0;
  int use_bf16_gemm = 1;           // bf16 mode: LSTM-sized GEMMs read bf16 operand copies (gemm_bf16.hip)
  int gate_minor = 0;              // layout of `gates` chosen by the last forward (see mmda_lstm_desc.gate_minor)
  int inference = 0, last_fwd_inference = 0;   // evaluation passes: no stash, no copies that only the backward pass reads
  int zero_grad_pending = 0;       // train_step: the gradient bucket is cleared inside forward(), beside the fusion block
  // train_step: the losses that read only the private/shared representations (diff, CMD) are issued by forward() on the side
  // stream as soon as those exist, beside the transformer layer and the heads; mmda_misa_losses() then adds the rest
  int eager_losses = 0, eager_done = 0;
  // Data-parallel "global statistics" mode (mmda_misa_set_external_batch_losses): the batch-statistic losses -- DiffLoss, CMD and the
  // confidence loss -- were computed by the caller on the batch of ALL ranks (all-gathered inputs, the same loss entry points), their
  // sums written into losses[1], [2], [4] and their gradient rows of THIS rank added into d_x6 / d_scores / d_tcp behind
  // mmda_misa_zero_act_grads: mmda_misa_losses then only adds what is a mean over samples (cls, recon) and the weighted total.
  int ext_batch_losses = 0;
  // Loss seeds (training step, fused row-local stretches): the forward stretches store the gradient seeds of the reconstruction loss
  // (d_recon, d_orig) and -- without ConfidNet, whose loss adds into the same buffer -- of the classification loss (d_scores) where
  // they produce recon / scores, so the launch that computes cls / conf / recon and the weighted total has no gradient to seed and
  // leaves the critical path between the forward and the backward pass (14 us at B=32): it runs on the side stream beside the
  // layer-2 backward recurrence.  Values are bit-identical to the loss launch's (same expressions on the same operands).
  const float* emo_eager = nullptr;          // labels of the step in flight (train_step)
  int seed_recon = 0, seed_cls = 0;          // this step's forward wrote those seeds
  const float* misc_deferred = nullptr;      // labels: the loss-value launch is still to be issued (backward's first side fork)
  // Flag joins (training step; common.h: flag_wait): where the main stream needs a side-stream chain's results, the consuming KERNEL
  // waits on the device for a word that a one-thread launch behind the chain sets, instead of the stream waiting for an event -- an
  // event wait costs the main stream 9 - 12 us of packet processing even when the chain finished long ago (tools/micro/fork_cost.hip),
  // twice per step.  Forward: the loss chain's gradients (d_x6) are first read by the LayerNorm-1 stretch of the backward pass
  // (fused_bwd_a_kernel).  Backward: nothing on the main stream reads what the side stream's weight-gradient GEMMs and early
  // optimizer pass write; the step's last optimizer launch simply does not complete before they have.  ev_join is still recorded
  // behind each chain for the paths that cannot wait on the device.  MMDA_FLAG_JOIN=0: event joins as before.
  unsigned* jflags = nullptr;                // device: [0] forward chain done, [1] backward chain done, [2] a wait timed out
  unsigned jval[2] = {0u, 0u};
  // The sort-based embedding scatter of a large batch in two halves: the sorted (id, position) list depends on the ids only and is
  // made on the side stream beside the layer-2 backward recurrence (33 us of launches at B=256 that used to sit between the last
  // GEMM and the optimizer); word [3] tells the main stream it is there (a one-wave wait launch in front of the sums).
  int64_t esort = -1; int esort_valid = 0; unsigned esort_val = 0u;
  int64_t rec_part = -1;
  int flag_join_ok = 0, fj1 = 0, fj2 = 0, fj1_armed = 0;
  int ldR = 0;
  // cluster-exchange regions: at the front of the workspace, sized by B alone, so a change of T (every batch under the reference's
  // collate) neither moves nor clears them -- flags are monotonic epochs.  Cleared (on the caller's stream) only when the buffer
  // or B changes; the abort words found there before a clear are kept in `abort_sticky`.
  float* xchg_ws = nullptr; int xchg_B = 0; int abort_sticky = 0;
  std::vector<hipEvent_t> ev;      // [step][slot][start/stop]
  int ev_steps = 0, ev_fwd = 0, ev_bwd = 0;
  int ev_stride = 1, ev_seen_f = 0, ev_seen_b = 0;   // record every ev_stride-th step (the event pairs cost ~35 us per step)
  int ev_rotate = 0;                                 // 1: a sampled step brackets ONE of the four recurrent launches (sample index % 4): ~9 us
  std::vector<char> ev_done;                         // (step, slot) was recorded
};

namespace {

int64_t add_param(mmda_misa* m, const std::string& name, int rows, int cols) {
  ParamInfo p{name, m->flat, rows, cols};
  m->index[name] = (int)m->params.size();
  m->params.push_back(p);
  m->flat += (int64_t)rows * (cols > 0 ? cols : 1);
  return p.off;
}

void build_params(mmda_misa* m) {
  const mmda_misa_config& c = m->cfg;
  const int dims[3] = {c.d_t, c.d_v, c.d_a};
  const char* mn[3] = {"t", "v", "a"};
  const int hs = c.hidden;
  // Order of the dense bucket = the order in which the backward pass completes the gradients (data-parallel ranks start
  // reducing a prefix while the rest is still being computed, mmda_misa_early_grad_floats): fusion block and LayerNorms first,
  // then the layer-2 recurrent layers, then layer 1; the embedding matrix closes the bucket.
  for (int i = 0; i < 3; ++i) { Mod& md = m->mod[i]; md.D = md.H = dims[i]; }
  for (int i = 0; i < 3; ++i) {
    Mod& md = m->mod[i];
    std::string p = std::string("project_") + mn[i] + ".project_" + mn[i];
    md.pw = add_param(m, p + ".weight", hs, 4 * md.H);
    md.pb = add_param(m, p + ".bias", hs, 0);
    md.plw = add_param(m, p + "_layer_norm.weight", hs, 0);
    md.plb = add_param(m, p + "_layer_norm.bias", hs, 0);
  }
  // batched groups: three (hs,hs) weights back to back, then their three biases (uniform strides for batched GEMMs)
  m->priv_w = add_param(m, "private_t.private_t_1.weight", hs, hs);
  add_param(m, "private_v.private_v_1.weight", hs, hs);
  add_param(m, "private_a.private_a_3.weight", hs, hs);          // sic: reference models.py:95
  m->priv_b = add_param(m, "private_t.private_t_1.bias", hs, 0);
  add_param(m, "private_v.private_v_1.bias", hs, 0);
  add_param(m, "private_a.private_a_3.bias", hs, 0);
  m->sh_w = add_param(m, "shared.shared_1.weight", hs, hs);
  m->sh_b = add_param(m, "shared.shared_1.bias", hs, 0);
  m->rec_w = add_param(m, "recon_t.recon_t_1.weight", hs, hs);
  add_param(m, "recon_v.recon_v_1.weight", hs, hs);
  add_param(m, "recon_a.recon_a_1.weight", hs, hs);
  m->rec_b = add_param(m, "recon_t.recon_t_1.bias", hs, 0);
  add_param(m, "recon_v.recon_v_1.bias", hs, 0);
  add_param(m, "recon_a.recon_a_1.bias", hs, 0);
  if (!c.use_cmd_sim) {
    m->d1_w = add_param(m, "discriminator.discriminator_layer_1.weight", hs, hs);
    m->d1_b = add_param(m, "discriminator.discriminator_layer_1.bias", hs, 0);
    m->d2_w = add_param(m, "discriminator.discriminator_layer_2.weight", 3, hs);
    m->d2_b = add_param(m, "discriminator.discriminator_layer_2.bias", 3, 0);
  }
  m->sp_w = add_param(m, "sp_discriminator.sp_discriminator_layer_1.weight", 4, hs);
  m->sp_b = add_param(m, "sp_discriminator.sp_discriminator_layer_1.bias", 4, 0);
  // [confidence; classifier] adjacent -> one (6+ncls, 6hs) head GEMM
  m->head_w = add_param(m, "confidence.confidence_layer_1.weight", 6, 6 * hs);
  add_param(m, "classifier.classifier_layer.weight", c.ncls, 6 * hs);
  m->head_b = add_param(m, "confidence.confidence_layer_1.bias", 6, 0);
  add_param(m, "classifier.classifier_layer.bias", c.ncls, 0);
  for (int i = 0; i < 3; ++i) {
    m->mod[i].ln_w = add_param(m, std::string(mn[i]) + "layer_norm.weight", 2 * m->mod[i].H, 0);
    m->mod[i].ln_b = add_param(m, std::string(mn[i]) + "layer_norm.bias", 2 * m->mod[i].H, 0);
  }
  const std::string te = "transformer_encoder.layers.0.";
  m->in_w = add_param(m, te + "self_attn.in_proj_weight", 3 * hs, hs);
  m->in_b = add_param(m, te + "self_attn.in_proj_bias", 3 * hs, 0);
  m->out_w = add_param(m, te + "self_attn.out_proj.weight", hs, hs);
  m->out_b = add_param(m, te + "self_attn.out_proj.bias", hs, 0);
  m->l1_w = add_param(m, te + "linear1.weight", FFN, hs);
  m->l1_b = add_param(m, te + "linear1.bias", FFN, 0);
  m->l2_w = add_param(m, te + "linear2.weight", hs, FFN);
  m->l2_b = add_param(m, te + "linear2.bias", hs, 0);
  m->n1_w = add_param(m, te + "norm1.weight", hs, 0);
  m->n1_b = add_param(m, te + "norm1.bias", hs, 0);
  m->n2_w = add_param(m, te + "norm2.weight", hs, 0);
  m->n2_b = add_param(m, te + "norm2.bias", hs, 0);
  if (c.act == MMDA_ACT_PRELU) m->prelu_a = add_param(m, "activation.weight", 1, 0);      // last of the block: what follows is re-aligned
  for (int l = 1; l >= 0; --l) {
    m->flat = (m->flat + 3) & ~(int64_t)3;
    (l == 1 ? m->rnn2_begin : m->rnn1_begin) = m->flat;
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i];
      Rnn& r = md.rnn[l];
      r.H = md.H; r.D = l == 0 ? md.D : 2 * md.H;
      const int ng = c.rnncell == MMDA_CELL_GRU ? 3 : 4;      // gate blocks per direction in the torch-layout parameters
      std::string pre = std::string(mn[i]) + "rnn" + (l == 0 ? "1" : "2") + ".";
      r.w_ih = add_param(m, pre + "weight_ih_l0", ng * r.H, r.D);
      add_param(m, pre + "weight_ih_l0_reverse", ng * r.H, r.D);
      r.w_hh[0] = add_param(m, pre + "weight_hh_l0", ng * r.H, r.H);
      r.w_hh[1] = add_param(m, pre + "weight_hh_l0_reverse", ng * r.H, r.H);
      r.b_ih = add_param(m, pre + "bias_ih_l0", ng * r.H, 0);
      add_param(m, pre + "bias_ih_l0_reverse", ng * r.H, 0);
      r.b_hh = add_param(m, pre + "bias_hh_l0", ng * r.H, 0);
      add_param(m, pre + "bias_hh_l0_reverse", ng * r.H, 0);
    }
  }
  m->flat = (m->flat + 3) & ~(int64_t)3;
  m->dense = m->flat;
  m->embed = add_param(m, "embed.weight", c.vocab, c.d_t);
  m->flat = (m->flat + 3) & ~(int64_t)3;
}

struct Carver {
  int64_t cur = 0;
  int64_t take(int64_t n) { int64_t o = cur; cur += (n + 3) & ~(int64_t)3; return o; }   // 16-B aligned
};

// lays out the workspace for (B,T); returns total floats.  With m == nullptr-like dry run when commit == false.
int64_t layout(mmda_misa* m, int B, int T, bool commit) {
  const mmda_misa_config& c = m->cfg;
  const int hs = c.hidden, NC = 6 + c.ncls;
  const int64_t R = (int64_t)T * B;
  Carver k;
  mmda_misa tmp_store;               // only used to keep the code path identical in dry runs
  mmda_misa* o = commit ? m : &tmp_store;
  if (!commit) { o->cfg = m->cfg; for (int i = 0; i < 3; ++i) o->mod[i] = m->mod[i]; }
  for (int i = 0; i < 3; ++i) {       // exchange buffers of the recurrences first: their offsets depend on B only
    Mod& md = o->mod[i];
    md.xchg_floats = (mmda_lstm_xchg_bytes(md.H, B) + 3) / 4;
    md.xchg = md.xchg_floats > 0 ? k.take(md.xchg_floats) : -1;
  }
  for (int i = 0; i < 3; ++i) {
    Mod& md = o->mod[i];
    for (int l = 0; l < 2; ++l) {
      Rnn& r = md.rnn[l];
      for (int d = 0; d < 2; ++d) {     // sized for the larger (fp32) packing so the mode can be switched in place
        r.pack_f[d] = k.take(mmda_lstm_packed_bytes(MMDA_F32, r.H, 0) / 4);
        r.pack_b[d] = k.take(mmda_lstm_packed_bytes(MMDA_F32, r.H, 1) / 4);
        r.pack_c[d] = k.take(mmda_lstm_packed_bytes(MMDA_BF16, r.H, 2) / 4);
      }
    }
    for (int l = 0; l < 2; ++l) {
      Rnn& r = md.rnn[l];
      const int ldR = round_up((int)R, 8);
      r.ldD = round_up(r.D, 8); r.ldG = round_up(8 * r.H, 8);
      r.wb = k.take((int64_t)8 * r.H * r.ldD / 2); r.wbT = k.take((int64_t)r.D * r.ldG / 2);
      r.xb = k.take(R * r.ldD / 2); r.xbT = k.take((int64_t)r.D * ldR / 2);
      r.dgb = k.take(R * r.ldG / 2); r.dgbT = k.take((int64_t)8 * r.H * ldR / 2);
      r.hbT = k.take((int64_t)2 * r.H * ldR / 2);
      r.ldH = round_up(r.H, 8);
      for (int d = 0; d < 2; ++d) r.hbp[d] = k.take(R * r.ldH / 2);
    }
    if (c.rnncell == MMDA_CELL_GRU) {
      for (int l = 0; l < 2; ++l) {
        Rnn& r = md.rnn[l];
        r.pw_ih = k.take((int64_t)8 * r.H * r.D); r.pw_hh[0] = k.take((int64_t)4 * r.H * r.H); r.pw_hh[1] = k.take((int64_t)4 * r.H * r.H);
        r.pb_ih = k.take(8 * r.H); r.pb_hh = k.take(8 * r.H);
      }
    }
    md.x = (i == 0) ? k.take(R * md.D) : -1;
    for (int l = 0; l < 2; ++l) {
      md.gates[l] = k.take(R * 8 * md.H);
      md.c[l] = k.take((int64_t)T * round_up(B, 4) * 2 * md.H);     // batch-minor-by-4 under the gate-minor layout: rows rounded up
      md.hseq[l] = k.take(R * 2 * md.H);
    }
    md.normed = k.take(R * 2 * md.H);
    md.ln_mean = k.take(R);
    md.ln_rstd = k.take(R);
    md.utt = k.take((int64_t)B * 4 * md.H);
    md.d_utt = k.take((int64_t)B * 4 * md.H);
    md.d_hseq1 = k.take(R * 2 * md.H);
    md.d_normed = k.take(R * 2 * md.H);
    md.d_x = (i == 0) ? k.take(R * md.D) : -1;
  }
  const int64_t BH = (int64_t)B * hs;
  o->z = k.take(3 * BH); o->pmean = k.take(3 * B); o->prstd = k.take(3 * B);
  // public outputs (what the reference's solver reads off the module) are contiguous so the host can snapshot them at once
  const int64_t pub_begin = k.cur;
  o->orig = k.take(3 * BH); o->x6 = k.take(6 * BH); o->recon = k.take(3 * BH); o->dom = k.take((int64_t)3 * B * 3);
  o->tcp = k.take((int64_t)B * 6); o->scores = k.take((int64_t)B * c.ncls); o->labels = k.take((int64_t)B * c.ncls);
  const int64_t pub_end = k.cur;
  o->rsum = k.take(3 * BH); o->dom_z = k.take(3 * BH); o->dom_h = k.take(3 * BH);
  o->qkv = k.take(6 * BH * 3); o->probs = k.take((int64_t)B * NHEAD * S6 * S6); o->ctx = k.take(6 * BH);
  o->attn_out = k.take(6 * BH); o->ln1_mean = k.take(6 * B); o->ln1_rstd = k.take(6 * B); o->x1 = k.take(6 * BH);
  o->f1 = k.take((int64_t)6 * B * FFN); o->f2 = k.take(6 * BH); o->ln2_mean = k.take(6 * B); o->ln2_rstd = k.take(6 * B);
  o->hfused = k.take(6 * BH); o->logits = k.take((int64_t)B * NC);
  o->x1q = k.take(6 * BH / 4); o->x1s = k.take(6 * BH / 128 + 4); o->w1q = k.take((int64_t)FFN * hs / 4); o->w1s = k.take((int64_t)FFN * hs / 128 + 4);
  o->f1q = k.take((int64_t)6 * B * FFN / 4); o->f1s = k.take((int64_t)6 * B * FFN / 128 + 4);
  o->w2q = k.take((int64_t)hs * FFN / 4); o->w2s = k.take((int64_t)hs * FFN / 128 + 4);
  o->diff_work = k.take(mmda_loss_diff_work_floats(B, hs));
  o->ffn_parts = k.take((int64_t)(FFN / 32) * 6 * BH);      // partial products of the hidden-sliced feed-forward kernels (fused_rows.hip)
  o->ln_parts = k.take((int64_t)512 * 2 * 2 * (o->mod[0].H + o->mod[1].H + o->mod[2].H));   // <= 512 block partials of the three inter-layer LayerNorms' gamma / beta gradients (norm.hip)
  o->rec_part = k.take(3 * BH);                             // d_recon W_rec, made beside the backward pass's first stretch for its third (fused_rows.h)
  o->esort = k.take(2 * (int64_t)B * T + 64);              // sorted (id, position) list of the step's text ids (dist.hip)
  o->pg_parts = k.take((int64_t)B * FUSED_PG_SLOTS * 2 * 128);      // per-sample LayerNorm gamma / beta gradient partials of the fused backward stretches
  o->touched = k.take((c.vocab + 3) / 4);              // one byte per embedding row: occurs in this batch (see mmda_clamp_adam_rows)
  o->head_wT = k.take((int64_t)6 * hs * NC); o->l2_wT = k.take((int64_t)FFN * hs); o->l1_wT = k.take((int64_t)hs * FFN);
  o->out_wT = k.take((int64_t)hs * hs); o->in_wT = k.take((int64_t)hs * 3 * hs); o->rec_wT = k.take((int64_t)3 * hs * hs);
  o->priv_wT = k.take((int64_t)3 * hs * hs); o->sh_wT = k.take((int64_t)hs * hs);
  if (!c.use_cmd_sim) { o->d1_wT = k.take((int64_t)hs * hs); o->d2_wT = k.take((int64_t)hs * 3); }
  for (int i = 0; i < 3; ++i) o->pwT[i] = k.take((int64_t)4 * o->mod[i].H * hs);
  o->gpad_begin = k.cur;
  if (c.rnncell == MMDA_CELL_GRU) {
    for (int i = 0; i < 3; ++i)
      for (int l = 0; l < 2; ++l) {
        Rnn& r = o->mod[i].rnn[l];
        r.gw_ih = k.take((int64_t)8 * r.H * r.D); r.gw_hh[0] = k.take((int64_t)4 * r.H * r.H); r.gw_hh[1] = k.take((int64_t)4 * r.H * r.H);
        r.gb = k.take(8 * r.H);
      }
  }
  o->gpad_end = k.cur;
  // ---- the loss sums and the activation gradients seeded by the losses (zeroed every step by ONE memset, contiguous)
  o->zero_begin = k.cur;
  o->losses = k.take(8);
  o->d_tcp = k.take((int64_t)B * 6); o->d_x6 = k.take(6 * BH); o->d_dom = k.take((int64_t)3 * B * 3);
  o->zero_cls = k.cur;                       // (a step whose forward stores these seeds itself clears up to here only)
  o->d_scores = k.take((int64_t)B * c.ncls);
  o->zero_recon = k.cur;
  o->d_orig = k.take(3 * BH); o->d_recon = k.take(3 * BH);
  o->zero_end = k.cur;
  // ---- fully overwritten gradients
  o->d_logits = k.take((int64_t)B * NC); o->d_hfused = k.take(6 * BH); o->d_x1 = k.take(6 * BH); o->d_f2 = k.take(6 * BH);
  o->d_f1 = k.take((int64_t)6 * B * FFN); o->d_attn_out = k.take(6 * BH); o->d_ctx = k.take(6 * BH);
  o->d_qkv = k.take(6 * BH * 3); o->d_z = k.take(3 * BH); o->d_dom_h = k.take(3 * BH); o->d_dom_z = k.take(3 * BH);
  if (commit) {
    std::map<std::string, int64_t>& t = m->tens;
    t.clear();
    t["scores"] = m->scores; t["labels"] = m->labels; t["tcp"] = m->tcp; t["logits"] = m->logits; t["hfused"] = m->hfused;
    t["x6"] = m->x6; t["orig"] = m->orig; t["recon"] = m->recon; t["dom"] = m->dom; t["losses"] = m->losses;
    t["utt_t"] = m->mod[0].utt; t["utt_v"] = m->mod[1].utt; t["utt_a"] = m->mod[2].utt;
    t["d_scores"] = m->d_scores; t["d_tcp"] = m->d_tcp; t["d_x6"] = m->d_x6; t["d_orig"] = m->d_orig;
    t["d_recon"] = m->d_recon; t["d_dom"] = m->d_dom;
    t["pub_begin"] = pub_begin; t["pub_end"] = pub_end; t["zero_begin"] = m->zero_begin; t["zero_end"] = m->zero_end;
    t["hseq1_t"] = m->mod[0].hseq[0]; t["hseq1_v"] = m->mod[1].hseq[0]; t["hseq1_a"] = m->mod[2].hseq[0];
    t["d_x_t"] = m->mod[0].d_x;      // gradient w.r.t. the gathered embedding rows (T*B, d_t): the sparse form of embed.weight.grad
    t["xchg_t"] = m->mod[0].xchg; t["xchg_v"] = m->mod[1].xchg; t["xchg_a"] = m->mod[2].xchg;   // word 0 of each = its abort word
  }
  return k.cur;
}

// ---------------------------------------------------------------------------------------------- GEMM shorthands
struct Ctx {
  mmda_misa* m; void* s; int rc = 0;
  bool grouping = false; std::vector<mmda_gemm_args> pending;
  bool deferring = false; std::vector<mmda_gemm_args> deferred;   // weight-gradient GEMMs: off the critical path, run on the side stream
};

// independent GEMMs issued between group_begin/group_end go out as ONE grouped launch
void group_begin(Ctx& c) { c.grouping = true; c.pending.clear(); }
void group_end(Ctx& c) {
  c.grouping = false;
  if (!c.rc && !c.pending.empty()) c.rc = mmda_gemm_grouped(c.pending.data(), (int)c.pending.size(), c.s);
  c.pending.clear();
}

void gemm(Ctx& c, int mode, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* Bp, int ldb, float* C,
          int ldc, const float* bias = nullptr, const float* bias2 = nullptr, int acc = 0, int act = 0, int batch = 1,
          int64_t sA = 0, int64_t sB = 0, int64_t sC = 0, int64_t sBias = 0, mmda_gemm_args* extra = nullptr) {
  if (c.rc) return;
  mmda_gemm_args g = {};
  if (extra) g = *extra;
  g.mode = mode; g.transA = tA; g.transB = tB; g.M = M; g.N = N; g.K = K; g.batch = batch;
  g.A = A; g.lda = lda; g.strideA = sA; g.B = Bp; g.ldb = ldb; g.strideB = sB; g.C = C; g.ldc = ldc; g.strideC = sC;
  g.bias = bias; g.bias2 = bias2; g.strideBias = sBias; g.accumulate = acc; g.act = act;
  if (c.deferring && tA && acc) c.deferred.push_back(g);       // TN + accumulate == a weight gradient
  else if (c.grouping) c.pending.push_back(g);
  else c.rc = mmda_gemm(&g, c.s);
}

// Fork/join with the side stream: everything in `list` only has to be finished before the optimizer step, so it runs
// concurrently with the recurrent kernels (which occupy ~10 % of the CUs while they walk the serial chain).
int side_fork(mmda_misa* m, void* main_stream, void** out);
int side_launch(mmda_misa* m, std::vector<mmda_gemm_args>& list, void* main_stream);
int side_join(mmda_misa* m, void* main_stream);
// y(M,N) = x(M,K) W(N,K)^T + b
void lin_fwd(Ctx& c, int mode, int M, int N, int K, const float* x, const float* W, const float* b, float* y, int act = 0) {
  gemm(c, mode, 0, 1, M, N, K, x, K, W, K, y, N, b, nullptr, 0, act);
}
// dx(M,K) (+)= dy(M,N) W(N,K)
void lin_dx(Ctx& c, int mode, int M, int N, int K, const float* dy, const float* W, float* dx, int acc) {
  gemm(c, mode, 0, 0, M, K, N, dy, N, W, K, dx, K, nullptr, nullptr, acc);
}
// dW(N,K) += dy(M,N)^T x(M,K);  db(N) += colsum(dy)
void lin_dw(Ctx& c, int mode, int M, int N, int K, const float* dy, const float* x, float* dW, float* db) {
  mmda_gemm_args e = {};
  e.bias_grad = db;                 // the bias gradient rides along as a virtual ones-column of the same GEMM
  gemm(c, mode, 1, 0, N, K, M, dy, N, x, K, dW, K, nullptr, nullptr, 1, 0, 1, 0, 0, 0, 0, &e);
}

// ---- row-skinny forms (fusion block at B <= SKINNY_MAX_B): see gemm_skinny.hip
static const int SKINNY_MAX_B = getenv("MMDA_SKINNY_MAX_B") ? atoi(getenv("MMDA_SKINNY_MAX_B")) : 256;   // (env: tile-shape experiments)
// y(M,N) = act(x(M,K) W(N,K)^T + b)
mmda_skinny_args sk_nt(int M, int N, int K, const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int act = 0) {
  mmda_skinny_args g = {};
  g.M = M; g.N = N; g.K = K; g.transB = 1; g.A = x; g.lda = ldx; g.B = W; g.ldb = K; g.C = y; g.ldc = ldy; g.bias = b; g.act = act;
  return g;
}
// dx(M,K) (+)= dy(M,N) W(N,K)
mmda_skinny_args sk_nn(int M, int N, int K, const float* dy, int lddy, const float* W, float* dx, int lddx, int acc) {
  mmda_skinny_args g = {};
  g.M = M; g.N = K; g.K = N; g.transB = 0; g.A = dy; g.lda = lddy; g.B = W; g.ldb = K; g.C = dx; g.ldc = lddx; g.accumulate = acc;
  return g;
}
// dx(M,K) (+)= dy(M,N) W(N,K) through the K-major copy WT(K,N): an NT problem (float4 loads along the reduction for both operands)
mmda_skinny_args sk_dx(int M, int N, int K, const float* dy, int lddy, const float* WT, float* dx, int lddx, int acc) {
  mmda_skinny_args g = {};
  g.M = M; g.N = K; g.K = N; g.transB = 1; g.A = dy; g.lda = lddy; g.B = WT; g.ldb = N; g.C = dx; g.ldc = lddx; g.accumulate = acc;
  return g;
}
void sk_launch(Ctx& c, const mmda_skinny_args* p, int n) {
  if (!c.rc) c.rc = mmda_gemm_skinny(p, n, c.s);
}


void ev_rec(mmda_misa* m, int step, int slot, int which, void* stream) {
  if (m->ev.empty() || step >= m->ev_steps) return;
  if ((slot < 2 ? m->ev_seen_f : m->ev_seen_b) % m->ev_stride) return;
  if (m->ev_rotate && slot != (step & 3)) return;
  (void)hipEventRecord(m->ev[(step * 4 + slot) * 2 + which], (hipStream_t)stream);
  if (which == 1 && (size_t)(step * 4 + slot) < m->ev_done.size()) m->ev_done[step * 4 + slot] = 1;
}

// Fork: work issued on the returned stream starts after everything already on `main_stream` and runs beside what follows
// there; side_join() makes `main_stream` wait for it.  Without overlap the main stream itself is returned.
int side_fork(mmda_misa* m, void* main_stream, void** out) {
  *out = main_stream;
  static const int no_side = getenv("MMDA_NO_SIDE") ? atoi(getenv("MMDA_NO_SIDE")) : 0;      // diagnostics: everything on one stream
  if (!m->use_side || no_side) return MMDA_OK;
  if (!m->side) {
    if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) return MMDA_ELAUNCH;
    // The fork / join events order streams of ONE device: no system-scope fence with them (hipEventDisableSystemFence -- "device
    // memory may not be visible to the host and other devices", neither of which waits on these events; every kernel still ends with
    // its own device-scope release).  With the fence a recorded event costs the recording stream 6 us between two launches, without
    // it 3 (tools/micro/fork_cost.hip).  MMDA_EVENT_SYSFENCE=1: with the fence.
    static const int sysfence = getenv("MMDA_EVENT_SYSFENCE") ? atoi(getenv("MMDA_EVENT_SYSFENCE")) : 0;
    const unsigned evf = hipEventDisableTiming | (sysfence ? 0u : hipEventDisableSystemFence);
    if (hipEventCreateWithFlags(&m->ev_fork, evf) != hipSuccess) return MMDA_ELAUNCH;
    if (hipEventCreateWithFlags(&m->ev_join, evf) != hipSuccess) return MMDA_ELAUNCH;
    if (hipEventCreateWithFlags(&m->ev_pack, evf) != hipSuccess) return MMDA_ELAUNCH;
    if (hipMalloc(reinterpret_cast<void**>(&m->jflags), 64) != hipSuccess) { m->jflags = nullptr; return MMDA_ELAUNCH; }
    if (hipMemset(m->jflags, 0, 64) != hipSuccess) return MMDA_ELAUNCH;
  }
  if (hipEventRecord(m->ev_fork, (hipStream_t)main_stream) != hipSuccess) return MMDA_ELAUNCH;
  if (hipStreamWaitEvent(m->side, m->ev_fork, 0) != hipSuccess) return MMDA_ELAUNCH;
  m->side_pending = 1;
  *out = m->side;
  return MMDA_OK;
}
int side_launch(mmda_misa* m, std::vector<mmda_gemm_args>& list, void* main_stream) {
  if (list.empty()) return MMDA_OK;
  void* ss = nullptr;
  int rc = side_fork(m, main_stream, &ss);
  if (!rc) rc = mmda_gemm_grouped(list.data(), (int)list.size(), ss);
  list.clear();
  return rc;
}
__global__ void flag_wait_kernel(const unsigned* flag, unsigned value, unsigned* err) { flag_wait(flag, value, err); }
__global__ void flag_set_kernel(unsigned* flag, unsigned value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// the side stream's chain ends here for the main stream: word `idx` is set behind it (flag join) -- or was, by the chain's last launch
// itself (by_kernel) --; ev_join is recorded too
int side_flag_signal(mmda_misa* m, int idx, bool by_kernel = false) {
  if (!m->side || !m->jflags) return MMDA_EINVAL;
  ++m->jval[idx];
  if (!by_kernel) {
    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, m->side, m->jflags + idx, m->jval[idx]);
    MMDA_CHECK_LAUNCH("side_flag_signal");
  }
  if (hipEventRecord(m->ev_join, m->side) != hipSuccess) return MMDA_ELAUNCH;
  m->side_pending = 0;
  return MMDA_OK;
}
// a flag join whose consumer kernel will not run after all: the event recorded with it
int flag_join_fallback(mmda_misa* m, void* main_stream) {
  if (hipStreamWaitEvent((hipStream_t)main_stream, m->ev_join, 0) != hipSuccess) return MMDA_ELAUNCH;
  return MMDA_OK;
}
int side_join(mmda_misa* m, void* main_stream) {
  if (!m->side_pending) return MMDA_OK;
  if (hipEventRecord(m->ev_join, m->side) != hipSuccess) return MMDA_ELAUNCH;
  if (hipStreamWaitEvent((hipStream_t)main_stream, m->ev_join, 0) != hipSuccess) return MMDA_ELAUNCH;
  m->side_pending = 0;
  return MMDA_OK;
}

#define WS(off) (m->ws + (off))
#define PP(off) (m->P + (off))
#define GG(off) (m->G + (off))

// Recurrent-layer parameters / gradients as the kernels see them: the bound flat buffers (LSTM) or the four-slot workspace
// copies (GRU; filled by mmda_gru_pad_params at the top of forward, folded back by mmda_gru_unpad_grads at the end of backward).
inline bool is_gru(const mmda_misa* m) { return m->cfg.rnncell == MMDA_CELL_GRU; }
inline float* rW_ih(mmda_misa* m, const Rnn& r) { return is_gru(m) ? WS(r.pw_ih) : PP(r.w_ih); }
inline float* rW_hh(mmda_misa* m, const Rnn& r, int d) { return is_gru(m) ? WS(r.pw_hh[d]) : PP(r.w_hh[d]); }
inline float* rB_ih(mmda_misa* m, const Rnn& r) { return is_gru(m) ? WS(r.pb_ih) : PP(r.b_ih); }
inline float* rB_hh(mmda_misa* m, const Rnn& r) { return is_gru(m) ? WS(r.pb_hh) : PP(r.b_hh); }
inline float* gW_ih(mmda_misa* m, const Rnn& r) { return is_gru(m) ? WS(r.gw_ih) : GG(r.w_ih); }
inline float* gW_hh(mmda_misa* m, const Rnn& r, int d) { return is_gru(m) ? WS(r.gw_hh[d]) : GG(r.w_hh[d]); }
inline float* gB_ih(mmda_misa* m, const Rnn& r) { return is_gru(m) ? WS(r.gb) : GG(r.b_ih); }
inline float* gB_hh(mmda_misa* m, const Rnn& r) { return is_gru(m) ? nullptr : GG(r.b_hh); }

// GRU: jobs for the pad / unpad kernels; `base` is the flat parameter (pad) or gradient (unpad) buffer
int gru_jobs(mmda_misa* m, float* base, bool grads, mmda_gru_pad_job* j) {
  int n = 0;
  for (int i = 0; i < 3; ++i)
    for (int l = 0; l < 2; ++l, ++n) {
      const Rnn& r = m->mod[i].rnn[l];
      j[n] = mmda_gru_pad_job{};
      j[n].H = r.H; j[n].D = r.D;
      for (int d = 0; d < 2; ++d) {
        j[n].w_ih[d] = base + r.w_ih + (int64_t)d * 3 * r.H * r.D; j[n].w_hh[d] = base + r.w_hh[d];
        j[n].b_ih[d] = base + r.b_ih + (int64_t)d * 3 * r.H; j[n].b_hh[d] = base + r.b_hh + (int64_t)d * 3 * r.H;
        j[n].pw_hh[d] = grads ? WS(r.gw_hh[d]) : WS(r.pw_hh[d]);
      }
      j[n].pw_ih = grads ? WS(r.gw_ih) : WS(r.pw_ih);
      j[n].pb_ih = grads ? WS(r.gb) : WS(r.pb_ih);
      j[n].pb_hh = grads ? nullptr : WS(r.pb_hh);
    }
  return n;
}

// parameters of a parametrised activation (prelu / rrelu) at one of its uses; zeros for every other activation
mmda_act_params act_params(mmda_misa* m, int training, uint64_t seed, int site, bool grads) {
  mmda_act_params p = {};
  if (m->cfg.act == MMDA_ACT_PRELU) { p.slope = m->P + m->prelu_a; p.dslope = grads ? m->G + m->prelu_a : nullptr; }
  if (m->cfg.act == MMDA_ACT_RRELU) { p.lo = 1.0f / 8.0f; p.hi = 1.0f / 3.0f; p.rand = training ? 1 : 0; p.seed = seed; p.site = site; }   // torch defaults
  return p;
}

int check_ready(const mmda_misa* m) {
  if (!m || !m->P || !m->ws) return MMDA_EINVAL;
  return MMDA_OK;
}

}  // namespace

// =============================================================================================== lifecycle
extern "C" int mmda_misa_create(const mmda_misa_config* cfg, mmda_misa** out) {
  if (!cfg || !out) return MMDA_EINVAL;
  if (cfg->vocab <= 0 || cfg->d_t <= 0 || cfg->d_v <= 0 || cfg->d_a <= 0 || cfg->hidden <= 0 || cfg->ncls <= 0) return MMDA_EINVAL;
  if (cfg->d_t > 512 || cfg->d_v > 512 || cfg->d_a > 512 || cfg->hidden % NHEAD || cfg->hidden > 1024) return MMDA_EINVAL;
  if (cfg->mode != MMDA_F32 && cfg->mode != MMDA_BF16) return MMDA_EINVAL;
  if (cfg->rnncell != MMDA_CELL_LSTM && cfg->rnncell != MMDA_CELL_GRU) return MMDA_EINVAL;
  mmda_misa* m = new mmda_misa();
  m->cfg = *cfg;
  build_params(m);
  *out = m;
  return MMDA_OK;
}
extern "C" void mmda_misa_destroy(mmda_misa* m) {
  if (!m) return;
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  if (m->ev_join) (void)hipEventDestroy(m->ev_join);
  if (m->ev_pack) (void)hipEventDestroy(m->ev_pack);
  if (m->ev_early) (void)hipEventDestroy(m->ev_early);
  if (m->side) (void)hipStreamDestroy(m->side);
  if (m->jflags) (void)hipFree(m->jflags);
  for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
  delete m;
}
extern "C" int mmda_misa_num_params(const mmda_misa* m) { return m ? (int)m->params.size() : MMDA_EINVAL; }
extern "C" int mmda_misa_param_info(const mmda_misa* m, int i, const char** name, int64_t* offset, int* rows, int* cols) {
  if (!m || i < 0 || i >= (int)m->params.size()) return MMDA_EINVAL;
  const ParamInfo& p = m->params[i];
  if (name) *name = p.name.c_str();
  if (offset) *offset = p.off;
  if (rows) *rows = p.rows;
  if (cols) *cols = p.cols;
  return MMDA_OK;
}
extern "C" int64_t mmda_misa_flat_floats(const mmda_misa* m) { return m ? m->flat : MMDA_EINVAL; }
extern "C" int64_t mmda_misa_dense_floats(const mmda_misa* m) { return m ? m->dense : MMDA_EINVAL; }
extern "C" int mmda_misa_bind(mmda_misa* m, float* params, float* grads, float* adam_m, float* adam_v) {
  if (!m || !params) return MMDA_EINVAL;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)adam_m | (uintptr_t)adam_v) & 15) return MMDA_EINVAL;
  m->P = params; m->G = grads; m->M1 = adam_m; m->V1 = adam_v;
  return MMDA_OK;
}
extern "C" int64_t mmda_misa_workspace_floats(const mmda_misa* m, int B, int T) {
  if (!m || B <= 0 || T <= 0) return MMDA_EINVAL;
  return layout(const_cast<mmda_misa*>(m), B, T, false);
}
namespace {
// abort words of the CURRENT exchange regions -> abort_sticky (synchronous device->host copies)
int harvest_abort(mmda_misa* m) {
  if (!m->ws || !m->xchg_ws || m->xchg_ws != m->ws) return MMDA_OK;
  for (int i = 0; i < 3; ++i) {
    if (m->mod[i].xchg < 0) continue;
    unsigned w = 0;
    if (hipMemcpy(&w, m->ws + m->mod[i].xchg, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return MMDA_ELAUNCH;
    if (w) m->abort_sticky = 1;
  }
  return MMDA_OK;
}
}  // namespace

extern "C" int mmda_misa_set_workspace_async(mmda_misa* m, float* ws, int64_t floats, int B, int T, void* stream) {
  if (!m || !ws || B <= 0 || T <= 0 || ((uintptr_t)ws & 15)) return MMDA_EINVAL;
  int64_t need = layout(m, B, T, false);
  if (floats < need) return MMDA_EINVAL;
  // The exchange regions keep their place and their contents while the buffer and B stay the same (a new T only): nothing to clear,
  // the flags there are monotonic epochs.  Otherwise they start from zero -- after the abort words of the old ones were looked at
  // (same buffer, new B: one synchronous read per region, rare; a NEW buffer: the caller reads mmda_misa_cluster_status first).
  const bool keep = m->xchg_ws == ws && m->xchg_B == B;
  if (!keep && m->xchg_ws == ws) { int rc = harvest_abort(m); if (rc) return rc; }
  layout(m, B, T, true);
  m->ws = ws; m->ws_floats = floats; m->B = B; m->T = T; m->ldR = round_up(B * T, 8);
  hipStream_t s = (hipStream_t)stream;
  if (!keep) {
    for (int i = 0; i < 3; ++i)
      if (m->mod[i].xchg >= 0 && hipMemsetAsync(ws + m->mod[i].xchg, 0, sizeof(float) * m->mod[i].xchg_floats, s) != hipSuccess) return MMDA_ELAUNCH;
    m->xchg_ws = ws; m->xchg_B = B;
  }
  if (m->gpad_end > m->gpad_begin && hipMemsetAsync(ws + m->gpad_begin, 0, sizeof(float) * (m->gpad_end - m->gpad_begin), s) != hipSuccess)
    return MMDA_ELAUNCH;
  return MMDA_OK;
}
extern "C" int mmda_misa_set_workspace(mmda_misa* m, float* ws, int64_t floats, int B, int T) {
  return mmda_misa_set_workspace_async(m, ws, floats, B, T, nullptr);       // the null stream: ordered against every blocking stream
}
extern "C" int64_t mmda_misa_tensor_offset(const mmda_misa* m, const char* name) {
  if (!m || !name) return -1;
  auto it = m->tens.find(name);
  return it == m->tens.end() ? -1 : it->second;
}
extern "C" int mmda_misa_set_mode(mmda_misa* m, int mode) {
  if (!m || (mode != MMDA_F32 && mode != MMDA_BF16)) return MMDA_EINVAL;
  m->cfg.mode = mode;
  return MMDA_OK;
}
extern "C" int mmda_misa_set_overlap(mmda_misa* m, int side_stream) {
  if (!m) return MMDA_EINVAL;
  m->use_side = side_stream ? 1 : 0;
  return MMDA_OK;
}
extern "C" int mmda_misa_set_recurrence(mmda_misa* m, int resident_weights) {
  if (!m) return MMDA_EINVAL;
  m->use_cluster = resident_weights ? 1 : 0;
  return MMDA_OK;
}
extern "C" int mmda_misa_set_gemm_operands(mmda_misa* m, int bf16_copies) {
  if (!m) return MMDA_EINVAL;
  m->use_bf16_gemm = bf16_copies ? 1 : 0;
  return MMDA_OK;
}
extern "C" int64_t mmda_misa_early_grad_floats(const mmda_misa* m) {
  return (m && m->early_valid) ? m->early_floats : 0;
}
extern "C" int mmda_misa_wait_early_grads(mmda_misa* m, void* stream) {
  if (!m || !m->early_valid || !m->ev_early) return MMDA_EINVAL;
  if (hipStreamWaitEvent((hipStream_t)stream, m->ev_early, 0) != hipSuccess) return MMDA_ELAUNCH;
  return MMDA_OK;
}
extern "C" int mmda_misa_set_external_batch_losses(mmda_misa* m, int on) {
  if (!m) return MMDA_EINVAL;
  m->ext_batch_losses = on ? 1 : 0;
  return MMDA_OK;
}
extern "C" int mmda_misa_set_inference(mmda_misa* m, int forward_only) {
  if (!m) return MMDA_EINVAL;
  m->inference = forward_only ? 1 : 0;
  return MMDA_OK;
}
extern "C" int mmda_misa_cluster_status(const mmda_misa* m, int* aborted_host) {
  // reads the sticky abort words of the three exchange buffers (device->host copy: call it off the step path)
  if (!m || !m->ws || !aborted_host) return MMDA_EINVAL;
  *aborted_host = m->abort_sticky;          // seen in an exchange region that has since been cleared (B changed)
  if (m->jflags) {                          // a flag join timed out (see mmda_misa::jflags)
    unsigned w = 0;
    if (hipMemcpy(&w, m->jflags + 2, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return MMDA_ELAUNCH;
    if (w) *aborted_host |= 2;                 // (bit 1: a stream-to-stream flag wait, not a recurrence)
  }
  for (int i = 0; i < 3; ++i) {
    if (m->mod[i].xchg < 0) continue;
    unsigned w = 0;
    if (hipMemcpy(&w, m->ws + m->mod[i].xchg, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return MMDA_ELAUNCH;
    if (w) *aborted_host = 1;
  }
  return MMDA_OK;
}

// =============================================================================================== forward
namespace {
// K-major (transposed) fp32 copies of the fusion block's weights for its input-gradient GEMMs in the backward pass; issued on a side
// stream that the end of forward() joins
void weight_transpose_jobs(mmda_misa* m, std::vector<mmda_transpose_job>& tj) {
  const mmda_misa_config& c = m->cfg;
  const int hs_ = c.hidden, NC_ = 6 + c.ncls;
  auto T_ = [&](int64_t src, int rows, int cols, int64_t dst) { tj.push_back(mmda_transpose_job{PP(src), rows, cols, cols, WS(dst), rows}); };
  T_(m->head_w, NC_, 6 * hs_, m->head_wT); T_(m->l2_w, hs_, FFN, m->l2_wT); T_(m->l1_w, FFN, hs_, m->l1_wT);
  T_(m->out_w, hs_, hs_, m->out_wT); T_(m->in_w, 3 * hs_, hs_, m->in_wT); T_(m->sh_w, hs_, hs_, m->sh_wT);
  for (int i = 0; i < 3; ++i) {
    T_(m->rec_w + (int64_t)i * hs_ * hs_, hs_, hs_, m->rec_wT + (int64_t)i * hs_ * hs_);
    T_(m->priv_w + (int64_t)i * hs_ * hs_, hs_, hs_, m->priv_wT + (int64_t)i * hs_ * hs_);
    T_(m->mod[i].pw, hs_, 4 * m->mod[i].H, m->pwT[i]);
  }
  if (!c.use_cmd_sim) { T_(m->d1_w, hs_, hs_, m->d1_wT); T_(m->d2_w, 3, hs_, m->d2_wT); }
}
int weight_transposes(mmda_misa* m, void* ss) {
  std::vector<mmda_transpose_job> tj;
  weight_transpose_jobs(m, tj);
  const int rc = mmda_transpose_f32(tj.data(), (int)tj.size(), ss);
  m->wT_valid = rc ? 0 : 1;
  m->wT_pending = 0;
  return rc;
}

// bf16 operand copies that only the backward pass reads -- hseq of both layers (per direction for the tn weight-gradient GEMMs,
// transposed for the nt ones) and, nt only, the transposed layer-2 inputs: made on the side stream beside the fusion block.  <= 15 jobs.
int backward_only_jobs(mmda_misa* m, mmda_convert_job* cj) {
  const int R = m->B * m->T;
  int n = 0;
  for (int i = 0; i < 3; ++i) {
    Mod& md = m->mod[i];
    for (int l = 0; l < 2; ++l) {
      Rnn& r = md.rnn[l];
      if ((m->tn_wgrad >> l) & 1) {
        for (int d = 0; d < 2; ++d)
          cj[n++] = mmda_convert_job{WS(md.hseq[l]) + d * md.H, 2 * md.H, R, md.H, nullptr, WS(r.hbp[d]), r.ldH, nullptr, 0};
      } else {
        cj[n++] = mmda_convert_job{WS(md.hseq[l]), 2 * md.H, R, 2 * md.H, nullptr, nullptr, 0, WS(r.hbT), m->ldR};
      }
    }
    if (!(m->tn_wgrad & 2)) cj[n++] = mmda_convert_job{WS(md.normed), 2 * md.H, R, 2 * md.H, nullptr, nullptr, 0, WS(md.rnn[1].xbT), m->ldR};
  }
  return n;
}

// side stream, right after x6 = [private x3, shared x3] exists: clear the loss sums and the loss-seeded activation gradients,
// then DiffLoss and CMD with their gradients (they read x6 only), then the gradient bucket if train_step left that to forward()
int eager_side_losses(mmda_misa* m, void* stream, bool hseq2_t) {
  if (!m->eager_losses) return MMDA_OK;
  const mmda_misa_config& c = m->cfg;
  const int B = m->B, hs = c.hidden;
  const int64_t BH = (int64_t)B * hs;
  void* ss = nullptr;
  int rc = side_fork(m, stream, &ss);
  // (large batches: the loss chain on the side stream is the longer one by far -- the weight transposes go to the main stream)
  if (!rc && m->wT_pending) rc = weight_transposes(m, B >= 128 ? stream : ss);
  (void)hseq2_t;                   // (the backward pass's operand copies are made by backward() itself: backward_only_jobs)
  // (the forward stretches on the main stream STORE the seeds they own: clearing those here would race with them)
  const int64_t zend = m->seed_cls ? m->zero_cls : m->seed_recon ? m->zero_recon : m->zero_end;
  // The gradient bucket (43 MB, 11 us) is cleared on the MAIN stream behind the fork: since the row-local stretches were fused the
  // side stream's loss chain (72 us at B=32), not the main stream's fusion block (55 us), is what the join at the end of forward() waits
  // for.  (Nothing on either stream touches the bucket before the backward pass; MMDA_ZERO_GRAD_SIDE=1: the old place.)
  // ... unless the main stream will not wait for this chain before the LayerNorm-1 stretch of the backward pass (flag join on the
  // device, small batches: see mmda_misa::jflags) -- the chain then has two launches of slack and the clear comes back here, in ONE
  // launch with the activation-gradient region, at the head of the chain.
  static const int zg_side = getenv("MMDA_ZERO_GRAD_SIDE") ? atoi(getenv("MMDA_ZERO_GRAD_SIDE")) : -1;
  const bool fj_plan = m->flag_join_ok && m->seed_recon && m->seed_cls && ss != stream && m->jflags;      // (what forward()'s end will decide)
  const bool zg_here = zg_side >= 0 ? zg_side != 0 : (fj_plan && B <= 64);
  if (!rc && zg_here && m->zero_grad_pending) {
    rc = mmda_zero2(WS(m->zero_begin), zend - m->zero_begin, m->G, m->flat, ss);
    m->zero_grad_pending = 0;
  } else if (!rc) {
    if (hipMemsetAsync(WS(m->zero_begin), 0, sizeof(float) * (zend - m->zero_begin), (hipStream_t)ss) != hipSuccess) rc = MMDA_ELAUNCH;
  }
  float* L = WS(m->losses);
  if (!rc) rc = mmda_loss_diff(WS(m->x6), BH, B, hs, c.diff_weight, L + 1, WS(m->d_x6), WS(m->diff_work), ss);
  // the chain's last launch sets the flag-join word itself where it can (single-workgroup CMD: no one-thread launch behind it)
  m->fj1_armed = 0;
  if (!rc && c.use_cmd_sim && fj_plan && mmda_loss_cmd_sets_flag(B, hs)) {
    mmda_loss_cmd_arm_flag(m->jflags + 0, m->jval[0] + 1);
    m->fj1_armed = 1;
  }
  if (!rc && c.use_cmd_sim) rc = mmda_loss_cmd(WS(m->x6 + 3 * BH), BH, B, hs, c.sim_weight, L + 2, WS(m->d_x6 + 3 * BH), ss);
  if (!rc && m->zero_grad_pending) { rc = mmda_misa_zero_grad(m, stream); m->zero_grad_pending = 0; }
  m->eager_done = 1;
  return rc;
}
}  // namespace

namespace {
// feed-forward of the fusion transformer layer on block-scaled fp8 operands (models.py:160-161; torch's linear1 -> relu -> dropout ->
// linear2): x1 (6B, hs) -> f1 (6B, FFN) -> f2 (6B, hs).  Four launches: quantise {x1, W1, W2}, product 1 (+bias, relu, dropout),
// quantise f1, product 2 (+bias).  The f32 tensors f1 / f2 the backward pass reads are written as on the exact path.
int ffn_fp8(mmda_misa* m, float p_tf, uint64_t seed, void* stream) {
  const mmda_misa_config& c = m->cfg;
  const int hs = c.hidden, R6 = 6 * m->B;
  if ((hs % 128) != 0) return MMDA_EINVAL;               // K of product 1 must be whole 128-deep MFMA steps
  auto U8 = [&](int64_t off) { return reinterpret_cast<unsigned char*>(m->ws + off); };
  mmda_mx8_quant_job q[3] = {
      {m->ws + m->x1, hs, R6, hs, U8(m->x1q), U8(m->x1s)},
      {m->P + m->l1_w, hs, FFN, hs, U8(m->w1q), U8(m->w1s)},
      {m->P + m->l2_w, FFN, hs, FFN, U8(m->w2q), U8(m->w2s)}};
  int rc = mmda_mx8_quant(q, 3, stream);
  if (rc) return rc;
  mmda_mx8_args g = {};
  g.M = R6; g.N = FFN; g.K = hs; g.Aq = U8(m->x1q); g.As = U8(m->x1s); g.Bq = U8(m->w1q); g.Bs = U8(m->w1s);
  g.C = m->ws + m->f1; g.ldc = FFN; g.bias = m->P + m->l1_b; g.act = MMDA_ACT_RELU; g.drop_p = p_tf; g.drop_seed = seed; g.drop_site = SITE_FFN;
  rc = mmda_gemm_mx8(&g, stream);
  if (rc) return rc;
  mmda_mx8_quant_job qf = {m->ws + m->f1, FFN, R6, FFN, U8(m->f1q), U8(m->f1s)};
  rc = mmda_mx8_quant(&qf, 1, stream);
  if (rc) return rc;
  mmda_mx8_args h = {};
  h.M = R6; h.N = hs; h.K = FFN; h.Aq = U8(m->f1q); h.As = U8(m->f1s); h.Bq = U8(m->w2q); h.Bs = U8(m->w2s);
  h.C = m->ws + m->f2; h.ldc = hs; h.bias = m->P + m->l2_b;
  return mmda_gemm_mx8(&h, stream);
}
}  // namespace

extern "C" int mmda_misa_set_fusion_fp8(mmda_misa* m, int on) {
  if (!m) return MMDA_EINVAL;
  if (on && (m->cfg.hidden % 128) != 0) return MMDA_EINVAL;
  m->fusion_fp8 = on ? 1 : 0;
  return MMDA_OK;
}

extern "C" int mmda_misa_forward(mmda_misa* m, const int64_t* t_ids, const float* v, const float* a, const int32_t* lengths,
                                 int training, uint64_t seed, void* stream) {
  if (check_ready(m) || !t_ids || !v || !a || !lengths) return MMDA_EINVAL;
  const mmda_misa_config& c = m->cfg;
  const int B = m->B, T = m->T, hs = c.hidden, mode = c.mode, NC = 6 + c.ncls;
  // The fusion block (projections, private/shared/recon, transformer layer, heads) is 2 % of the FLOPs and feeds the
  // batch-statistic losses: it always runs on the exact f32 MFMA path.  `mode` (bf16) covers the LSTM GEMMs + recurrences.
  const int fmode = MMDA_F32;
  const int R = T * B;
  Ctx x{m, stream};
  m->training = training; m->seed = seed;
  const float p_tf = training ? c.fusion_dropout : 0.f, p_cls = training ? c.dropout : 0.f;

  // W_hh -> MFMA fragment order (weights changed since the last step): all twelve matrices in one launch, on the side stream
  // underneath the embedding gather and the first input GEMM; joined before the first recurrent kernel.
  if (is_gru(m)) {          // GRU parameters -> four-slot layout (everything below reads the padded copies)
    mmda_gru_pad_job gj[MMDA_GRU_PAD_MAX];
    int n = gru_jobs(m, m->P, false, gj);
    x.rc = mmda_gru_pad_params(gj, n, stream);
    if (x.rc) return x.rc;
  }
  // which recurrent kernels will run: decides the packings of W_hh that are made and the layout of `gates`
  auto probe_resident = [&](int gate_minor, int backward) -> bool {
    if (T <= 0 || mode != MMDA_BF16) return false;
    mmda_lstm_desc probe[3];
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i]; Rnn& r = md.rnn[0];
      probe[i] = mmda_lstm_desc{};
      probe[i].H = r.H; probe[i].gates = WS(md.gates[0]); probe[i].cstash = WS(md.c[0]); probe[i].hseq = WS(md.hseq[0]);
      probe[i].wpack[0] = WS(r.pack_f[0]); probe[i].wpack[1] = WS(r.pack_f[1]);
      probe[i].wpack_c[0] = WS(r.pack_c[0]); probe[i].wpack_c[1] = WS(r.pack_c[1]); probe[i].utt = WS(md.utt);
      probe[i].xchg = (m->use_cluster && md.xchg >= 0) ? (void*)WS(md.xchg) : nullptr; probe[i].gate_minor = gate_minor;
      probe[i].cell = c.rnncell;
    }
    if (backward == 2) return mmda_lstm_bwd_emits_dg_bf16(mode, 3, probe, B, T) != 0;      // ... and writes the gate gradients as bf16
    return mmda_lstm_resident_applicable(mode, 3, probe, B, T, backward) != 0;
  };
  // bf16 mode: the input GEMMs read bf16 operand copies (K-major, 16-B rows): W_ih of both layers (plain for the forward,
  // transposed for dX) and the layer-1 inputs -- the text rows are gathered from the embedding matrix by the conversion itself
  // (models.py:201), so no fp32 copy of them is made.
  const bool bfg = mode == MMDA_BF16 && m->use_bf16_gemm;
  const int ldR = m->ldR;
  const float* xin[3] = {WS(m->mod[0].x), v, a};
  // Gate-minor layout of the pre-activations / stash / gate gradients ([dir][unit][gate], 16-byte accesses in the recurrent
  // kernels): possible when the bf16 GEMMs produce and consume them (the interleave rides on the W_ih conversion and on the
  // GEMM epilogues) AND the resident-weights kernels will run, forward and backward.
  int gm = 0;
  static const int no_gm = getenv("MMDA_NO_GATE_MINOR") ? 1 : 0;     // ablation switch
  if (bfg && (B % 8) == 0 && T > 0 && !no_gm) gm = probe_resident(1, 0) && probe_resident(1, 1);
  m->gate_minor = gm;
  const bool inf = m->inference != 0;                   // no backward follows: transposed copies and stashes are not needed
  m->last_fwd_inference = inf;
  // The transposed copies are read by the backward pass only: with a side stream they are made there, beside the recurrent
  // kernels (`late_t`), and the main stream converts just what the forward GEMMs read.
  // (measured: the fork's marker packet on the main stream and the L2 traffic beside the forward recurrences cost ~15 us more
  // than the smaller main-stream conversions save, so this stays an ablation switch, off by default)
  static const int late_t_on = getenv("MMDA_LATE_T") ? atoi(getenv("MMDA_LATE_T")) : 0;
  const bool late_t = bfg && m->use_side && !inf && late_t_on;
  // Weight gradients in the tn form of the bf16 GEMM (dW = dG^T X on row-major dG, X, hseq): the transposed copies of the inputs, of
  // hseq and of the gate gradients are not made at all (B=256: 0.33 ms of conversions per step).  Needs the gate gradients as bf16
  // from the recurrent kernel (gate-minor resident path).  Up to T*B = 4096 rows: measured (step, ms) B=32 0.759 -> 0.745, B=64 0.928 ->
  // 0.917, but B=128 1.362 -> 1.391, B=256 2.39 -> 2.55, T=500 4.06 -> 4.08 -- in isolation the tn kernel matches the nt one at K = 1600
  // and runs 15 - 20 % slower at K = 12800 (twice the LDS read instructions per k-tile), which at large batches outweighs the
  // conversions it saves.  MMDA_GEMM_TN=0: the transposed-copy (nt) form everywhere; MMDA_GEMM_TN_MAX_ROWS moves the limit.
  static const int tn_on = getenv("MMDA_GEMM_TN") ? atoi(getenv("MMDA_GEMM_TN")) : 1;
  // Round 3: the LDS-DMA pipelined GEMM (gemm_bf16_dma_kernel) runs the tn form at K = 12800 as fast as the nt form, so the limit is
  // gone by default (B=256: the 0.47 ms of transposing conversions per step with it).
  static const int tn_max_rows = getenv("MMDA_GEMM_TN_MAX_ROWS") ? atoi(getenv("MMDA_GEMM_TN_MAX_ROWS")) : (1 << 30);
  // (Layer 1 alone in the tn form beyond that limit -- its transposed gate-gradient copy, 0.2 ms at B=256, is the one that cannot hide
  // beside a recurrence -- measured slower too: B=128 1.285 -> 1.315 ms, B=256 2.20 -> 2.29, T=500 3.92 -> 3.96.  MMDA_GEMM_TN_L1=1.)
  static const int tn_l1 = getenv("MMDA_GEMM_TN_L1") ? atoi(getenv("MMDA_GEMM_TN_L1")) : 0;
  const bool tn_any = bfg && gm && tn_on && !late_t && (B % 8) == 0 && probe_resident(1, 2);
  const bool tnw = tn_any && R <= tn_max_rows;
  const bool tn_ok = tnw || (tn_any && tn_l1);           // layer 1 in the tn form
  m->tn_wgrad = (tn_ok && !inf) ? (tnw ? 3 : 1) : 0;
  auto first_jobs = [&](bool plain, bool transposed, mmda_convert_job* cj) -> int {
    int n = 0;
    for (int i = 0; i < 3; ++i) {
      for (int l = 0; l < 2; ++l) {
        Rnn& r = m->mod[i].rnn[l];
        cj[n++] = mmda_convert_job{rW_ih(m, r), r.D, 8 * r.H, r.D, nullptr, plain ? WS(r.wb) : nullptr, plain ? r.ldD : 0,
                                   transposed ? WS(r.wbT) : nullptr, transposed ? r.ldG : 0, gm ? r.H : 0};
      }
      Rnn& r0 = m->mod[i].rnn[0];
      const float* src = i == 0 ? PP(m->embed) : xin[i];
      const bool xt = transposed && !tn_ok;              // layer-1 inputs transposed: only the nt form of layer 1's dW_ih reads them
      if (plain || xt)
        cj[n++] = mmda_convert_job{src, r0.D, R, r0.D, i == 0 ? t_ids : nullptr, plain ? WS(r0.xb) : nullptr, plain ? r0.ldD : 0,
                                   xt ? WS(r0.xbT) : nullptr, xt ? ldR : 0};
    }
    return n;
  };
  auto first_converts = [&](bool plain, bool transposed, void* st) -> int {
    mmda_convert_job cj[9];
    const int n = first_jobs(plain, transposed, cj);
    return mmda_convert_bf16(cj, n, st);
  };
  // Merged form (bf16 operand copies in use): the twelve packings and the first conversions go out as ONE launch on the main stream
  // (mmda_lstm_pack_whh_and_convert) -- no fork, no cross-stream wait in front of the first recurrent kernel (each costs the main stream
  // 4 - 14 us).  The K-major copies of the fusion block's weights, which only the backward pass reads, then ride on the first fork that
  // happens anyway (wT_pending).  MMDA_PACK_MERGE=0: round 1's form (packing on the side stream, joined by an event).
  static const int pack_merge = getenv("MMDA_PACK_MERGE") ? atoi(getenv("MMDA_PACK_MERGE")) : 1;
  const bool merged = bfg && pack_merge && !late_t;
  m->wT_pending = 0;
  {
    // The forward packing always; the resident-weights backward packing when those kernels will run the backward pass and the
    // streaming backward packing only when they will not; neither for an evaluation pass.
    const bool infer = m->inference != 0;
    const bool want_c = !infer && m->use_cluster && mode == MMDA_BF16;
    const bool want_b = !infer && !(want_c && probe_resident(0, 1));
    int Hs[12]; const float* Wp[12]; void* Fp[12]; void* Bp[12]; void* Cp[12];
    int k = 0;
    for (int i = 0; i < 3; ++i)
      for (int l = 0; l < 2; ++l)
        for (int d = 0; d < 2; ++d, ++k) {
          Rnn& r = m->mod[i].rnn[l];
          Hs[k] = r.H; Wp[k] = rW_hh(m, r, d); Fp[k] = WS(r.pack_f[d]); Bp[k] = want_b ? WS(r.pack_b[d]) : nullptr; Cp[k] = WS(r.pack_c[d]);
        }
    m->pack_b_valid = want_b ? 1 : 0;
    m->packed_c_valid = want_c ? 1 : 0;
    m->wT_valid = 0;
    const bool want_wT = B <= SKINNY_MAX_B && !m->inference;
    if (merged) {
      mmda_convert_job cj[9];
      const int nj = first_jobs(true, !inf, cj);
      // ... and the K-major copies of the fusion block's weights (backward pass) in the same launch: 6 MB of traffic that cost a
      // launch of its own 7 - 9 us at the head of the loss chain (side stream, the longer of the two chains beside the fusion block)
      // or, at B >= 128, 15 us with its gap on the main stream.  MMDA_WT_MERGE=0: on the first fork as before.
      static const int wt_merge = getenv("MMDA_WT_MERGE") ? atoi(getenv("MMDA_WT_MERGE")) : 1;
      std::vector<mmda_transpose_job> tj;
      if (want_wT && wt_merge) weight_transpose_jobs(m, tj);
      if (tj.size() > 20) tj.clear();
      x.rc = mmda_lstm_pack_convert_transpose(12, Hs, Wp, Fp, Bp, want_c ? Cp : nullptr, cj, nj, tj.data(), (int)tj.size(), stream);
      m->wT_pending = (want_wT && tj.empty()) ? 1 : 0;
      if (!tj.empty()) m->wT_valid = x.rc ? 0 : 1;
    } else {
      void* ss = nullptr;
      x.rc = side_fork(m, stream, &ss);
      if (!x.rc) x.rc = mmda_lstm_pack_whh_multi(mode, 12, Hs, Wp, Fp, Bp, want_c ? Cp : nullptr, ss);
      // the first recurrent kernel waits for the packing only, not for the transposes issued behind it
      if (!x.rc && ss != stream && hipEventRecord(m->ev_pack, (hipStream_t)ss) != hipSuccess) x.rc = MMDA_ELAUNCH;
      if (!x.rc && want_wT) x.rc = weight_transposes(m, ss);
    }
  }
  if (x.rc) return x.rc;
  if (merged) {
    // (done by the merged launch above)
  } else if (bfg) {
    x.rc = first_converts(true, !inf && !late_t, stream);
  } else {
    // embedding rows (models.py:201)
    x.rc = mmda_embed_gather(PP(m->embed), t_ids, R, c.d_t, WS(m->mod[0].x), stream);
  }
  for (int l = 0; l < 2; ++l) {
    mmda_lstm_desc desc[3];
    mmda_gemm_bf16_args bg[3];
    group_begin(x);
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i]; Rnn& r = md.rnn[l];
      const float* in = l == 0 ? xin[i] : WS(md.normed);
      // time-batched input-to-hidden GEMM for both directions: (R, D) x (8H, D)^T + b_ih + b_hh
      if (bfg) {
        bg[i] = mmda_gemm_bf16_args{};
        bg[i].M = R; bg[i].N = 8 * r.H; bg[i].K = r.D; bg[i].A = WS(r.xb); bg[i].lda = r.ldD; bg[i].B = WS(r.wb); bg[i].ldb = r.ldD;
        bg[i].C = WS(md.gates[l]); bg[i].ldc = 8 * r.H; bg[i].bias = rB_ih(m, r); bg[i].bias2 = rB_hh(m, r);
        bg[i].perm_n_H = gm ? r.H : 0;
      } else {
        gemm(x, mode, 0, 1, R, 8 * r.H, r.D, in, r.D, rW_ih(m, r), r.D, WS(md.gates[l]), 8 * r.H, rB_ih(m, r), rB_hh(m, r));
      }
      desc[i] = mmda_lstm_desc{};
      desc[i].H = r.H; desc[i].gates = WS(md.gates[l]); desc[i].cstash = WS(md.c[l]); desc[i].hseq = WS(md.hseq[l]);
      desc[i].wpack[0] = WS(r.pack_f[0]); desc[i].wpack[1] = WS(r.pack_f[1]);
      desc[i].utt = WS(md.utt); desc[i].layer = l; desc[i].d_hseq = nullptr;
      desc[i].xchg = (m->use_cluster && md.xchg >= 0) ? (void*)WS(md.xchg) : nullptr; desc[i].epoch_base = m->epoch;
      desc[i].gate_minor = gm; desc[i].forward_only = inf; desc[i].cell = c.rnncell;
    }
    if (bfg && !x.rc) x.rc = mmda_gemm_bf16_grouped(bg, 3, stream);
    m->epoch += (unsigned)T + 2u;
    group_end(x);
    if (!x.rc && l == 0 && m->side_pending && m->use_side) {   // packed W_hh ready (the side stream carries on with its transposes)
      if (hipStreamWaitEvent((hipStream_t)stream, m->ev_pack, 0) != hipSuccess) x.rc = MMDA_ELAUNCH;
    }
    if (x.rc) return x.rc;
    if (late_t) {
      // Side stream, beside this layer's recurrent kernel (which leaves ~145 CUs idle), joined at the end of forward():
      //   layer 1: W_ih^T of both layers (dX) and the layer-1 inputs transposed (dW_ih)
      //   layer 2: its inputs transposed (dW_ih) and hseq^T of layer 1 (dW_hh)
      // A join costs the main stream ~10 us however early the side stream finished, so only work that an existing join covers
      // is moved there.
      void* ss = nullptr;
      x.rc = side_fork(m, stream, &ss);
      if (!x.rc && l == 0) x.rc = first_converts(false, true, ss);
      if (!x.rc && l == 1) {
        mmda_convert_job cj[6];
        for (int i = 0; i < 3; ++i) {
          Mod& md = m->mod[i];
          cj[i] = mmda_convert_job{WS(md.hseq[0]), 2 * md.H, R, 2 * md.H, nullptr, nullptr, 0, WS(md.rnn[0].hbT), ldR};
          cj[3 + i] = mmda_convert_job{WS(md.normed), 2 * md.H, R, 2 * md.H, nullptr, nullptr, 0, WS(md.rnn[1].xbT), ldR};
        }
        x.rc = mmda_convert_bf16(cj, 6, ss);
      }
      if (x.rc) return x.rc;
    }
    ev_rec(m, m->ev_fwd, l, 0, stream);
    x.rc = mmda_lstm_fwd(mode, 3, desc, B, T, lengths, stream);
    ev_rec(m, m->ev_fwd, l, 1, stream);
    if (x.rc) return x.rc;
    if (l == 0) {
      mmda_ln_args ln[3];
      for (int i = 0; i < 3; ++i) {
        Mod& md = m->mod[i];
        ln[i] = mmda_ln_args{};
        ln[i].rows = R; ln[i].n = 2 * md.H; ln[i].x = WS(md.hseq[0]); ln[i].gamma = PP(md.ln_w); ln[i].beta = PP(md.ln_b);
        ln[i].y = WS(md.normed); ln[i].mean = WS(md.ln_mean); ln[i].rstd = WS(md.ln_rstd); ln[i].eps = 1e-5f;
      }
      // (bf16 GEMMs: the LayerNorm also writes its output as the bf16 operand copy layer 2's input GEMM reads; the copies that only
      //  the backward pass needs -- hseq, transposed inputs -- are made later on the side stream: backward_only_jobs)
      if (bfg)
        for (int i = 0; i < 3; ++i) { ln[i].y_bf16 = WS(m->mod[i].rnn[1].xb); ln[i].ld_bf16 = m->mod[i].rnn[1].ldD; }
      // ... and ONLY as that copy where nothing reads the fp32 output: an evaluation pass, or a training step whose layer-2 weight
      // gradients take the tn form (they read the bf16 copy; the nt form converts the fp32 output into a transposed copy)
      if (bfg && (inf || (m->tn_wgrad & 2)))
        for (int i = 0; i < 3; ++i) ln[i].y = nullptr;
      x.rc = mmda_layernorm_fwd_multi(ln, 3, stream);
    } else if (!x.rc && !m->eager_losses && ((bfg && !inf) || m->zero_grad_pending || m->wT_pending)) {
      // side stream, beside the fusion block: the gradient bucket is cleared (train_step) and hseq^T of layer 2 is made for its
      // dW_hh.  Joined at the end of forward(), so everything the backward pass issues on either stream is ordered behind both.
      void* ss = nullptr;
      x.rc = side_fork(m, stream, &ss);
      if (!x.rc && m->wT_pending) x.rc = weight_transposes(m, ss);
      if (!x.rc && m->zero_grad_pending && !m->eager_losses) { x.rc = mmda_misa_zero_grad(m, ss); m->zero_grad_pending = 0; }

    }
  }
  if (x.rc) return x.rc;
  // shared_private (models.py:265-279)
  const int64_t BH = (int64_t)B * hs;
  if (B <= SKINNY_MAX_B) {
    // ---- few rows: row-skinny GEMMs, independent ones grouped per launch (12 launches for the whole block)
    mmda_skinny_args g[8];
    mmda_ln_args ln[3];
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i];
      g[i] = sk_nt(B, hs, 4 * md.H, WS(md.utt), 4 * md.H, PP(md.pw), PP(md.pb), WS(m->z + i * BH), hs);
      ln[i] = mmda_ln_args{};
      ln[i].rows = B; ln[i].n = hs; ln[i].x = WS(m->z + i * BH); ln[i].gamma = PP(md.plw); ln[i].beta = PP(md.plb);
      ln[i].y = WS(m->orig + i * BH); ln[i].mean = WS(m->pmean + i * B); ln[i].rstd = WS(m->prstd + i * B); ln[i].act = c.act;
      ln[i].eps = 1e-5f; ln[i].actp = act_params(m, training, seed, SITE_RRELU + i, false);
    }
    sk_launch(x, g, 3);
    if (!x.rc) x.rc = mmda_layernorm_fwd_multi(ln, 3, stream);
    // private x3 and shared (one weight over the stacked 3B rows), sigmoid epilogue
    for (int i = 0; i < 3; ++i)
      g[i] = sk_nt(B, hs, hs, WS(m->orig + i * BH), hs, PP(m->priv_w + (int64_t)i * hs * hs), PP(m->priv_b + i * hs), WS(m->x6 + i * BH), hs,
                   MMDA_ACT_SIGMOID);
    g[3] = sk_nt(3 * B, hs, hs, WS(m->orig), hs, PP(m->sh_w), PP(m->sh_b), WS(m->x6 + 3 * BH), hs, MMDA_ACT_SIGMOID);
    sk_launch(x, g, 4);
    // The row-local stretches as one launch each (fused_rows.hip): recon + qkv -> attention -> out-proj -> LayerNorm 1, and
    // LayerNorm 2 -> heads.  MMDA_ROW_FUSE=0: the launches they replace (4 and 3).
    static const int row_fuse_on = getenv("MMDA_ROW_FUSE") ? atoi(getenv("MMDA_ROW_FUSE")) : 1;
    const bool row_fuse = row_fuse_on && c.use_cmd_sim && hs == 128 && NHEAD == 2;
    // loss seeds by the stretches (see mmda_misa::emo_eager); MMDA_LOSS_SEEDS=0: by the loss launch behind the forward pass, as before
    static const int loss_seeds_on = getenv("MMDA_LOSS_SEEDS") ? atoi(getenv("MMDA_LOSS_SEEDS")) : 1;
    m->seed_recon = (loss_seeds_on && row_fuse && m->eager_losses && m->emo_eager) ? 1 : 0;
    m->seed_cls = (m->seed_recon && !c.use_confidNet) ? 1 : 0;
    if (!x.rc) x.rc = eager_side_losses(m, stream, bfg && !inf);
    static const int fuse_nb_env = getenv("MMDA_ROW_FUSE_NB") ? atoi(getenv("MMDA_ROW_FUSE_NB")) : 1;      // samples per workgroup (B=32: 0.722 ms with 2, 0.712 with 1; B=256 equal)
    const int fuse_nb = ((B % 2) == 0 && fuse_nb_env == 2) ? 2 : 1;
    mmda_ln_args l1 = {};
    l1.rows = 6 * B; l1.n = hs; l1.x = WS(m->x6); l1.res = WS(m->attn_out); l1.gamma = PP(m->n1_w); l1.beta = PP(m->n1_b);
    l1.y = WS(m->x1); l1.mean = WS(m->ln1_mean); l1.rstd = WS(m->ln1_rstd); l1.drop_p = p_tf; l1.drop_seed = seed;
    l1.drop_site = SITE_DROP1; l1.eps = 1e-5f;
    if (row_fuse) {
      if (!x.rc) {
        FusedFwdA f = {};
        f.B = B; f.hs = hs; f.nhead = NHEAD; f.nb = fuse_nb; f.x6 = WS(m->x6);
        f.rec_w = PP(m->rec_w); f.rec_b = PP(m->rec_b); f.recon = WS(m->recon);
        f.in_w = PP(m->in_w); f.in_b = PP(m->in_b); f.qkv = WS(m->qkv);
        f.ctx = WS(m->ctx); f.probs = WS(m->probs); f.p_tf = p_tf; f.seed = seed; f.site_attn = SITE_ATTN;
        f.out_w = PP(m->out_w); f.out_b = PP(m->out_b); f.attn_out = WS(m->attn_out); f.ln1 = l1;
        if (m->seed_recon) {
          f.orig = WS(m->orig); f.d_recon = WS(m->d_recon); f.d_orig = WS(m->d_orig);
          f.recon_inv_n = 1.0f / (float)(3 * BH); f.recon_scale = c.recon_weight;
        }
        x.rc = mmda_fused_fwd_a(&f, stream);
      }
    } else {
      // reconstruct from private + shared (models.py:254-262), the q/k/v projection of the six tokens (models.py:243) and the
      // discriminator's first layer all read x6 only
      int n = 0;
      for (int i = 0; i < 3; ++i) {
        g[n] = sk_nt(B, hs, hs, WS(m->x6 + i * BH), hs, PP(m->rec_w + (int64_t)i * hs * hs), PP(m->rec_b + i * hs), WS(m->recon + i * BH), hs);
        g[n++].A2 = WS(m->x6 + (3 + i) * BH);
      }
      g[n++] = sk_nt(6 * B, 3 * hs, hs, WS(m->x6), hs, PP(m->in_w), PP(m->in_b), WS(m->qkv), 3 * hs);
      if (!c.use_cmd_sim) g[n++] = sk_nt(3 * B, hs, hs, WS(m->x6 + 3 * BH), hs, PP(m->d1_w), PP(m->d1_b), WS(m->dom_z), hs);
      sk_launch(x, g, n);
      if (!c.use_cmd_sim && !x.rc)
        {
        const mmda_act_params ap = act_params(m, training, seed, SITE_RRELU_DISC, false);
        x.rc = mmda_act_dropout_fwd_p(WS(m->dom_z), WS(m->dom_h), 3 * BH, c.act, &ap, p_cls, seed, SITE_DISC, stream);
      }
      if (!x.rc) x.rc = mmda_attn_fwd(WS(m->qkv), S6, B, hs, NHEAD, WS(m->ctx), WS(m->probs), p_tf, seed, SITE_ATTN, stream);
      n = 0;
      g[n++] = sk_nt(6 * B, hs, hs, WS(m->ctx), hs, PP(m->out_w), PP(m->out_b), WS(m->attn_out), hs);
      if (!c.use_cmd_sim) g[n++] = sk_nt(3 * B, 3, hs, WS(m->dom_h), hs, PP(m->d2_w), PP(m->d2_b), WS(m->dom), 3);
      sk_launch(x, g, n);
      if (!x.rc) x.rc = mmda_layernorm_fwd(&l1, stream);
    }
    // feed-forward pair: one launch split over the hidden units (fused_rows.hip), its partial products summed by the stretch behind it
    static const int ffn_fuse_on = getenv("MMDA_FFN_FUSE") ? atoi(getenv("MMDA_FFN_FUSE")) : 1;
    const bool ffn_fuse = row_fuse && ffn_fuse_on && !m->fusion_fp8 && (FFN % 32) == 0;
    if (m->fusion_fp8) {
      if (!x.rc) x.rc = ffn_fp8(m, p_tf, seed, stream);
    } else if (ffn_fuse) {
      if (!x.rc) {
        FusedFfnFwd f = {};
        f.M = 6 * B; f.hs = hs; f.F = FFN; f.S = 32; f.x1 = WS(m->x1); f.w1 = PP(m->l1_w); f.b1 = PP(m->l1_b); f.f1 = WS(m->f1);
        f.p = p_tf; f.seed = seed; f.site = SITE_FFN; f.w2 = PP(m->l2_w); f.parts = WS(m->ffn_parts);
        x.rc = mmda_fused_ffn_fwd(&f, stream);
      }
    } else {
      g[0] = sk_nt(6 * B, FFN, hs, WS(m->x1), hs, PP(m->l1_w), PP(m->l1_b), WS(m->f1), FFN, MMDA_ACT_RELU);
      g[0].drop_p = p_tf; g[0].drop_seed = seed; g[0].drop_site = SITE_FFN;
      sk_launch(x, g, 1);
      g[0] = sk_nt(6 * B, hs, FFN, WS(m->f1), FFN, PP(m->l2_w), PP(m->l2_b), WS(m->f2), hs);
      sk_launch(x, g, 1);
    }
    mmda_ln_args l2 = {};
    l2.rows = 6 * B; l2.n = hs; l2.x = WS(m->x1); l2.res = WS(m->f2); l2.gamma = PP(m->n2_w); l2.beta = PP(m->n2_b);
    l2.y = WS(m->hfused); l2.mean = WS(m->ln2_mean); l2.rstd = WS(m->ln2_rstd); l2.drop_p = p_tf; l2.drop_seed = seed;
    l2.drop_site = SITE_DROP2; l2.permute_S = S6; l2.permute_B = B; l2.eps = 1e-5f;   // emits h = cat(h[0..5], dim=1)
    if (row_fuse) {
      if (!x.rc) {
        FusedFwdC f = {};
        f.B = B; f.hs = hs; f.ncls = c.ncls; f.nb = fuse_nb; f.ln2 = l2;
        if (ffn_fuse) { f.ffn_parts = WS(m->ffn_parts); f.n_parts = FFN / 32; f.b2 = PP(m->l2_b); f.f2 = WS(m->f2); }
        f.hfused = WS(m->hfused); f.head_w = PP(m->head_w); f.head_b = PP(m->head_b); f.logits = WS(m->logits);
        f.threshold = c.threshold; f.tcp = WS(m->tcp); f.scores = WS(m->scores); f.labels = WS(m->labels);
        f.p_cls = p_cls; f.seed = seed; f.site_cls = SITE_CLS;
        if (m->seed_cls) { f.emo = m->emo_eager; f.d_scores = WS(m->d_scores); }
        x.rc = mmda_fused_fwd_c(&f, stream);
      }
    } else {
      if (!x.rc) x.rc = mmda_layernorm_fwd(&l2, stream);
      g[0] = sk_nt(B, NC, 6 * hs, WS(m->hfused), 6 * hs, PP(m->head_w), PP(m->head_b), WS(m->logits), NC);
      sk_launch(x, g, 1);
      if (!x.rc)
        x.rc = mmda_heads_fwd(WS(m->logits), B, c.ncls, c.threshold, WS(m->tcp), WS(m->scores), WS(m->labels), p_cls, seed, SITE_CLS,
                              stream);
    }
  } else {
    // ---- many rows: the tiled generic kernel
    for (int i = 0; i < 3 && !x.rc; ++i) {
      Mod& md = m->mod[i];
      lin_fwd(x, fmode, B, hs, 4 * md.H, WS(md.utt), PP(md.pw), PP(md.pb), WS(m->z + i * BH));
      if (x.rc) break;
      mmda_ln_args ln = {};
      ln.rows = B; ln.n = hs; ln.x = WS(m->z + i * BH); ln.gamma = PP(md.plw); ln.beta = PP(md.plb); ln.y = WS(m->orig + i * BH);
      ln.mean = WS(m->pmean + i * B); ln.rstd = WS(m->prstd + i * B); ln.act = c.act; ln.eps = 1e-5f;
      ln.actp = act_params(m, training, seed, SITE_RRELU + i, false);
      x.rc = mmda_layernorm_fwd(&ln, stream);
    }
    // private (three weights, batched) and shared (one weight over the stacked 3B rows), sigmoid epilogue
    gemm(x, fmode, 0, 1, B, hs, hs, WS(m->orig), hs, PP(m->priv_w), hs, WS(m->x6), hs, PP(m->priv_b), nullptr, 0, MMDA_ACT_SIGMOID, 3,
         BH, (int64_t)hs * hs, BH, hs);
    gemm(x, fmode, 0, 1, 3 * B, hs, hs, WS(m->orig), hs, PP(m->sh_w), hs, WS(m->x6 + 3 * BH), hs, PP(m->sh_b), nullptr, 0,
         MMDA_ACT_SIGMOID);
    m->seed_recon = m->seed_cls = 0;
    if (!x.rc) x.rc = eager_side_losses(m, stream, bfg && !inf);
    // reconstruct (models.py:254-262)
    if (!x.rc) x.rc = mmda_add(WS(m->x6), WS(m->x6 + 3 * BH), WS(m->rsum), 3 * BH, stream);
    gemm(x, fmode, 0, 1, B, hs, hs, WS(m->rsum), hs, PP(m->rec_w), hs, WS(m->recon), hs, PP(m->rec_b), nullptr, 0, 0, 3, BH,
         (int64_t)hs * hs, BH, hs);
    // adversarial discriminator behind the gradient-reversal layer (models.py:219-227); identity in forward
    if (!c.use_cmd_sim) {
      lin_fwd(x, fmode, 3 * B, hs, hs, WS(m->x6 + 3 * BH), PP(m->d1_w), PP(m->d1_b), WS(m->dom_z));
      if (!x.rc) {
      const mmda_act_params ap = act_params(m, training, seed, SITE_RRELU_DISC, false);
      x.rc = mmda_act_dropout_fwd_p(WS(m->dom_z), WS(m->dom_h), 3 * BH, c.act, &ap, p_cls, seed, SITE_DISC, stream);
    }
      lin_fwd(x, fmode, 3 * B, 3, hs, WS(m->dom_h), PP(m->d2_w), PP(m->d2_b), WS(m->dom));
    }
    // 1-layer transformer fusion over the six tokens (models.py:243-245; torch post-norm encoder layer)
    lin_fwd(x, fmode, 6 * B, 3 * hs, hs, WS(m->x6), PP(m->in_w), PP(m->in_b), WS(m->qkv));
    if (!x.rc) x.rc = mmda_attn_fwd(WS(m->qkv), S6, B, hs, NHEAD, WS(m->ctx), WS(m->probs), p_tf, seed, SITE_ATTN, stream);
    lin_fwd(x, fmode, 6 * B, hs, hs, WS(m->ctx), PP(m->out_w), PP(m->out_b), WS(m->attn_out));
    if (!x.rc) {
      mmda_ln_args ln = {};
      ln.rows = 6 * B; ln.n = hs; ln.x = WS(m->x6); ln.res = WS(m->attn_out); ln.gamma = PP(m->n1_w); ln.beta = PP(m->n1_b);
      ln.y = WS(m->x1); ln.mean = WS(m->ln1_mean); ln.rstd = WS(m->ln1_rstd); ln.drop_p = p_tf; ln.drop_seed = seed;
      ln.drop_site = SITE_DROP1; ln.eps = 1e-5f;
      x.rc = mmda_layernorm_fwd(&ln, stream);
    }
    if (m->fusion_fp8) {
      if (!x.rc) x.rc = ffn_fp8(m, p_tf, seed, stream);
    } else {
      mmda_gemm_args e = {};
      e.drop_p = p_tf; e.drop_seed = seed; e.drop_site = SITE_FFN;
      gemm(x, fmode, 0, 1, 6 * B, FFN, hs, WS(m->x1), hs, PP(m->l1_w), hs, WS(m->f1), FFN, PP(m->l1_b), nullptr, 0, MMDA_ACT_RELU, 1, 0,
           0, 0, 0, &e);
      lin_fwd(x, fmode, 6 * B, hs, FFN, WS(m->f1), PP(m->l2_w), PP(m->l2_b), WS(m->f2));
    }
    if (!x.rc) {
      mmda_ln_args ln = {};
      ln.rows = 6 * B; ln.n = hs; ln.x = WS(m->x1); ln.res = WS(m->f2); ln.gamma = PP(m->n2_w); ln.beta = PP(m->n2_b);
      ln.y = WS(m->hfused); ln.mean = WS(m->ln2_mean); ln.rstd = WS(m->ln2_rstd); ln.drop_p = p_tf; ln.drop_seed = seed;
      ln.drop_site = SITE_DROP2; ln.permute_S = S6; ln.permute_B = B; ln.eps = 1e-5f;   // emits h = cat(h[0..5], dim=1)
      x.rc = mmda_layernorm_fwd(&ln, stream);
    }
    // heads (models.py:247-249)
    lin_fwd(x, fmode, B, NC, 6 * hs, WS(m->hfused), PP(m->head_w), PP(m->head_b), WS(m->logits));
    if (!x.rc)
      x.rc = mmda_heads_fwd(WS(m->logits), B, c.ncls, c.threshold, WS(m->tcp), WS(m->scores), WS(m->labels), p_cls, seed, SITE_CLS,
                            stream);
  }
  if (!m->ev.empty()) { if (m->ev_seen_f % m->ev_stride == 0) m->ev_fwd++; m->ev_seen_f++; }
  if (!x.rc && m->wT_pending) {                // no fork came by (the eager losses are off and nothing else was pending)
    void* ss = nullptr;
    x.rc = side_fork(m, stream, &ss);
    if (!x.rc) x.rc = weight_transposes(m, ss);
  }
  if (!x.rc && m->flag_join_ok && m->seed_recon && m->seed_cls && m->side_pending && m->use_side && m->jflags) {
    x.rc = side_flag_signal(m, 0, m->fj1_armed != 0);      // (see mmda_misa::jflags: fused_bwd_a_kernel waits for the loss chain)
    m->fj1 = x.rc ? 0 : 1; m->fj1_armed = 0;
    return x.rc;
  }
  if (m->fj1_armed) return MMDA_ELAUNCH;       // (the CMD launch was armed on the same condition: cannot happen)
  if (!x.rc) x.rc = side_join(m, stream);      // (the side stream finished long ago: this only orders later work behind it)
  return x.rc;
}

// =============================================================================================== losses
extern "C" int mmda_misa_zero_act_grads(mmda_misa* m, void* stream) {
  if (check_ready(m)) return MMDA_EINVAL;
  if (hipMemsetAsync(WS(m->zero_begin), 0, sizeof(float) * (m->zero_end - m->zero_begin), (hipStream_t)stream) != hipSuccess)
    return MMDA_ELAUNCH;
  return MMDA_OK;
}

extern "C" int mmda_misa_losses(mmda_misa* m, const float* emo, int with_grads, void* stream) {
  if (check_ready(m) || !emo) return MMDA_EINVAL;
  const mmda_misa_config& c = m->cfg;
  const int B = m->B, hs = c.hidden;
  const int64_t BH = (int64_t)B * hs;
  hipStream_t s = (hipStream_t)stream;
  int rc = MMDA_OK;
  float* L = WS(m->losses);
  const bool ext = m->ext_batch_losses && with_grads && c.use_cmd_sim;      // the caller did (see mmda_misa::ext_batch_losses)
  const bool eager = (m->eager_done || ext) && with_grads;        // forward() already cleared the region and ran diff (+ CMD) on the side stream
  m->eager_done = 0;
  const bool s_recon = eager && m->seed_recon, s_cls = eager && m->seed_cls;      // seeds the forward stretches stored already
  m->seed_recon = m->seed_cls = 0;
  if (eager) {
    rc = side_join(m, stream);                           // (forward() joined already; kept for callers that split the calls)
    if (rc) return rc;
  } else {
    if (with_grads) { rc = mmda_misa_zero_act_grads(m, stream); if (rc) return rc; }      // covers the loss sums too
    else if (hipMemsetAsync(WS(m->losses), 0, sizeof(float) * 8, s) != hipSuccess) return MMDA_ELAUNCH;
    rc = mmda_loss_diff(WS(m->x6), BH, B, hs, c.diff_weight, L + 1, with_grads ? WS(m->d_x6) : nullptr, WS(m->diff_work), stream);
    if (rc) return rc;
  }
  if (c.use_cmd_sim) {
    if (!eager) rc = mmda_loss_cmd(WS(m->x6 + 3 * BH), BH, B, hs, c.sim_weight, L + 2, with_grads ? WS(m->d_x6 + 3 * BH) : nullptr, stream);
  } else {
    rc = mmda_loss_domain(WS(m->dom), B, c.sim_weight, L + 2, with_grads ? WS(m->d_dom) : nullptr, stream);
  }
  if (rc) return rc;
  // cls, conf (computed every step like solver.py:168; it only seeds gradients with use_confidNet, solver.py:180-181), recon and
  // the weighted total in one launch
  if (s_recon && s_cls && c.use_cmd_sim) {
    // no gradient left to seed: the launch only computes loss values -- issued by backward() on the side stream (train_step calls it next)
    m->misc_deferred = emo;
    return MMDA_OK;
  }
  return mmda_loss_misc(WS(m->scores), WS(m->tcp), emo, B, c.ncls, (with_grads && !s_cls) ? WS(m->d_scores) : nullptr,
                        (with_grads && !s_cls) ? WS(m->d_tcp) : nullptr, c.ncls == 6 && !ext, with_grads && c.use_confidNet && !ext, c.conf_weight,
                        WS(m->recon), WS(m->orig), 3 * BH, c.recon_weight, (with_grads && !s_recon) ? WS(m->d_recon) : nullptr,
                        (with_grads && !s_recon) ? WS(m->d_orig) : nullptr, L, c.diff_weight, c.sim_weight, c.recon_weight, c.conf_weight,
                        c.use_confidNet, stream);
}

// =============================================================================================== backward
extern "C" int mmda_misa_zero_grad(mmda_misa* m, void* stream) {
  if (!m || !m->G) return MMDA_EINVAL;
  if (hipMemsetAsync(m->G, 0, sizeof(float) * m->flat, (hipStream_t)stream) != hipSuccess) return MMDA_ELAUNCH;
  return MMDA_OK;
}

extern "C" int mmda_misa_backward(mmda_misa* m, const int64_t* t_ids, const float* v, const float* a, const int32_t* lengths,
                                  void* stream) {
  if (check_ready(m) || !m->G || !t_ids || !v || !a || !lengths) return MMDA_EINVAL;
  if (m->last_fwd_inference) return MMDA_EINVAL;        // the last forward was an evaluation pass: nothing was stashed
  m->early_valid = 0;
  const mmda_misa_config& c = m->cfg;
  const int B = m->B, T = m->T, hs = c.hidden, mode = c.mode, NC = 6 + c.ncls;
  const int fmode = MMDA_F32;       // fusion block: exact path (see mmda_misa_forward)
  const int R = T * B;
  const int64_t BH = (int64_t)B * hs;
  Ctx x{m, stream};
  const int training = m->training; const uint64_t seed = m->seed;
  const float p_tf = training ? c.fusion_dropout : 0.f, p_cls = training ? c.dropout : 0.f;

  // Weight-gradient GEMMs of the fusion block are collected and issued as one grouped launch on the side stream once
  // the dX chain (the critical path into the encoders) is through; their inputs are not modified afterwards.
  x.deferring = true;
  bool pg_pending = false;               // the fused stretches left LayerNorm parameter-gradient partials (mmda_fused_pg_finish)
  bool rec_hoisted = false;              // stretch C's launch made d_recon W_rec for stretch A
  if (B <= SKINNY_MAX_B) {
    // ---- few rows: the dX chain on row-skinny GEMMs (13 launches); weight gradients deferred exactly as below
    mmda_skinny_args g[8];
    const bool wt = m->wT_valid != 0;                   // K-major weight copies from this step's forward (side stream, joined there)
    // The row-local stretches as one launch each (fused_rows.hip): heads' sigmoid' -> d_hfused -> LayerNorm 2, and LayerNorm 1 ->
    // ... -> the projection LayerNorms.  MMDA_ROW_FUSE=0: the launches they replace (3 and 6).
    static const int row_fuse_on = getenv("MMDA_ROW_FUSE") ? atoi(getenv("MMDA_ROW_FUSE")) : 1;
    const bool row_fuse = row_fuse_on && wt && c.use_cmd_sim && hs == 128 && NHEAD == 2;
    if (m->fj1 && !row_fuse) { x.rc = flag_join_fallback(m, stream); m->fj1 = 0; if (x.rc) return x.rc; }     // (no kernel here waits on the device)
    static const int fuse_nb_env = getenv("MMDA_ROW_FUSE_NB") ? atoi(getenv("MMDA_ROW_FUSE_NB")) : 1;      // samples per workgroup (B=32: 0.722 ms with 2, 0.712 with 1; B=256 equal)
    const int fuse_nb = ((B % 2) == 0 && fuse_nb_env == 2) ? 2 : 1;
    mmda_ln_bwd_args l2a = {};
    l2a.rows = 6 * B; l2a.n = hs; l2a.dy = WS(m->d_hfused); l2a.x = WS(m->x1); l2a.res = WS(m->f2); l2a.gamma = PP(m->n2_w);
    l2a.mean = WS(m->ln2_mean); l2a.rstd = WS(m->ln2_rstd); l2a.d_x = WS(m->d_x1); l2a.d_res = WS(m->d_f2);
    l2a.dgamma = GG(m->n2_w); l2a.dbeta = GG(m->n2_b); l2a.drop_p = p_tf; l2a.drop_seed = seed; l2a.drop_site = SITE_DROP2;
    l2a.permute_S = S6; l2a.permute_B = B;
    if (row_fuse) {
      FusedBwdC f = {};
      f.B = B; f.hs = hs; f.ncls = c.ncls; f.nb = fuse_nb;
      f.tcp = WS(m->tcp); f.scores = WS(m->scores); f.d_tcp = WS(m->d_tcp); f.d_scores = WS(m->d_scores); f.d_logits = WS(m->d_logits);
      // flag join pending: d_tcp is all zeros (no ConfidNet gradients in that mode), but cleared by the side stream's chain, which
      // this launch does not wait for -- NULL reads as zero
      if (m->fj1) f.d_tcp = nullptr;
      // the reconstruction term of stretch A's d_x6 chain, in workgroups of this launch (small batches: both sets fit the chip twice
      // over; MMDA_FUSED_SPLIT=0: inside stretch A as before)
      static const int fsplit = getenv("MMDA_FUSED_SPLIT") ? atoi(getenv("MMDA_FUSED_SPLIT")) : 1;
      rec_hoisted = fsplit && m->rec_part >= 0 && ceil_div(B, fuse_nb) <= 64;
      if (rec_hoisted) { f.d_recon = WS(m->d_recon); f.rec_wT = WS(m->rec_wT); f.rec_part = WS(m->rec_part); }
      f.p_cls = p_cls; f.seed = seed; f.site_cls = SITE_CLS; f.head_w = PP(m->head_w); f.d_hfused = WS(m->d_hfused); f.ln2 = l2a;
      f.pg_parts = WS(m->pg_parts);
      x.rc = mmda_fused_bwd_c(&f, stream);
    } else {
      x.rc = mmda_heads_bwd(WS(m->tcp), WS(m->scores), WS(m->d_tcp), WS(m->d_scores), B, c.ncls, WS(m->d_logits), p_cls, seed, SITE_CLS,
                            stream);
      g[0] = wt ? sk_dx(B, NC, 6 * hs, WS(m->d_logits), NC, WS(m->head_wT), WS(m->d_hfused), 6 * hs, 0)
                : sk_nn(B, NC, 6 * hs, WS(m->d_logits), NC, PP(m->head_w), WS(m->d_hfused), 6 * hs, 0);
      sk_launch(x, g, 1);
      if (!x.rc) x.rc = mmda_layernorm_bwd(&l2a, stream);       // norm2
    }
    lin_dw(x, fmode, B, NC, 6 * hs, WS(m->d_logits), WS(m->hfused), GG(m->head_w), GG(m->head_b));
    // FFN
    // d f1 = (d f2 W2) * [f1 > 0] / (1-p): f1 is stored post-relu, post-dropout, so f1 > 0 <=> kept and pre-activation > 0
    static const int ffn_fuse_on = getenv("MMDA_FFN_FUSE") ? atoi(getenv("MMDA_FFN_FUSE")) : 1;
    const bool ffn_fuse = row_fuse && ffn_fuse_on && (FFN % 32) == 0;
    if (ffn_fuse) {
      if (!x.rc) {
        FusedFfnBwd f = {};
        f.M = 6 * B; f.hs = hs; f.F = FFN; f.S = 32; f.d_f2 = WS(m->d_f2); f.f1 = WS(m->f1);
        f.gate_scale = p_tf > 0.f ? 1.f / (1.f - p_tf) : 1.f; f.l2_wT = WS(m->l2_wT); f.d_f1 = WS(m->d_f1); f.l1_wT = WS(m->l1_wT);
        f.parts = WS(m->ffn_parts);
        x.rc = mmda_fused_ffn_bwd(&f, stream);
      }
    } else {
      g[0] = wt ? sk_dx(6 * B, hs, FFN, WS(m->d_f2), hs, WS(m->l2_wT), WS(m->d_f1), FFN, 0)
                : sk_nn(6 * B, hs, FFN, WS(m->d_f2), hs, PP(m->l2_w), WS(m->d_f1), FFN, 0);
      g[0].gate = WS(m->f1); g[0].ldgate = FFN; g[0].gate_scale = p_tf > 0.f ? 1.f / (1.f - p_tf) : 1.f;
      sk_launch(x, g, 1);
      g[0] = wt ? sk_dx(6 * B, FFN, hs, WS(m->d_f1), FFN, WS(m->l1_wT), WS(m->d_x1), hs, 1)
                : sk_nn(6 * B, FFN, hs, WS(m->d_f1), FFN, PP(m->l1_w), WS(m->d_x1), hs, 1);
      sk_launch(x, g, 1);
    }
    lin_dw(x, fmode, 6 * B, hs, FFN, WS(m->d_f2), WS(m->f1), GG(m->l2_w), GG(m->l2_b));
    lin_dw(x, fmode, 6 * B, FFN, hs, WS(m->d_f1), WS(m->x1), GG(m->l1_w), GG(m->l1_b));
    // norm1 + self-attention ... projection LayerNorms
    if (row_fuse && !x.rc) {
      FusedBwdA f = {};
      f.B = B; f.hs = hs; f.nhead = NHEAD; f.nb = fuse_nb;
      if (ffn_fuse) { f.ffn_parts = WS(m->ffn_parts); f.n_parts = FFN / 32; f.d_x1 = WS(m->d_x1); }
      mmda_ln_bwd_args& l = f.ln1;
      l.rows = 6 * B; l.n = hs; l.dy = WS(m->d_x1); l.x = WS(m->x6); l.res = WS(m->attn_out); l.gamma = PP(m->n1_w);
      l.mean = WS(m->ln1_mean); l.rstd = WS(m->ln1_rstd); l.d_x = WS(m->d_x6); l.accumulate_dx = 1; l.d_res = WS(m->d_attn_out);
      l.dgamma = GG(m->n1_w); l.dbeta = GG(m->n1_b); l.drop_p = p_tf; l.drop_seed = seed; l.drop_site = SITE_DROP1;
      f.d_attn_out = WS(m->d_attn_out); f.out_wT = WS(m->out_wT); f.d_ctx = WS(m->d_ctx);
      f.qkv = WS(m->qkv); f.probs = WS(m->probs); f.d_qkv = WS(m->d_qkv); f.p_tf = p_tf; f.seed = seed; f.site_attn = SITE_ATTN;
      f.in_wT = WS(m->in_wT); f.d_recon = WS(m->d_recon); f.rec_wT = WS(m->rec_wT); f.x6 = WS(m->x6); f.d_x6 = WS(m->d_x6);
      if (rec_hoisted) f.rec_part = WS(m->rec_part);
      f.priv_wT = WS(m->priv_wT); f.sh_wT = WS(m->sh_wT); f.d_orig = WS(m->d_orig);
      for (int i = 0; i < 3; ++i) {
        Mod& md = m->mod[i];
        mmda_ln_bwd_args& lp = f.lnp[i];
        lp.rows = B; lp.n = hs; lp.dy = WS(m->d_orig + i * BH); lp.x = WS(m->z + i * BH); lp.gamma = PP(md.plw);
        lp.mean = WS(m->pmean + i * B); lp.rstd = WS(m->prstd + i * B); lp.d_x = WS(m->d_z + i * BH);
        lp.dgamma = GG(md.plw); lp.dbeta = GG(md.plb); lp.act = c.act;
        lp.actp = act_params(m, training, seed, SITE_RRELU + i, true);
      }
      f.pg_parts = WS(m->pg_parts);
      if (m->fj1) {
        // Waiting workgroups hold their CU's LDS (104 KB each): with one on every CU the side stream's kernels could not start, and
        // the wait would never end -- on the device only while the stretch leaves most of the chip free; otherwise the event, here
        // (two launches later than the end of the forward pass, where it used to be: the loss chain is the longer one at large B)
        if (ceil_div(B, fuse_nb) <= 64) { f.wait_flag = m->jflags; f.wait_value = m->jval[0]; f.wait_err = m->jflags + 2; }
        else x.rc = flag_join_fallback(m, stream);
        m->fj1 = 0;
      }
      if (!x.rc) x.rc = mmda_fused_bwd_a(&f, stream);
      pg_pending = true;
      // the weight gradients of the stretch (deferred: one grouped launch on the side stream, as below)
      lin_dw(x, fmode, 6 * B, hs, hs, WS(m->d_attn_out), WS(m->ctx), GG(m->out_w), GG(m->out_b));
      lin_dw(x, fmode, 6 * B, 3 * hs, hs, WS(m->d_qkv), WS(m->x6), GG(m->in_w), GG(m->in_b));
      {
        mmda_gemm_args e = {};
        e.bias_grad = GG(m->rec_b);      // strideBias = hs: one bias gradient per batched problem
        gemm(x, fmode, 1, 0, hs, hs, B, WS(m->d_recon), hs, WS(m->rsum), hs, GG(m->rec_w), hs, nullptr, nullptr, 1, 0, 3, BH, BH, (int64_t)hs * hs,
             hs, &e);
      }
      {
        mmda_gemm_args e = {};
        e.bias_grad = GG(m->priv_b);
        gemm(x, fmode, 1, 0, hs, hs, B, WS(m->d_x6), hs, WS(m->orig), hs, GG(m->priv_w), hs, nullptr, nullptr, 1, 0, 3, BH, BH, (int64_t)hs * hs,
             hs, &e);
      }
      lin_dw(x, fmode, 3 * B, hs, hs, WS(m->d_x6 + 3 * BH), WS(m->orig), GG(m->sh_w), GG(m->sh_b));
    } else {
      // norm1 + self-attention
      if (!x.rc) {
        mmda_ln_bwd_args l = {};
        l.rows = 6 * B; l.n = hs; l.dy = WS(m->d_x1); l.x = WS(m->x6); l.res = WS(m->attn_out); l.gamma = PP(m->n1_w);
        l.mean = WS(m->ln1_mean); l.rstd = WS(m->ln1_rstd); l.d_x = WS(m->d_x6); l.accumulate_dx = 1; l.d_res = WS(m->d_attn_out);
        l.dgamma = GG(m->n1_w); l.dbeta = GG(m->n1_b); l.drop_p = p_tf; l.drop_seed = seed; l.drop_site = SITE_DROP1;
        x.rc = mmda_layernorm_bwd(&l, stream);
      }
      int n = 0;
      g[n++] = wt ? sk_dx(6 * B, hs, hs, WS(m->d_attn_out), hs, WS(m->out_wT), WS(m->d_ctx), hs, 0)
                  : sk_nn(6 * B, hs, hs, WS(m->d_attn_out), hs, PP(m->out_w), WS(m->d_ctx), hs, 0);
      if (!c.use_cmd_sim) g[n++] = wt ? sk_dx(3 * B, 3, hs, WS(m->d_dom), 3, WS(m->d2_wT), WS(m->d_dom_h), hs, 0)
                                      : sk_nn(3 * B, 3, hs, WS(m->d_dom), 3, PP(m->d2_w), WS(m->d_dom_h), hs, 0);
      sk_launch(x, g, n);
      lin_dw(x, fmode, 6 * B, hs, hs, WS(m->d_attn_out), WS(m->ctx), GG(m->out_w), GG(m->out_b));
      if (!x.rc) x.rc = mmda_attn_bwd(WS(m->qkv), WS(m->probs), WS(m->d_ctx), S6, B, hs, NHEAD, WS(m->d_qkv), p_tf, seed, SITE_ATTN, stream);
      lin_dw(x, fmode, 6 * B, 3 * hs, hs, WS(m->d_qkv), WS(m->x6), GG(m->in_w), GG(m->in_b));
      // adversarial branch: discriminator grads, then the REVERSED gradient into the shared codes (functions.py:17-21)
      if (!c.use_cmd_sim) {
        lin_dw(x, fmode, 3 * B, 3, hs, WS(m->d_dom), WS(m->dom_h), GG(m->d2_w), GG(m->d2_b));
        if (!x.rc) {
          const mmda_act_params ap = act_params(m, training, seed, SITE_RRELU_DISC, true);
          x.rc = mmda_act_dropout_bwd_p(WS(m->d_dom_h), WS(m->dom_z), WS(m->d_dom_z), 3 * BH, c.act, &ap, p_cls, seed, SITE_DISC, stream);
        }
        lin_dw(x, fmode, 3 * B, hs, hs, WS(m->d_dom_z), WS(m->x6 + 3 * BH), GG(m->d1_w), GG(m->d1_b));
        g[0] = wt ? sk_dx(3 * B, hs, hs, WS(m->d_dom_z), hs, WS(m->d1_wT), WS(m->d_x6 + 3 * BH), hs, 1)
                  : sk_nn(3 * B, hs, hs, WS(m->d_dom_z), hs, PP(m->d1_w), WS(m->d_x6 + 3 * BH), hs, 1);
        g[0].alpha = -c.reverse_grad_weight;
        sk_launch(x, g, 1);
      }
      // d_x6[token j] = (d_x6 + d_qkv[j] W_in + d_recon[j % 3] W_rec[j % 3]) * s (1 - s): the q/k/v projection's and the
      // reconstruction's input gradients (the latter flows into BOTH private and shared) and the sigmoid backward, six problems
      for (int j = 0; j < 6; ++j) {
        const int i = j % 3;
        g[j] = wt ? sk_dx(B, 3 * hs, hs, WS(m->d_qkv + (int64_t)j * B * 3 * hs), 3 * hs, WS(m->in_wT), WS(m->d_x6 + j * BH), hs, 1)
                  : sk_nn(B, 3 * hs, hs, WS(m->d_qkv + (int64_t)j * B * 3 * hs), 3 * hs, PP(m->in_w), WS(m->d_x6 + j * BH), hs, 1);
        g[j].K2 = hs; g[j].A_2nd = WS(m->d_recon + i * BH); g[j].lda_2nd = hs; g[j].ldb_2nd = hs;
        g[j].B_2nd = wt ? WS(m->rec_wT + (int64_t)i * hs * hs) : PP(m->rec_w + (int64_t)i * hs * hs);
        g[j].dsig = WS(m->x6 + j * BH); g[j].lddsig = hs;
      }
      sk_launch(x, g, 6);
      {
        mmda_gemm_args e = {};
        e.bias_grad = GG(m->rec_b);      // strideBias = hs: one bias gradient per batched problem
        gemm(x, fmode, 1, 0, hs, hs, B, WS(m->d_recon), hs, WS(m->rsum), hs, GG(m->rec_w), hs, nullptr, nullptr, 1, 0, 3, BH, BH, (int64_t)hs * hs,
             hs, &e);
      }
      // d_orig[i] += d_private[i] W_priv[i] + d_shared[i] W_shared
      for (int i = 0; i < 3; ++i) {
        g[i] = wt ? sk_dx(B, hs, hs, WS(m->d_x6 + i * BH), hs, WS(m->priv_wT + (int64_t)i * hs * hs), WS(m->d_orig + i * BH), hs, 1)
                  : sk_nn(B, hs, hs, WS(m->d_x6 + i * BH), hs, PP(m->priv_w + (int64_t)i * hs * hs), WS(m->d_orig + i * BH), hs, 1);
        g[i].K2 = hs; g[i].A_2nd = WS(m->d_x6 + (3 + i) * BH); g[i].lda_2nd = hs; g[i].B_2nd = wt ? WS(m->sh_wT) : PP(m->sh_w); g[i].ldb_2nd = hs;
      }
      sk_launch(x, g, 3);
      {
        mmda_gemm_args e = {};
        e.bias_grad = GG(m->priv_b);
        gemm(x, fmode, 1, 0, hs, hs, B, WS(m->d_x6), hs, WS(m->orig), hs, GG(m->priv_w), hs, nullptr, nullptr, 1, 0, 3, BH, BH, (int64_t)hs * hs,
             hs, &e);
      }
      lin_dw(x, fmode, 3 * B, hs, hs, WS(m->d_x6 + 3 * BH), WS(m->orig), GG(m->sh_w), GG(m->sh_b));
      // projections
      if (!x.rc) {
        mmda_ln_bwd_args l[3];
        for (int i = 0; i < 3; ++i) {
          Mod& md = m->mod[i];
          l[i] = mmda_ln_bwd_args{};
          l[i].rows = B; l[i].n = hs; l[i].dy = WS(m->d_orig + i * BH); l[i].x = WS(m->z + i * BH); l[i].gamma = PP(md.plw);
          l[i].mean = WS(m->pmean + i * B); l[i].rstd = WS(m->prstd + i * B); l[i].d_x = WS(m->d_z + i * BH);
          l[i].dgamma = GG(md.plw); l[i].dbeta = GG(md.plb); l[i].act = c.act;
          l[i].actp = act_params(m, training, seed, SITE_RRELU + i, true);
        }
        x.rc = mmda_layernorm_bwd_multi(l, 3, stream);
      }
    }
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i];
      g[i] = wt ? sk_dx(B, hs, 4 * md.H, WS(m->d_z + i * BH), hs, WS(m->pwT[i]), WS(md.d_utt), 4 * md.H, 0)
                : sk_nn(B, hs, 4 * md.H, WS(m->d_z + i * BH), hs, PP(md.pw), WS(md.d_utt), 4 * md.H, 0);
      lin_dw(x, fmode, B, hs, 4 * md.H, WS(m->d_z + i * BH), WS(md.utt), GG(md.pw), GG(md.pb));
    }
    sk_launch(x, g, 3);
  } else {
    if (m->fj1) { x.rc = flag_join_fallback(m, stream); m->fj1 = 0; if (x.rc) return x.rc; }      // (cannot happen: the seeds need the fused stretches)
    // heads
    x.rc = mmda_heads_bwd(WS(m->tcp), WS(m->scores), WS(m->d_tcp), WS(m->d_scores), B, c.ncls, WS(m->d_logits), p_cls, seed, SITE_CLS,
                          stream);
    lin_dx(x, fmode, B, NC, 6 * hs, WS(m->d_logits), PP(m->head_w), WS(m->d_hfused), 0);
    lin_dw(x, fmode, B, NC, 6 * hs, WS(m->d_logits), WS(m->hfused), GG(m->head_w), GG(m->head_b));
    // norm2 + FFN
    if (!x.rc) {
      mmda_ln_bwd_args l = {};
      l.rows = 6 * B; l.n = hs; l.dy = WS(m->d_hfused); l.x = WS(m->x1); l.res = WS(m->f2); l.gamma = PP(m->n2_w);
      l.mean = WS(m->ln2_mean); l.rstd = WS(m->ln2_rstd); l.d_x = WS(m->d_x1); l.d_res = WS(m->d_f2);
      l.dgamma = GG(m->n2_w); l.dbeta = GG(m->n2_b); l.drop_p = p_tf; l.drop_seed = seed; l.drop_site = SITE_DROP2;
      l.permute_S = S6; l.permute_B = B;
      x.rc = mmda_layernorm_bwd(&l, stream);
    }
    {
      // d f1 = (d f2 W2) * [f1 > 0] / (1-p): f1 is stored post-relu, post-dropout, so f1 > 0 <=> kept and pre-activation > 0
      mmda_gemm_args e = {};
      e.gate = WS(m->f1); e.ldgate = FFN; e.gate_scale = p_tf > 0.f ? 1.f / (1.f - p_tf) : 1.f;
      gemm(x, fmode, 0, 0, 6 * B, FFN, hs, WS(m->d_f2), hs, PP(m->l2_w), FFN, WS(m->d_f1), FFN, nullptr, nullptr, 0, 0, 1, 0, 0, 0, 0, &e);
    }
    lin_dw(x, fmode, 6 * B, hs, FFN, WS(m->d_f2), WS(m->f1), GG(m->l2_w), GG(m->l2_b));
    lin_dx(x, fmode, 6 * B, FFN, hs, WS(m->d_f1), PP(m->l1_w), WS(m->d_x1), 1);
    lin_dw(x, fmode, 6 * B, FFN, hs, WS(m->d_f1), WS(m->x1), GG(m->l1_w), GG(m->l1_b));
    // norm1 + self-attention
    if (!x.rc) {
      mmda_ln_bwd_args l = {};
      l.rows = 6 * B; l.n = hs; l.dy = WS(m->d_x1); l.x = WS(m->x6); l.res = WS(m->attn_out); l.gamma = PP(m->n1_w);
      l.mean = WS(m->ln1_mean); l.rstd = WS(m->ln1_rstd); l.d_x = WS(m->d_x6); l.accumulate_dx = 1; l.d_res = WS(m->d_attn_out);
      l.dgamma = GG(m->n1_w); l.dbeta = GG(m->n1_b); l.drop_p = p_tf; l.drop_seed = seed; l.drop_site = SITE_DROP1;
      x.rc = mmda_layernorm_bwd(&l, stream);
    }
    lin_dx(x, fmode, 6 * B, hs, hs, WS(m->d_attn_out), PP(m->out_w), WS(m->d_ctx), 0);
    lin_dw(x, fmode, 6 * B, hs, hs, WS(m->d_attn_out), WS(m->ctx), GG(m->out_w), GG(m->out_b));
    if (!x.rc) x.rc = mmda_attn_bwd(WS(m->qkv), WS(m->probs), WS(m->d_ctx), S6, B, hs, NHEAD, WS(m->d_qkv), p_tf, seed, SITE_ATTN, stream);
    lin_dx(x, fmode, 6 * B, 3 * hs, hs, WS(m->d_qkv), PP(m->in_w), WS(m->d_x6), 1);
    lin_dw(x, fmode, 6 * B, 3 * hs, hs, WS(m->d_qkv), WS(m->x6), GG(m->in_w), GG(m->in_b));
    // adversarial branch: discriminator grads, then the REVERSED gradient into the shared codes (functions.py:17-21)
    if (!c.use_cmd_sim) {
      lin_dx(x, fmode, 3 * B, 3, hs, WS(m->d_dom), PP(m->d2_w), WS(m->d_dom_h), 0);
      lin_dw(x, fmode, 3 * B, 3, hs, WS(m->d_dom), WS(m->dom_h), GG(m->d2_w), GG(m->d2_b));
      if (!x.rc) {
        const mmda_act_params ap = act_params(m, training, seed, SITE_RRELU_DISC, true);
        x.rc = mmda_act_dropout_bwd_p(WS(m->d_dom_h), WS(m->dom_z), WS(m->d_dom_z), 3 * BH, c.act, &ap, p_cls, seed, SITE_DISC, stream);
      }
      lin_dw(x, fmode, 3 * B, hs, hs, WS(m->d_dom_z), WS(m->x6 + 3 * BH), GG(m->d1_w), GG(m->d1_b));
      mmda_gemm_args e = {};
      e.alpha = -c.reverse_grad_weight;
      gemm(x, fmode, 0, 0, 3 * B, hs, hs, WS(m->d_dom_z), hs, PP(m->d1_w), hs, WS(m->d_x6 + 3 * BH), hs, nullptr, nullptr, 1, 0, 1, 0, 0, 0, 0, &e);
    }
    // reconstruct: d(private+shared) goes to both halves of d_x6
    gemm(x, fmode, 0, 0, B, hs, hs, WS(m->d_recon), hs, PP(m->rec_w), hs, WS(m->d_x6), hs, nullptr, nullptr, 1, 0, 3, BH, (int64_t)hs * hs, BH);
    gemm(x, fmode, 0, 0, B, hs, hs, WS(m->d_recon), hs, PP(m->rec_w), hs, WS(m->d_x6 + 3 * BH), hs, nullptr, nullptr, 1, 0, 3, BH,
         (int64_t)hs * hs, BH);
    {
      mmda_gemm_args e = {};
      e.bias_grad = GG(m->rec_b);      // strideBias = hs: one bias gradient per batched problem
      gemm(x, fmode, 1, 0, hs, hs, B, WS(m->d_recon), hs, WS(m->rsum), hs, GG(m->rec_w), hs, nullptr, nullptr, 1, 0, 3, BH, BH, (int64_t)hs * hs,
           hs, &e);
    }
    // sigmoid of private/shared
    if (!x.rc) x.rc = mmda_sigmoid_bwd_inplace(WS(m->d_x6), WS(m->x6), 6 * BH, stream);
    gemm(x, fmode, 0, 0, B, hs, hs, WS(m->d_x6), hs, PP(m->priv_w), hs, WS(m->d_orig), hs, nullptr, nullptr, 1, 0, 3, BH, (int64_t)hs * hs, BH);
    {
      mmda_gemm_args e = {};
      e.bias_grad = GG(m->priv_b);
      gemm(x, fmode, 1, 0, hs, hs, B, WS(m->d_x6), hs, WS(m->orig), hs, GG(m->priv_w), hs, nullptr, nullptr, 1, 0, 3, BH, BH, (int64_t)hs * hs,
           hs, &e);
    }
    lin_dx(x, fmode, 3 * B, hs, hs, WS(m->d_x6 + 3 * BH), PP(m->sh_w), WS(m->d_orig), 1);
    lin_dw(x, fmode, 3 * B, hs, hs, WS(m->d_x6 + 3 * BH), WS(m->orig), GG(m->sh_w), GG(m->sh_b));
    // projections
    for (int i = 0; i < 3 && !x.rc; ++i) {
      Mod& md = m->mod[i];
      mmda_ln_bwd_args l = {};
      l.rows = B; l.n = hs; l.dy = WS(m->d_orig + i * BH); l.x = WS(m->z + i * BH); l.gamma = PP(md.plw);
      l.mean = WS(m->pmean + i * B); l.rstd = WS(m->prstd + i * B); l.d_x = WS(m->d_z + i * BH);
      l.dgamma = GG(md.plw); l.dbeta = GG(md.plb); l.act = c.act;
      l.actp = act_params(m, training, seed, SITE_RRELU + i, true);
      x.rc = mmda_layernorm_bwd(&l, stream);
      lin_dx(x, fmode, B, hs, 4 * md.H, WS(m->d_z + i * BH), PP(md.pw), WS(md.d_utt), 0);
      lin_dw(x, fmode, B, hs, 4 * md.H, WS(m->d_z + i * BH), WS(md.utt), GG(md.pw), GG(md.pb));
    }
  }
  if (x.rc) return x.rc;
  x.deferring = false;
  {
    // side stream: private + shared (the reconstruction's input, needed only by its weight gradient) and every deferred
    // weight-gradient GEMM of the block
    void* ss = nullptr;
    x.rc = side_fork(m, stream, &ss);
    if (!x.rc && m->misc_deferred) {
      // loss values of the step (the forward stretches stored every gradient seed: see mmda_misa::emo_eager)
      x.rc = mmda_loss_misc(WS(m->scores), WS(m->tcp), m->misc_deferred, B, c.ncls, nullptr, nullptr, c.ncls == 6, 0, c.conf_weight,
                            WS(m->recon), WS(m->orig), 3 * BH, c.recon_weight, nullptr, nullptr, WS(m->losses), c.diff_weight,
                            c.sim_weight, c.recon_weight, c.conf_weight, c.use_confidNet, ss);
      m->misc_deferred = nullptr;
    }
    if (!x.rc && B <= SKINNY_MAX_B) x.rc = mmda_add(WS(m->x6), WS(m->x6 + 3 * BH), WS(m->rsum), 3 * BH, ss);
    // the sorted id list of the embedding scatter (see mmda_misa::esort); MMDA_SORT_EARLY=0: made where the scatter runs
    static const int sort_early = getenv("MMDA_SORT_EARLY") ? atoi(getenv("MMDA_SORT_EARLY")) : 1;
    m->esort_valid = 0;
    if (!x.rc && sort_early && T > 0 && m->esort >= 0 && mmda_embed_scatter_sorts(R)) {
      x.rc = mmda_embed_sort_ids(t_ids, R, lengths, B, c.vocab, reinterpret_cast<unsigned*>(WS(m->esort)), ss);
      if (!x.rc && ss != stream) {
        hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)ss, m->jflags + 3, ++m->esort_val);
        if (hipGetLastError() != hipSuccess) x.rc = MMDA_ELAUNCH;
      }
      m->esort_valid = x.rc ? 0 : (ss != stream ? 2 : 1);
    }
    if (!x.rc && pg_pending) {
      // gamma / beta gradients of the five LayerNorms the fused stretches walked: per-sample partials added in sample order
      float* dg[FUSED_PG_SLOTS] = {GG(m->n2_w), GG(m->n1_w), GG(m->mod[0].plw), GG(m->mod[1].plw), GG(m->mod[2].plw)};
      float* db[FUSED_PG_SLOTS] = {GG(m->n2_b), GG(m->n1_b), GG(m->mod[0].plb), GG(m->mod[1].plb), GG(m->mod[2].plb)};
      x.rc = mmda_fused_pg_finish(WS(m->pg_parts), B, hs, dg, db, ss);
    }
    if (!x.rc && !x.deferred.empty()) x.rc = mmda_gemm_grouped(x.deferred.data(), (int)x.deferred.size(), ss);
    x.deferred.clear();
    // the bf16 operand copies that only the weight-gradient GEMMs read (hseq of both layers, nt form: transposed layer-2 inputs): here,
    // where the side stream is idle beside the layer-2 recurrence, instead of in front of the loss kernels of the forward pass
    if (!x.rc && mode == MMDA_BF16 && m->use_bf16_gemm && T > 0) {
      mmda_convert_job cj[16];
      const int nj = backward_only_jobs(m, cj);
      x.rc = mmda_convert_bf16(cj, nj, ss);
    }
    // Experiment, OFF by default (MMDA_ADAM_EMBED_SPLIT=1): clip + Adam of the embedding rows this batch does NOT touch, here, beside
    // the layer-2 recurrence (the side stream is idle for ~100 us behind the GEMMs above).  Their gradient is zero whatever the rest of
    // the backward pass does (the bucket was cleared during the forward pass, the scatter at the end adds into the touched rows only),
    // and torch's dense Adam moves them by their momentum all the same: 6 of the 10.8 M parameters would leave the tail of the step
    // (-32 us).  Measured at B=32: the 170 MB it streams beside the recurrence slow that kernel's hand-offs by 24 us (0.071 -> 0.095 ms)
    // and the step ends up 14 us LONGER (0.769 -> 0.783 ms); behind the early optimizer pass beside the layer-1 recurrence instead, the
    // side stream becomes the longer of the two and the join waits (0.766 -> 0.834 ms).
    static const int embed_split = getenv("MMDA_ADAM_EMBED_SPLIT") ? atoi(getenv("MMDA_ADAM_EMBED_SPLIT")) : 0;
    if (!x.rc && m->adam_early_on && embed_split == 1 && m->use_side && m->M1 && m->V1 && m->embed + (int64_t)c.vocab * c.d_t == m->flat) {
      unsigned char* mask = reinterpret_cast<unsigned char*>(WS(m->touched));
      x.rc = mmda_mark_rows(mask, c.vocab, t_ids, R, ss);
      if (!x.rc) x.rc = mmda_clamp_adam_rows(PP(m->embed), GG(m->embed), m->M1 + m->embed, m->V1 + m->embed, c.vocab, c.d_t, mask, 0,
                                             m->ae_lr, 0.9f, 0.999f, 1e-8f, m->ae_clip, 1.0f, m->ae_step, ss);
      if (!x.rc) m->embed_early_done = 1;
    }
  }
  if (x.rc) return x.rc;
  // encoders, top layer first
  const float* xin[3] = {WS(m->mod[0].x), v, a};
  // Layer 2's weight-gradient GEMMs run beside the layer-1 recurrent kernel on the side stream (MMDA_DW_OVERLAP=0: held back and
  // issued with layer 1's in one grouped launch after it).  The wave-autonomous recurrence runs one wave on each of ~110 CUs,
  // raises its priority and reserves those CUs' whole LDS, so the GEMM's workgroups land on the other ~145 CUs; what the two still
  // share is L2 and fabric bandwidth (the recurrence slows by ~30 us, the GEMMs' ~65 us leave the critical path).
  static const int dw_overlap = getenv("MMDA_DW_OVERLAP") ? atoi(getenv("MMDA_DW_OVERLAP")) : 1;
  std::vector<mmda_gemm_bf16_args> bside;
  for (int l = 1; l >= 0; --l) {
    mmda_lstm_desc desc[3];
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i]; Rnn& r = md.rnn[l];
      desc[i] = mmda_lstm_desc{};
      desc[i].H = r.H; desc[i].gates = WS(md.gates[l]); desc[i].cstash = WS(md.c[l]); desc[i].hseq = WS(md.hseq[l]);
      desc[i].wpack[0] = WS(r.pack_b[0]); desc[i].wpack[1] = WS(r.pack_b[1]);
      desc[i].wpack_c[0] = m->packed_c_valid ? WS(r.pack_c[0]) : nullptr; desc[i].wpack_c[1] = m->packed_c_valid ? WS(r.pack_c[1]) : nullptr;
      desc[i].utt = WS(md.d_utt); desc[i].layer = l; desc[i].d_hseq = l == 0 ? WS(md.d_hseq1) : nullptr;
      desc[i].xchg = (m->use_cluster && md.xchg >= 0) ? (void*)WS(md.xchg) : nullptr; desc[i].epoch_base = m->epoch;
      desc[i].gate_minor = m->gate_minor; desc[i].cell = c.rnncell;
    }
    // The wave-autonomous gate-minor kernel emits the gate gradients as bf16 (the operand of the input-gradient GEMM) -- and only
    // as bf16: every consumer of dG in this mode is a bf16 GEMM (the transposed operand is re-laid-out from it).  The other
    // kernels (barrier form: H > 320, ablation switches) write fp32 `gates` as before.
    const bool kdg = mode == MMDA_BF16 && m->use_bf16_gemm && mmda_lstm_bwd_emits_dg_bf16(mode, 3, desc, B, T) != 0;
    if (kdg)
      for (int i = 0; i < 3; ++i) { desc[i].dg_bf16 = WS(m->mod[i].rnn[l].dgb); desc[i].dg_bf16_only = 1; }
    m->epoch += (unsigned)T + 2u;
    // the forward pass skipped the streaming backward packing because the resident-weights kernels were going to run: they must
    if (!m->pack_b_valid && !mmda_lstm_resident_applicable(mode, 3, desc, B, T, 1)) return MMDA_EINVAL;
    ev_rec(m, m->ev_bwd, l == 1 ? 2 : 3, 0, stream);
    x.rc = mmda_lstm_bwd(mode, 3, desc, B, T, lengths, stream);
    ev_rec(m, m->ev_bwd, l == 1 ? 2 : 3, 1, stream);
    if (x.rc) return x.rc;
    // All weight / input gradient GEMMs of this layer (three modalities) are independent.  Layer 2: d(normed) feeds the
    // next recurrent kernel (main stream); its weight gradients run on the side stream underneath that kernel.
    const bool bfg = mode == MMDA_BF16 && m->use_bf16_gemm;
    const bool bf_hh = bfg && (B % 8) == 0;        // the time-shifted views of dG^T / hseq^T start B elements into a row
    std::vector<mmda_gemm_bf16_args> bmain;
    // gate gradients -> bf16: transposed (A of every dW) and, where an input gradient is needed and the recurrent kernel did not
    // write it itself, plain (A of dX).  Layer 2 with the kernel-written plain copy: only the weight-gradient GEMMs (side stream)
    // read the transposed one, so the conversion goes to the side stream with them.
    mmda_convert_job dgj[3];
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i]; Rnn& r = md.rnn[l];
      const bool plain = (l == 1 || i == 0) && !kdg;
      dgj[i] = mmda_convert_job{WS(md.gates[l]), 8 * r.H, R, 8 * r.H, nullptr, plain ? WS(r.dgb) : nullptr, plain ? r.ldG : 0, WS(r.dgbT),
                                m->ldR};
      if (kdg) { dgj[i].src = WS(r.dgb); dgj[i].ld = r.ldG; dgj[i].src_bf16 = 1; }
    }
    const bool tn = bfg && ((m->tn_wgrad >> l) & 1);   // weight gradients straight from dG / inputs / hseq as they lie (no dG^T)
    if (tn && !kdg) return MMDA_EINVAL;                // (forward() set tn_wgrad only where the recurrent kernel writes bf16 dG)
    const bool dg_on_side = bfg && kdg && l == 1 && dw_overlap && m->use_side && !tn;
    if (bfg && !dg_on_side && !tn) {
      x.rc = mmda_convert_bf16(dgj, 3, stream);
      if (x.rc) return x.rc;
    }
    x.deferring = (l == 1);
    group_begin(x);
    for (int i = 0; i < 3; ++i) {
      Mod& md = m->mod[i]; Rnn& r = md.rnn[l];
      const int H = r.H, G8 = 8 * H;
      const float* dG = WS(md.gates[l]);
      const float* in = l == 0 ? xin[i] : WS(md.normed);
      std::vector<mmda_gemm_bf16_args>& wq = (l == 1) ? bside : bmain;      // weight gradients: side stream for layer 2
      const unsigned short* dgT = reinterpret_cast<const unsigned short*>(WS(r.dgbT));
      const unsigned short* hT = reinterpret_cast<const unsigned short*>(WS(r.hbT));
      // dW_ih (both directions stacked); db_ih = db_hh = column sums of dG ride along as a virtual ones-column
      if (bfg) {
        mmda_gemm_bf16_args g = {};
        g.M = G8; g.N = r.D; g.K = R; g.A = dgT; g.lda = m->ldR; g.B = WS(r.xbT); g.ldb = m->ldR; g.C = gW_ih(m, r); g.ldc = r.D;
        g.accumulate = 1; g.bias_grad = gB_ih(m, r); g.bias_grad2 = gB_hh(m, r); g.perm_m_H = m->gate_minor ? H : 0;
        if (tn) { g.tn = 1; g.A = WS(r.dgb); g.lda = r.ldG; g.B = WS(r.xb); g.ldb = r.ldD; }
        wq.push_back(g);
      } else {
        mmda_gemm_args e = {};
        e.bias_grad = gB_ih(m, r); e.bias_grad2 = gB_hh(m, r);
        gemm(x, mode, 1, 0, G8, r.D, R, dG, G8, in, r.D, gW_ih(m, r), r.D, nullptr, nullptr, 1, 0, 1, 0, 0, 0, 0, &e);
      }
      // dW_hh: forward direction pairs dG[t] with h[t-1]; reverse direction pairs dG[t] with h[t+1] (zero past len)
      if (T > 1) {
        if (bf_hh) {
          mmda_gemm_bf16_args g = {};
          g.M = 4 * H; g.N = H; g.K = (T - 1) * B; g.lda = m->ldR; g.ldb = m->ldR; g.ldc = H; g.accumulate = 1;
          g.perm_m_H = m->gate_minor ? H : 0;
          if (tn) {
            // rows of dG / hseq are (t, b): the forward direction pairs rows t B + b of dG with rows (t - 1) B + b of h, the reverse
            // direction rows t B + b with rows (t + 1) B + b
            const unsigned short* dg = reinterpret_cast<const unsigned short*>(WS(r.dgb));
            const unsigned short* h0 = reinterpret_cast<const unsigned short*>(WS(r.hbp[0]));
            const unsigned short* h1 = reinterpret_cast<const unsigned short*>(WS(r.hbp[1]));
            g.tn = 1; g.lda = r.ldG; g.ldb = r.ldH;
            g.A = dg + (int64_t)B * r.ldG; g.B = h0; g.C = gW_hh(m, r, 0);
            wq.push_back(g);
            g.A = dg + 4 * H; g.B = h1 + (int64_t)B * r.ldH; g.C = gW_hh(m, r, 1);
            wq.push_back(g);
          } else {
          g.A = dgT + B; g.B = hT; g.C = gW_hh(m, r, 0);
          wq.push_back(g);
          g.A = dgT + (int64_t)4 * H * m->ldR; g.B = hT + (int64_t)H * m->ldR + B; g.C = gW_hh(m, r, 1);
          wq.push_back(g);
          }
        } else {
          const float* hs_ = WS(md.hseq[l]);
          gemm(x, mode, 1, 0, 4 * H, H, (T - 1) * B, dG + (int64_t)B * G8, G8, hs_, 2 * H, gW_hh(m, r, 0), H, nullptr, nullptr, 1);
          gemm(x, mode, 1, 0, 4 * H, H, (T - 1) * B, dG + 4 * H, G8, hs_ + (int64_t)B * 2 * H + H, 2 * H, gW_hh(m, r, 1), H, nullptr,
               nullptr, 1);
        }
      }
      // d(normed) = dG W_ih (layer 2) / d(embedding rows) (text layer 1)
      if (l == 1 || i == 0) {
        float* dst = l == 1 ? WS(md.d_normed) : WS(md.d_x);
        if (bfg) {
          mmda_gemm_bf16_args g = {};
          g.M = R; g.N = r.D; g.K = G8; g.A = WS(r.dgb); g.lda = r.ldG; g.B = WS(r.wbT); g.ldb = r.ldG; g.C = dst; g.ldc = r.D;
          bmain.push_back(g);
        } else {
          gemm(x, mode, 0, 0, R, r.D, G8, dG, G8, rW_ih(m, r), r.D, dst, r.D);
        }
      }
    }
    group_end(x);
    x.deferring = false;
    if (l == 0 && !dw_overlap) { bmain.insert(bmain.end(), bside.begin(), bside.end()); bside.clear(); }
    if (!x.rc && !bmain.empty()) x.rc = mmda_gemm_bf16_grouped(bmain.data(), (int)bmain.size(), stream);
    if (l == 1 && !x.rc) {
      // the inter-layer LayerNorm backward gives d(hseq of layer 1): input gradients on the main stream (they feed the next
      // recurrent kernel); gamma/beta gradients and this layer's weight-gradient GEMMs on the side stream underneath it
      mmda_ln_bwd_args lb[3];
      for (int i = 0; i < 3; ++i) {
        Mod& md = m->mod[i];
        lb[i] = mmda_ln_bwd_args{};
        lb[i].rows = R; lb[i].n = 2 * md.H; lb[i].dy = WS(md.d_normed); lb[i].x = WS(md.hseq[0]); lb[i].gamma = PP(md.ln_w);
        lb[i].mean = WS(md.ln_mean); lb[i].rstd = WS(md.ln_rstd); lb[i].d_x = WS(md.d_hseq1);
      }
      // (the input-gradient launch goes out BEFORE the fork: it reads the LayerNorm weights, which the early optimizer step below may
      // update on the side stream -- the fork orders the side stream behind it)
      // 16-byte form (norm.hip): the d_x launch leaves the gamma / beta gradients as per-block partials on its way (it reads dy and x
      // anyway) and the side stream only adds them up -- instead of a second pass over dy and x there (100 us at B = 256)
      const bool ln_split = mmda_ln_bwd_parts_applies(lb, 3) &&
                            mmda_ln_parts_floats(lb, 3) <= (int64_t)512 * 2 * 2 * (m->mod[0].H + m->mod[1].H + m->mod[2].H);
      if (ln_split) {
        for (int i = 0; i < 3; ++i) { lb[i].dgamma = GG(m->mod[i].ln_w); lb[i].dbeta = GG(m->mod[i].ln_b); }
        x.rc = mmda_ln_bwd_parts(lb, 3, WS(m->ln_parts), stream);
      } else {
        x.rc = mmda_layernorm_bwd_multi(lb, 3, stream);
      }
      void* ss = nullptr;
      if (!x.rc) x.rc = side_fork(m, stream, &ss);
      for (int i = 0; i < 3; ++i) { lb[i].dgamma = GG(m->mod[i].ln_w); lb[i].dbeta = GG(m->mod[i].ln_b); if (!ln_split) lb[i].d_x = nullptr; }
      if (!x.rc && dg_on_side) x.rc = mmda_convert_bf16(dgj, 3, ss);
      if (!x.rc) x.rc = ln_split ? mmda_ln_parts_finish(lb, 3, WS(m->ln_parts), ss) : mmda_layernorm_param_grads(lb, 3, ss);
      if (!x.rc && !x.deferred.empty()) x.rc = mmda_gemm_grouped(x.deferred.data(), (int)x.deferred.size(), ss);
      const bool l2_early = !bside.empty() && dw_overlap;
      if (!x.rc && l2_early) { x.rc = mmda_gemm_bf16_grouped(bside.data(), (int)bside.size(), ss); bside.clear(); }
      x.deferred.clear();
      // Everything issued so far on either stream is final for the fusion block, the LayerNorms and -- when its weight-gradient
      // GEMMs just went out and no GRU re-layout follows -- layer 2: data-parallel ranks may start reducing that prefix now,
      // beside the layer-1 recurrence (mmda_misa_wait_early_grads).
      if (!x.rc) {
        if (!m->ev_early && hipEventCreateWithFlags(&m->ev_early, hipEventDisableTiming) != hipSuccess) x.rc = MMDA_ELAUNCH;
        if (!x.rc && hipEventRecord(m->ev_early, (hipStream_t)ss) != hipSuccess) x.rc = MMDA_ELAUNCH;
        m->early_floats = (l2_early && !is_gru(m) && mode == MMDA_BF16 && m->use_bf16_gemm) ? m->rnn1_begin : m->rnn2_begin;
        m->early_valid = 1;
        // Single-GPU fused step: clip + Adam of that prefix right here, beside the layer-1 recurrence (nothing issued after this
        // point reads those fp32 parameters: the remaining GEMMs of a bf16 step read bf16 copies made in the forward pass, and in
        // the other modes the prefix ends in front of the recurrent layers).  The side stream is a real second stream only when
        // use_side is on; otherwise this is simply the same work in front of the recurrence.
        if (!x.rc && m->adam_early_on && m->early_floats > 0 && m->M1 && m->V1) {
          x.rc = mmda_clamp_adam(m->P, m->G, m->M1, m->V1, m->early_floats, m->ae_lr, 0.9f, 0.999f, 1e-8f, m->ae_clip, 1.0f, m->ae_step, ss);
          if (!x.rc) m->adam_early_done = m->early_floats;
          // (MMDA_ADAM_EMBED_SPLIT=2: the untouched embedding rows behind it, i.e. beside the tail GEMMs rather than a recurrence)
          static const int embed_split2 = getenv("MMDA_ADAM_EMBED_SPLIT") ? atoi(getenv("MMDA_ADAM_EMBED_SPLIT")) : 0;
          if (!x.rc && embed_split2 == 2 && m->use_side && m->embed + (int64_t)c.vocab * c.d_t == m->flat) {
            unsigned char* mask = reinterpret_cast<unsigned char*>(WS(m->touched));
            x.rc = mmda_mark_rows(mask, c.vocab, t_ids, R, ss);
            if (!x.rc) x.rc = mmda_clamp_adam_rows(PP(m->embed), GG(m->embed), m->M1 + m->embed, m->V1 + m->embed, c.vocab, c.d_t, mask, 0,
                                                   m->ae_lr, 0.9f, 0.999f, 1e-8f, m->ae_clip, 1.0f, m->ae_step, ss);
            if (!x.rc) m->embed_early_done = 1;
          }
        }
      }
    } else if (!x.rc) {
      // text: gradient w.r.t. the embedding rows, scattered densely into embed.weight.grad (sparse=False)
      if (m->esort_valid) {
        if (m->esort_valid == 2) {                         // made on the side stream: its word, waited for by one wave
          hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, m->jflags + 3, m->esort_val, m->jflags + 2);
          if (hipGetLastError() != hipSuccess) x.rc = MMDA_ELAUNCH;
        }
        if (!x.rc)
          x.rc = mmda_embed_scatter_presorted(GG(m->embed), reinterpret_cast<const unsigned*>(WS(m->esort)), R, c.d_t, c.vocab,
                                              WS(m->mod[0].d_x), stream);
        m->esort_valid = 0;
      } else {
        x.rc = mmda_embed_scatter_add_masked(GG(m->embed), t_ids, R, c.d_t, WS(m->mod[0].d_x), lengths, B, stream);
      }
    }
    if (x.rc) return x.rc;
  }
  if (!m->ev.empty()) { if (m->ev_seen_b % m->ev_stride == 0) m->ev_bwd++; m->ev_seen_b++; }
  // every gradient is complete on `stream` when backward returns -- or, in a fused training step whose last optimizer launch can wait
  // on the device (see mmda_misa::jflags), when that launch completes
  // (only where every gradient the side stream computes lies in the prefix it also stepped -- the bf16 step, whose layer-2 weight
  //  gradients ran there in front of the early optimizer pass: the launch that waits READS the rest of the bucket before it waits)
  if (!x.rc && m->flag_join_ok && m->adam_early_on && !is_gru(m) && m->use_side && m->side_pending && m->jflags &&
      m->adam_early_done > 0 && m->adam_early_done == m->rnn1_begin) {
    x.rc = side_flag_signal(m, 1);
    m->fj2 = x.rc ? 0 : 1;
    return x.rc;
  }
  if (!x.rc) x.rc = side_join(m, stream);
  if (!x.rc && is_gru(m)) {                    // fold the four-slot weight gradients into the torch-layout gradient buffer
    mmda_gru_pad_job gj[MMDA_GRU_PAD_MAX];
    int n = gru_jobs(m, m->G, true, gj);
    x.rc = mmda_gru_unpad_grads(gj, n, stream);
  }
  return x.rc;
}

extern "C" int mmda_misa_timing_end(mmda_misa* m) {
  if (!m) return MMDA_EINVAL;
  for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
  m->ev.clear(); m->ev_steps = m->ev_fwd = m->ev_bwd = 0; m->ev_seen_f = m->ev_seen_b = 0;
  return MMDA_OK;
}
extern "C" int mmda_misa_timing_stride(mmda_misa* m, int stride) {
  if (!m || stride < 1) return MMDA_EINVAL;
  m->ev_stride = stride;
  return MMDA_OK;
}
extern "C" int mmda_misa_timing_begin(mmda_misa* m, int max_steps) {
  if (!m || max_steps <= 0 || max_steps > 4096) return MMDA_EINVAL;
  mmda_misa_timing_end(m);
  m->ev.resize((size_t)max_steps * 8);
  // (timing events only: without the system-scope fence -- cache write-back and invalidation -- a default event performs when it is
  //  recorded, which is what hipEventDisableSystemFence is for: a sampled step costs the run half as much)
  for (auto& e : m->ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return MMDA_ELAUNCH;
  m->ev_steps = max_steps;
  m->ev_done.assign((size_t)max_steps * 4, 0);
  return MMDA_OK;
}
extern "C" int mmda_misa_timing_rotate(mmda_misa* m, int on) {
  if (!m) return MMDA_EINVAL;
  m->ev_rotate = on ? 1 : 0;
  return MMDA_OK;
}
extern "C" int mmda_misa_timing_collect(mmda_misa* m, float mean_ms[4], int* steps) {
  if (!m || !mean_ms || m->ev.empty()) return MMDA_EINVAL;
  int n = m->ev_fwd < m->ev_bwd ? m->ev_fwd : m->ev_bwd;
  if (n > m->ev_steps) n = m->ev_steps;
  double acc[4] = {0, 0, 0, 0};
  int cnt[4] = {0, 0, 0, 0};
  for (int s = 0; s < n; ++s)
    for (int k = 0; k < 4; ++k) {
      if ((size_t)(s * 4 + k) >= m->ev_done.size() || !m->ev_done[s * 4 + k]) continue;      // (rotation: one launch per sampled step)
      float ms = 0.f;
      if (hipEventSynchronize(m->ev[(s * 4 + k) * 2 + 1]) != hipSuccess) return MMDA_ELAUNCH;
      if (hipEventElapsedTime(&ms, m->ev[(s * 4 + k) * 2], m->ev[(s * 4 + k) * 2 + 1]) != hipSuccess) return MMDA_ELAUNCH;
      acc[k] += ms; cnt[k]++;
    }
  int least = n;
  for (int k = 0; k < 4; ++k) { mean_ms[k] = cnt[k] > 0 ? (float)(acc[k] / cnt[k]) : 0.f; least = cnt[k] < least ? cnt[k] : least; }
  if (steps) *steps = least;                            // samples behind every mean (the least-sampled launch)
  return MMDA_OK;
}

// =============================================================================================== optimizer / step
extern "C" int mmda_misa_adam_step(mmda_misa* m, float lr, float clip, float grad_scale, int step, void* stream) {
  if (!m || !m->P || !m->G || !m->M1 || !m->V1) return MMDA_EINVAL;
  return mmda_clamp_adam(m->P, m->G, m->M1, m->V1, m->flat, lr, 0.9f, 0.999f, 1e-8f, clip, grad_scale, step, stream);
}

extern "C" int mmda_misa_train_step(mmda_misa* m, const int64_t* t_ids, const float* v, const float* a, const int32_t* lengths,
                                    const float* emo, int training, uint64_t seed, int do_adam, float lr, float clip, int step,
                                    void* stream) {
  // the gradient bucket is cleared on the side stream beside the forward pass's fusion block (not at the start of the step: the
  // side stream's first job there, packing W_hh, is what the first recurrent kernel waits for)
  if (check_ready(m) || !m->G) return MMDA_EINVAL;
  m->inference = 0;                                     // a training step always stashes
  m->zero_grad_pending = m->T > 0 ? 1 : 0;
  m->eager_losses = 1; m->eager_done = 0;
  m->emo_eager = emo; m->misc_deferred = nullptr;
  static const int flag_join_on = getenv("MMDA_FLAG_JOIN") ? atoi(getenv("MMDA_FLAG_JOIN")) : 1;
  m->flag_join_ok = (flag_join_on && m->use_side) ? 1 : 0; m->fj1 = m->fj2 = 0;
  int rc = m->zero_grad_pending ? MMDA_OK : mmda_misa_zero_grad(m, stream);
  if (rc) return rc;
  rc = mmda_misa_forward(m, t_ids, v, a, lengths, training, seed, stream);
  m->eager_losses = 0; m->emo_eager = nullptr;
  if (rc) return rc;
  if (m->zero_grad_pending) return MMDA_ELAUNCH;        // forward() always reaches its fusion block
  rc = mmda_misa_losses(m, emo, 1, stream);
  if (rc) return rc;
  static const int adam_split = getenv("MMDA_ADAM_SPLIT") ? atoi(getenv("MMDA_ADAM_SPLIT")) : 1;     // 0: ablation (one launch at the end)
  m->adam_early_on = (do_adam && adam_split) ? 1 : 0; m->ae_lr = lr; m->ae_clip = clip; m->ae_step = step; m->adam_early_done = 0;
  m->embed_early_done = 0;
  rc = mmda_misa_backward(m, t_ids, v, a, lengths, stream);
  m->adam_early_on = 0; m->flag_join_ok = 0;
  if (rc) return rc;
  if (m->fj1) { rc = flag_join_fallback(m, stream); m->fj1 = 0; if (rc) return rc; }      // (no stretch took it over: cannot happen)
  if (m->fj2 && !do_adam) { rc = flag_join_fallback(m, stream); m->fj2 = 0; if (rc) return rc; }
  if (do_adam) {
    // the rest of the bucket (layer-1 recurrent layers, embedding -- or everything, if the backward pass stepped nothing early)
    const int64_t o = m->adam_early_done;
    const int64_t end = m->embed_early_done ? m->embed : m->flat;
    // (flag join: this launch does not complete before the side stream's weight-gradient GEMMs and early optimizer pass have)
    const bool fj = m->fj2 != 0;
    m->fj2 = 0;
    rc = mmda_clamp_adam_wait(m->P + o, m->G + o, m->M1 + o, m->V1 + o, end - o, lr, 0.9f, 0.999f, 1e-8f, clip, 1.0f, step,
                              fj ? m->jflags + 1 : nullptr, m->jval[1], fj ? m->jflags + 2 : nullptr, stream);
    if (!rc && m->embed_early_done)          // the rows of this batch (their gradient has just been scattered)
      rc = mmda_clamp_adam_rows(m->P + m->embed, m->G + m->embed, m->M1 + m->embed, m->V1 + m->embed, m->cfg.vocab, m->cfg.d_t,
                                reinterpret_cast<const unsigned char*>(m->ws + m->touched), 1, lr, 0.9f, 0.999f, 1e-8f, clip, 1.0f, step, stream);
  }
  return rc;
}
