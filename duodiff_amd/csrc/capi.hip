// extern "C" surface of libduodiff.so (include/duodiff.h): context, model packing, the U-ViT
// forward as a sequence of HIP kernels, the fused sampling step and the hipGraph-replayed loop.
//
// Host-side restatement of: UViT.__init__ / forward wiring (reference models/uvit.py:228-383),
// get_samples DDPM branch and backbone switch (sampler.py:128-139), schedule constants
// (sampler.py:40-44, ddpm_core.py:64-70).
#include "../../include/duodiff.h"
#include "../../include/duodiff_dev.h"
#include "dd_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace dd;

namespace dd {
hipError_t init_gemm_kernels();
hipError_t init_attention_kernels();
hipError_t init_rowops_kernels();
int device_num_cus();
}  // namespace dd

// ------------------------------------------------------------------------------------------
struct dd_ctx {
    int device = 0;
    std::string err;
    StepState* st = nullptr;     // device
    StepState* st2 = nullptr;    // device: the second half-batch chain of dd_sample (its own timestep / counter)
    hipStream_t side = nullptr;  // that chain's stream (context-owned, non-blocking)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_ee_fork = nullptr, ev_ee_join = nullptr;   // early-exit heads / probes of a layer on the side stream, beside the block's attention launch
    StepCoef* coef = nullptr;    // device [1000]
    StepCoef coef_host[1000];
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    float timing[3] = {0, 0, 0};
    int num_cus = 256;           // CU count the persistent GEMM grids are sized for (dd_set_num_cus), a multiple of 8
    int base_cus = 256;          // ... as set; num_cus itself is halved while a chained call captures the graphs of a large GEMM-path batch
    // dd_sample's graphs run on context-owned staging copies of x / y, so a captured step does not depend on the caller's
    // tensor addresses (reference get_samples allocates a fresh x per call: the graphs would be re-captured every time)
    float* x_stage = nullptr;
    int64_t* y_stage = nullptr;
    size_t x_stage_elems = 0, y_stage_elems = 0;
    long long graph_captures = 0;
    int last_chains = 1;         // chains the last dd_sample call ran (dd_dev_last_sample_chains)
    // dd_sample_affine: the step table on the device (+ its host staging copy, which must outlive the async upload)
    AffineRow* atab = nullptr;
    size_t atab_rows = 0;
    std::vector<AffineRow> atab_host;
    bool ee_inline = false;      // early-exit heads / probes stay on the launch stream (set while the two chains of dd_sample_early_exit are enqueued: `side` is a chain's stream then)
    int prof_kind = 0;           // dd_profile_select: which launches dd_profile_steps brackets (DD_PROF_*)
    unsigned dev_flags = 0;      // dd_dev_set_flags (include/duodiff_dev.h): kernel-variant switches of the development harness
};

namespace {

struct HostParam {
    std::vector<float> data;
    std::vector<int64_t> shape;
    bool set = false;
};

struct BlockW {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *proj_b, *fc1_b, *fc2_b, *skip_b, *qkv_b;
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w, *skip_w;
    const char* mlp_img;     // fused-MLP weight image (mlp_fused.hip) or null
    const float* mlp_b1p;    // fc1 bias in accumulator-register order
    const bf16_t* qa_img;    // attn.qkv weight per head in fragment order (qkv_attention_pack) or null
    const char* rl_img = nullptr;   // mlp.fc2 weight as the row-resident launch streams it (rowlin_pack) or null
    const char* rlp_img = nullptr;  // attn.proj weight, ditto
    const char* rls_img = nullptr;  // skip_linear weight [D, 2 D], ditto (out-blocks)
};

// head_dec_kernel operands from a head's LayerNorm (gamma, beta) and decoder_pred (W [pd, D], b): wg = W diag(gamma); dc = c [pd] = b + W . beta,
// then the row sums of wg [pd] (the kernel multiplies the un-normalised rows: dec = rstd (wg . d - mean_d wsum) + c, rowops.hip)
static void fold_head_norm(int D, int pd, const float* wd, const float* bd, const float* ng, const float* nb, std::vector<float>& wg, std::vector<float>& dc) {
    wg.assign((size_t)pd * D, 0.f);
    dc.assign(2 * (size_t)pd, 0.f);
    for (int r = 0; r < pd; ++r) {
        double acc = bd[r], wsum = 0.0;
        for (int k = 0; k < D; ++k) {
            wg[(size_t)r * D + k] = wd[(size_t)r * D + k] * ng[k];
            acc += (double)wd[(size_t)r * D + k] * (double)nb[k];
            wsum += (double)wg[(size_t)r * D + k];
        }
        dc[r] = (float)acc;
        dc[pd + r] = (float)wsum;
    }
}

struct HeadW { const float *ng, *nb, *wdec, *bdec, *wconv, *bconv; const float *wg = nullptr, *dc = nullptr; const float *wsplit = nullptr, *dcs = nullptr; };   // wg / dc: head_dec_kernel operands (norm folded into decoder_pred; dc = c [pd], row sums of wg [pd]) or null; wsplit / dcs: the same for the split-bf16 product (bf16 engine, embed_dim 256 / 512) or null

struct GraphKey {
    const void* x; const void* y; int B, noise, variance, num_cus;   // num_cus: the captured persistent grids are sized from it
    const void* atab;                                                // dd_sample_affine's table (null: the DDPM update)
    const void *aux0 = nullptr, *aux1 = nullptr;                     // dd_sample_early_exit: the two log tables
    float thr = 0.f;                                                 //                        and the threshold
    int b0 = 0;                                                      // first image of a half-batch chain within the whole batch
    bool operator==(const GraphKey& o) const {
        return x == o.x && y == o.y && B == o.B && noise == o.noise && variance == o.variance && num_cus == o.num_cus &&
               atab == o.atab && aux0 == o.aux0 && aux1 == o.aux1 && thr == o.thr && b0 == o.b0;
    }
};

}  // namespace

// The activation workspace of one chain: dd_sample runs a large batch as TWO independent half-batch chains on two streams (one
// chain's HBM-bound kernel phases then run under the other's MFMA phases); the second chain has its own copy of every buffer.
struct WsPtrs {
    float* x = nullptr; void *h = nullptr, *ao = nullptr, *qkv = nullptr, *hid = nullptr, *xb = nullptr;
    std::vector<void*> skips;
    float* dec = nullptr; float* mlp_partial = nullptr; bf16_t* qkv_dump = nullptr; bf16_t* hfrag = nullptr; float* ytap = nullptr;
};
struct WsOffsets { size_t x, h, ao, qkv, hid, xb, dec, part, dump, hf, bytes; std::vector<size_t> sk; bool has_part, has_dump, has_hf; size_t part_bytes = 0; size_t tap = 0; bool has_tap = false; };

struct dd_model {
    dd_ctx* ctx = nullptr;
    dd_config cfg{};
    int D = 0, L = 0, N = 0, extras = 0, pd = 0, pdp = 0, H = 0, hidden = 0, hid_ld = 0, half_depth = 0, Mp_max = 0;
    std::map<std::string, HostParam> params;
    bool finalized = false;
    int prec = DD_PREC_BF16;
    size_t esize = 2;
    char* warena = nullptr;    // weights
    char* wsarena = nullptr;   // activations
    std::vector<BlockW> blocks;  // in.., mid, out..
    const float *emb_wt = nullptr, *emb_b = nullptr, *pos = nullptr, *label = nullptr;
    const float *tm_w1t = nullptr, *tm_b1 = nullptr, *tm_w2t = nullptr, *tm_b2 = nullptr;   // time_embed MLP (mlp_time_embed)
    const float *norm_g = nullptr, *norm_b = nullptr, *wdec = nullptr, *bdec = nullptr, *wconv = nullptr, *bconv = nullptr;
    const float *wdec_g = nullptr, *dec_c = nullptr;   // head_dec_kernel operands (decoder weight * norm gamma; bias + W . beta [pd], then the row sums of wdec_g [pd]) or null
    float* x = nullptr; void* h = nullptr; void* ao = nullptr; void* qkv = nullptr; void* hid = nullptr; void* xb = nullptr;
    std::vector<void*> skips;
    float* dec = nullptr;
    // early-exit baseline (models/early_exit.py:193-268): per-layer output heads + MLP probes; ee_type < 0: plain U-ViT
    int ee_type = -1, n_probe = 0;
    std::vector<HeadW> heads;             // head i is applied to the input of block i
    const float *probe_w = nullptr, *probe_b = nullptr;   // [n_probe, D], [n_probe]
    std::vector<AttnProbeW> attn_probes;                  // DD_EE_ATTENTION_PROBE: one per layer
    bool ee_conv_stride_ok = false;                       // the heads' conv weights / biases sit at a constant stride in the weight arena (one batched conv launch)
    long long ee_wconv_stride = 0, ee_bconv_stride = 0;
    bool fused_mlp = false;               // bf16 mode, D in {64,128,256,512}: fc1+GELU+fc2+residual in one launch
    bool fused_proj = false;              // ... and attn.proj + residual in front of it (D % 128 == 0): patch rows only
    bool fused_skip = false;              // ... and the NEXT block's skip_linear + norm1 behind it (mid / out blocks; early-exit models tap y on the way,
                                          //     whose heads read every block's output)
    bool fused_qkv = false;               // ... and the NEXT block's attn.qkv Linear last of all (no qkv bias; not for early-exit models)
    bool splitk = false;                  // GEMM-path models whose N = embed_dim Linears have too few 256 x 256 tiles at max_batch: split-K + reduce_ln launches
    bool rowlin_skip = false;             // ... and the out-blocks' skip_linear + norm1
    bool rowlin_proj = false;             // ... and attn.proj + residual + norm2 likewise
    bool rowlin_fc2 = false;              // embed_dim 768 on the GEMM path: mlp.fc2 + residual + the next block's norm1 in one row-resident launch (rowlin.hip)
    bool fused_qa = false;                // attn.qkv computed inside the attention launch (attention.hip qkv_attention_kernel): takes precedence over
                                          // fused_qkv wherever the previous block's fused launch leaves norm1 in h
    bf16_t* qkv_dump = nullptr;           // scratch for the qkv stores of rows past the end of a ragged tile
    bf16_t* hfrag = nullptr;              // fused_qa: norm1 of the patch rows in MFMA fragment order (MlpFusedArgs::ln_out_frag)
    float* ytap = nullptr;                // early-exit models with fused_skip: the block output y of the launches that run the next skip_linear (MlpFusedArgs::y_tap)
    float* mlp_partial = nullptr;         // partial slabs of hidden-split leftover tiles (mlp_fused_plan)
    size_t mlp_partial_bytes = 0;
    hipGraphExec_t graph[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [0] DDPM step (dd_sample), [1] table-driven step (dd_sample_affine),
    GraphKey gkey[6]{};                                      // [2] early-exit step (dd_sample_early_exit), [3] / [4] / [5]: [0] / [1] / [2] of the second chain
    WsOffsets wsoff{}, wsoff2{};                             // layout of the main workspace (max_batch) and of the second chain's (half of it)
    char* wsarena2 = nullptr;                                // the second chain's workspace (allocated by the first chained dd_sample)
    WsPtrs ws2;
    float* ee_ws = nullptr;                                  // dd_sample_early_exit scratch: eps | model_output | cls | outs (two chains: one such block per chain, half the batch each)
    float* ee_sums = nullptr;                                // ... and the chains' per-step sums of the predicted errors [2][1000][depth]
    size_t ee_ws_elems = 0;
    // in-context timing (dd_profile_steps): event pairs recorded around each launch of kind ctx->prof_kind when enabled
    bool time_fc1 = false;
    std::vector<hipEvent_t> fc1_events;   // pairs, grown on demand
    size_t fc1_used = 0;
};

namespace dd {
// shared with vae.hip (dd_ctx is defined in this translation unit)
int ctx_fail(dd_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int ctx_device(dd_ctx* c) { return c->device; }
int ctx_num_cus(dd_ctx* c) { return c->num_cus; }
}  // namespace dd

namespace {

int fail(dd_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int fail_hip(dd_ctx* c, hipError_t e, const char* what) {
    return fail(c, DD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define DD_HIP(c, expr)                                          \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return fail_hip((c), _e, #expr);   \
    } while (0)

// ---- schedule: bit-for-bit the fp32 tables of sampler.py:40-44 / ddpm_core.py:64-70 (see oracle/schedule_oracle.py)
// betas = torch.linspace(beta_init, beta_final, n): ATen rounds once per element, forward from start in the first half,
// backward from end in the second; alphas_bar = torch.cumprod: the CPU scan accumulates in double, rounds each output.
void base_tables(float beta_init, float beta_final, int n, float* betas, float* alphas, float* abar, float* abar_prev) {
    const float start = beta_init, end = beta_final;
    const float stepf = n > 1 ? (end - start) / (float)(n - 1) : 0.0f;
    const double step = (double)stepf;
    const int half = n / 2;
    for (int i = 0; i < n; ++i) {
        betas[i] = i < half ? (float)((double)start + step * i) : (float)((double)end - step * (n - 1 - i));
        alphas[i] = 1.0f - betas[i];
    }
    double run = 1.0;
    for (int i = 0; i < n; ++i) {
        run *= (double)alphas[i];
        abar[i] = (float)run;
    }
    for (int i = 0; i < n; ++i) abar_prev[i] = i ? abar[i - 1] : 1.0f;
}
float bt_sampler_order(float beta, float abar_prev, float abar) {   // sampler.py:44: betas * (1 - abar_prev) / (1 - abar)
    volatile float num = beta * (1.0f - abar_prev);
    return num / (1.0f - abar);
}
float bt_scheduler_order(float beta, float abar_prev, float abar) {   // ddpm_core.py:68-70: (1 - abar_prev) / (1 - abar) * betas
    volatile float ratio = (1.0f - abar_prev) / (1.0f - abar);
    return ratio * beta;
}

struct Schedule {
    float betas[1000], alphas[1000], abar[1000], abar_prev[1000], bt_sampler[1000], bt_sched[1000];
    float c1[1000], c2[1000], sigma[1000], sigma_beta[1000];
    Schedule() {
        base_tables(1e-4f, 0.02f, 1000, betas, alphas, abar, abar_prev);
        for (int i = 0; i < 1000; ++i) {
            bt_sampler[i] = bt_sampler_order(betas[i], abar_prev[i], abar[i]);
            bt_sched[i] = bt_scheduler_order(betas[i], abar_prev[i], abar[i]);
            c1[i] = sqrtf(1.0f / alphas[i]);
            c2[i] = (1.0f - alphas[i]) / sqrtf(1.0f - abar[i]);
            sigma[i] = sqrtf(bt_sampler[i]);
            sigma_beta[i] = sqrtf(betas[i]);
        }
    }
};
const Schedule& schedule() {
    static const Schedule s;
    return s;
}

unsigned short host_f2bf(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ModuleDict key of EarlyExitUViT.matrix -> row of the probe table (early_exit.py:194-204, 219-240); -1: not a key
int probe_index(const dd_model* m, const std::string& key) {
    auto num = [](const std::string& v, int& out) {
        if (v.empty() || v.size() > 6) return false;
        for (char ch : v) if (ch < '0' || ch > '9') return false;
        out = std::atoi(v.c_str());
        return true;
    };
    int i = 0, t = 0;
    switch (m->ee_type) {
        case DD_EE_MLP_PER_LAYER:
        case DD_EE_ATTENTION_PROBE: return num(key, i) && i < m->cfg.depth ? i : -1;
        case DD_EE_MLP_PER_TIMESTEP: return num(key, t) && t < 1000 ? t : -1;
        case DD_EE_MLP_PER_LAYER_PER_TIMESTEP: {
            const size_t cpos = key.find(", ");
            if (cpos == std::string::npos) return -1;
            if (!num(key.substr(0, cpos), i) || !num(key.substr(cpos + 2), t) || i >= m->cfg.depth || t >= 1000) return -1;
            return t * m->cfg.depth + i;
        }
    }
    return -1;
}
std::string probe_key(const dd_model* m, int layer, int t) {
    switch (m->ee_type) {
        case DD_EE_MLP_PER_LAYER:
        case DD_EE_ATTENTION_PROBE: return std::to_string(layer);
        case DD_EE_MLP_PER_TIMESTEP: return std::to_string(t);
        default: return std::to_string(layer) + ", " + std::to_string(t);
    }
}
std::string head_prefix(const dd_model* m, int layer) {
    if (layer < m->half_depth) return "in_blocks_heads." + std::to_string(layer) + ".";
    if (layer == m->half_depth) return "mid_block_head.";
    return "out_blocks_heads." + std::to_string(layer - m->half_depth - 1) + ".";
}
// "in_blocks_heads.<i>.<rest>" / "mid_block_head.<rest>" / "out_blocks_heads.<i>.<rest>" -> layer index, or -1
int head_index(const dd_model* m, const std::string& name, std::string& rest) {
    for (int layer = 0; layer < m->cfg.depth; ++layer) {
        const std::string pre = head_prefix(m, layer);
        if (name.compare(0, pre.size(), pre) == 0) { rest = name.substr(pre.size()); return layer; }
    }
    return -1;
}

// expected shapes by reference state_dict name (models/uvit.py:228-336)
bool expected_shape(const dd_model* m, const std::string& name, std::vector<int64_t>& shp) {
    const int64_t D = m->D, C = m->cfg.in_chans, P = m->cfg.patch_size, hid = m->hidden;
    auto is = [&](const char* s) { return name == s; };
    if (is("pos_embed")) { shp = {1, m->L, D}; return true; }
    if (is("patch_embed.proj.weight")) { shp = {D, C, P, P}; return true; }
    if (is("patch_embed.proj.bias")) { shp = {D}; return true; }
    if (m->cfg.mlp_time_embed) {
        if (is("time_embed.0.weight")) { shp = {4 * D, D}; return true; }
        if (is("time_embed.0.bias")) { shp = {4 * D}; return true; }
        if (is("time_embed.2.weight")) { shp = {D, 4 * D}; return true; }
        if (is("time_embed.2.bias")) { shp = {D}; return true; }
    }
    if (is("label_emb.weight")) { if (m->cfg.num_classes <= 0) return false; shp = {m->cfg.num_classes, D}; return true; }
    if (is("norm.weight") || is("norm.bias")) { shp = {D}; return true; }
    if (is("decoder_pred.weight")) { shp = {m->pd, D}; return true; }
    if (is("decoder_pred.bias")) { shp = {m->pd}; return true; }
    if (is("final_layer.weight")) { shp = {C, C, 3, 3}; return true; }
    if (is("final_layer.bias")) { shp = {C}; return true; }
    std::string rest;
    bool out_blk = false;
    if (m->ee_type >= 0) {
        if (name.compare(0, 7, "matrix.") == 0 && m->ee_type == DD_EE_ATTENTION_PROBE) {   // AttentionProbe, early_exit.py:46-58
            const size_t e = name.find('.', 7);
            if (e == std::string::npos || probe_index(m, name.substr(7, e - 7)) < 0) return false;
            const std::string tail = name.substr(e + 1);
            if (tail == "q") { shp = {1, 1, 1, D}; return true; }
            if (tail == "weight_kv.weight") { shp = {2 * D, D}; return true; }
            if (tail == "weight_kv.bias") { shp = {2 * D}; return true; }
            if (tail == "classification.0.weight") { shp = {D, D}; return true; }
            if (tail == "classification.0.bias") { shp = {D}; return true; }
            if (tail == "classification.2.weight") { shp = {1, D}; return true; }
            if (tail == "classification.2.bias") { shp = {1}; return true; }
            return false;
        }
        if (name.compare(0, 7, "matrix.") == 0) {               // matrix.<key>.classifier.0.{weight,bias}
            const size_t e = name.find(".classifier.0.");
            if (e == std::string::npos || probe_index(m, name.substr(7, e - 7)) < 0) return false;
            const std::string tail = name.substr(e + 14);
            if (tail == "weight") { shp = {1, D}; return true; }
            if (tail == "bias") { shp = {1}; return true; }
            return false;
        }
        std::string hrest;
        if (head_index(m, name, hrest) >= 0) {
            if (hrest == "norm.weight" || hrest == "norm.bias") { shp = {D}; return true; }
            if (hrest == "decoder_pred.weight") { shp = {m->pd, D}; return true; }
            if (hrest == "decoder_pred.bias") { shp = {m->pd}; return true; }
            if (hrest == "final_layer.weight") { shp = {C, C, 3, 3}; return true; }
            if (hrest == "final_layer.bias") { shp = {C}; return true; }
            return false;
        }
    }
    auto strip = [&](const char* pre, bool indexed) -> bool {
        const size_t n = std::strlen(pre);
        if (name.compare(0, n, pre) != 0) return false;
        size_t p = n;
        if (indexed) {
            size_t q = p;
            while (q < name.size() && name[q] >= '0' && name[q] <= '9') ++q;
            if (q == p || q >= name.size() || name[q] != '.') return false;
            const int idx = std::atoi(name.substr(p, q - p).c_str());
            if (idx >= m->half_depth) return false;
            p = q + 1;
        }
        rest = name.substr(p);
        return true;
    };
    if (strip("in_blocks.", true)) out_blk = false;
    else if (strip("out_blocks.", true)) out_blk = true;
    else if (strip("mid_block.", false)) out_blk = false;
    else return false;
    if (rest == "norm1.weight" || rest == "norm1.bias" || rest == "norm2.weight" || rest == "norm2.bias" ||
        rest == "attn.proj.bias" || rest == "mlp.fc2.bias") { shp = {D}; return true; }
    if (rest == "attn.qkv.weight") { shp = {3 * D, D}; return true; }
    if (m->cfg.qkv_bias && rest == "attn.qkv.bias") { shp = {3 * D}; return true; }
    if (rest == "attn.proj.weight") { shp = {D, D}; return true; }
    if (rest == "mlp.fc1.weight") { shp = {hid, D}; return true; }
    if (rest == "mlp.fc1.bias") { shp = {hid}; return true; }
    if (rest == "mlp.fc2.weight") { shp = {D, hid}; return true; }
    if (out_blk && rest == "skip_linear.weight") { shp = {D, 2 * D}; return true; }
    if (out_blk && rest == "skip_linear.bias") { shp = {D}; return true; }
    return false;
}

std::vector<std::string> required_names(const dd_model* m) {
    std::vector<std::string> v = {"pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias", "norm.weight",
                                  "norm.bias", "decoder_pred.weight", "decoder_pred.bias", "final_layer.weight",
                                  "final_layer.bias"};
    if (m->cfg.num_classes > 0) v.push_back("label_emb.weight");
    if (m->cfg.mlp_time_embed)
        for (const char* s : {"time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias"}) v.push_back(s);
    auto blk = [&](const std::string& p, bool skip) {
        for (const char* s : {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.proj.weight", "attn.proj.bias",
                              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                              "mlp.fc2.bias"})
            v.push_back(p + s);
        if (skip) { v.push_back(p + "skip_linear.weight"); v.push_back(p + "skip_linear.bias"); }
        if (m->cfg.qkv_bias) v.push_back(p + "attn.qkv.bias");
    };
    for (int i = 0; i < m->half_depth; ++i) blk("in_blocks." + std::to_string(i) + ".", false);
    blk("mid_block.", false);
    for (int i = 0; i < m->half_depth; ++i) blk("out_blocks." + std::to_string(i) + ".", true);
    if (m->ee_type >= 0) {
        for (int layer = 0; layer < m->cfg.depth; ++layer)
            for (const char* s : {"norm.weight", "norm.bias", "decoder_pred.weight", "decoder_pred.bias",
                                  "final_layer.weight", "final_layer.bias"})
                v.push_back(head_prefix(m, layer) + s);
        const bool per_layer_only = m->ee_type == DD_EE_MLP_PER_LAYER || m->ee_type == DD_EE_ATTENTION_PROBE;
        const int nt = per_layer_only ? 1 : 1000, nl = m->ee_type == DD_EE_MLP_PER_TIMESTEP ? 1 : m->cfg.depth;
        for (int t = 0; t < nt; ++t)
            for (int layer = 0; layer < nl; ++layer) {
                const std::string pre = "matrix." + probe_key(m, layer, t) + ".";
                if (m->ee_type == DD_EE_ATTENTION_PROBE) {
                    for (const char* s : {"q", "weight_kv.weight", "weight_kv.bias", "classification.0.weight", "classification.0.bias",
                                          "classification.2.weight", "classification.2.bias"})
                        v.push_back(pre + s);
                } else {
                    v.push_back(pre + "classifier.0.weight");
                    v.push_back(pre + "classifier.0.bias");
                }
            }
    }
    return v;
}

void bind_ws(const WsOffsets& o, char* arena, WsPtrs& w) {
    w.x = (float*)(arena + o.x); w.h = arena + o.h; w.ao = arena + o.ao; w.qkv = arena + o.qkv; w.hid = arena + o.hid; w.xb = arena + o.xb;
    w.dec = (float*)(arena + o.dec);
    w.skips.clear();
    for (size_t v : o.sk) w.skips.push_back(arena + v);
    w.mlp_partial = o.has_part ? (float*)(arena + o.part) : nullptr;
    w.qkv_dump = o.has_dump ? (bf16_t*)(arena + o.dump) : nullptr;
    w.hfrag = o.has_hf ? (bf16_t*)(arena + o.hf) : nullptr;
    w.ytap = o.has_tap ? (float*)(arena + o.tap) : nullptr;
}
// Layout of one chain's activation workspace for batches up to `batch` (the model's max_batch; half of it, rounded up, for the second
// half-batch chain of dd_sample, which never runs more).
WsOffsets ws_layout(const dd_model* m, int batch) {
    const int D = m->D, L = m->L, hid = m->hidden;
    const size_t es = m->esize;
    const size_t Mp = (size_t)round_up(batch * L, 256);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    // qkv is head-major: B * 3H units of Lp rows x 64 (rows [L, Lp) of a unit are never written: the bf16 attention launch that stages K / V
    // from this tensor zeroes them in its LDS images itself)
    const size_t qkv_elems = (size_t)batch * 3 * D * (size_t)make_head_major(L, m->H).Lp;
    const size_t o_x = take(Mp * D * 4), o_h = take(Mp * D * es), o_ao = take(Mp * D * es), o_qkv = take(std::max(Mp * 3 * D, qkv_elems) * es);
    const size_t o_hid = take(Mp * (size_t)m->hid_ld * es), o_xb = take(Mp * D * es);
    std::vector<size_t> o_sk;
    for (int i = 0; i < m->half_depth; ++i) o_sk.push_back(take(Mp * D * es));
    const size_t o_dec = take(Mp * m->pd * 4);
    const size_t part_bytes = m->fused_mlp ? mlp_fused_partial_bytes(batch, m->extras, D, hid)
                              : m->splitk ? (size_t)2 * Mp * D * 4
                              : m->rowlin_fc2 ? std::max(std::max(rowlin_partial_bytes(batch, m->extras, hid), rowlin_partial_bytes(batch, m->extras, 2 * D)), rowlin_partial_bytes(batch, m->extras, D)) : 0;
    const size_t o_part = take(part_bytes);
    const size_t o_dump = take(m->fused_qkv ? 16384 : 0);
    const size_t o_hf = take(m->fused_qa ? (size_t)batch * m->N * D * 2 : 0);
    const bool has_tap = m->ee_type >= 0 && m->fused_skip;
    const size_t o_tap = take(has_tap ? Mp * D * 4 : 0);
    WsOffsets o{o_x, o_h, o_ao, o_qkv, o_hid, o_xb, o_dec, o_part, o_dump, o_hf, off, o_sk, part_bytes != 0, m->fused_qkv, m->fused_qa};
    o.bytes = off;
    o.part_bytes = part_bytes;
    o.tap = o_tap; o.has_tap = has_tap;
    return o;
}
// exchange the model's workspace pointers with the second chain's (the caller swaps the context's step state too): the launch sequence of a step is
// enqueued / captured for that chain by the same code, on the same weights
void swap_chain(dd_model* m) {
    WsPtrs& w = m->ws2;
    std::swap(m->x, w.x); std::swap(m->h, w.h); std::swap(m->ao, w.ao); std::swap(m->qkv, w.qkv); std::swap(m->hid, w.hid);
    std::swap(m->xb, w.xb); std::swap(m->dec, w.dec); std::swap(m->skips, w.skips); std::swap(m->mlp_partial, w.mlp_partial);
    std::swap(m->qkv_dump, w.qkv_dump); std::swap(m->hfrag, w.hfrag); std::swap(m->ytap, w.ytap);
}

// ---- the forward: tokens -> blocks -> decoder_pred patches (m->dec) ---------------------------
// early-exit taps of one forward (EarlyExitUViT.forward, early_exit.py:290-313): cls [depth, B], outs [depth, B, C, S, S]
struct EeTaps { float* cls; float* outs; int t; };

template <typename T>
int run_backbone(dd_model* m, const float* x_img, const float* t_vec, const int64_t* y_dev, int B, hipStream_t s,
                 const EeTaps* ee = nullptr) {
    dd_ctx* c = m->ctx;
    const int D = m->D, L = m->L, M = B * L;
    const int Mp = round_up(M, 256);
    EmbedArgs ea{x_img, m->emb_wt, m->emb_b, m->pos, m->label, (const long long*)y_dev, t_vec, c->st, m->x,
                 B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, D, L, m->extras,
                 m->cfg.num_classes, m->cfg.normalize_timesteps, Mp, (c->dev_flags & DD_DEV_GENERIC_EMBED) ? 1 : 0};
    // the first block's norm1 of the patch rows from the embed launch's registers (where the attention launch computes attn.qkv itself and
    // normalises the extra-token rows from the residual stream: the time_embed MLP below only rewrites such a row)
    bool ln1_done = false;
    if constexpr (sizeof(T) == 2) {
        if (m->fused_qa && !(c->dev_flags & DD_DEV_NO_EMBED_LN) && embed_ln_supported(ea)) {
            ea.ln_g = m->blocks[0].ln1_g; ea.ln_b = m->blocks[0].ln1_b; ea.ln_frag = m->hfrag;
            ln1_done = true;
        }
    }
    DD_HIP(c, launch_embed(ea, s));
    if (m->tm_w1t) {   // mlp_time_embed: the time token goes through Linear -> SiLU -> Linear (models/uvit.py:264-272, 358)
        TimeMlpArgs ta{m->tm_w1t, m->tm_b1, m->tm_w2t, m->tm_b2, m->pos, t_vec, c->st, m->x, B, D, L, m->extras, m->cfg.normalize_timesteps};
        DD_HIP(c, launch_time_mlp(ta, s));
    }

    // in-context timing (dd_profile_steps): an event on s in front of and behind every launch of the selected kind
    const int prof = c->prof_kind == DD_PROF_DOMINANT ? ((sizeof(T) == 2 && m->fused_mlp) ? DD_PROF_BLOCK_TAIL : DD_PROF_FC1) : c->prof_kind;
    auto mark = [&](int kind) -> int {
        if (!m->time_fc1 || kind != prof) return DD_OK;
        while (m->fc1_events.size() < m->fc1_used + 1) {
            hipEvent_t e;
            DD_HIP(c, hipEventCreate(&e));
            m->fc1_events.push_back(e);
        }
        DD_HIP(c, hipEventRecord(m->fc1_events[m->fc1_used++], s));
        return DD_OK;
    };
    // 128 x 128 or 256 x 256 tiles for a bf16 Linear: decided for the model's max_batch on the context's CU count -- never for the batch (or the
    // chain-halved grid) of this call, so that a row takes the same kernel alone, in a full batch and in a half-batch chain
    auto tile128 = [&](int N, int K, int K1) { return sizeof(T) == 2 && gemm_prefers_128(m->cfg.max_batch * L, N, K, K1, c->base_cus) ? 1 : 0; };
    T* h = (T*)m->h; T* ao = (T*)m->ao; T* qkv = (T*)m->qkv; T* hid = (T*)m->hid; T* xb = (T*)m->xb;
    const int nb = (int)m->blocks.size();
    bool h_ready = ln1_done;   // h already holds norm1 of the coming block (written by the fused MLP of the previous one / the embed launch)
    bool skip_done = false; // ... and x already holds that block's skip_linear output (the previous fused launch ran it too)
    bool qkv_done = false;  // ... and qkv already holds that block's attn.qkv output (ditto)
    bool qa_ready = ln1_done;  // ... or only the extra-token rows of it: the patch rows' qkv is computed inside the attention launch
    bool ee_side = false;   // this block's early-exit head / probe launches are in flight on the side stream
    // Early-exit heads: every layer's LayerNorm + decoder_pred launch writes its own slice of a [depth][B L, pd] buffer and every MLP probe's row
    // launch its own [B, L] slice; ONE unpatchify / conv launch and ONE probe reduce launch behind the last block finish all layers (the
    // per-layer arithmetic is unchanged; 2 x 12 launches less on each step's critical path).  The buffers live in the MLP hidden buffer, which
    // the fused block tail never touches; models on the GEMM path (and images below one 16 x 16 tile) keep the per-layer launches.
    float* ee_dec_all = nullptr;
    float* ee_srow_all = nullptr;
    if (ee && sizeof(T) == 2 && m->fused_mlp && m->ee_conv_stride_ok && m->heads[0].wg && m->cfg.img_size >= 16 &&
        (size_t)nb * ((size_t)M * m->pd + (size_t)B * L) * sizeof(float) <= (size_t)Mp * m->hid_ld * m->esize) {
        ee_dec_all = (float*)m->hid;
        ee_srow_all = ee_dec_all + (size_t)nb * M * m->pd;
    }
    for (int bi = 0; bi < nb; ++bi) {
        const BlockW& w = m->blocks[bi];
        const bool is_in = bi < m->half_depth, is_out = bi > m->half_depth;
        if (ee) {
            // output head and uncertainty probe on the INPUT of block bi (for out-blocks: before skip_linear, as the
            // reference taps x before blk(x, skip)); both read the fp32 residual stream.  In- and mid-blocks: on the context's side
            // stream, beside this block's norm1 / qkv / attention launches (which only read x); joined before the first launch that
            // writes x (attn.proj).  Out-blocks start with skip_linear, which overwrites x: their heads stay in line.
            const HeadW& hd = m->heads[bi];
            // (the previous launch ran this block's skip_linear already: x holds its output, the tapped y is in ytap)
            const float* xin = (skip_done && m->ytap) ? m->ytap : m->x;
            ee_side = !is_out && c->side && s != c->side && !c->ee_inline;
            hipStream_t hs = ee_side ? c->side : s;
            if (ee_side) {
                DD_HIP(c, hipEventRecord(c->ev_ee_fork, s));
                DD_HIP(c, hipStreamWaitEvent(c->side, c->ev_ee_fork, 0));
            }
            // probe row: layer bi | timestep t | (t, layer): t is read from the step state inside the launch (a captured
            // step replays for every t); dd_forward_early_exit has put int(t) there
            const int t_mul = m->ee_type == DD_EE_MLP_PER_LAYER ? 0 : m->ee_type == DD_EE_MLP_PER_TIMESTEP ? 1 : nb;
            const int add = m->ee_type == DD_EE_MLP_PER_TIMESTEP ? 0 : bi;
            bool probe_done = false;
            if (hd.wg) {   // the head's LayerNorm + decoder_pred in one exact-fp32 launch (the final head's kernel), patch rows only
                HeadDecArgs ha{xin, hd.wg, hd.dc, ee_dec_all ? ee_dec_all + (size_t)bi * M * m->pd : m->dec, M, m->pd, (L - m->extras) % 16 == 0 ? L : 0, m->extras};
                if (hd.wsplit) { ha.wg = hd.wsplit; ha.c = hd.dcs; ha.split = 1; }      // (bf16 engine: the split-bf16 product, rowops.hip SPLIT)
                if (ee_srow_all && m->ee_type != DD_EE_ATTENTION_PROBE && head_dec_probe_supported(D)) {   // ... and the MLP probe's per-token values of the same rows
                    ha.srow = ee_srow_all + (size_t)bi * B * L; ha.pw_base = m->probe_w; ha.pb_base = m->probe_b; ha.st = c->st; ha.t_mul = t_mul; ha.add = add;
                    probe_done = true;
                }
                DD_HIP(c, launch_head_dec(ha, D, c->num_cus, hs));
            } else {
                float* hf = (float*)m->hid;   // the MLP hidden buffer is free between blocks
                DD_HIP(c, launch_layernorm<float>(xin, hd.ng, hd.nb, hf, M, D, hs));
                GemmArgs<float> g{hf, nullptr, hd.wdec, hd.bdec, m->dec, nullptr, M, m->pd, D, D, D, 0, m->pd};
                DD_HIP(c, launch_gemm<float>(g, EPI_BIAS_SET, hs, c->num_cus));
            }
            const long long chw = (long long)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
            if (!ee_dec_all) {
                FinalArgs fa{m->dec, hd.wconv, hd.bconv, nullptr, nullptr, ee->outs + (long long)bi * B * chw, nullptr, c->st,
                             c->coef, B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, m->L, m->extras, DD_NOISE_NONE, 0, 0};
                DD_HIP(c, launch_final(fa, hs));
            }
            if (m->ee_type == DD_EE_ATTENTION_PROBE) {
                DD_HIP(c, launch_ee_attn_probe(xin, m->attn_probes[bi], ee->cls + (long long)bi * B, B, L, D, hs));
            } else if (!probe_done) {
                if (ee_srow_all) DD_HIP(c, launch_ee_probe(xin, m->probe_w, m->probe_b, nullptr, ee_srow_all + (size_t)bi * B * L, B, L, D, c->st, t_mul, add, hs));   // rows only: reduced behind the last block
                else DD_HIP(c, launch_ee_probe(xin, m->probe_w, m->probe_b, ee->cls + (long long)bi * B, (float*)m->hid, B, L, D, c->st, t_mul, add, hs));   // (the MLP hidden buffer is free between blocks)
            }
            if (ee_side) DD_HIP(c, hipEventRecord(c->ev_ee_join, c->side));
        }
        if (is_out && !skip_done) {
            const int oi = bi - m->half_depth - 1;
            const T* skip = (const T*)m->skips[m->half_depth - 1 - oi];  // LIFO (uvit.py:374-375)
            GemmArgs<T> g{xb, skip, (const T*)w.skip_w, w.skip_b, m->x, nullptr, M, D, 2 * D, D, D, D, D};
            g.tile128 = tile128(D, 2 * D, D);
            bool done = false;
            if constexpr (sizeof(T) == 2) {
                if (m->rowlin_skip) {     // (embed_dim 768) x = skip_linear(cat[x, skip]) and this block's norm1 in one row-resident launch
                    RowLinArgs ra{};
                    ra.A = (const bf16_t*)xb; ra.A2 = (const bf16_t*)skip; ra.k_split = D; ra.set_x = 1; ra.lda = D; ra.K = 2 * D;
                    ra.wimg = w.rls_img; ra.bias = w.skip_b; ra.xres = m->x; ra.partial = m->mlp_partial; ra.ln_g = w.ln1_g; ra.ln_b = w.ln1_b;
                    if (m->fused_qa) ra.h_frag = m->hfrag; else ra.h_out = (bf16_t*)h;
                    rowlin_plan(B, m->N, m->extras, L, 2 * D, ra);
                    if (int rc = mark(DD_PROF_ROWLIN)) return rc;
                    DD_HIP(c, launch_rowlin(ra, s));
                    if (int rc = mark(DD_PROF_ROWLIN)) return rc;
                    MlpFusedArgs fr{};
                    fr.b2 = w.skip_b; fr.xres = m->x; fr.partial = m->mlp_partial; fr.ldo = D; fr.reduce_set = 1;
                    fr.tok_n = ra.tok_n; fr.tok_e = ra.tok_e; fr.tok_l = ra.tok_l; fr.n_extra = ra.n_extra; fr.tiles_left = ra.tiles_extra;
                    fr.groups = ra.groups; fr.prows = 128;
                    if (!m->fused_qa) { fr.ln_out_g = w.ln1_g; fr.ln_out_b = w.ln1_b; fr.ln_out = (bf16_t*)h; }
                    DD_HIP(c, launch_mlp_reduce(fr, D, s));
                    h_ready = true; qa_ready = m->fused_qa; done = true;
                }
            }
            if constexpr (sizeof(T) == 2) {
                if (m->splitk && !done) {     // split-K halves -> slabs; x = bias + slabs and this block's norm1 in the row pass behind it
                    g.partial = m->mlp_partial; g.splits = 2;
                    if (int rc = mark(DD_PROF_SPLITK)) return rc;
                    DD_HIP(c, launch_gemm_splitk(g, s, c->num_cus));
                    if (int rc = mark(DD_PROF_SPLITK)) return rc;
                    ReduceLnArgs ra{m->x, m->mlp_partial, (long long)M * D, 2, 0, w.skip_b, nullptr, D, w.ln1_g, w.ln1_b, (bf16_t*)h,
                                    m->fused_qa ? m->hfrag : nullptr, L, m->extras, M};
                    DD_HIP(c, launch_reduce_ln(ra, D, s));
                    h_ready = true; qa_ready = m->fused_qa; done = true;
                }
            }
            if (!done) DD_HIP(c, launch_gemm<T>(g, EPI_BIAS_SET, s, c->num_cus));
        }
        skip_done = false;
        if (!h_ready && !qkv_done) {     // else: written by the previous block's fused MLP
            if (sizeof(T) == 2 && m->fused_qa) {
                // (the first block; blocks behind a skip_linear GEMM when that fusion is off) norm1 straight into the order the
                // attention launch loads it, so that these blocks take the same launch as the others
                DD_HIP(c, launch_layernorm_frag(m->x, w.ln1_g, w.ln1_b, (bf16_t*)h, m->hfrag, M, D, L, m->extras, s));
                qa_ready = true;
            } else {
                DD_HIP(c, launch_layernorm<T>(m->x, w.ln1_g, w.ln1_b, h, M, D, s));
            }
        }
        h_ready = false;
        if (qa_ready) {
            if (int rc = mark(DD_PROF_QKV_ATTENTION)) return rc;
            if constexpr (sizeof(T) == 2)
                DD_HIP(c, launch_qkv_attention(m->hfrag, w.qa_img, w.qkv_b, nullptr, m->x, w.ln1_g, w.ln1_b, (bf16_t*)ao, B, L, m->H, D, m->extras, s));
            if (int rc = mark(DD_PROF_QKV_ATTENTION)) return rc;
        } else {
            if (!qkv_done) {
                GemmArgs<T> g{h, nullptr, (const T*)w.qkv_w, w.qkv_b, nullptr, qkv, M, 3 * D, D, D, D, 0, 3 * D};
                g.hm = make_head_major(L, m->H);     // head-major: each (q | k | v, head) unit of an image is contiguous (attention.hip)
                DD_HIP(c, launch_gemm<T>(g, w.qkv_b ? EPI_BIAS_STORE : EPI_STORE, s, c->num_cus));
            }
            if (int rc = mark(DD_PROF_QKV_ATTENTION)) return rc;
            DD_HIP(c, launch_attention<T>(qkv, ao, B, L, m->H, D, s));
            if (int rc = mark(DD_PROF_QKV_ATTENTION)) return rc;
        }
        qkv_done = false; qa_ready = false;
        if (ee_side) {     // the head / probe launches have read x: from here on the block updates it
            DD_HIP(c, hipStreamWaitEvent(s, c->ev_ee_join, 0));
            ee_side = false;
        }
        bool ln2_done = false;
        if constexpr (sizeof(T) == 2) {
            if (m->rowlin_proj) {
                // (embed_dim 768) x += proj(ao) + b and norm2 of the updated rows in one row-resident launch (see mlp.fc2 below)
                RowLinArgs ra{};
                ra.A = (const bf16_t*)ao; ra.lda = D; ra.K = D; ra.wimg = w.rlp_img; ra.bias = w.proj_b; ra.xres = m->x;
                ra.partial = m->mlp_partial; ra.ln_g = w.ln2_g; ra.ln_b = w.ln2_b; ra.h_out = (bf16_t*)h;
                rowlin_plan(B, m->N, m->extras, L, D, ra);
                if (int rc = mark(DD_PROF_ROWLIN)) return rc;
                DD_HIP(c, launch_rowlin(ra, s));
                if (int rc = mark(DD_PROF_ROWLIN)) return rc;
                MlpFusedArgs fr{};
                fr.b2 = w.proj_b; fr.xres = m->x; fr.partial = m->mlp_partial; fr.ldo = D;
                fr.tok_n = ra.tok_n; fr.tok_e = ra.tok_e; fr.tok_l = ra.tok_l; fr.n_extra = ra.n_extra; fr.tiles_left = ra.tiles_extra;
                fr.groups = ra.groups; fr.prows = 128;
                fr.ln_out_g = w.ln2_g; fr.ln_out_b = w.ln2_b; fr.ln_out = (bf16_t*)h;
                DD_HIP(c, launch_mlp_reduce(fr, D, s));
                ln2_done = true;
            }
        }
        if constexpr (sizeof(T) == 2) {
            if (m->splitk && !ln2_done) {     // attn.proj as split-K halves; x += bias + slabs and norm2 in the row pass behind it
                GemmArgs<T> g{ao, nullptr, (const T*)w.proj_w, nullptr, nullptr, nullptr, M, D, D, D, D, 0, D};
                g.partial = m->mlp_partial; g.splits = 2;
                if (int rc = mark(DD_PROF_SPLITK)) return rc;
                DD_HIP(c, launch_gemm_splitk(g, s, c->num_cus));
                if (int rc = mark(DD_PROF_SPLITK)) return rc;
                ReduceLnArgs ra{m->x, m->mlp_partial, (long long)M * D, 2, 1, w.proj_b, nullptr, D, w.ln2_g, w.ln2_b, (bf16_t*)h, nullptr, L, m->extras, M};
                DD_HIP(c, launch_reduce_ln(ra, D, s));
                ln2_done = true;
            }
        }
        if (!ln2_done && !(sizeof(T) == 2 && m->fused_proj)) {   // fused: x += proj(ao) + b happens inside the fused MLP launch below
            GemmArgs<T> g{ao, nullptr, (const T*)w.proj_w, w.proj_b, m->x, nullptr, M, D, D, D, D, 0, D};
            g.tile128 = tile128(D, D, D);
            DD_HIP(c, launch_gemm<T>(g, EPI_BIAS_RESID, s, c->num_cus));
        }
        if (!ln2_done && !(sizeof(T) == 2 && m->fused_mlp)) DD_HIP(c, launch_layernorm<T>(m->x, w.ln2_g, w.ln2_b, h, M, D, s));   // fused MLP: norm2 in its prologue
        // the T-typed copy of the block output feeds a later skip_linear: as the `skip`
        // operand (in-blocks) or as the `x` operand (mid / out blocks, except the last)
        T* copy = is_in ? (T*)m->skips[bi] : (bi + 1 < nb ? xb : nullptr);
        if constexpr (sizeof(T) == 2) {
            if (m->fused_mlp) {
                MlpFusedArgs fa{};
                fa.X = nullptr; fa.ldx = D; fa.wimg = w.mlp_img; fa.b1p = w.mlp_b1p; fa.b2 = w.fc2_b;
                fa.ln_in_g = w.ln2_g; fa.ln_in_b = w.ln2_b;                       // norm2 of this block, in the prologue
                if (bi + 1 < nb && bi + 1 <= m->half_depth) {                    // next block starts with norm1 (no skip_linear in between)
                    fa.ln_out_g = m->blocks[bi + 1].ln1_g; fa.ln_out_b = m->blocks[bi + 1].ln1_b; fa.ln_out = (bf16_t*)h;
                    h_ready = true;
                }
                fa.xres = m->x; fa.out = (bf16_t*)copy; fa.ldo = D; fa.partial = m->mlp_partial;
                if (m->fused_proj) { fa.ao = (const bf16_t*)ao; fa.bproj = w.proj_b; fa.nproj = D / 32; }
                const bool skip_next = m->fused_skip && bi >= m->half_depth && bi + 1 < nb;   // the next block starts with skip_linear
                if (skip_next) {
                    const BlockW& wn = m->blocks[bi + 1];
                    const int oi = bi - m->half_depth;                                // index of the NEXT block among the out-blocks
                    fa.skip = (const bf16_t*)m->skips[m->half_depth - 1 - oi];      // LIFO (uvit.py:374-375)
                    if (ee) fa.y_tap = m->ytap;                                       // the next block's head / probe read y, which this launch consumes
                    fa.bskip = wn.skip_b; fa.nskip = D / 16;
                    fa.ln_out_g = wn.ln1_g; fa.ln_out_b = wn.ln1_b; fa.ln_out = (bf16_t*)h;
                    h_ready = true; skip_done = true;
                }
                // the next block's attn.qkv last of all (where that block's skip_linear, if it has one, runs in here as well)
                // the next block's attn.qkv: inside its attention launch (fused_qa), else last of all in this launch (fused_qkv) --
                // wherever this launch leaves that block's norm1
                const bool h_next = bi + 1 < nb && (bi < m->half_depth || skip_next);
                const bool qa_next = m->fused_qa && h_next;
                const bool qkv_next = !qa_next && m->fused_qkv && h_next;
                if (qa_next) fa.ln_out_frag = m->hfrag;      // the patch rows' norm1 in the order the attention launch loads it
                if (qkv_next) {
                    const BlockW& wn = m->blocks[bi + 1];
                    fa.ln_out_g = wn.ln1_g; fa.ln_out_b = wn.ln1_b; fa.ln_out = (bf16_t*)h;   // (written for the extra-token rows only)
                    fa.qkv_out = (bf16_t*)qkv; fa.qkv_dump = m->qkv_dump; fa.hm = make_head_major(L, m->H); fa.nqkv = 3 * D / 32;
                    h_ready = true; qkv_done = true;
                }
                mlp_fused_plan(B, m->N, m->extras, L, m->hidden, fa);
                // The LAST block's projection / MLP of the extra-token rows feed nothing: the output head decodes the patch rows only
                // (models/uvit.py:377-380 slices the extras off), and those rows' K / V went into this block's attention before.  No
                // proj_rows / hidden-split workgroups / reduce launch for them.
                if (bi + 1 == nb && m->fused_proj) { fa.n_extra = 0; fa.tiles_left = 0; }
                if (m->fused_proj) fa.reduce_set = 1;   // the extra-token rows' projection runs in their hidden-split workgroups: the first group's slab carries x + proj(ao) + b
                if (int rc = mark(DD_PROF_BLOCK_TAIL)) return rc;
                DD_HIP(c, launch_mlp_fused(fa, D, s));
                if (int rc = mark(DD_PROF_BLOCK_TAIL)) return rc;              // (the event pair brackets the fused kernel alone)
                if (skip_next) {
                    MlpFusedArgs fr = fa;                    // the reduce kernel finishes y of the extra-token rows (fp32 + the bf16
                    fr.ln_out = nullptr;                     // copy in xb); their skip_linear + norm1 follow in one small launch
                    DD_HIP(c, launch_mlp_reduce(fr, D, s));
                    // (fused_qa: the attention launch normalises the extra-token rows itself, so the launch is split by columns)
                    DD_HIP(c, launch_skip_rows_ln(fa, D, s, !qa_next));
                } else if (qa_next) {
                    MlpFusedArgs fr = fa;
                    fr.ln_out = nullptr;                     // (ditto: no norm1 rows needed)
                    DD_HIP(c, launch_mlp_reduce(fr, D, s));
                } else {
                    DD_HIP(c, launch_mlp_reduce(fa, D, s));
                }
                if (qkv_next) DD_HIP(c, launch_qkv_rows(fa, D, s));   // the extra-token rows' qkv, from the norm1 rows the launch above wrote
                qa_ready = qa_next;    // (the extra-token rows reach the attention launch through the residual stream x)
                continue;
            }
        }
        {
            GemmArgs<T> g{h, nullptr, (const T*)w.fc1_w, w.fc1_b, nullptr, hid, M, m->hidden, D, D, D, 0, m->hid_ld};
            g.tile128 = tile128(m->hidden, D, D);
            if (int rc = mark(DD_PROF_FC1)) return rc;
            DD_HIP(c, launch_gemm<T>(g, EPI_BIAS_GELU, s, c->num_cus));
            if (int rc = mark(DD_PROF_FC1)) return rc;
        }
        if constexpr (sizeof(T) == 2) {
            if (m->rowlin_fc2) {
                // x += fc2(hid) + b with each wave's 32 residual rows resident in registers; where the next block starts with norm1 (in- and
                // mid-blocks) that LayerNorm leaves from the same registers -- in the attention launch's fragment order under fused_qa
                RowLinArgs ra{};
                ra.A = (const bf16_t*)hid; ra.lda = m->hid_ld; ra.K = m->hidden; ra.wimg = w.rl_img; ra.bias = w.fc2_b; ra.xres = m->x;
                ra.x_copy = (bf16_t*)copy; ra.partial = m->mlp_partial;
                const bool ln_next = bi + 1 < nb && bi + 1 <= m->half_depth;
                if (ln_next) {
                    ra.ln_g = m->blocks[bi + 1].ln1_g; ra.ln_b = m->blocks[bi + 1].ln1_b;
                    if (m->fused_qa) ra.h_frag = m->hfrag; else ra.h_out = (bf16_t*)h;
                }
                rowlin_plan(B, m->N, m->extras, L, m->hidden, ra);
                if (int rc = mark(DD_PROF_ROWLIN)) return rc;
                DD_HIP(c, launch_rowlin(ra, s));
                if (int rc = mark(DD_PROF_ROWLIN)) return rc;
                MlpFusedArgs fr{};           // the extra-token rows: bias + residual + the K-split slabs in a fixed order (+ their norm1 rows)
                fr.b2 = w.fc2_b; fr.xres = m->x; fr.out = (bf16_t*)copy; fr.ldo = D; fr.partial = m->mlp_partial;
                fr.tok_n = ra.tok_n; fr.tok_e = ra.tok_e; fr.tok_l = ra.tok_l; fr.n_extra = ra.n_extra; fr.tiles_left = ra.tiles_extra;
                fr.groups = ra.groups; fr.prows = 128;
                if (ln_next && !m->fused_qa) { fr.ln_out_g = ra.ln_g; fr.ln_out_b = ra.ln_b; fr.ln_out = (bf16_t*)h; }
                DD_HIP(c, launch_mlp_reduce(fr, D, s));
                h_ready = ln_next;
                qa_ready = ln_next && m->fused_qa;    // (the extra-token rows reach the attention launch through the residual stream x)
                continue;
            }
        }
        if constexpr (sizeof(T) == 2) {
            if (m->splitk) {     // mlp.fc2 as split-K halves; x += bias + slabs, the bf16 copy and (in- / mid-blocks) the next block's norm1 in the row pass
                GemmArgs<T> g{hid, nullptr, (const T*)w.fc2_w, nullptr, nullptr, nullptr, M, D, m->hidden, m->hidden, m->hid_ld, m->hid_ld, D};
                g.partial = m->mlp_partial; g.splits = 2;
                if (int rc = mark(DD_PROF_SPLITK)) return rc;
                DD_HIP(c, launch_gemm_splitk(g, s, c->num_cus));
                if (int rc = mark(DD_PROF_SPLITK)) return rc;
                const bool ln_next = bi + 1 < nb && bi + 1 <= m->half_depth;
                ReduceLnArgs ra{m->x, m->mlp_partial, (long long)M * D, 2, 1, w.fc2_b, (bf16_t*)copy, D,
                                ln_next ? m->blocks[bi + 1].ln1_g : nullptr, ln_next ? m->blocks[bi + 1].ln1_b : nullptr, (bf16_t*)h,
                                ln_next && m->fused_qa ? m->hfrag : nullptr, L, m->extras, M};
                DD_HIP(c, launch_reduce_ln(ra, D, s));
                h_ready = ln_next;
                qa_ready = ln_next && m->fused_qa;
                continue;
            }
        }
        {
            GemmArgs<T> g{hid, nullptr, (const T*)w.fc2_w, w.fc2_b, m->x, copy, M, D, m->hidden, m->hidden,
                          m->hid_ld, m->hid_ld, D};
            g.tile128 = tile128(D, m->hidden, m->hidden);
            DD_HIP(c, launch_gemm<T>(g, EPI_BIAS_RESID, s, c->num_cus));
        }
    }
    if (ee_dec_all) {     // every layer's unpatchify + conv in one launch (layer i: images [i B, (i + 1) B) of nb B, its own conv weights), every MLP probe's mean in one
        const long long chw = (long long)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
        FinalArgs fa{ee_dec_all, m->heads[0].wconv, m->heads[0].bconv, nullptr, nullptr, ee->outs, nullptr, c->st,
                     c->coef, nb * B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, m->L, m->extras, DD_NOISE_NONE, 0, 0};
        fa.layer_B = B; fa.w_stride = m->ee_wconv_stride; fa.b_stride = m->ee_bconv_stride;
        DD_HIP(c, launch_final(fa, s));
        if (m->ee_type != DD_EE_ATTENTION_PROBE) DD_HIP(c, launch_ee_probe_reduce(ee_srow_all, ee->cls, nb * B, L, s));
        (void)chw;
    }
    // output head (uvit.py:377-378): final LayerNorm in fp32 into scratch (the MLP hidden buffer is
    // free here), then decoder_pred as an exact-fp32 MFMA GEMM in BOTH precision modes, so eps is
    // never rounded to bf16.  dec holds all L tokens per image; the extras are skipped downstream.
    if (m->wdec_g) {   // fused: rows read once, normalised rows never written
        HeadDecArgs ha{m->x, m->wdec_g, m->dec_c, m->dec, M, m->pd, (L - m->extras) % 16 == 0 ? L : 0, m->extras};   // (only the patch rows)
        DD_HIP(c, launch_head_dec(ha, D, c->num_cus, s));
        return DD_OK;
    }
    float* hf = (float*)m->hid;
    DD_HIP(c, launch_layernorm<float>(m->x, m->norm_g, m->norm_b, hf, M, D, s));
    GemmArgs<float> g{hf, nullptr, m->wdec, m->bdec, m->dec, nullptr, M, m->pd, D, D, D, 0, m->pd};
    DD_HIP(c, launch_gemm<float>(g, EPI_BIAS_SET, s, c->num_cus));
    return DD_OK;
}

int run_model(dd_model* m, const float* x_img, const float* t_vec, const int64_t* y_dev, int B, hipStream_t s,
              const EeTaps* ee = nullptr) {
    return m->prec == DD_PREC_BF16 ? run_backbone<bf16_t>(m, x_img, t_vec, y_dev, B, s, ee)
                                   : run_backbone<float>(m, x_img, t_vec, y_dev, B, s, ee);
}

int check_call(dd_ctx* c, dd_model* m, int B, const int64_t* y_dev) {
    if (!c || !m) return DD_ERR_INVALID;
    if (m->ctx != c) return fail(c, DD_ERR_INVALID, "model belongs to another context");
    if (!m->finalized) return fail(c, DD_ERR_STATE, "dd_model_finalize has not been called");
    if (B < 1 || B > m->cfg.max_batch) return fail(c, DD_ERR_INVALID, "batch size outside [1, max_batch]");
    if (m->cfg.num_classes > 0 && !y_dev)
        return fail(c, DD_ERR_INVALID, "class-conditional model called without labels (pos_embed has L=extras+N rows)");
    if (m->cfg.num_classes <= 0 && y_dev)
        return fail(c, DD_ERR_INVALID, "unconditional model called with labels");
    return DD_OK;
}

// one sampling step enqueued on s: x <- update(x, model(x, t)) ; t comes from ctx->st
// advance != 0: the step's last kernel also decrements the device-resident timestep (graph replays / dd_sample)
int enqueue_step(dd_ctx* c, dd_model* m, float* x_dev, const int64_t* y_dev, int noise_mode, const float* z_dev,
                 int variance, float* eps_out, int B, hipStream_t s, int advance = 0, const AffineRow* atab = nullptr, int b0 = 0) {
    int rc = run_model(m, x_dev, nullptr, y_dev, B, s);
    if (rc) return rc;
    FinalArgs fa{m->dec, m->wconv, m->bconv, x_dev, z_dev, eps_out, x_dev, c->st, c->coef,
                 B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, m->L, m->extras, noise_mode, variance, advance, atab, b0};
    DD_HIP(c, launch_final(fa, s));
    return DD_OK;
}

// dd_sample / dd_sample_affine with graphs: the loop runs on context-owned staging copies of x / y, so the captured step
// does not depend on the caller's tensor addresses
int stage_inputs(dd_ctx* c, const float* x_dev, const int64_t* y_dev, int B, size_t x_elems, hipStream_t s, float** x_run,
                 const int64_t** y_run) {
    if (c->x_stage_elems < x_elems) {       // grows only: model graphs keyed on the old address are re-captured once
        if (c->x_stage) (void)hipFree(c->x_stage);
        c->x_stage = nullptr; c->x_stage_elems = 0;
        DD_HIP(c, hipMalloc((void**)&c->x_stage, x_elems * sizeof(float)));
        c->x_stage_elems = x_elems;
    }
    if (y_dev && c->y_stage_elems < (size_t)B) {
        if (c->y_stage) (void)hipFree(c->y_stage);
        c->y_stage = nullptr; c->y_stage_elems = 0;
        DD_HIP(c, hipMalloc((void**)&c->y_stage, (size_t)B * sizeof(int64_t)));
        c->y_stage_elems = (size_t)B;
    }
    DD_HIP(c, hipMemcpyAsync(c->x_stage, x_dev, x_elems * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (y_dev) DD_HIP(c, hipMemcpyAsync(c->y_stage, y_dev, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    *x_run = c->x_stage;
    *y_run = y_dev ? c->y_stage : nullptr;
    return DD_OK;
}

// dd_sample splits an even batch of at least 32 images into two half-batch chains (same-box A/B, profiles/r04/ab_chains.txt): the
// fused-path models (their kernels have strong MFMA / HBM phase structure and one tile per CU at the benchmark batch: CelebA B = 128
// +10.5 %) and GEMM-path models whose launches leave CUs idle (ImageNet-256 latents, B = 32: +9 %) as they are; GEMM-path models at
// large batches (ImageNet-64, B = 256) with the persistent GEMM grids of both chains sized for HALF the CUs (chain_gemm_cus): a full-size
// grid holds every CU's LDS, two of them only queue behind each other (-4.7 %), two half-size ones run side by side (+4.7 %).
// Development flags force the split on for any even batch, or switch it off.
bool use_chains(dd_ctx* c, dd_model* m, int B, bool early_exit_loop = false) {
    if ((c->dev_flags & DD_DEV_NO_CHAINS) || (B & 1) || B < 2 || ((m->ee_type >= 0) != early_exit_loop)) return false;
    return B >= 32 || (c->dev_flags & DD_DEV_FORCE_CHAINS);
}
int chain_gemm_cus(dd_ctx* c, dd_model* m, int B) {
    const bool fused = m->prec == DD_PREC_BF16 && m->fused_mlp;
    if (fused || (long long)B * m->L <= 32768) return c->num_cus;
    const int half = c->num_cus / 2 / 8 * 8;
    return half >= 8 ? half : c->num_cus;
}
// (zeroed ON THE LAUNCH STREAM: the second chain's stream is non-blocking, so a null-stream memset is not ordered before its first kernels --
// the chain's first step ran while the memset was still sweeping the arena: the first images of the second half came out wrong on the very
// first chained call of a model, intermittently)
int ensure_chain_ws(dd_ctx* c, dd_model* m, hipStream_t s) {
    if (m->wsarena2) return DD_OK;
    m->wsoff2 = ws_layout(m, (m->cfg.max_batch + 1) / 2);     // a chain never runs more than half of max_batch
    DD_HIP(c, hipMalloc((void**)&m->wsarena2, m->wsoff2.bytes));
    DD_HIP(c, hipMemsetAsync(m->wsarena2, 0, m->wsoff2.bytes, s));
    bind_ws(m->wsoff2, m->wsarena2, m->ws2);
    return DD_OK;
}
// a failure between the fork and the join of a chained call must not leave the side stream running on the second chain's buffers behind
// the caller's back: armed after the fork, disarmed by the regular join
struct SideJoin {
    dd_ctx* c; bool armed;
    ~SideJoin() { if (armed && c->side) (void)hipStreamSynchronize(c->side); }
};

// the model's captured step of kind `which` (0 DDPM, 1 table-driven) for this key: reused, or captured now from enqueue(m)
template <typename F>
int get_graph(dd_ctx* c, dd_model* m, int which, const GraphKey& key, hipStream_t s, F&& enqueue) {
    if (m->graph[which] && m->gkey[which] == key) return DD_OK;
    if (m->graph[which]) { (void)hipGraphExecDestroy(m->graph[which]); m->graph[which] = nullptr; }
    hipGraph_t g = nullptr;
    DD_HIP(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int r = enqueue(m);
    const hipError_t e2 = hipStreamEndCapture(s, &g);
    if (r) { if (g) (void)hipGraphDestroy(g); return r; }
    if (e2 != hipSuccess) return fail_hip(c, e2, "hipStreamEndCapture");
    const hipError_t e3 = hipGraphInstantiate(&m->graph[which], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e3 != hipSuccess) { m->graph[which] = nullptr; return fail_hip(c, e3, "hipGraphInstantiate"); }
    m->gkey[which] = key;
    ++c->graph_captures;
    return DD_OK;
}

}  // namespace

// ==========================================================================================
extern "C" {

int dd_abi_version(void) { return DD_ABI_VERSION; }

#ifndef DD_BUILD_ID
#define DD_BUILD_ID "unknown"
#endif
const char* dd_build_id(void) { return DD_BUILD_ID; }

int dd_schedule_table(int which, float* out) {
    if (!out) return DD_ERR_INVALID;
    const Schedule& s = schedule();
    const float* src = nullptr;
    switch (which) {
        case 0: src = s.betas; break;
        case 1: src = s.alphas; break;
        case 2: src = s.abar; break;
        case 3: src = s.abar_prev; break;
        case 4: src = s.bt_sampler; break;
        case 5: src = s.bt_sched; break;
        case 6: src = s.c1; break;
        case 7: src = s.c2; break;
        case 8: src = s.sigma; break;
        default: return DD_ERR_INVALID;
    }
    std::memcpy(out, src, 1000 * sizeof(float));
    return DD_OK;
}

int dd_schedule_build(float beta_init, float beta_final, int steps, float* betas, float* alphas, float* alphas_bar,
                      float* alphas_bar_prev, float* betas_tilde) {
    if (steps < 1 || steps > (1 << 20)) return DD_ERR_INVALID;
    std::vector<float> b(steps), a(steps), ab(steps), abp(steps);
    base_tables(beta_init, beta_final, steps, b.data(), a.data(), ab.data(), abp.data());
    const size_t bytes = (size_t)steps * sizeof(float);
    if (betas) std::memcpy(betas, b.data(), bytes);
    if (alphas) std::memcpy(alphas, a.data(), bytes);
    if (alphas_bar) std::memcpy(alphas_bar, ab.data(), bytes);
    if (alphas_bar_prev) std::memcpy(alphas_bar_prev, abp.data(), bytes);
    if (betas_tilde)
        for (int i = 0; i < steps; ++i) betas_tilde[i] = bt_scheduler_order(b[i], abp[i], ab[i]);
    return DD_OK;
}

int dd_ctx_create(int device, dd_ctx** out) {
    if (!out) return DD_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DD_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return DD_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DD_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return DD_ERR_UNSUPPORTED;  // gfx950 code objects only
    dd_ctx* c = new (std::nothrow) dd_ctx();
    if (!c) return DD_ERR_NOMEM;
    c->device = device;
    c->num_cus = device_num_cus();
    c->base_cus = c->num_cus;
    const Schedule& s = schedule();
    for (int i = 0; i < 1000; ++i) c->coef_host[i] = StepCoef{s.c1[i], s.c2[i], s.sigma[i], s.sigma_beta[i]};
    bool ok = hipMalloc(&c->st, sizeof(StepState)) == hipSuccess && hipMalloc(&c->coef, sizeof(StepCoef) * 1000) == hipSuccess &&
              hipMemcpy(c->coef, c->coef_host, sizeof(StepCoef) * 1000, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(c->st, 0, sizeof(StepState)) == hipSuccess && hipMalloc(&c->st2, sizeof(StepState)) == hipSuccess &&
              hipMemset(c->st2, 0, sizeof(StepState)) == hipSuccess &&
              hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_ee_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_ee_join, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 3; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    ok = ok && hipStreamSynchronize(nullptr) == hipSuccess;     // the null-stream memsets above, before any caller stream touches the state
    ok = ok && init_gemm_kernels() == hipSuccess && init_attention_kernels() == hipSuccess &&
         init_rowops_kernels() == hipSuccess && init_mlp_fused_kernels() == hipSuccess && init_rowlin_kernels() == hipSuccess;
    if (!ok) { dd_ctx_destroy(c); return DD_ERR_HIP; }
    *out = c;
    return DD_OK;
}

void dd_ctx_destroy(dd_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_ee_fork) (void)hipEventDestroy(c->ev_ee_fork);
    if (c->ev_ee_join) (void)hipEventDestroy(c->ev_ee_join);
    if (c->st) (void)hipFree(c->st);
    if (c->st2) (void)hipFree(c->st2);
    if (c->coef) (void)hipFree(c->coef);
    if (c->x_stage) (void)hipFree(c->x_stage);
    if (c->y_stage) (void)hipFree(c->y_stage);
    if (c->atab) (void)hipFree(c->atab);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

const char* dd_last_error(dd_ctx* c) { return c ? c->err.c_str() : "null context"; }

int dd_sync(dd_ctx* c, void* stream) {
    if (!c) return DD_ERR_INVALID;
    DD_HIP(c, hipStreamSynchronize((hipStream_t)stream));
    return DD_OK;
}

int dd_model_create(dd_ctx* c, const dd_config* cfg, dd_model** out) {
    if (!c || !cfg || !out) return DD_ERR_INVALID;
    *out = nullptr;
    const dd_config& g = *cfg;
    if (g.img_size <= 0 || g.patch_size <= 0 || g.img_size % g.patch_size) return fail(c, DD_ERR_INVALID, "img_size must be a positive multiple of patch_size");
    if (g.num_heads <= 0 || g.embed_dim != g.num_heads * 64) return fail(c, DD_ERR_UNSUPPORTED, "kernels require head_dim == 64 (all shipped configs)");
    if (g.depth < 1 || g.depth % 2 != 1) return fail(c, DD_ERR_INVALID, "depth must be odd");
    if (g.in_chans < 1 || g.in_chans > 4) return fail(c, DD_ERR_UNSUPPORTED, "in_chans must be 1..4");
    if (g.embed_dim > 1024) return fail(c, DD_ERR_UNSUPPORTED, "embed_dim must be <= 1024");
    if (g.max_batch < 1) return fail(c, DD_ERR_INVALID, "max_batch must be >= 1");
    dd_model* m = new (std::nothrow) dd_model();
    if (!m) return fail(c, DD_ERR_NOMEM, "out of host memory");
    m->ctx = c; m->cfg = g;
    m->D = g.embed_dim; m->H = g.num_heads;
    m->N = (g.img_size / g.patch_size) * (g.img_size / g.patch_size);
    m->extras = g.num_classes > 0 ? 2 : 1;
    m->L = m->N + m->extras;
    m->pd = g.patch_size * g.patch_size * g.in_chans;
    m->pdp = round_up(m->pd, 8);
    m->hidden = g.embed_dim * (g.mlp_ratio > 0 ? g.mlp_ratio : 4);
    m->hid_ld = m->hidden;   // row stride of the MLP hidden activation (a +64 pad against power-of-two strides measured no gain)
    m->half_depth = g.depth / 2;
    m->Mp_max = round_up(g.max_batch * m->L, 256);
    if (m->L > 288) { delete m; return fail(c, DD_ERR_UNSUPPORTED, "sequence length must be <= 288 tokens"); }
    if (m->pd > 64) { delete m; return fail(c, DD_ERR_UNSUPPORTED, "patch_size^2 * in_chans must be <= 64"); }
    *out = m;
    return DD_OK;
}

int dd_model_set_param(dd_model* m, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!m || !name || !host || !shape || ndim < 1) return DD_ERR_INVALID;
    dd_ctx* c = m->ctx;
    if (m->finalized) return fail(c, DD_ERR_STATE, "model already finalized");
    std::vector<int64_t> want;
    if (!expected_shape(m, name, want)) return fail(c, DD_ERR_NOT_FOUND, std::string("unexpected key in state_dict: ") + name);
    std::vector<int64_t> got(shape, shape + ndim);
    if (got != want) {
        std::string msg = std::string("size mismatch for ") + name + ": expected [";
        for (size_t i = 0; i < want.size(); ++i) msg += (i ? "," : "") + std::to_string(want[i]);
        msg += "], got [";
        for (size_t i = 0; i < got.size(); ++i) msg += (i ? "," : "") + std::to_string(got[i]);
        return fail(c, DD_ERR_INVALID, msg + "]");
    }
    size_t n = 1;
    for (auto d : want) n *= (size_t)d;
    HostParam& p = m->params[name];
    p.data.assign(host, host + n);
    p.shape = want;
    p.set = true;
    return DD_OK;
}

int64_t dd_model_num_params(const dd_model* m) {
    if (!m) return 0;
    int64_t n = 0;
    for (auto& kv : m->params) n += (int64_t)kv.second.data.size();
    return n;
}

int dd_model_finalize(dd_model* m, int precision) {
    if (!m) return DD_ERR_INVALID;
    dd_ctx* c = m->ctx;
    if (m->finalized) return fail(c, DD_ERR_STATE, "model already finalized");
    if (precision != DD_PREC_BF16 && precision != DD_PREC_FP32) return fail(c, DD_ERR_INVALID, "unknown precision");
    for (const std::string& nm : required_names(m)) {
        auto it = m->params.find(nm);
        if (it == m->params.end() || !it->second.set) return fail(c, DD_ERR_NOT_FOUND, "missing key in state_dict: " + nm);
    }
    DD_HIP(c, hipSetDevice(c->device));
    m->prec = precision;
    m->esize = precision == DD_PREC_BF16 ? 2 : 4;
    const size_t es = m->esize;
    const int D = m->D, hid = m->hidden, L = m->L;

    // ---- pack weights into one arena: fp32 vectors/tables + T-typed GEMM matrices
    std::vector<char> host;
    auto align = [&]() { host.resize((host.size() + 255) / 256 * 256); };
    auto put_f32 = [&](const float* src, size_t n) -> size_t {
        align(); const size_t off = host.size(); host.resize(off + n * 4); std::memcpy(&host[off], src, n * 4); return off;
    };
    auto put_mat = [&](const std::vector<float>& src) -> size_t {
        align(); const size_t off = host.size(); host.resize(off + src.size() * es);
        if (es == 4) std::memcpy(&host[off], src.data(), src.size() * 4);
        else { unsigned short* d = (unsigned short*)&host[off]; for (size_t i = 0; i < src.size(); ++i) d[i] = host_f2bf(src[i]); }
        return off;
    };
    auto P = [&](const std::string& nm) -> const std::vector<float>& { return m->params[nm].data; };

    // fused block tail (mlp_fused.hip): bf16 mode only (development A/B runs can switch it off: dd_dev_set_flags)
    m->fused_mlp = precision == DD_PREC_BF16 && mlp_fused_supported(D, hid) && !(c->dev_flags & DD_DEV_NO_FUSED_MLP);
    m->fused_proj = m->fused_mlp && D % 128 == 0 && !(c->dev_flags & DD_DEV_NO_FUSED_PROJ);
    m->fused_skip = m->fused_proj && (hid / 32) % 2 == 0 && !(c->dev_flags & DD_DEV_NO_FUSED_SKIP);     // (early-exit models too: y leaves through MlpFusedArgs::y_tap)
    m->fused_qkv = m->fused_proj && m->ee_type < 0 && (hid / 32) % 2 == 0 && !m->cfg.qkv_bias && !(c->dev_flags & DD_DEV_NO_FUSED_QKV);
    // (early-exit models too: their heads and probes read the residual stream between blocks, which this launch does not touch)
    // (embed_dim 768 / 1024 too, which have no fused block tail: their norm1 launch writes the fragment order, the qkv tensor is gone)
    m->fused_qa = precision == DD_PREC_BF16 && qkv_attention_supported(D, m->H, L, m->extras) && !(c->dev_flags & DD_DEV_NO_FUSED_QA);
    // (embed_dim 768, no fused block tail) mlp.fc2 with the residual rows resident in registers: x read and written once, the next norm1 from registers
    m->rowlin_fc2 = precision == DD_PREC_BF16 && !m->fused_mlp && rowlin_supported(D, hid) && m->N % 32 == 0 && !(c->dev_flags & DD_DEV_NO_ROWLIN);
    m->rowlin_proj = m->rowlin_fc2 && !(c->dev_flags & DD_DEV_NO_ROWLIN_PROJ);
    m->rowlin_skip = m->rowlin_fc2 && rowlin_supported(D, 2 * D) && !(c->dev_flags & DD_DEV_NO_ROWLIN_SKIP);
    // split-K for the N = embed_dim Linears (skip_linear, attn.proj, mlp.fc2) where even max_batch leaves half of the CUs without a 256 x 256
    // tile (ImageNet-256 latents: 32 x 4 tiles): a function of the model (max_batch), never of a call's batch
    m->splitk = precision == DD_PREC_BF16 && !m->fused_mlp && !m->rowlin_fc2 && D % 256 == 0 && !(c->dev_flags & DD_DEV_NO_SPLITK) &&
                (long long)(m->cfg.max_batch * L / 256) * (D / 256) * 2 <= device_num_cus() &&
                (hid / 64) % 2 == 0;
    for (int b = 1; m->splitk && b <= m->cfg.max_batch; ++b)     // every batch this model can be called with must fit the kernel's row partition
        m->splitk = gemm_splitk_supported(b * L, D, D, D, 2);
    auto put_raw = [&](size_t bytes) -> size_t { align(); const size_t off = host.size(); host.resize(off + bytes, 0); return off; };
    struct BlockOff { size_t ln1_g, ln1_b, ln2_g, ln2_b, proj_b, fc1_b, fc2_b, skip_b, qkv_w, proj_w, fc1_w, fc2_w, skip_w, mlp_img, mlp_b1p, qkv_b, qa_img, rl_img, rlp_img, rls_img; bool skip; };
    std::vector<BlockOff> boffs;
    // next_skip: prefix of the block whose skip_linear runs in THIS block's fused launch ("" = none)
    // next_qkv: prefix of the block whose attn.qkv runs in THIS block's fused launch ("" = none: the last block)
    auto pack_block = [&](const std::string& p, bool skip, const std::string& next_skip, const std::string& next_qkv) {
        BlockOff o{};
        o.skip = skip;
        o.ln1_g = put_f32(P(p + "norm1.weight").data(), D); o.ln1_b = put_f32(P(p + "norm1.bias").data(), D);
        o.ln2_g = put_f32(P(p + "norm2.weight").data(), D); o.ln2_b = put_f32(P(p + "norm2.bias").data(), D);
        o.proj_b = put_f32(P(p + "attn.proj.bias").data(), D);
        o.fc1_b = put_f32(P(p + "mlp.fc1.bias").data(), hid); o.fc2_b = put_f32(P(p + "mlp.fc2.bias").data(), D);
        o.qkv_w = put_mat(P(p + "attn.qkv.weight")); o.proj_w = put_mat(P(p + "attn.proj.weight"));
        if (m->cfg.qkv_bias) o.qkv_b = put_f32(P(p + "attn.qkv.bias").data(), 3 * (size_t)D);
        o.fc1_w = put_mat(P(p + "mlp.fc1.weight")); o.fc2_w = put_mat(P(p + "mlp.fc2.weight"));
        if (skip) { o.skip_b = put_f32(P(p + "skip_linear.bias").data(), D); o.skip_w = put_mat(P(p + "skip_linear.weight")); }
        if (m->fused_mlp) {
            const bool with_skip = m->fused_skip && !next_skip.empty();
            // (an out-block's qkv needs its skip_linear in here too; with fused_qa the attention launch computes qkv and no section is packed)
            const bool with_qkv = m->fused_qkv && !m->fused_qa && !next_qkv.empty() && (next_skip.empty() || with_skip);
            o.mlp_img = put_raw(mlp_fused_image_bytes(D, hid, m->fused_proj, with_skip, with_qkv));
            o.mlp_b1p = put_raw((size_t)hid * 4);
            const size_t proj_bytes = m->fused_proj ? (size_t)D * D * 2 : 0;      // D/32 blocks of Wproj lead the stream
            if (m->fused_proj) mlp_fused_pack_proj(D, P(p + "attn.proj.weight").data(), host_f2bf, (unsigned short*)&host[o.mlp_img]);
            mlp_fused_pack(D, hid, P(p + "mlp.fc1.weight").data(), P(p + "mlp.fc1.bias").data(), P(p + "mlp.fc2.weight").data(),
                           true, host_f2bf, (unsigned short*)&host[o.mlp_img + proj_bytes], (float*)&host[o.mlp_b1p]);
            if (with_skip)    // the next block's skip_linear: 2 D/32 blocks behind the MLP blocks
                mlp_fused_pack_skip(D, P(next_skip + "skip_linear.weight").data(), host_f2bf,
                                    (unsigned short*)&host[o.mlp_img + proj_bytes + (size_t)(hid / 32) * 2 * (D / 16) * 1024]);
            if (with_qkv)     // the next block's attn.qkv: 3 D/32 blocks closing the image
                mlp_fused_pack_rows(D, 3 * D, P(next_qkv + "attn.qkv.weight").data(), host_f2bf,
                                    (unsigned short*)&host[o.mlp_img + proj_bytes + ((size_t)(hid / 32) * 2 + (with_skip ? D / 16 : 0)) * (D / 16) * 1024]);
        }
        if (m->fused_qa) {
            o.qa_img = put_raw((size_t)3 * D * D * 2);
            qkv_attention_pack(D, m->H, P(p + "attn.qkv.weight").data(), host_f2bf, (unsigned short*)&host[o.qa_img]);
        }
        if (m->rowlin_fc2) {
            o.rl_img = put_raw((size_t)D * hid * 2);
            rowlin_pack(hid, P(p + "mlp.fc2.weight").data(), host_f2bf, (unsigned short*)&host[o.rl_img]);
            o.rlp_img = put_raw((size_t)D * D * 2);
            rowlin_pack(D, P(p + "attn.proj.weight").data(), host_f2bf, (unsigned short*)&host[o.rlp_img]);
            if (skip) {
                o.rls_img = put_raw((size_t)D * 2 * D * 2);
                rowlin_pack(2 * D, P(p + "skip_linear.weight").data(), host_f2bf, (unsigned short*)&host[o.rls_img]);
            }
        }
        boffs.push_back(o);
    };
    auto in_name = [&](int i) { return "in_blocks." + std::to_string(i) + "."; };
    auto out_name = [&](int i) { return "out_blocks." + std::to_string(i) + "."; };
    for (int i = 0; i < m->half_depth; ++i) pack_block(in_name(i), false, "", i + 1 < m->half_depth ? in_name(i + 1) : std::string("mid_block."));
    pack_block("mid_block.", false, m->half_depth > 0 ? out_name(0) : "", m->half_depth > 0 ? out_name(0) : "");
    for (int i = 0; i < m->half_depth; ++i)
        pack_block(out_name(i), true, i + 1 < m->half_depth ? out_name(i + 1) : "", i + 1 < m->half_depth ? out_name(i + 1) : "");

    // patch-embed weight [D, pd] -> transposed [pd, D] (coalesced over D in the embed kernel)
    std::vector<float> wt((size_t)m->pd * D);
    const std::vector<float>& pe = P("patch_embed.proj.weight");
    for (int d = 0; d < D; ++d) for (int k = 0; k < m->pd; ++k) wt[(size_t)k * D + d] = pe[(size_t)d * m->pd + k];
    const size_t o_wt = put_f32(wt.data(), wt.size()), o_eb = put_f32(P("patch_embed.proj.bias").data(), D);
    const size_t o_pos = put_f32(P("pos_embed").data(), (size_t)L * D);
    const size_t o_lab = m->cfg.num_classes > 0 ? put_f32(P("label_emb.weight").data(), (size_t)m->cfg.num_classes * D) : 0;
    size_t o_tm[4] = {0, 0, 0, 0};
    if (m->cfg.mlp_time_embed) {   // transposed: the kernel's threads run over the OUTPUT index
        const std::vector<float>&w1 = P("time_embed.0.weight"), &w2 = P("time_embed.2.weight");
        std::vector<float> w1t((size_t)D * 4 * D), w2t((size_t)4 * D * D);
        for (int j = 0; j < 4 * D; ++j) for (int k = 0; k < D; ++k) w1t[(size_t)k * 4 * D + j] = w1[(size_t)j * D + k];
        for (int d = 0; d < D; ++d) for (int k = 0; k < 4 * D; ++k) w2t[(size_t)k * D + d] = w2[(size_t)d * 4 * D + k];
        o_tm[0] = put_f32(w1t.data(), w1t.size()); o_tm[1] = put_f32(P("time_embed.0.bias").data(), 4 * (size_t)D);
        o_tm[2] = put_f32(w2t.data(), w2t.size()); o_tm[3] = put_f32(P("time_embed.2.bias").data(), D);
    }
    const size_t o_ng = put_f32(P("norm.weight").data(), D), o_nb = put_f32(P("norm.bias").data(), D);
    // head_dec_kernel operands: the final norm's affine part folded into decoder_pred (dec = Wg . xn + c)
    const bool fused_head = head_dec_supported(D, m->pd) && !(c->dev_flags & DD_DEV_NO_FUSED_HEAD);
    size_t o_wg = 0, o_dc = 0;
    if (fused_head) {
        const std::vector<float>&wd = P("decoder_pred.weight"), &bd = P("decoder_pred.bias"), &ng = P("norm.weight"), &nbv = P("norm.bias");
        std::vector<float> wg, dc;
        fold_head_norm(D, m->pd, wd.data(), bd.data(), ng.data(), nbv.data(), wg, dc);
        o_wg = put_f32(wg.data(), wg.size()); o_dc = put_f32(dc.data(), dc.size());
    }
    const size_t o_wdec = put_f32(P("decoder_pred.weight").data(), (size_t)m->pd * D), o_bd = put_f32(P("decoder_pred.bias").data(), m->pd);
    const size_t o_wc = put_f32(P("final_layer.weight").data(), P("final_layer.weight").size());
    const size_t o_bc = put_f32(P("final_layer.bias").data(), m->cfg.in_chans);
    struct HeadOff { size_t ng, nb, wdec, bdec, wconv, bconv, wg, dc, ws, dcs; };
    const bool split_heads = fused_head && es == 2 && head_dec_probe_supported(D) && !(c->dev_flags & DD_DEV_NO_SPLIT_HEADS);
    struct AttnProbeOff { size_t u, wvt, bv, w0t, b0, w2, b2; };
    std::vector<AttnProbeOff> aoffs;
    std::vector<HeadOff> hoffs;
    size_t o_pw = 0, o_pb = 0;
    if (m->ee_type >= 0) {
        for (int layer = 0; layer < m->cfg.depth; ++layer) {
            const std::string p = head_prefix(m, layer);
            HeadOff o{};
            o.ng = put_f32(P(p + "norm.weight").data(), D); o.nb = put_f32(P(p + "norm.bias").data(), D);
            o.wdec = put_f32(P(p + "decoder_pred.weight").data(), (size_t)m->pd * D); o.bdec = put_f32(P(p + "decoder_pred.bias").data(), m->pd);
            o.wconv = put_f32(P(p + "final_layer.weight").data(), P(p + "final_layer.weight").size());
            o.bconv = put_f32(P(p + "final_layer.bias").data(), m->cfg.in_chans);
            if (fused_head) {   // as the final head: dec = (W . diag(gamma)) xn + (b + W . beta)
                const std::vector<float>&wd = P(p + "decoder_pred.weight"), &bd = P(p + "decoder_pred.bias"), &ng = P(p + "norm.weight"), &nbv = P(p + "norm.bias");
                std::vector<float> wg, dc;
                fold_head_norm(D, m->pd, wd.data(), bd.data(), ng.data(), nbv.data(), wg, dc);
                o.wg = put_f32(wg.data(), wg.size()); o.dc = put_f32(dc.data(), dc.size());
                if (split_heads) {      // the early-exit heads of the bf16 engine: Wg as hi + lo bf16 halves in the SPLIT kernel's fragment order
                    const int nt = (m->pd + 15) / 16;
                    std::vector<unsigned short> img((size_t)(D / 32) * nt * 2 * 64 * 8);
                    std::vector<float> dcs(dc);
                    pack_head_split(D, m->pd, wg.data(), host_f2bf, img.data(), dcs.data() + m->pd);
                    o.ws = put_f32(reinterpret_cast<const float*>(img.data()), img.size() / 2); o.dcs = put_f32(dcs.data(), dcs.size());
                }
            }
            hoffs.push_back(o);
        }
        if (m->ee_type == DD_EE_ATTENTION_PROBE) {
            // AttentionProbe (early_exit.py:40-80), one learned query, one head.  q . (Wk x + bk) = (Wk^T q) . x + const, and the
            // constant cancels in the softmax; sum_l p_l (Wv x_l + bv) = Wv (sum_l p_l x_l) + bv.  So the probe needs u = Wk^T q /
            // sqrt(D) (folded here, in double), and Wv / classification.0 transposed for coalesced mat-vecs -- never the [L, 2D] kv.
            for (int layer = 0; layer < m->cfg.depth; ++layer) {
                const std::string pre = "matrix." + std::to_string(layer) + ".";
                const std::vector<float>&q = P(pre + "q"), &wkv = P(pre + "weight_kv.weight"), &bkv = P(pre + "weight_kv.bias"),
                                        &w0 = P(pre + "classification.0.weight");
                std::vector<float> u(D), wvt((size_t)D * D), w0t((size_t)D * D);
                const double scale = 1.0 / std::sqrt((double)D);
                for (int k = 0; k < D; ++k) {
                    double acc = 0.0;
                    for (int j = 0; j < D; ++j) acc += (double)q[j] * (double)wkv[(size_t)j * D + k];
                    u[k] = (float)(acc * scale);
                }
                for (int j = 0; j < D; ++j)
                    for (int k = 0; k < D; ++k) {
                        wvt[(size_t)k * D + j] = wkv[(size_t)(D + j) * D + k];
                        w0t[(size_t)k * D + j] = w0[(size_t)j * D + k];
                    }
                AttnProbeOff o{};
                o.u = put_f32(u.data(), D); o.wvt = put_f32(wvt.data(), wvt.size()); o.bv = put_f32(bkv.data() + D, D);
                o.w0t = put_f32(w0t.data(), w0t.size()); o.b0 = put_f32(P(pre + "classification.0.bias").data(), D);
                o.w2 = put_f32(P(pre + "classification.2.weight").data(), D); o.b2 = put_f32(P(pre + "classification.2.bias").data(), 1);
                aoffs.push_back(o);
            }
        } else {
        const int nt = m->ee_type == DD_EE_MLP_PER_LAYER ? 1 : 1000, nl = m->ee_type == DD_EE_MLP_PER_TIMESTEP ? 1 : m->cfg.depth;
        m->n_probe = nt * nl;
        std::vector<float> pw((size_t)m->n_probe * D), pb(m->n_probe);
        for (int t = 0; t < nt; ++t)
            for (int layer = 0; layer < nl; ++layer) {
                const std::string key = probe_key(m, layer, t);
                const int row = probe_index(m, key);
                std::memcpy(&pw[(size_t)row * D], P("matrix." + key + ".classifier.0.weight").data(), (size_t)D * 4);
                pb[row] = P("matrix." + key + ".classifier.0.bias")[0];
            }
        o_pw = put_f32(pw.data(), pw.size());
        o_pb = put_f32(pb.data(), pb.size());
        }
    }
    align();

    DD_HIP(c, hipMalloc((void**)&m->warena, host.size()));
    DD_HIP(c, hipMemcpy(m->warena, host.data(), host.size(), hipMemcpyHostToDevice));
    auto F = [&](size_t off) { return (const float*)(m->warena + off); };
    auto V = [&](size_t off) { return (const void*)(m->warena + off); };
    for (const BlockOff& o : boffs) {
        BlockW w{F(o.ln1_g), F(o.ln1_b), F(o.ln2_g), F(o.ln2_b), F(o.proj_b), F(o.fc1_b), F(o.fc2_b),
                 o.skip ? F(o.skip_b) : nullptr, m->cfg.qkv_bias ? F(o.qkv_b) : nullptr, V(o.qkv_w), V(o.proj_w), V(o.fc1_w), V(o.fc2_w),
                 o.skip ? V(o.skip_w) : nullptr, m->fused_mlp ? (const char*)V(o.mlp_img) : nullptr,
                 m->fused_mlp ? F(o.mlp_b1p) : nullptr, m->fused_qa ? (const bf16_t*)V(o.qa_img) : nullptr,
                 m->rowlin_fc2 ? (const char*)V(o.rl_img) : nullptr, m->rowlin_fc2 ? (const char*)V(o.rlp_img) : nullptr,
                 m->rowlin_fc2 && o.skip ? (const char*)V(o.rls_img) : nullptr};
        m->blocks.push_back(w);
    }
    if (m->cfg.mlp_time_embed) { m->tm_w1t = F(o_tm[0]); m->tm_b1 = F(o_tm[1]); m->tm_w2t = F(o_tm[2]); m->tm_b2 = F(o_tm[3]); }
    m->emb_wt = F(o_wt); m->emb_b = F(o_eb); m->pos = F(o_pos); m->label = m->cfg.num_classes > 0 ? F(o_lab) : nullptr;
    for (const HeadOff& o : hoffs) m->heads.push_back(HeadW{F(o.ng), F(o.nb), F(o.wdec), F(o.bdec), F(o.wconv), F(o.bconv), fused_head ? F(o.wg) : nullptr, fused_head ? F(o.dc) : nullptr,
                                                            split_heads ? F(o.ws) : nullptr, split_heads ? F(o.dcs) : nullptr});
    if (m->heads.size() >= 2) {
        m->ee_wconv_stride = m->heads[1].wconv - m->heads[0].wconv; m->ee_bconv_stride = m->heads[1].bconv - m->heads[0].bconv;
        m->ee_conv_stride_ok = true;
        for (size_t i = 0; i < m->heads.size(); ++i)
            if (m->heads[i].wconv != m->heads[0].wconv + (long long)i * m->ee_wconv_stride || m->heads[i].bconv != m->heads[0].bconv + (long long)i * m->ee_bconv_stride) m->ee_conv_stride_ok = false;
    }
    if (m->ee_type >= 0 && m->ee_type != DD_EE_ATTENTION_PROBE) { m->probe_w = F(o_pw); m->probe_b = F(o_pb); }
    for (const AttnProbeOff& o : aoffs) m->attn_probes.push_back(AttnProbeW{F(o.u), F(o.wvt), F(o.bv), F(o.w0t), F(o.b0), F(o.w2), F(o.b2)});
    if (fused_head) { m->wdec_g = F(o_wg); m->dec_c = F(o_dc); }
    m->norm_g = F(o_ng); m->norm_b = F(o_nb); m->wdec = F(o_wdec); m->bdec = F(o_bd); m->wconv = F(o_wc); m->bconv = F(o_bc);

    // ---- activation workspace (HBM-resident for the life of the model)
    m->wsoff = ws_layout(m, m->cfg.max_batch);
    DD_HIP(c, hipMalloc((void**)&m->wsarena, m->wsoff.bytes));
    DD_HIP(c, hipMemset(m->wsarena, 0, m->wsoff.bytes));
    DD_HIP(c, hipStreamSynchronize(nullptr));    // (callers run the model on non-blocking streams, which a null-stream memset does not order itself before)
    {
        WsPtrs w;
        bind_ws(m->wsoff, m->wsarena, w);
        m->x = w.x; m->h = w.h; m->ao = w.ao; m->qkv = w.qkv; m->hid = w.hid; m->xb = w.xb; m->dec = w.dec; m->skips = w.skips;
        m->mlp_partial = w.mlp_partial; m->qkv_dump = w.qkv_dump; m->hfrag = w.hfrag; m->ytap = w.ytap;
    }
    m->mlp_partial_bytes = m->wsoff.part_bytes;

    // host copies are no longer needed
    for (auto& kv : m->params) { std::vector<float>().swap(kv.second.data); }
    m->finalized = true;
    return DD_OK;
}

void dd_model_destroy(dd_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    for (auto g : m->graph) if (g) (void)hipGraphExecDestroy(g);
    for (hipEvent_t e : m->fc1_events) (void)hipEventDestroy(e);
    if (m->ee_ws) (void)hipFree(m->ee_ws);
    if (m->ee_sums) (void)hipFree(m->ee_sums);
    if (m->warena) (void)hipFree(m->warena);
    if (m->wsarena) (void)hipFree(m->wsarena);
    if (m->wsarena2) (void)hipFree(m->wsarena2);
    delete m;
}

int dd_forward(dd_ctx* c, dd_model* m, const float* x_dev, float t, const float* t_dev, const int64_t* y_dev,
               float* eps_dev, int B, void* stream) {
    int rc = check_call(c, m, B, y_dev);
    if (rc) return rc;
    if (!x_dev || !eps_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    DD_HIP(c, launch_set_state_float(c->st, t, s));
    rc = run_model(m, x_dev, t_dev, y_dev, B, s);
    if (rc) return rc;
    FinalArgs fa{m->dec, m->wconv, m->bconv, nullptr, nullptr, eps_dev, nullptr, c->st, c->coef,
                 B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, m->L, m->extras, DD_NOISE_NONE, 0, 0};
    DD_HIP(c, launch_final(fa, s));
    return DD_OK;
}

int dd_model_enable_early_exit(dd_model* m, int classifier_type) {
    if (!m) return DD_ERR_INVALID;
    dd_ctx* c = m->ctx;
    if (m->finalized || !m->params.empty()) return fail(c, DD_ERR_STATE, "enable early exit before any parameter is set");
    if (classifier_type < DD_EE_MLP_PER_LAYER || classifier_type > DD_EE_ATTENTION_PROBE)
        return fail(c, DD_ERR_UNSUPPORTED, "unknown classifier type");
    m->ee_type = classifier_type;
    return DD_OK;
}

int dd_forward_early_exit(dd_ctx* c, dd_model* m, const float* x_dev, float t, const float* t_dev, const int64_t* y_dev,
                          float* eps_dev, float* classifier_dev, float* outputs_dev, int B, void* stream) {
    int rc = check_call(c, m, B, y_dev);
    if (rc) return rc;
    if (m->ee_type < 0) return fail(c, DD_ERR_STATE, "model was not created with dd_model_enable_early_exit");
    if (!x_dev || !eps_dev || !classifier_dev || !outputs_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    const int ti = (int)t;                                   // t = int(timesteps[0]) (early_exit.py:271)
    if (m->ee_type != DD_EE_MLP_PER_LAYER && m->ee_type != DD_EE_ATTENTION_PROBE && (ti < 0 || ti > 999)) return fail(c, DD_ERR_NOT_FOUND, "no probe for this timestep (KeyError in the reference)");
    hipStream_t s = (hipStream_t)stream;
    DD_HIP(c, launch_set_state_float(c->st, t, s));
    const EeTaps ee{classifier_dev, outputs_dev, ti};
    rc = run_model(m, x_dev, t_dev, y_dev, B, s, &ee);
    if (rc) return rc;
    FinalArgs fa{m->dec, m->wconv, m->bconv, nullptr, nullptr, eps_dev, nullptr, c->st, c->coef,
                 B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, m->L, m->extras, DD_NOISE_NONE, 0, 0};
    DD_HIP(c, launch_final(fa, s));
    return DD_OK;
}

int dd_early_exit_select(dd_ctx* c, const float* outputs_dev, const float* eps_dev, const float* classifier_dev,
                         float threshold, int depth, int B, int64_t chw, float* model_output_dev, int32_t* indices_dev,
                         float* err_mean_dev, void* stream) {
    if (!c) return DD_ERR_INVALID;
    if (!outputs_dev || !eps_dev || !classifier_dev || !model_output_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    if (depth < 1 || B < 1 || chw < 1) return fail(c, DD_ERR_INVALID, "depth, B and chw must be positive");
    DD_HIP(c, launch_ee_select(outputs_dev, eps_dev, classifier_dev, threshold, depth, B, (long long)chw, model_output_dev,
                               indices_dev, err_mean_dev, nullptr, (hipStream_t)stream));
    return DD_OK;
}

int dd_ddpm_step(dd_ctx* c, const float* x_dev, const float* eps_dev, const float* z_dev, int t, int variance,
                 float* x_out_dev, int64_t n, void* stream) {
    if (!c) return DD_ERR_INVALID;
    if (!x_dev || !eps_dev || !x_out_dev || n < 0) return fail(c, DD_ERR_INVALID, "null tensor");
    if (t < 0 || t > 999) return fail(c, DD_ERR_INVALID, "timestep outside [0, 999]");
    StepCoef cf = c->coef_host[t];
    if (variance == DD_VAR_BETA) cf.sigma_tilde = cf.sigma_beta;
    const int use_noise = (t > 0 && z_dev) ? 1 : 0;
    if (n == 0) return DD_OK;
    DD_HIP(c, launch_ddpm_step(x_dev, eps_dev, z_dev, x_out_dev, cf, use_noise, (long long)n, (hipStream_t)stream));
    return DD_OK;
}

int dd_ddpm_step_coef(dd_ctx* c, const float* x_dev, const float* eps_dev, const float* z_dev, float c1, float c2,
                      float sigma, float* x_out_dev, int64_t n, void* stream) {
    if (!c) return DD_ERR_INVALID;
    if (!x_dev || !eps_dev || !x_out_dev || n < 0) return fail(c, DD_ERR_INVALID, "null tensor");
    if (n == 0) return DD_OK;
    const StepCoef cf{c1, c2, sigma, sigma};
    DD_HIP(c, launch_ddpm_step(x_dev, eps_dev, z_dev, x_out_dev, cf, z_dev ? 1 : 0, (long long)n, (hipStream_t)stream));
    return DD_OK;
}

int dd_affine_step(dd_ctx* c, const float* x_dev, const float* m_dev, const float* z_dev, float a, float b, float cc,
                   float* out_dev, int64_t n, void* stream) {
    if (!c) return DD_ERR_INVALID;
    if (!x_dev || !m_dev || !out_dev || n < 0) return fail(c, DD_ERR_INVALID, "null tensor");
    if (n == 0) return DD_OK;
    DD_HIP(c, launch_affine_step(x_dev, m_dev, z_dev, out_dev, a, b, cc, (long long)n, (hipStream_t)stream));
    return DD_OK;
}

int dd_to_images(dd_ctx* c, const float* x_dev, float* images_dev, int B, int C, int S, void* stream) {
    if (!c) return DD_ERR_INVALID;
    if (!x_dev || !images_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    if (B < 0 || C < 1 || S < 1) return fail(c, DD_ERR_INVALID, "bad image shape");
    if (B == 0) return DD_OK;
    DD_HIP(c, launch_to_images(x_dev, images_dev, B, C, S, (hipStream_t)stream));
    return DD_OK;
}

int dd_sample_step(dd_ctx* c, dd_model* m, float* x_dev, int t, const int64_t* y_dev, int noise_mode, const float* z_dev,
                   uint64_t seed, int variance, float* eps_out_dev, int B, void* stream) {
    int rc = check_call(c, m, B, y_dev);
    if (rc) return rc;
    if (!x_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    if (t < 0 || t > 999) return fail(c, DD_ERR_INVALID, "timestep outside [0, 999]");
    if (noise_mode == DD_NOISE_BUFFER && !z_dev && t > 0) return fail(c, DD_ERR_INVALID, "DD_NOISE_BUFFER needs z_dev");
    hipStream_t s = (hipStream_t)stream;
    DD_HIP(c, launch_set_state(c->st, t, (unsigned long long)seed, s));
    return enqueue_step(c, m, x_dev, y_dev, noise_mode, z_dev, variance, eps_out_dev, B, s);
}

int dd_sample(dd_ctx* c, const dd_sample_args* a, void* stream) {
    if (!c || !a) return DD_ERR_INVALID;
    int rc = check_call(c, a->first, a->B, a->y_dev);
    if (rc) return rc;
    if (a->late && (rc = check_call(c, a->late, a->B, a->y_dev))) return rc;
    if (!a->x_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    if (a->t_start > 999 || a->t_end < 0 || a->t_end > a->t_start) return fail(c, DD_ERR_INVALID, "need 999 >= t_start >= t_end >= 0");
    if (a->noise_mode != DD_NOISE_PHILOX && a->noise_mode != DD_NOISE_NONE)
        return fail(c, DD_ERR_INVALID, "dd_sample generates noise on the device; for host noise drive dd_sample_step");
    if (a->late) {
        const dd_config &f = a->first->cfg, &l = a->late->cfg;
        if (f.img_size != l.img_size || f.in_chans != l.in_chans) return fail(c, DD_ERR_INVALID, "first and late model disagree on image shape");
    }
    hipStream_t s = (hipStream_t)stream;
    const bool switching = a->late && a->t_switch > 0 && a->t_switch <= 1000;
    const int t_sw = 1000 - a->t_switch;  // the late model takes over AFTER this step (sampler.py:135-136)

    // the loop runs on x_run / y_run: with graphs the context's staging buffers (copied in here, copied back at the end)
    float* x_run = a->x_dev;
    const int64_t* y_run = a->y_dev;
    const size_t chw = (size_t)a->first->cfg.in_chans * a->first->cfg.img_size * a->first->cfg.img_size;
    const size_t x_elems = (size_t)a->B * chw;
    // Two half-batch chains (graph replays only).  Images are independent and a row's path through the kernels does not depend on the
    // batch size, so chain 0 = images [0, B/2) on the caller's stream and chain 1 = images [B/2, B) on the context's side stream
    // compute bit for bit what the undivided batch computes (Philox pixel ids carry the image offset) -- with the two chains free
    // to drift apart, so that one's HBM-bound phases (row prologues / epilogues, attention row fetch) run under the other's MFMA phases.
    const bool chained = a->use_graph && use_chains(c, a->first, a->B) && (!switching || use_chains(c, a->late, a->B));
    const int B0 = chained ? a->B / 2 : a->B, B1 = a->B - B0;
    c->last_chains = chained ? 2 : 1;
    if (a->use_graph) {
        if ((rc = stage_inputs(c, a->x_dev, a->y_dev, a->B, x_elems, s, &x_run, &y_run))) return rc;
        // (the captured persistent GEMM grids are sized from c->num_cus: halved for both chains of a large GEMM-path batch)
        struct CusGuard { dd_ctx* c; int saved; ~CusGuard() { c->num_cus = saved; } } cus_guard{c, c->num_cus};
        if (chained) c->num_cus = std::min(chain_gemm_cus(c, a->first, a->B), switching ? chain_gemm_cus(c, a->late, a->B) : c->num_cus);
        GraphKey key{x_run, y_run, B0, a->noise_mode, a->variance, c->num_cus, nullptr};
        auto step = [&](dd_model* m) { return enqueue_step(c, m, x_run, y_run, a->noise_mode, nullptr, a->variance, nullptr, B0, s, 1); };
        if ((rc = get_graph(c, a->first, 0, key, s, step))) return rc;
        if (switching && (rc = get_graph(c, a->late, 0, key, s, step))) return rc;
        if (chained) {
            float* x1 = x_run + (size_t)B0 * chw;
            const int64_t* y1 = y_run ? y_run + B0 : nullptr;
            GraphKey key1{x1, y1, B1, a->noise_mode, a->variance, c->num_cus, nullptr};
            key1.b0 = B0;
            auto step1 = [&](dd_model* m) {     // the same launch sequence on the second chain's workspace and step state
                swap_chain(m); std::swap(c->st, c->st2);
                const int r = enqueue_step(c, m, x1, y1, a->noise_mode, nullptr, a->variance, nullptr, B1, s, 1, nullptr, B0);
                swap_chain(m); std::swap(c->st, c->st2);
                return r;
            };
            if ((rc = ensure_chain_ws(c, a->first, s)) || (rc = get_graph(c, a->first, 3, key1, s, step1))) return rc;
            if (switching && ((rc = ensure_chain_ws(c, a->late, s)) || (rc = get_graph(c, a->late, 3, key1, s, step1)))) return rc;
        }
    }
    DD_HIP(c, launch_set_state(c->st, a->t_start, (unsigned long long)a->seed, s));
    if (chained) {
        DD_HIP(c, launch_set_state(c->st2, a->t_start, (unsigned long long)a->seed, s));
        DD_HIP(c, hipEventRecord(c->ev_fork, s));                 // the side stream starts behind the staging copies and the state
        DD_HIP(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    SideJoin side_join{c, chained};
    DD_HIP(c, hipEventRecord(c->ev[0], s));
    bool marked = false;
    dd_model* cur = a->first;
    for (int t = a->t_start; t >= a->t_end; --t) {
        if (a->use_graph) {
            DD_HIP(c, hipGraphLaunch(cur->graph[0], s));
            if (chained) DD_HIP(c, hipGraphLaunch(cur->graph[3], c->side));
        } else {
            rc = enqueue_step(c, cur, x_run, y_run, a->noise_mode, nullptr, a->variance, nullptr, a->B, s, 1);
            if (rc) return rc;
        }
        if (switching && t == t_sw) {
            cur = a->late;
            DD_HIP(c, hipEventRecord(c->ev[1], s));
            marked = true;
        }
    }
    if (chained) {
        DD_HIP(c, hipEventRecord(c->ev_join, c->side));
        DD_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
        side_join.armed = false;
    }
    if (!marked) DD_HIP(c, hipEventRecord(c->ev[1], s));
    DD_HIP(c, hipEventRecord(c->ev[2], s));
    if (x_run != a->x_dev) DD_HIP(c, hipMemcpyAsync(a->x_dev, x_run, x_elems * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DD_OK;
}

int dd_sample_affine(dd_ctx* c, const dd_affine_sample_args* a, void* stream) {
    if (!c || !a) return DD_ERR_INVALID;
    int rc = check_call(c, a->first, a->B, a->y_dev);
    if (rc) return rc;
    if (a->late && (rc = check_call(c, a->late, a->B, a->y_dev))) return rc;
    if (!a->x_dev || !a->t || !a->a || !a->b || !a->c || !a->noise) return fail(c, DD_ERR_INVALID, "null tensor / table");
    if (a->n_steps < 1 || a->n_steps > (1 << 20)) return fail(c, DD_ERR_INVALID, "n_steps outside [1, 2^20]");
    if (a->counter_base < 0 || a->counter_base > (1 << 20)) return fail(c, DD_ERR_INVALID, "counter_base outside [0, 2^20]");
    if (a->noise_mode != DD_NOISE_PHILOX && a->noise_mode != DD_NOISE_NONE)
        return fail(c, DD_ERR_INVALID, "dd_sample_affine generates noise on the device; for host noise drive dd_forward + dd_affine_step");
    if (a->late) {
        const dd_config &f = a->first->cfg, &l = a->late->cfg;
        if (f.img_size != l.img_size || f.in_chans != l.in_chans) return fail(c, DD_ERR_INVALID, "first and late model disagree on image shape");
    }
    hipStream_t s = (hipStream_t)stream;
    const int n = a->n_steps;
    const bool switching = a->late && a->switch_after >= 0 && a->switch_after < n;
    // the table: n rows + one more whose timestep the last step hands on (never used)
    if (c->atab_rows < (size_t)n + 1) {
        if (c->atab) (void)hipFree(c->atab);
        c->atab = nullptr; c->atab_rows = 0;
        DD_HIP(c, hipMalloc((void**)&c->atab, ((size_t)n + 1) * sizeof(AffineRow)));
        c->atab_rows = (size_t)n + 1;
    }
    DD_HIP(c, hipStreamSynchronize(s));            // a previous call's upload may still read the host staging copy
    c->atab_host.assign((size_t)n + 1, AffineRow{0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0});
    for (int k = 0; k < n; ++k) c->atab_host[k] = AffineRow{a->t[k], a->a[k], a->b[k], a->c[k], a->noise[k] ? 1 : 0, a->counter_base + k, 0, 0};
    DD_HIP(c, hipMemcpyAsync(c->atab, c->atab_host.data(), ((size_t)n + 1) * sizeof(AffineRow), hipMemcpyHostToDevice, s));

    float* x_run = a->x_dev;
    const int64_t* y_run = a->y_dev;
    const size_t chw = (size_t)a->first->cfg.in_chans * a->first->cfg.img_size * a->first->cfg.img_size;
    const size_t x_elems = (size_t)a->B * chw;
    // two half-batch chains, as dd_sample (both read the one step table; each has its own step index and Philox image offset)
    const bool chained = a->use_graph && use_chains(c, a->first, a->B) && (!switching || use_chains(c, a->late, a->B));
    const int B0 = chained ? a->B / 2 : a->B, B1 = a->B - B0;
    c->last_chains = chained ? 2 : 1;
    if (a->use_graph) {
        if ((rc = stage_inputs(c, a->x_dev, a->y_dev, a->B, x_elems, s, &x_run, &y_run))) return rc;
        struct CusGuard { dd_ctx* c; int saved; ~CusGuard() { c->num_cus = saved; } } cus_guard{c, c->num_cus};
        if (chained) c->num_cus = std::min(chain_gemm_cus(c, a->first, a->B), switching ? chain_gemm_cus(c, a->late, a->B) : c->num_cus);
        const GraphKey key{x_run, y_run, B0, a->noise_mode, 0, c->num_cus, c->atab};
        auto step = [&](dd_model* m) { return enqueue_step(c, m, x_run, y_run, a->noise_mode, nullptr, 0, nullptr, B0, s, 1, c->atab); };
        if ((rc = get_graph(c, a->first, 1, key, s, step))) return rc;
        if (switching && (rc = get_graph(c, a->late, 1, key, s, step))) return rc;
        if (chained) {
            float* x1 = x_run + (size_t)B0 * chw;
            const int64_t* y1 = y_run ? y_run + B0 : nullptr;
            GraphKey key1{x1, y1, B1, a->noise_mode, 0, c->num_cus, c->atab};
            key1.b0 = B0;
            auto step1 = [&](dd_model* m) {
                swap_chain(m); std::swap(c->st, c->st2);
                const int r = enqueue_step(c, m, x1, y1, a->noise_mode, nullptr, 0, nullptr, B1, s, 1, c->atab, B0);
                swap_chain(m); std::swap(c->st, c->st2);
                return r;
            };
            if ((rc = ensure_chain_ws(c, a->first, s)) || (rc = get_graph(c, a->first, 4, key1, s, step1))) return rc;
            if (switching && ((rc = ensure_chain_ws(c, a->late, s)) || (rc = get_graph(c, a->late, 4, key1, s, step1)))) return rc;
        }
    }
    DD_HIP(c, launch_set_state_table(c->st, c->atab, (unsigned long long)a->seed, s));
    if (chained) {
        DD_HIP(c, launch_set_state_table(c->st2, c->atab, (unsigned long long)a->seed, s));
        DD_HIP(c, hipEventRecord(c->ev_fork, s));
        DD_HIP(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    SideJoin side_join{c, chained};
    DD_HIP(c, hipEventRecord(c->ev[0], s));
    bool marked = false;
    dd_model* cur = a->first;
    for (int k = 0; k < n; ++k) {
        if (switching && k == a->switch_after) {
            cur = a->late;
            DD_HIP(c, hipEventRecord(c->ev[1], s));
            marked = true;
        }
        if (a->use_graph) {
            DD_HIP(c, hipGraphLaunch(cur->graph[1], s));
            if (chained) DD_HIP(c, hipGraphLaunch(cur->graph[4], c->side));
        } else {
            rc = enqueue_step(c, cur, x_run, y_run, a->noise_mode, nullptr, 0, nullptr, a->B, s, 1, c->atab);
            if (rc) return rc;
        }
    }
    if (chained) {
        DD_HIP(c, hipEventRecord(c->ev_join, c->side));
        DD_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
        side_join.armed = false;
    }
    if (!marked) DD_HIP(c, hipEventRecord(c->ev[1], s));
    DD_HIP(c, hipEventRecord(c->ev[2], s));
    if (x_run != a->x_dev) DD_HIP(c, hipMemcpyAsync(a->x_dev, x_run, x_elems * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DD_OK;
}

// One early-exit sampling step on the device (reference eesampler.py:56-81): EarlyExitUViT.forward with every head and
// probe -> per-sample exit selection -> DDPM update with the selected output; rows t of the two log tables are written.
// ws: the chain's scratch block (eps | model_output | cls | outs for B images); b0 / B_all: the chain's first image within the whole batch and
// the whole batch (idx_tab rows are B_all wide); sums: err_tab is the chain's table of per-layer SUMS (launch_ee_mean_combine joins the chains)
static int enqueue_ee_step(dd_ctx* c, dd_model* m, float* ws, float* x, const int64_t* y, float thr, float* err_tab, int32_t* idx_tab,
                           int noise_mode, int B, hipStream_t s, int b0 = 0, int B_all = 0, bool sums = false) {
    const long long chw = (long long)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
    const int depth = m->cfg.depth;
    float* eps = ws;
    float* mo = eps + (size_t)B * chw;
    float* cls = mo + (size_t)B * chw;
    float* outs = cls + (size_t)depth * B;
    const EeTaps ee{cls, outs, 0};
    int rc = run_model(m, x, nullptr, y, B, s, &ee);
    if (rc) return rc;
    FinalArgs fa{m->dec, m->wconv, m->bconv, nullptr, nullptr, eps, nullptr, c->st, c->coef,
                 B, m->cfg.in_chans, m->cfg.img_size, m->cfg.patch_size, m->L, m->extras, DD_NOISE_NONE, 0, 0};
    DD_HIP(c, launch_final(fa, s));
    // exit layer per image, the selected output and the DDPM update in one launch (dd_early_exit_select + the step kernel, fused: same arithmetic)
    DD_HIP(c, launch_ee_select_step(x, outs, eps, cls, thr, depth, idx_tab, err_tab, B_all > 0 ? B_all : B, b0, sums, c->st, c->coef, B,
                                    m->cfg.in_chans, m->cfg.img_size, noise_mode, 1, s));
    (void)mo;
    return DD_OK;
}

int dd_sample_early_exit(dd_ctx* c, const dd_ee_sample_args* a, void* stream) {
    if (!c || !a) return DD_ERR_INVALID;
    dd_model* m = a->model;
    int rc = check_call(c, m, a->B, a->y_dev);
    if (rc) return rc;
    if (m->ee_type < 0) return fail(c, DD_ERR_STATE, "model was not created with dd_model_enable_early_exit");
    if (!a->x_dev) return fail(c, DD_ERR_INVALID, "null tensor");
    if (a->t_start > 999 || a->t_end < 0 || a->t_end > a->t_start) return fail(c, DD_ERR_INVALID, "need 999 >= t_start >= t_end >= 0");
    if (a->noise_mode != DD_NOISE_PHILOX && a->noise_mode != DD_NOISE_NONE)
        return fail(c, DD_ERR_INVALID, "dd_sample_early_exit generates noise on the device; for host noise drive dd_forward_early_exit");
    hipStream_t s = (hipStream_t)stream;
    const size_t chw = (size_t)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
    const size_t need = (size_t)m->cfg.max_batch * ((2 + m->cfg.depth) * chw + m->cfg.depth);
    if (m->ee_ws_elems < need) {
        if (m->ee_ws) (void)hipFree(m->ee_ws);
        m->ee_ws = nullptr; m->ee_ws_elems = 0;
        DD_HIP(c, hipMalloc((void**)&m->ee_ws, need * sizeof(float)));
        m->ee_ws_elems = need;
    }
    float* x_run = a->x_dev;
    const int64_t* y_run = a->y_dev;
    // Two half-batch chains, as dd_sample: the samples are independent (every exit decision is per sample); the one quantity over the whole
    // batch -- the logged per-layer mean of the predicted errors, eesampler.py:70 -- becomes per-chain sums that one small launch joins
    // behind the loop, (chain 0 + chain 1) / B.  Heads and probes stay on each chain's own stream.
    const bool chained = a->use_graph && use_chains(c, m, a->B, true);
    const int B0 = chained ? a->B / 2 : a->B, B1 = a->B - B0;
    c->last_chains = chained ? 2 : 1;
    float* ws1 = m->ee_ws + (size_t)B0 * ((2 + m->cfg.depth) * chw + m->cfg.depth);     // (the block's size is linear in the batch: two halves fit)
    float *sum0 = nullptr, *sum1 = nullptr;
    if (chained && a->err_dev) {
        if (!m->ee_sums) DD_HIP(c, hipMalloc((void**)&m->ee_sums, (size_t)2 * 1000 * m->cfg.depth * sizeof(float)));
        sum0 = m->ee_sums; sum1 = m->ee_sums + (size_t)1000 * m->cfg.depth;
    }
    struct InlineGuard { dd_ctx* c; bool saved; ~InlineGuard() { c->ee_inline = saved; } } inline_guard{c, c->ee_inline};
    c->ee_inline = chained;      // (measured: the 13 forks as parallel branches of each chain's graph cost +0.9 ms per step -- 5.22 against 4.23 ms; profiles/r05/ab_round5.txt)
    if (a->use_graph) {
        if ((rc = stage_inputs(c, a->x_dev, a->y_dev, a->B, (size_t)a->B * chw, s, &x_run, &y_run))) return rc;
        GraphKey key{x_run, y_run, B0, a->noise_mode, 0, c->num_cus, nullptr};
        key.aux0 = chained ? (const void*)sum0 : (const void*)a->err_dev; key.aux1 = a->idx_dev; key.thr = a->threshold; key.b0 = chained ? -1 : 0;   // (-1: chain 0 of two -- not the whole batch's graph)
        auto step = [&](dd_model* mm) {
            return enqueue_ee_step(c, mm, mm->ee_ws, x_run, y_run, a->threshold, chained ? sum0 : a->err_dev, a->idx_dev, a->noise_mode, B0, s, 0, a->B, chained);
        };
        if ((rc = get_graph(c, m, 2, key, s, step))) return rc;
        if (chained) {
            float* x1 = x_run + (size_t)B0 * chw;
            const int64_t* y1 = y_run ? y_run + B0 : nullptr;
            GraphKey key1{x1, y1, B1, a->noise_mode, 0, c->num_cus, nullptr};
            key1.aux0 = sum1; key1.aux1 = a->idx_dev; key1.thr = a->threshold; key1.b0 = B0;
            auto step1 = [&](dd_model* mm) {
                swap_chain(mm); std::swap(c->st, c->st2);
                const int r = enqueue_ee_step(c, mm, ws1, x1, y1, a->threshold, sum1, a->idx_dev, a->noise_mode, B1, s, B0, a->B, true);
                swap_chain(mm); std::swap(c->st, c->st2);
                return r;
            };
            if ((rc = ensure_chain_ws(c, m, s)) || (rc = get_graph(c, m, 5, key1, s, step1))) return rc;
        }
    }
    DD_HIP(c, launch_set_state(c->st, a->t_start, (unsigned long long)a->seed, s));
    if (chained) {
        DD_HIP(c, launch_set_state(c->st2, a->t_start, (unsigned long long)a->seed, s));
        DD_HIP(c, hipEventRecord(c->ev_fork, s));
        DD_HIP(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    }
    SideJoin side_join{c, chained};
    DD_HIP(c, hipEventRecord(c->ev[0], s));
    for (int t = a->t_start; t >= a->t_end; --t) {
        if (a->use_graph) {
            DD_HIP(c, hipGraphLaunch(m->graph[2], s));
            if (chained) DD_HIP(c, hipGraphLaunch(m->graph[5], c->side));
        } else {
            rc = enqueue_ee_step(c, m, m->ee_ws, x_run, y_run, a->threshold, a->err_dev, a->idx_dev, a->noise_mode, a->B, s);
            if (rc) return rc;
        }
    }
    if (chained) {
        DD_HIP(c, hipEventRecord(c->ev_join, c->side));
        DD_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
        side_join.armed = false;
        if (a->err_dev) DD_HIP(c, launch_ee_mean_combine(sum0, sum1, a->err_dev, m->cfg.depth, a->t_end, a->t_start, a->B, s));
    }
    DD_HIP(c, hipEventRecord(c->ev[1], s));
    DD_HIP(c, hipEventRecord(c->ev[2], s));
    if (x_run != a->x_dev) DD_HIP(c, hipMemcpyAsync(a->x_dev, x_run, (size_t)a->B * chw * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DD_OK;
}

long long dd_dev_graph_captures(dd_ctx* c) { return c ? c->graph_captures : -1; }
int dd_dev_last_sample_chains(dd_ctx* c) { return c ? c->last_chains : -1; }

int dd_dev_poison_workspaces(dd_ctx* c, dd_model* m, void* stream) {
    if (!c || !m || m->ctx != c || !m->finalized) return DD_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = ensure_chain_ws(c, m, s)) return rc;
    DD_HIP(c, hipMemsetAsync(m->wsarena, 0xFF, m->wsoff.bytes, s));
    DD_HIP(c, hipMemsetAsync(m->wsarena2, 0xFF, m->wsoff2.bytes, s));
    return DD_OK;
}

int dd_dev_set_flags(dd_ctx* c, unsigned flags) {
    if (!c) return DD_ERR_INVALID;
    c->dev_flags = flags;
    return DD_OK;
}

int dd_profile_select(dd_ctx* c, int kind) {
    if (!c || kind < DD_PROF_DOMINANT || kind > DD_PROF_SPLITK) return DD_ERR_INVALID;
    c->prof_kind = kind;
    return DD_OK;
}

int dd_last_sample_timing(dd_ctx* c, float out3[3]) {
    if (!c || !out3) return DD_ERR_INVALID;
    DD_HIP(c, hipEventSynchronize(c->ev[2]));
    DD_HIP(c, hipEventElapsedTime(&out3[0], c->ev[0], c->ev[2]));
    DD_HIP(c, hipEventElapsedTime(&out3[1], c->ev[0], c->ev[1]));
    DD_HIP(c, hipEventElapsedTime(&out3[2], c->ev[1], c->ev[2]));
    return DD_OK;
}

int dd_profile_steps(dd_ctx* c, dd_model* m, float* x_dev, const int64_t* y_dev, int t_start, int steps, int B,
                     void* stream, float* fc1_ms_out, int* launches_out) {
    int rc = check_call(c, m, B, y_dev);
    if (rc) return rc;
    if (!x_dev || !fc1_ms_out || steps < 1 || t_start > 999 || t_start - steps + 1 < 0) return fail(c, DD_ERR_INVALID, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    m->time_fc1 = true;
    m->fc1_used = 0;
    for (int i = 0; i < steps && !rc; ++i) {
        hipError_t e = launch_set_state(c->st, t_start - i, 12345ull, s);
        if (e != hipSuccess) { m->time_fc1 = false; return fail_hip(c, e, "set_state"); }
        rc = enqueue_step(c, m, x_dev, y_dev, DD_NOISE_PHILOX, nullptr, DD_VAR_BETA_TILDE, nullptr, B, s);
    }
    m->time_fc1 = false;
    if (rc) return rc;
    DD_HIP(c, hipStreamSynchronize(s));
    double total = 0.0;
    for (size_t i = 0; i + 1 < m->fc1_used; i += 2) {
        float ms = 0.f;
        DD_HIP(c, hipEventElapsedTime(&ms, m->fc1_events[i], m->fc1_events[i + 1]));
        total += ms;
    }
    const int n = (int)(m->fc1_used / 2);
    *fc1_ms_out = n ? (float)(total / n) : 0.f;
    if (launches_out) *launches_out = n;
    return DD_OK;
}

// dd_profile_steps for the way dd_sample runs a large batch: the two half-batch chains enqueued eagerly on `stream` and on the
// context's side stream, step by step, with an event pair around every launch of the dominant kernel in BOTH chains -- the
// launches overlap the other chain's kernels exactly as the graph replays of the timed loop do.
int dd_profile_steps_chained(dd_ctx* c, dd_model* m, float* x_dev, const int64_t* y_dev, int t_start, int steps, int B,
                             void* stream, float* ms_out, int* launches_out) {
    int rc = check_call(c, m, B, y_dev);
    if (rc) return rc;
    if (!x_dev || !ms_out || steps < 1 || t_start > 999 || t_start - steps + 1 < 0 || (B & 1) || B < 2) return fail(c, DD_ERR_INVALID, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = ensure_chain_ws(c, m, s))) return rc;
    const int B0 = B / 2, B1 = B - B0;
    const size_t chw = (size_t)m->cfg.in_chans * m->cfg.img_size * m->cfg.img_size;
    // (the persistent GEMM grids of both chains sized as dd_sample sizes them: halved for a large GEMM-path batch)
    struct CusGuard { dd_ctx* c; int saved; ~CusGuard() { c->num_cus = saved; } } cus_guard{c, c->num_cus};
    c->num_cus = chain_gemm_cus(c, m, B);
    DD_HIP(c, hipEventRecord(c->ev_fork, s));
    DD_HIP(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    SideJoin side_join{c, true};
    m->time_fc1 = true;
    m->fc1_used = 0;
    for (int i = 0; i < steps && !rc; ++i) {
        hipError_t e = launch_set_state(c->st, t_start - i, 12345ull, s);
        if (e == hipSuccess) e = launch_set_state(c->st2, t_start - i, 12345ull, c->side);
        if (e != hipSuccess) { m->time_fc1 = false; return fail_hip(c, e, "set_state"); }
        rc = enqueue_step(c, m, x_dev, y_dev, DD_NOISE_PHILOX, nullptr, DD_VAR_BETA_TILDE, nullptr, B0, s);
        if (rc) break;
        swap_chain(m); std::swap(c->st, c->st2);
        rc = enqueue_step(c, m, x_dev + (size_t)B0 * chw, y_dev ? y_dev + B0 : nullptr, DD_NOISE_PHILOX, nullptr, DD_VAR_BETA_TILDE, nullptr,
                          B1, c->side, 0, nullptr, B0);
        swap_chain(m); std::swap(c->st, c->st2);
    }
    m->time_fc1 = false;
    if (rc) return rc;
    DD_HIP(c, hipEventRecord(c->ev_join, c->side));
    DD_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
    side_join.armed = false;
    DD_HIP(c, hipStreamSynchronize(s));
    double total = 0.0;
    for (size_t i = 0; i + 1 < m->fc1_used; i += 2) {
        float ms = 0.f;
        DD_HIP(c, hipEventElapsedTime(&ms, m->fc1_events[i], m->fc1_events[i + 1]));
        total += ms;
    }
    const int n = (int)(m->fc1_used / 2);
    *ms_out = n ? (float)(total / n) : 0.f;
    if (launches_out) *launches_out = n;
    return DD_OK;
}

int dd_bench_gemm(dd_ctx* c, dd_model* m, int B, int iters, void* stream, float* ms_out, double* flops_out) {
    int rc = check_call(c, m, B, m && m->cfg.num_classes > 0 ? (const int64_t*)1 : nullptr);
    if (rc) return rc;
    if (iters < 1 || !ms_out) return fail(c, DD_ERR_INVALID, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int M = B * m->L, D = m->D;
    const BlockW& w = m->blocks[0];
    auto once = [&]() -> hipError_t {
        if (m->prec == DD_PREC_BF16) {
            GemmArgs<bf16_t> g{(const bf16_t*)m->h, nullptr, (const bf16_t*)w.fc1_w, w.fc1_b, nullptr, (bf16_t*)m->hid,
                               M, m->hidden, D, D, D, 0, m->hid_ld};
            return launch_gemm<bf16_t>(g, EPI_BIAS_GELU, s, c->num_cus);
        }
        GemmArgs<float> g{(const float*)m->h, nullptr, (const float*)w.fc1_w, w.fc1_b, nullptr, (float*)m->hid,
                          M, m->hidden, D, D, D, 0, m->hid_ld};
        return launch_gemm<float>(g, EPI_BIAS_GELU, s, c->num_cus);
    };
    DD_HIP(c, once());
    hipEvent_t e0, e1;
    DD_HIP(c, hipEventCreate(&e0));
    DD_HIP(c, hipEventCreate(&e1));
    DD_HIP(c, hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) DD_HIP(c, once());
    DD_HIP(c, hipEventRecord(e1, s));
    DD_HIP(c, hipEventSynchronize(e1));
    float ms = 0.f;
    DD_HIP(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_out = ms / (float)iters;
    if (flops_out) *flops_out = 2.0 * (double)M * (double)m->hidden * (double)D;
    return DD_OK;
}

int dd_dev_mlp(dd_ctx* c, int M, int D, int hidden, int extras, const float* x_host, const float* w1, const float* b1, const float* w2,
               const float* b2, float* xres_host, unsigned short* out_host, const float* ln_in, const float* ln_out,
               unsigned short* ln_out_host, int iters, void* stream, float* ms_out, const float* ao_host, const float* wproj,
               const float* bproj, const float* skip_host, const float* wskip, const float* bskip, const float* wqkv,
               unsigned short* qkv_out_host) {
    const bool proj = ao_host && wproj && bproj;
    const bool skp = skip_host && wskip && bskip;
    const bool qk = wqkv && qkv_out_host;
    if (qk && (!proj || !ln_out || (hidden / 32) % 2)) return DD_ERR_INVALID;   // the qkv phases ride on the proj-fused launch, behind norm1
    if (proj && (!ln_in || D % 128)) return DD_ERR_INVALID;   // the projection rides in the LayerNorm-in kernel only
    if (skp && (!proj || !ln_out || !ln_out_host || (hidden / 32) % 2)) return DD_ERR_INVALID;   // the skip phases ride on the proj-fused launch and end in norm1
    if (!c || !x_host || !w1 || !b1 || !w2 || !b2 || !xres_host || M < 1 || iters < 0 || extras < 0 || (extras > 0 && M % (1 + extras))) return DD_ERR_INVALID;
    if (!mlp_fused_supported(D, hidden)) return fail(c, DD_ERR_UNSUPPORTED, "fused MLP: D in {64,128,256,512}, hidden % 64 == 0");
    hipStream_t s = (hipStream_t)stream;
    const size_t Mp = (size_t)round_up(M, 256);
    std::vector<unsigned short> xh(Mp * D, 0), img(mlp_fused_image_bytes(D, hidden, proj, skp, qk) / 2, 0);
    std::vector<float> b1p(hidden), xr(Mp * D, 0.f);
    for (size_t i = 0; i < (size_t)M * D; ++i) { xh[i] = host_f2bf(x_host[i]); xr[i] = xres_host[i]; }
    if (proj) mlp_fused_pack_proj(D, wproj, host_f2bf, img.data());
    mlp_fused_pack(D, hidden, w1, b1, w2, ln_in != nullptr, host_f2bf, img.data() + (proj ? (size_t)D * D : 0), b1p.data());
    if (skp) mlp_fused_pack_skip(D, wskip, host_f2bf, img.data() + (proj ? (size_t)D * D : 0) + (size_t)(hidden / 32) * 2 * (D / 16) * 512);
    if (qk) mlp_fused_pack_rows(D, 3 * D, wqkv, host_f2bf, img.data() + (proj ? (size_t)D * D : 0) + ((size_t)(hidden / 32) * 2 + (skp ? D / 16 : 0)) * (D / 16) * 512);
    // extras > 0: the M rows are `M / (1 + extras)` images of one patch token each (drives the hidden-split path);
    // extras == 0: one image of M patch tokens (main tiles only)
    MlpFusedArgs a{};
    if (extras > 0) mlp_fused_plan(M / (1 + extras), 1, extras, 1 + extras, hidden, a);
    else mlp_fused_plan(1, M, 0, M, hidden, a);
    if (c->dev_flags & DD_DEV_MLP_EXTRAS_ONLY) { a.tiles_main = 0; a.n_main = 0; }   // time the hidden-split workgroups alone
    const size_t part = (size_t)a.tiles_left * a.groups * 128 * D * sizeof(float);
    void *dX = nullptr, *dI = nullptr, *dB1 = nullptr, *dB2 = nullptr, *dXr = nullptr, *dO = nullptr, *dP = nullptr, *dLn = nullptr, *dH = nullptr;
    void *dAo = nullptr, *dBp = nullptr, *dSk = nullptr, *dBs = nullptr, *dQ = nullptr, *dQd = nullptr;
    auto cleanup = [&]() { for (void* p : {dX, dI, dB1, dB2, dXr, dO, dP, dLn, dH, dAo, dBp, dSk, dBs, dQ, dQd}) if (p) (void)hipFree(p); };
#define DD_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail_hip(c, _e, #expr); } } while (0)
    DD_TRY(hipMalloc(&dX, xh.size() * 2)); DD_TRY(hipMalloc(&dI, img.size() * 2)); DD_TRY(hipMalloc(&dB1, hidden * 4));
    DD_TRY(hipMalloc(&dB2, D * 4)); DD_TRY(hipMalloc(&dXr, xr.size() * 4)); DD_TRY(hipMalloc(&dO, xh.size() * 2));
    if (part) DD_TRY(hipMalloc(&dP, part));
    DD_TRY(hipMemcpy(dX, xh.data(), xh.size() * 2, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dI, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dB1, b1p.data(), hidden * 4, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dB2, b2, D * 4, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dXr, xr.data(), xr.size() * 4, hipMemcpyHostToDevice));
    DD_TRY(hipMemset(dO, 0, xh.size() * 2));
    // ln_in / ln_out: [2, D] gamma then beta of the LayerNorm fused into the prologue / epilogue (or NULL)
    DD_TRY(hipMalloc(&dLn, (size_t)4 * D * 4));
    DD_TRY(hipMalloc(&dH, xh.size() * 2));
    DD_TRY(hipMemset(dH, 0, xh.size() * 2));
    if (ln_in) { DD_TRY(hipMemcpy(dLn, ln_in, (size_t)2 * D * 4, hipMemcpyHostToDevice)); a.ln_in_g = (const float*)dLn; a.ln_in_b = (const float*)dLn + D; }
    if (ln_out && ln_out_host) {
        DD_TRY(hipMemcpy((float*)dLn + 2 * D, ln_out, (size_t)2 * D * 4, hipMemcpyHostToDevice));
        a.ln_out_g = (const float*)dLn + 2 * D; a.ln_out_b = (const float*)dLn + 3 * D; a.ln_out = (bf16_t*)dH;
    }
    a.X = (const bf16_t*)dX; a.ldx = D; a.wimg = (const char*)dI; a.b1p = (const float*)dB1; a.b2 = (const float*)dB2;
    a.xres = (float*)dXr; a.out = out_host ? (bf16_t*)dO : nullptr; a.ldo = D; a.partial = (float*)dP;
    if (proj) {   // as run_backbone does it: patch rows in the main tiles, extra-token rows in their hidden-split workgroups
        std::vector<unsigned short> ah(Mp * D, 0);
        for (size_t i = 0; i < (size_t)M * D; ++i) ah[i] = host_f2bf(ao_host[i]);
        DD_TRY(hipMalloc(&dAo, ah.size() * 2)); DD_TRY(hipMalloc(&dBp, D * 4));
        DD_TRY(hipMemcpy(dAo, ah.data(), ah.size() * 2, hipMemcpyHostToDevice));
        DD_TRY(hipMemcpy(dBp, bproj, D * 4, hipMemcpyHostToDevice));
        a.ao = (const bf16_t*)dAo; a.bproj = (const float*)dBp; a.nproj = D / 32;
        a.reduce_set = 1;        // (the extra-token rows' projection runs in their hidden-split workgroups, run_backbone)
    }
    MlpFusedArgs ar = a;       // (what the reduce launch gets: see run_backbone)
    if (skp) {
        std::vector<unsigned short> sh(Mp * D, 0);
        for (size_t i = 0; i < (size_t)M * D; ++i) sh[i] = host_f2bf(skip_host[i]);
        DD_TRY(hipMalloc(&dSk, sh.size() * 2)); DD_TRY(hipMalloc(&dBs, D * 4));
        DD_TRY(hipMemcpy(dSk, sh.data(), sh.size() * 2, hipMemcpyHostToDevice));
        DD_TRY(hipMemcpy(dBs, bskip, D * 4, hipMemcpyHostToDevice));
        a.skip = (const bf16_t*)dSk; a.bskip = (const float*)dBs; a.nskip = D / 16;
        a.out = (bf16_t*)dO;   // y of the extra-token rows travels through the bf16 copy
        ar = a; ar.ln_out = nullptr;
    }
    size_t qkv_elems = 0;
    if (qk) {   // head-major qkv of the rows as images of a.tok_l tokens (HeadMajor, dd_internal.h): [images][3 D / 64 units][Lp][64]
        a.hm = make_head_major(a.tok_l, D / 64);
        const size_t images = (size_t)(M / a.tok_l);
        qkv_elems = images * 3 * D * (size_t)a.hm.Lp;
        DD_TRY(hipMalloc(&dQ, qkv_elems * 2)); DD_TRY(hipMalloc(&dQd, 16384));
        DD_TRY(hipMemset(dQ, 0, qkv_elems * 2));
        a.qkv_out = (bf16_t*)dQ; a.qkv_dump = (bf16_t*)dQd; a.nqkv = 3 * D / 32;
        ar.qkv_out = a.qkv_out; ar.qkv_dump = a.qkv_dump; ar.nqkv = a.nqkv; ar.hm = a.hm;
        if (!skp) ar = a;
    }
    DD_TRY(launch_mlp_fused(a, D, s));
    DD_TRY(launch_mlp_reduce(ar, D, s));
    DD_TRY(launch_skip_rows_ln(a, D, s));
    DD_TRY(launch_qkv_rows(a, D, s));
    DD_TRY(hipStreamSynchronize(s));
    if (qk) DD_TRY(hipMemcpy(qkv_out_host, dQ, qkv_elems * 2, hipMemcpyDeviceToHost));
    DD_TRY(hipMemcpy(xres_host, dXr, (size_t)M * D * 4, hipMemcpyDeviceToHost));
    if (out_host) DD_TRY(hipMemcpy(out_host, dO, (size_t)M * D * 2, hipMemcpyDeviceToHost));
    if (a.ln_out) DD_TRY(hipMemcpy(ln_out_host, dH, (size_t)M * D * 2, hipMemcpyDeviceToHost));
    if (iters > 0 && ms_out) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        DD_TRY(hipEventCreate(&e0)); DD_TRY(hipEventCreate(&e1));
        DD_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) { DD_TRY(launch_mlp_fused(a, D, s)); DD_TRY(launch_mlp_reduce(ar, D, s)); DD_TRY(launch_skip_rows_ln(a, D, s)); DD_TRY(launch_qkv_rows(a, D, s)); }
        DD_TRY(hipEventRecord(e1, s));
        DD_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        DD_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms_out = ms / (float)iters;
    }
#undef DD_TRY
    cleanup();
    return DD_OK;
}

int dd_dev_qkv_attention(dd_ctx* c, int B, int L, int H, int extras, const float* h_host, const float* wqkv, const float* bqkv,
                         unsigned short* out_host, int iters, void* stream, float* ms_out) {
    if (!c || !h_host || !wqkv || !out_host) return DD_ERR_INVALID;
    const int D = 64 * H;
    if (!qkv_attention_supported(D, H, L, extras)) return fail(c, DD_ERR_UNSUPPORTED, "qkv_attention: D = 512 / 768 / 1024, L = 256 + 1 or 2 extra tokens only");
    hipStream_t s = (hipStream_t)stream;
    const size_t M = (size_t)B * L;
    std::vector<unsigned short> hb(M * D), hf((size_t)B * 256 * D), img((size_t)3 * D * D);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = host_f2bf(h_host[i]);
    for (int b = 0; b < B; ++b)          // the patch rows in fragment order (what the fused block tail writes: MlpFusedArgs::ln_out_frag)
        for (int n = 0; n < 256; ++n)
            for (int k = 0; k < D; ++k)
                hf[((((size_t)b * 8 + n / 32) * (D / 16) + k / 16) * 64 + (n % 32) + 32 * ((k % 16) / 8)) * 8 + k % 8] = hb[((size_t)b * L + extras + n) * D + k];
    qkv_attention_pack(D, H, wqkv, host_f2bf, img.data());
    void *dH = nullptr, *dW = nullptr, *dB = nullptr, *dQ = nullptr, *dO = nullptr;
    auto cleanup = [&]() { for (void* p : {dH, dW, dB, dQ, dO}) if (p) (void)hipFree(p); };
#define DD_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(c, DD_ERR_HIP, hipGetErrorString(e_)); } } while (0)
    DD_TRY(hipMalloc(&dH, hf.size() * 2)); DD_TRY(hipMalloc(&dW, img.size() * 2)); DD_TRY(hipMalloc(&dQ, hb.size() * 2)); DD_TRY(hipMalloc(&dO, M * D * 2));
    DD_TRY(hipMemcpy(dH, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dW, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dQ, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));     // row-major norm1 rows: the kernel reads the extra-token rows of it
    DD_TRY(hipMemset(dO, 0, M * D * 2));
    if (bqkv) { DD_TRY(hipMalloc(&dB, (size_t)3 * D * 4)); DD_TRY(hipMemcpy(dB, bqkv, (size_t)3 * D * 4, hipMemcpyHostToDevice)); }
    DD_TRY(launch_qkv_attention((const bf16_t*)dH, (const bf16_t*)dW, (const float*)dB, (const bf16_t*)dQ, nullptr, nullptr, nullptr, (bf16_t*)dO, B, L, H, D, extras, s));
    DD_TRY(hipStreamSynchronize(s));
    DD_TRY(hipMemcpy(out_host, dO, M * D * 2, hipMemcpyDeviceToHost));
    if (iters > 0 && ms_out) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        DD_TRY(hipEventCreate(&e0)); DD_TRY(hipEventCreate(&e1));
        DD_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) DD_TRY(launch_qkv_attention((const bf16_t*)dH, (const bf16_t*)dW, (const float*)dB, (const bf16_t*)dQ, nullptr, nullptr, nullptr, (bf16_t*)dO, B, L, H, D, extras, s));
        DD_TRY(hipEventRecord(e1, s));
        DD_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        DD_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms_out = ms / (float)iters;
    }
#undef DD_TRY
    cleanup();
    return DD_OK;
}

int dd_dev_head_dec(dd_ctx* c, int M, int D, int pd, int tok_l, int tok_e, const float* x_host, const float* norm_g, const float* norm_b,
                    const float* wdec, const float* bdec, float* dec_host, const float* probe_w, const float* probe_b, float* srow_host,
                    int split, int iters, void* stream, float* ms_out) {
    if (!c || !x_host || !norm_g || !norm_b || !wdec || !bdec || !dec_host || M < 1 || iters < 0) return DD_ERR_INVALID;
    if (!head_dec_supported(D, pd)) return fail(c, DD_ERR_UNSUPPORTED, "head_dec: D in {256, 512, 768, 1024}, pd % 4 == 0, pd <= 64");
    const bool probe = probe_w && probe_b && srow_host;
    if (probe && !head_dec_probe_supported(D)) return fail(c, DD_ERR_UNSUPPORTED, "head_dec with the probe: D in {256, 512}");
    hipStream_t s = (hipStream_t)stream;
    if (split && !head_dec_probe_supported(D)) return fail(c, DD_ERR_UNSUPPORTED, "head_dec as a split-bf16 product: D in {256, 512}");
    std::vector<float> wg, dc;
    fold_head_norm(D, pd, wdec, bdec, norm_g, norm_b, wg, dc);
    if (split) {      // wg becomes the packed image (as floats: two bf16 each), dc's second half the row sums of hi + lo
        const int nt = (pd + 15) / 16;
        std::vector<unsigned short> img((size_t)(D / 32) * nt * 2 * 64 * 8);
        pack_head_split(D, pd, wg.data(), host_f2bf, img.data(), dc.data() + pd);
        wg.assign(img.size() / 2, 0.f);
        std::memcpy(wg.data(), img.data(), img.size() * 2);
    }
    void *dX = nullptr, *dW = nullptr, *dC = nullptr, *dO = nullptr, *dP = nullptr, *dPb = nullptr, *dS = nullptr;
    auto cleanup = [&]() { for (void* p : {dX, dW, dC, dO, dP, dPb, dS}) if (p) (void)hipFree(p); };
#define DD_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(c, DD_ERR_HIP, hipGetErrorString(e_)); } } while (0)
    DD_TRY(hipMalloc(&dX, (size_t)M * D * 4)); DD_TRY(hipMalloc(&dW, wg.size() * 4)); DD_TRY(hipMalloc(&dC, dc.size() * 4)); DD_TRY(hipMalloc(&dO, (size_t)M * pd * 4));
    DD_TRY(hipMemcpy(dX, x_host, (size_t)M * D * 4, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dW, wg.data(), wg.size() * 4, hipMemcpyHostToDevice));
    DD_TRY(hipMemcpy(dC, dc.data(), dc.size() * 4, hipMemcpyHostToDevice));
    DD_TRY(hipMemset(dO, 0xFF, (size_t)M * pd * 4));      // NaN: rows the launch does not decode stay recognisable
    HeadDecArgs ha{(const float*)dX, (const float*)dW, (const float*)dC, (float*)dO, M, pd, tok_l, tok_e};
    ha.split = split ? 1 : 0;
    if (probe) {
        DD_TRY(hipMalloc(&dP, (size_t)D * 4)); DD_TRY(hipMalloc(&dPb, 4)); DD_TRY(hipMalloc(&dS, (size_t)M * 4));
        DD_TRY(hipMemcpy(dP, probe_w, (size_t)D * 4, hipMemcpyHostToDevice));
        DD_TRY(hipMemcpy(dPb, probe_b, 4, hipMemcpyHostToDevice));
        DD_TRY(hipMemset(dS, 0xFF, (size_t)M * 4));
        ha.srow = (float*)dS; ha.pw_base = (const float*)dP; ha.pb_base = (const float*)dPb;     // (probe row 0: t_mul = add = 0, the step state is not read)
    }
    DD_TRY(launch_head_dec(ha, D, c->num_cus, s));
    DD_TRY(hipStreamSynchronize(s));
    DD_TRY(hipMemcpy(dec_host, dO, (size_t)M * pd * 4, hipMemcpyDeviceToHost));
    if (probe) DD_TRY(hipMemcpy(srow_host, dS, (size_t)M * 4, hipMemcpyDeviceToHost));
    if (iters > 0 && ms_out) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        DD_TRY(hipEventCreate(&e0)); DD_TRY(hipEventCreate(&e1));
        DD_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) DD_TRY(launch_head_dec(ha, D, c->num_cus, s));
        DD_TRY(hipEventRecord(e1, s));
        DD_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        DD_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms_out = ms / (float)iters;
    }
#undef DD_TRY
    cleanup();
    return DD_OK;
}

int dd_plan_rows(int M, int N, int K, int num_cus, int* q_out, int* e_out) {
    if (!q_out || !e_out) return DD_ERR_INVALID;
    return plan_rows_256(M, N, K, num_cus, q_out, e_out) ? DD_OK : DD_ERR_UNSUPPORTED;
}

int dd_set_num_cus(dd_ctx* c, int n) {
    if (!c) return DD_ERR_INVALID;
    if (n < 8) return fail(c, DD_ERR_INVALID, "need at least 8 CUs");
    c->num_cus = n / 8 * 8;      // the persistent kernels deal tiles to workgroups in groups of 8 (one per XCD)
    c->base_cus = c->num_cus;
    return DD_OK;
}

}  // extern "C"
