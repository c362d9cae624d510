// KL-VAE decode behind the C ABI: reference models/utils/autoencoder.py
//   FrozenAutoencoderKL.decode :486-490  =  z/scale -> post_quant_conv -> Decoder.forward :403-449
//   Decoder: conv_in, mid (ResnetBlock, AttnBlock, ResnetBlock), 4 up levels x 3 ResnetBlocks (+ nearest-2x Upsample
//   with conv), GroupNorm(32, eps 1e-6) + swish + conv_out.  ddconfig of get_autoencoder :503-516 (ch 128, mult 1,2,4,4).
// Activations are NHWC with an fp32 residual stream; every convolution is an explicit im2col (upsample folded into
// the gather) + one MFMA GEMM of gemm.hip with the bias / residual add fused in its epilogue.  The decode runs once per
// image (0.5 % of the FLOPs of 1000 sampling steps), so the design goal is correctness on parity-proven kernels.
#include "../../include/duodiff.h"
#include "dd_internal.h"

#include <cmath>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace dd;

struct dd_ctx;
extern "C" const char* dd_last_error(dd_ctx*);
namespace dd { int ctx_fail(dd_ctx* c, int code, const std::string& msg); int ctx_device(dd_ctx* c); int ctx_num_cus(dd_ctx* c); }

namespace {

struct ConvW { const void* w = nullptr; const float* b = nullptr; int cin = 0, cout = 0, k = 0, kpad = 0; };
struct NormW { const float* g = nullptr; const float* b = nullptr; int c = 0; };
struct ResW { NormW n1, n2; ConvW c1, c2, nin; };

struct HostT { std::vector<float> d; std::vector<int64_t> shape; };

}  // namespace

struct dd_vae {
    dd_ctx* ctx = nullptr;
    int max_chunk = 4, max_latent = 32;
    std::map<std::string, HostT> params;
    bool finalized = false;
    int prec = DD_PREC_BF16;
    size_t es = 2;
    char* warena = nullptr;
    char* ws = nullptr;
    // weights
    const float *pq_w = nullptr, *pq_b = nullptr;
    ConvW conv_in, conv_out, up_conv[4];
    ResW mid1, mid2, up[4][3];
    NormW attn_norm, norm_out;
    ConvW attn_q, attn_k, attn_v, attn_proj;
    // workspace
    float *s0 = nullptr, *s1 = nullptr, *h1 = nullptr, *part = nullptr, *score = nullptr, *ao = nullptr, *z4 = nullptr;
    void *nb = nullptr, *col = nullptr, *q = nullptr, *kk = nullptr, *vt = nullptr, *pp = nullptr;
};

namespace {

const int kCh = 128, kMult[4] = {1, 2, 4, 4};

#define VHIP(c, expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) return ctx_fail((c), DD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// every decoder tensor the reference's state_dict holds, with its shape
std::map<std::string, std::vector<int64_t>> expected() {
    std::map<std::string, std::vector<int64_t>> m;
    auto conv = [&](const std::string& n, int co, int ci, int k) { m[n + ".weight"] = {co, ci, k, k}; m[n + ".bias"] = {co}; };
    auto norm = [&](const std::string& n, int c) { m[n + ".weight"] = {c}; m[n + ".bias"] = {c}; };
    auto res = [&](const std::string& n, int ci, int co) {
        norm(n + ".norm1", ci); conv(n + ".conv1", co, ci, 3); norm(n + ".norm2", co); conv(n + ".conv2", co, co, 3);
        if (ci != co) conv(n + ".nin_shortcut", co, ci, 1);
    };
    conv("post_quant_conv", 4, 4, 1);
    const int top = kCh * kMult[3];
    conv("decoder.conv_in", top, 4, 3);
    res("decoder.mid.block_1", top, top);
    norm("decoder.mid.attn_1.norm", top);
    for (const char* n : {"q", "k", "v", "proj_out"}) conv(std::string("decoder.mid.attn_1.") + n, top, top, 1);
    res("decoder.mid.block_2", top, top);
    int cin = top;
    for (int lv = 3; lv >= 0; --lv) {
        const int cout = kCh * kMult[lv];
        for (int j = 0; j < 3; ++j) { res("decoder.up." + std::to_string(lv) + ".block." + std::to_string(j), cin, cout); cin = cout; }
        if (lv != 0) conv("decoder.up." + std::to_string(lv) + ".upsample.conv", cin, cin, 3);
    }
    norm("decoder.norm_out", cin);
    conv("decoder.conv_out", 3, cin, 3);
    return m;
}

unsigned short h_f2bf(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T>
int run_decode(dd_vae* v, const float* z, float* out, int B, int HL, hipStream_t s) {
    dd_ctx* c = v->ctx;
    T* nb = (T*)v->nb; T* col = (T*)v->col;
    float* cur = v->s0; float* other = v->s1;
    int H = HL, Cc = 0;

    auto gemm = [&](const T* A, int M, int K, const ConvW& w, int epi, float* xres, T* o, int ldo) -> int {
        GemmArgs<T> g{A, nullptr, (const T*)w.w, w.b, xres, o, M, w.cout == 3 ? 4 : w.cout, K, K, K, 0, ldo};
        VHIP(c, launch_gemm<T>(g, epi, s, ctx_num_cus(c)));
        return DD_OK;
    };
    // 3x3 conv of the T-typed NHWC image `src` (C channels at Hs x Hs, optionally upsampled 2x first)
    auto conv3 = [&](const T* src, int Bn, int Hout, int C, int up, const ConvW& w, int epi, float* xres) -> int {
        VHIP(c, launch_im2col3x3<T>(src, col, Bn, Hout, Hout, C, up, w.kpad, s));
        return gemm(col, Bn * Hout * Hout, w.kpad, w, epi, xres, nullptr, 0);
    };
    auto gn = [&](const float* x, const NormW& n, int Bn, int HW, int swish) -> int {
        VHIP(c, launch_groupnorm<T>(x, v->part, n.g, n.b, nb, Bn, HW, n.c, swish, s));
        return DD_OK;
    };
    // ResnetBlock (autoencoder.py:121-136) on the fp32 stream `cur` [B*H*H, cin]
    auto resnet = [&](const ResW& r) -> int {
        const int M = B * H * H, cin = r.c1.cin, cout = r.c1.cout;
        int rc;
        float* dst = cur;
        if (cin != cout) {   // x <- nin_shortcut(x) into the other stream buffer (1x1 conv on the raw stream)
            VHIP(c, launch_cast<T>(cur, (T*)v->q, (long long)M * cin, s));   // q buffer doubles as the raw-copy scratch
            if ((rc = gemm((const T*)v->q, M, cin, r.nin, EPI_BIAS_SET, other, nullptr, 0))) return rc;
            dst = other;
        }
        if ((rc = gn(cur, r.n1, B, H * H, 1))) return rc;
        if ((rc = conv3(nb, B, H, cin, 0, r.c1, EPI_BIAS_SET, v->h1))) return rc;
        if ((rc = gn(v->h1, r.n2, B, H * H, 1))) return rc;
        if ((rc = conv3(nb, B, H, cout, 0, r.c2, EPI_BIAS_RESID, dst))) return rc;     // x + h
        if (dst != cur) std::swap(cur, other);
        Cc = cout;
        return DD_OK;
    };

    int rc;
    const int M0 = B * H * H;
    VHIP(c, launch_vae_input(z, v->pq_w, v->pq_b, 1.0f / 0.18215f, v->z4, B, H * H, s));               // :487-488
    VHIP(c, launch_im2col3x3_c4<T>(v->z4, col, B, H, H, v->conv_in.kpad, s));
    if ((rc = gemm(col, M0, v->conv_in.kpad, v->conv_in, EPI_BIAS_SET, cur, nullptr, 0))) return rc;   // conv_in :413
    Cc = v->conv_in.cout;
    if ((rc = resnet(v->mid1))) return rc;                                                               // :416
    {   // AttnBlock :155-185, single head over the H*H pixels of each image
        const int HW = H * H, C = Cc, M = B * HW;
        if ((rc = gn(cur, v->attn_norm, B, HW, 0))) return rc;
        if ((rc = gemm(nb, M, C, v->attn_q, EPI_BIAS_STORE, nullptr, (T*)v->q, C))) return rc;
        if ((rc = gemm(nb, M, C, v->attn_k, EPI_BIAS_STORE, nullptr, (T*)v->kk, C))) return rc;
        for (int b = 0; b < B; ++b) {
            const T* hb = nb + (long long)b * HW * C;
            // V^T[c][tok] = Wv[c][:] . h[tok][:]  (bias of v is added after P.V: softmax rows sum to 1)
            GemmArgs<T> gv{(const T*)v->attn_v.w, nullptr, hb, nullptr, nullptr, (T*)v->vt, C, HW, C, C, C, 0, HW};
            VHIP(c, launch_gemm<T>(gv, EPI_STORE, s, ctx_num_cus(c)));
            // S[i][j] = q_i . k_j  -> fp32
            GemmArgs<T> gs{(const T*)v->q + (long long)b * HW * C, nullptr, (const T*)v->kk + (long long)b * HW * C, nullptr,
                           v->score, nullptr, HW, HW, C, C, C, 0, 0};
            VHIP(c, launch_gemm<T>(gs, EPI_BIAS_SET, s, ctx_num_cus(c)));
            VHIP(c, launch_softmax_rows<T>(v->score, (T*)v->pp, HW, HW, 1.0f / sqrtf((float)C), s));
            // O[i][c] = sum_j P[i][j] V^T[c][j] + b_v[c]
            GemmArgs<T> go{(const T*)v->pp, nullptr, (const T*)v->vt, v->attn_v.b, v->ao + (long long)b * HW * C, nullptr,
                           HW, C, HW, HW, HW, 0, 0};
            VHIP(c, launch_gemm<T>(go, EPI_BIAS_SET, s, ctx_num_cus(c)));
        }
        VHIP(c, launch_cast<T>(v->ao, (T*)v->q, (long long)M * C, s));
        if ((rc = gemm((const T*)v->q, M, C, v->attn_proj, EPI_BIAS_RESID, cur, nullptr, 0))) return rc;  // x + proj_out(h_)
    }
    if ((rc = resnet(v->mid2))) return rc;                                                               // :418
    for (int lv = 3; lv >= 0; --lv) {                                                                    // :421-427
        for (int j = 0; j < 3; ++j)
            if ((rc = resnet(v->up[lv][j]))) return rc;
        if (lv != 0) {   // Upsample: nearest 2x + conv3x3 (:56-59); the upsample is folded into the im2col gather
            VHIP(c, launch_cast<T>(cur, nb, (long long)B * H * H * Cc, s));
            H *= 2;
            if ((rc = conv3(nb, B, H, Cc, 1, v->up_conv[lv], EPI_BIAS_SET, other))) return rc;
            std::swap(cur, other);
        }
    }
    if ((rc = gn(cur, v->norm_out, B, H * H, 1))) return rc;                                             // :433-434
    if ((rc = conv3(nb, B, H, Cc, 0, v->conv_out, EPI_BIAS_SET, v->h1))) return rc;                      // :435  -> [M, 4]
    VHIP(c, launch_vae_output(v->h1, out, B, 3, H * H, 4, s));
    return DD_OK;
}

}  // namespace

extern "C" {

int dd_vae_create(dd_ctx* c, int max_chunk, int max_latent, dd_vae** out) {
    if (!c || !out || max_chunk < 1 || max_latent < 1 || max_latent > 32) return DD_ERR_INVALID;
    dd_vae* v = new (std::nothrow) dd_vae();
    if (!v) return DD_ERR_NOMEM;
    v->ctx = c; v->max_chunk = max_chunk; v->max_latent = max_latent;
    *out = v;
    return DD_OK;
}

int dd_vae_set_param(dd_vae* v, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!v || !name || !data || !shape) return DD_ERR_INVALID;
    if (v->finalized) return ctx_fail(v->ctx, DD_ERR_STATE, "autoencoder already finalized");
    const std::string n(name);
    if (n.rfind("encoder.", 0) == 0 || n.rfind("quant_conv.", 0) == 0) return DD_OK;   // encode side: not on the sampling path
    static const auto exp = expected();
    auto it = exp.find(n);
    if (it == exp.end()) return ctx_fail(v->ctx, DD_ERR_NOT_FOUND, "unexpected key in autoencoder state_dict: " + n);
    std::vector<int64_t> got(shape, shape + ndim);
    if (got != it->second) return ctx_fail(v->ctx, DD_ERR_INVALID, "size mismatch for " + n);
    size_t cnt = 1;
    for (auto d : got) cnt *= (size_t)d;
    HostT& t = v->params[n];
    t.d.assign(data, data + cnt);
    t.shape = got;
    return DD_OK;
}

int dd_vae_finalize(dd_vae* v, int precision) {
    if (!v) return DD_ERR_INVALID;
    dd_ctx* c = v->ctx;
    if (v->finalized) return ctx_fail(c, DD_ERR_STATE, "autoencoder already finalized");
    if (precision != DD_PREC_BF16 && precision != DD_PREC_FP32) return ctx_fail(c, DD_ERR_INVALID, "unknown precision");
    for (auto& kv : expected())
        if (!v->params.count(kv.first)) return ctx_fail(c, DD_ERR_NOT_FOUND, "missing key in autoencoder state_dict: " + kv.first);
    VHIP(c, hipSetDevice(ctx_device(c)));
    v->prec = precision;
    v->es = precision == DD_PREC_BF16 ? 2 : 4;
    const size_t es = v->es;
    const int kt = 128 / (int)es;   // GEMM k-tile in elements

    std::vector<char> host;
    auto align = [&]() { host.resize((host.size() + 255) / 256 * 256); };
    auto put_f32 = [&](const std::vector<float>& a) { align(); const size_t o = host.size(); host.resize(o + a.size() * 4); std::memcpy(&host[o], a.data(), a.size() * 4); return o; };
    auto put_mat = [&](const std::vector<float>& a) {
        align(); const size_t o = host.size(); host.resize(o + a.size() * es);
        if (es == 4) std::memcpy(&host[o], a.data(), a.size() * 4);
        else { unsigned short* d = (unsigned short*)&host[o]; for (size_t i = 0; i < a.size(); ++i) d[i] = h_f2bf(a[i]); }
        return o;
    };
    struct Pend { ConvW* cw; NormW* nw; size_t ow, ob; };
    std::vector<Pend> pend;
    auto P = [&](const std::string& n) -> const std::vector<float>& { return v->params[n].d; };
    // conv weight [Cout, Cin, k, k] -> GEMM matrix [Cout(+pad), (ky, kx, ci) padded to the k-tile]
    auto conv = [&](const std::string& n, ConvW& cw) {
        const auto& shp = v->params[n + ".weight"].shape;
        const int co = (int)shp[0], ci = (int)shp[1], k = (int)shp[2];
        const int K = k * k * ci, kpad = (K + kt - 1) / kt * kt, rows = co == 3 ? 4 : co;
        std::vector<float> m((size_t)rows * kpad, 0.f);
        const auto& w = P(n + ".weight");
        for (int o = 0; o < co; ++o)
            for (int c2 = 0; c2 < ci; ++c2)
                for (int t = 0; t < k * k; ++t) m[(size_t)o * kpad + t * ci + c2] = w[((size_t)o * ci + c2) * k * k + t];
        std::vector<float> b(rows, 0.f);
        std::memcpy(b.data(), P(n + ".bias").data(), co * 4);
        cw.cin = ci; cw.cout = co; cw.k = k; cw.kpad = kpad;
        pend.push_back({&cw, nullptr, put_mat(m), put_f32(b)});
    };
    auto norm = [&](const std::string& n, NormW& nw) {
        nw.c = (int)P(n + ".weight").size();
        pend.push_back({nullptr, &nw, put_f32(P(n + ".weight")), put_f32(P(n + ".bias"))});
    };
    auto res = [&](const std::string& n, ResW& r) {
        norm(n + ".norm1", r.n1); conv(n + ".conv1", r.c1); norm(n + ".norm2", r.n2); conv(n + ".conv2", r.c2);
        if (v->params.count(n + ".nin_shortcut.weight")) conv(n + ".nin_shortcut", r.nin);
    };
    const size_t o_pqw = put_f32(P("post_quant_conv.weight")), o_pqb = put_f32(P("post_quant_conv.bias"));
    conv("decoder.conv_in", v->conv_in);
    res("decoder.mid.block_1", v->mid1);
    norm("decoder.mid.attn_1.norm", v->attn_norm);
    conv("decoder.mid.attn_1.q", v->attn_q); conv("decoder.mid.attn_1.k", v->attn_k);
    conv("decoder.mid.attn_1.v", v->attn_v); conv("decoder.mid.attn_1.proj_out", v->attn_proj);
    res("decoder.mid.block_2", v->mid2);
    for (int lv = 3; lv >= 0; --lv) {
        for (int j = 0; j < 3; ++j) res("decoder.up." + std::to_string(lv) + ".block." + std::to_string(j), v->up[lv][j]);
        if (lv != 0) conv("decoder.up." + std::to_string(lv) + ".upsample.conv", v->up_conv[lv]);
    }
    norm("decoder.norm_out", v->norm_out);
    conv("decoder.conv_out", v->conv_out);
    align();
    VHIP(c, hipMalloc((void**)&v->warena, host.size()));
    VHIP(c, hipMemcpy(v->warena, host.data(), host.size(), hipMemcpyHostToDevice));
    for (auto& p : pend) {
        if (p.cw) { p.cw->w = v->warena + p.ow; p.cw->b = (const float*)(v->warena + p.ob); }
        else { p.nw->g = (const float*)(v->warena + p.ow); p.nw->b = (const float*)(v->warena + p.ob); }
    }
    v->pq_w = (const float*)(v->warena + o_pqw); v->pq_b = (const float*)(v->warena + o_pqb);

    // workspace for one chunk of images at the largest resolution (8 * latent)
    const size_t Bc = v->max_chunk, HWmax = (size_t)(8 * v->max_latent) * (8 * v->max_latent), slack = 512 * 4608 * 4;
    const size_t stream_elems = Bc * HWmax * 256;            // the upsample conv at the last level keeps 256 channels
    const size_t col_elems = Bc * HWmax * 2304;              // widest im2col: 256 channels at full resolution
    const size_t HWm = (size_t)v->max_latent * v->max_latent, Cm = 512;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + slack + 255) / 256 * 256; return o; };
    const size_t o_s0 = take(stream_elems * 4), o_s1 = take(stream_elems * 4), o_h1 = take(stream_elems * 4);
    const size_t o_nb = take(stream_elems * es), o_col = take(col_elems * es);
    const size_t o_part = take((size_t)groupnorm_partials((int)Bc, (int)HWmax) * 4);
    const size_t o_q = take(std::max(Bc * HWm * Cm, stream_elems) * es), o_k = take(Bc * HWm * Cm * es);
    const size_t o_vt = take(Cm * HWm * es), o_pp = take(HWm * HWm * es), o_sc = take(HWm * HWm * 4);
    const size_t o_ao = take(Bc * HWm * Cm * 4), o_z4 = take(Bc * HWm * 4 * 4);
    VHIP(c, hipMalloc((void**)&v->ws, off));
    VHIP(c, hipMemset(v->ws, 0, off));
    VHIP(c, hipStreamSynchronize(nullptr));      // (ditto capi.hip: ordered before any stream the decoder is later run on)
    v->s0 = (float*)(v->ws + o_s0); v->s1 = (float*)(v->ws + o_s1); v->h1 = (float*)(v->ws + o_h1);
    v->nb = v->ws + o_nb; v->col = v->ws + o_col; v->part = (float*)(v->ws + o_part);
    v->q = v->ws + o_q; v->kk = v->ws + o_k; v->vt = v->ws + o_vt; v->pp = v->ws + o_pp;
    v->score = (float*)(v->ws + o_sc); v->ao = (float*)(v->ws + o_ao); v->z4 = (float*)(v->ws + o_z4);
    for (auto& kv : v->params) std::vector<float>().swap(kv.second.d);
    v->finalized = true;
    return DD_OK;
}

int dd_vae_decode(dd_ctx* c, dd_vae* v, const float* z_dev, float* out_dev, int B, int latent_hw, void* stream) {
    if (!c || !v || v->ctx != c) return DD_ERR_INVALID;
    if (!v->finalized) return ctx_fail(c, DD_ERR_STATE, "dd_vae_finalize has not been called");
    if (!z_dev || !out_dev || B < 1) return ctx_fail(c, DD_ERR_INVALID, "null tensor or empty batch");
    if (latent_hw < 1 || latent_hw > v->max_latent) return ctx_fail(c, DD_ERR_INVALID, "latent size outside [1, max_latent]");
    if ((latent_hw * latent_hw) % 64) return ctx_fail(c, DD_ERR_UNSUPPORTED, "latent pixel count must be a multiple of 64 (attention k-tile)");
    hipStream_t s = (hipStream_t)stream;
    const long long zin = 4LL * latent_hw * latent_hw, zout = 3LL * 64 * latent_hw * latent_hw;
    for (int b0 = 0; b0 < B; b0 += v->max_chunk) {
        const int bn = std::min(v->max_chunk, B - b0);
        const int rc = v->prec == DD_PREC_BF16 ? run_decode<bf16_t>(v, z_dev + b0 * zin, out_dev + b0 * zout, bn, latent_hw, s)
                                               : run_decode<float>(v, z_dev + b0 * zin, out_dev + b0 * zout, bn, latent_hw, s);
        if (rc) return rc;
    }
    return DD_OK;
}

void dd_vae_destroy(dd_vae* v) {
    if (!v) return;
    if (v->warena) (void)hipFree(v->warena);
    if (v->ws) (void)hipFree(v->ws);
    delete v;
}

}  // extern "C"
