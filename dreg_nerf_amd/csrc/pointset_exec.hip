// Native executor of the point-set half of the registration network on gfx950: the six pre-norm self/cross-attention encoder
// layers, the shared final norm, the correspondence decoder and the overlap head — forward in ONE C call, backward in ONE C call
// (about 120 + 330 kernel launches that were issued one Python autograd node at a time: ~8 ms of host time per training step,
// which had become the bound of the step: the host issued a step in 17.7 ms against 18.9 ms of GPU time).
//
// Reference: conerf/register/transformer.py:50-86 (TransformerCrossEncoder), :225-299 (forward_pre of a layer),
// conerf/register/nerf_regtr.py:170-206 (encode, final norm per layer output), :273-308,350-394 (CorrespondenceDecoder).
// Python description of record (same kernels, same order; the fp32 parity path): dreg_nerf_amd/transformer_ops.py
// encode_decode_batched.  With fuse = 0 this executor reproduces that path bit for bit (tests/test_hip_pointset_exec.py).
//
// Row space: R rows = the key points of every pair of the step, pair by pair, source set then target set; `probs` tables
// (attn_ops.ProblemTable) give the attention problems.  Activations of the whole pass are kept in a caller-owned arena
// (dreg_ps_arena_bytes(R): ~15 KB per row); parameter gradients are accumulated in place into the caller's fp32 gradient
// buffers; weight / bias gradients run on the caller's second stream next to the data-gradient chain, their split sums, all
// bias column sums and all LayerNorm parameter sums are ONE launch each at the end of the pass.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../../include/dreg_nerf.h"
#ifndef DREG_ELAUNCH
#define DREG_ELAUNCH (-2)
#endif

namespace {

constexpr int E = 256, E3 = 768, FF = 1024, NL = 6, NH = 8;
// parameter table order (value ptr, grad ptr) x PS_NPARAM: per layer, then the shared tail
enum { N1W, N1B, SIW, SIB, SOW, SOB, N2W, N2B, CIW, CIB, COW, COB, N3W, N3B, L1W, L1B, L2W, L2B, PL_COUNT };
enum { FNW = NL * PL_COUNT, FNB, QW, QB, KW, KB, CW, CB, PS_NPARAM };
// linear layers in pack-table order: per layer (in_proj_s, out_proj_s, in_proj_c, out_proj_c, linear1, linear2), then q_proj, k_proj
enum { LIN_SI, LIN_SO, LIN_CI, LIN_CO, LIN_1, LIN_2, LIN_PER_LAYER };
constexpr int NLIN = NL * LIN_PER_LAYER + 2;

struct Lin { int cout, cin, w, b; };   // parameter indices
struct ReduceRec { const float* part; float* dw; int nsplit, Cout, Kpad, ntaps, Cin, Cin_real, accumulate, block0; };
static_assert(sizeof(ReduceRec) == 48, "matches WgradReduceDesc of conv.hip");
struct ColsumRec { const void* g; float* out; float* partial; int M, C, rpc, nch, pblock0, fblock0, accumulate, pad; };
static_assert(sizeof(ColsumRec) == 56, "matches ColsumDesc of fpn_ops.hip");
struct LnRec { const float* part; float* dg; float* db; int nblk, accumulate; };
static_assert(sizeof(LnRec) == 32, "matches LnFinalDesc of pointset.hip");

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

struct LayerBuf {   // arena offsets of one encoder layer
    size_t xa, xb, st1, st2, st3, h1, h2, h3, qkv1, qkv2, o1, o2, lse1, lse2, f;            // forward (kept for the backward pass)
    size_t gF, gB, gQ2, gC, gQ1;                                                        // backward: the linears' output gradients (bf16)
    size_t wg[LIN_PER_LAYER], cs[LIN_PER_LAYER], lnp[3];                                    // split partials, column-sum partials, LayerNorm partials
};
struct Layout {
    int R = -1;
    size_t allx, stf, dec_in, q, k, corr_lse;                                               // forward tail
    LayerBuf L[NL];
    size_t dallx, dallx_bf, dq, dk, ddec, ddec2, dH, dO, dvec, gx[2], lnpf[2], wgq, wgk, csq, csk, ov_ws, ks_ws, cs_ws, tables;
    size_t ks_bytes = 0, total = 0;
    size_t wg_bytes[NLIN];
};

struct Ps {
    float* val[PS_NPARAM];
    float* grad[PS_NPARAM];
    Lin lin[NLIN];
    Layout lay;
    int fuse = 1;
    int group_wgrad = 1;   // with fuse: the linear layers' weight-gradient partials of a backward pass in one launch per tile shape (bit-identical)
    std::vector<hipEvent_t> ev;            // one per linear layer + spares: "its output gradient is complete" (recorded on the caller's stream)
    hipEvent_t ev_done = nullptr;
    // descriptor tables travel through pinned staging (a pageable host-to-device copy drains the stream on ROCm)
    static constexpr int RING = 8;
    static constexpr size_t TABLE_BYTES = 16384;
    char* pinned = nullptr;
    hipEvent_t ring_ev[RING] = {};
    unsigned ring_next = 0;
    // optional HIP-event brackets around every linear-layer launch (forward / data gradient / weight gradient): dreg_ps_set_timing
    struct Timed { hipEvent_t e0, e1; int kind, rows, cin, cout, flags; };     // kind 0 fwd, 1 dgrad, 2 wgrad; flags: 1 addend, 2 fp32 output, 4 split-K workspace offered
    bool timing = false;
    std::vector<Timed> timed;
    size_t timed_used = 0;
};

struct PsScope {   // bracket of one launch (no-op unless timing is on)
    Ps::Timed* t = nullptr; hipStream_t st;
    PsScope(Ps* p, void* stream, int kind, int rows, int cin, int cout, int flags) : st((hipStream_t)stream) {
        if (!p->timing) return;
        if (p->timed_used == p->timed.size()) {
            Ps::Timed n{};
            if (hipEventCreate(&n.e0) != hipSuccess || hipEventCreate(&n.e1) != hipSuccess) return;
            p->timed.push_back(n);
        }
        t = &p->timed[p->timed_used++];
        t->kind = kind; t->rows = rows; t->cin = cin; t->cout = cout; t->flags = flags;
        (void)hipEventRecord(t->e0, st);
    }
    ~PsScope() { if (t) (void)hipEventRecord(t->e1, st); }
};

inline int lin_index(int layer, int which) { return layer * LIN_PER_LAYER + which; }

void build_layout(Ps* p, int R)
{
    Layout& y = p->lay;
    if (y.R == R) return;
    y = Layout();
    y.R = R;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t r = (size_t)R;
    y.allx = take(NL * r * E * 4);
    y.stf = take(NL * r * 2 * 4);
    y.dec_in = take(NL * r * E * 2);
    y.q = take(NL * r * E * 2);
    y.k = take(NL * r * E * 2);
    y.corr_lse = take(NL * r * 4);
    for (int i = 0; i < NLIN; ++i) {
        const Lin& l = p->lin[i];
        const int rows = i < NL * LIN_PER_LAYER ? R : NL * R;
        y.wg_bytes[i] = dreg_conv3d_wgrad_workspace_bytes(rows, 1, 1, 1, l.cin, l.cout, 1, 0);
        const size_t k1 = dreg_conv3d_igemm_workspace_bytes(rows, 1, 1, 1, l.cin, 1, 1, 1, l.cout, 1, 1, 0, 0, 0, 0);
        const size_t k2 = dreg_conv3d_igemm_workspace_bytes(rows, 1, 1, 1, l.cout, 1, 1, 1, l.cin, 1, 1, 0, 1, 0, 0);
        if (k1 > y.ks_bytes) y.ks_bytes = k1;
        if (k2 > y.ks_bytes) y.ks_bytes = k2;
    }
    const size_t lnp = dreg_layernorm_bwd_workspace_bytes(R), lnpf = dreg_layernorm_bwd_workspace_bytes(NL * R);
    for (int l = 0; l < NL; ++l) {
        LayerBuf& b = y.L[l];
        b.xa = take(r * E * 4); b.xb = take(r * E * 4);
        b.st1 = take(r * 8); b.st2 = take(r * 8); b.st3 = take(r * 8);
        b.h1 = take(r * E * 2); b.h2 = take(r * E * 2); b.h3 = take(r * E * 2);
        b.qkv1 = take(r * E3 * 2); b.qkv2 = take(r * E3 * 2);
        b.o1 = take(r * E * 2); b.o2 = take(r * E * 2);
        b.lse1 = take(NH * r * 4); b.lse2 = take(NH * r * 4);
        b.f = take(r * FF * 2);
        b.gF = take(r * FF * 2); b.gB = take(r * E * 2); b.gQ2 = take(r * E3 * 2); b.gC = take(r * E * 2); b.gQ1 = take(r * E3 * 2);
        for (int w = 0; w < LIN_PER_LAYER; ++w) {
            b.wg[w] = take(y.wg_bytes[lin_index(l, w)]);
            b.cs[w] = take(dreg_colsum_workspace_bytes(r, p->lin[lin_index(l, w)].cout));
        }
        for (int k = 0; k < 3; ++k) b.lnp[k] = take(lnp);
    }
    y.dallx = take(NL * r * E * 4);
    y.dallx_bf = take(NL * r * E * 2);
    y.dq = take(NL * r * E * 2); y.dk = take(NL * r * E * 2);
    y.ddec = take(NL * r * E * 2); y.ddec2 = take(NL * r * E * 2);
    y.dH = take(r * E * 2); y.dO = take(r * FF * 2);
    y.dvec = take((size_t)NL * r * 4 > NH * r * 4 ? NL * r * 4 : NH * r * 4);
    y.gx[0] = take(r * E * 4); y.gx[1] = take(r * E * 4);
    y.lnpf[0] = take(lnpf); y.lnpf[1] = take(lnpf);
    y.wgq = take(y.wg_bytes[NLIN - 2]); y.wgk = take(y.wg_bytes[NLIN - 1]);
    y.csq = take(dreg_colsum_workspace_bytes(NL * r, E)); y.csk = take(dreg_colsum_workspace_bytes(NL * r, E));
    y.ov_ws = take(dreg_overlap_bwd_workspace_bytes(NL * R));
    y.ks_ws = take(y.ks_bytes);
    y.cs_ws = take(dreg_colsum_workspace_bytes(NL * r, FF));
    y.tables = take(Ps::TABLE_BYTES);
    y.total = off + 256;
}

#define CK(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// y = x W^T (+bias) (+residual, fp32 output) (relu): the 1x1x1 path of the implicit-GEMM kernels, rows as the batch dimension
int linear_fwd(Ps* p, const Layout& y, char* A, const void* x, const void* wpk, const float* bias, const void* residual, void* out,
               int rows, int cin, int cout, int relu, int out_f32, void* st)
{
    PsScope sc(p, st, 0, rows, cin, cout, (residual ? 1 : 0) | (out_f32 ? 2 : 0) | (y.ks_bytes ? 4 : 0));
    return dreg_conv3d_igemm_ws(x, wpk, out, bias, residual, rows, 1, 1, 1, cin, 1, 1, 1, cout, 1, 1, 0, 0, relu, residual ? 1 : 0, residual ? 1 : 0,
                                residual ? 1 : 0, residual ? 1 : 0, 0, out_f32, y.ks_bytes ? A + y.ks_ws : nullptr, y.ks_bytes, st);
}
// gx = g W  (the data gradient: transposed pack); mask (optional): the forward activation whose ReLU the gradient passes through;
// add (optional): a same-shape bf16 tensor added in the epilogue
int linear_dgrad(Ps* p, const Layout& y, char* A, const void* g, const void* wpk_t, void* gx, const void* mask, const void* add,
                 int rows, int cin, int cout, void* st)
{
    const void* addend = mask ? mask : add;
    PsScope sc(p, st, 1, rows, cin, cout, (addend ? 1 : 0) | (y.ks_bytes ? 4 : 0));
    return dreg_conv3d_igemm_ws(g, wpk_t, gx, nullptr, addend, rows, 1, 1, 1, cout, 1, 1, 1, cin, 1, 1, 0, 1, mask ? 2 : 0, addend ? 1 : 0, addend ? 1 : 0,
                                addend ? 1 : 0, addend ? 1 : 0, 0, 0, y.ks_bytes ? A + y.ks_ws : nullptr, y.ks_bytes, st);
}

}  // namespace

extern "C" {

int dreg_ps_num_params(void) { return PS_NPARAM; }
int dreg_ps_num_linears(void) { return NLIN; }

// params: int64 [dreg_ps_num_params()][2] = (fp32 value ptr, fp32 grad ptr or 0): per encoder layer norm1.{weight,bias},
// self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}, norm2.*, cross_attn.* (same four), norm3.*, linear1.*, linear2.*;
// then transformer_encoder.norm.*, correspondence_decoder.{q_proj,k_proj,conf_logits_decoder}.{weight,bias}.
void* dreg_ps_create(const int64_t* params)
{
    Ps* p = new Ps();
    for (int i = 0; i < PS_NPARAM; ++i) { p->val[i] = (float*)params[2 * i]; p->grad[i] = (float*)params[2 * i + 1]; }
    for (int l = 0; l < NL; ++l) {
        const int o = l * PL_COUNT;
        p->lin[lin_index(l, LIN_SI)] = {E3, E, o + SIW, o + SIB};
        p->lin[lin_index(l, LIN_SO)] = {E, E, o + SOW, o + SOB};
        p->lin[lin_index(l, LIN_CI)] = {E3, E, o + CIW, o + CIB};
        p->lin[lin_index(l, LIN_CO)] = {E, E, o + COW, o + COB};
        p->lin[lin_index(l, LIN_1)] = {FF, E, o + L1W, o + L1B};
        p->lin[lin_index(l, LIN_2)] = {E, FF, o + L2W, o + L2B};
    }
    p->lin[NLIN - 2] = {E, E, QW, QB};
    p->lin[NLIN - 1] = {E, E, KW, KB};
    if (hipHostMalloc((void**)&p->pinned, Ps::RING * Ps::TABLE_BYTES, hipHostMallocDefault) != hipSuccess) { delete p; return nullptr; }
    return p;
}
void dreg_ps_destroy(void* h)
{
    Ps* p = (Ps*)h;
    if (!p) return;
    for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
    if (p->ev_done) (void)hipEventDestroy(p->ev_done);
    for (auto& e : p->ring_ev) if (e) (void)hipEventDestroy(e);
    for (auto& t : p->timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    if (p->pinned) (void)hipHostFree(p->pinned);
    delete p;
}
// 0: the arithmetic of the per-op path, bit for bit (separate ReLU-mask / gradient-sum / cast launches where that path has them);
// 1 (default): ReLU mask and the decoder's gradient sum in the data-gradient epilogues, one LayerNorm backward for the final norm's two
// applications, all bias column sums in one batched launch pair
void dreg_ps_set_fuse(void* h, int fuse) { ((Ps*)h)->fuse = fuse ? 1 : 0; }
void dreg_ps_set_group_wgrad(void* h, int on) { ((Ps*)h)->group_wgrad = on ? 1 : 0; }   // per handle; read at every backward call
// HIP events around every linear-layer launch of the following passes (bench.py's bracketed step); read them back AFTER a device
// synchronisation: info[5 * i] = (kind 0 fwd / 1 dgrad / 2 wgrad, rows, cin, cout, flags: 1 addend, 2 fp32 output, 4 split-K workspace
// offered), ms[i] = the launch's duration.  Returns the number of records (and clears them).
void dreg_ps_set_timing(void* h, int enable) { ((Ps*)h)->timing = enable != 0; }     // (records stay until they are read)
int dreg_ps_read_timings(void* h, int* info, float* ms, int cap)
{
    Ps* p = (Ps*)h;
    int n = 0;
    for (size_t i = 0; i < p->timed_used && n < cap; ++i) {
        const Ps::Timed& t = p->timed[i];
        float v = 0.f;
        if (hipEventElapsedTime(&v, t.e0, t.e1) != hipSuccess) continue;
        info[5 * n] = t.kind; info[5 * n + 1] = t.rows; info[5 * n + 2] = t.cin; info[5 * n + 3] = t.cout; info[5 * n + 4] = t.flags;
        ms[n++] = v;
    }
    p->timed_used = 0;
    return n;
}
size_t dreg_ps_arena_bytes(void* h, int R) { Ps* p = (Ps*)h; build_layout(p, R); return p->lay.total; }

// packs: pointers [dreg_ps_num_linears()][2] = (forward pack [Cout][kpad(Cin)], data-gradient pack [Cin][kpad(Cout)]) of every linear
// layer in table order (dreg_pack_conv_weight, bf16).  feats / pe fp32 [R,256], xyz fp32 [R,3]; probs_self / probs_cross int32
// [nprob][4] (device).  Outputs (caller-owned): cond fp32 [6,R,256], corr fp32 [6,R,3], ov fp32 [6,R].
int dreg_ps_forward(void* h, void* arena, size_t arena_bytes, const int64_t* packs, const float* feats, const float* xyz, const float* pe,
                    const int* probs_self, const int* probs_cross, int nprob, int max_len, int R,
                    float* cond, float* corr, float* ov, void* stream)
{
    Ps* p = (Ps*)h;
    if (R <= 0) return DREG_EINVAL;
    build_layout(p, R);
    const Layout& y = p->lay;
    if (arena_bytes < y.total) return DREG_EINVAL;
    char* A = (char*)arena;
    const float sc = 0.17677669529663687f;   // 1 / sqrt(256 / 8)
    auto pk = [&](int li, int t) { return (const void*)packs[2 * li + t]; };
    const float* xin = feats;
    for (int l = 0; l < NL; ++l) {
        const LayerBuf& b = y.L[l];
        const int o = l * PL_COUNT;
        float* xout = (float*)(A + y.allx) + (size_t)l * R * E;
        // self attention: q = k = v = LN1(x) + pe (transformer.py:238-250)
        CK(dreg_layernorm_fwd(xin, p->val[o + N1W], p->val[o + N1B], pe, A + b.h1, (float*)(A + b.st1), R, E, 1e-5f, 0, stream));
        CK(linear_fwd(p, y, A, A + b.h1, pk(lin_index(l, LIN_SI), 0), p->val[o + SIB], nullptr, A + b.qkv1, R, E, E3, 0, 0, stream));
        CK(dreg_mha_varlen_fwd(A + b.qkv1, A + b.qkv1 + E * 2, A + b.qkv1 + 2 * E * 2, A + b.o1, (float*)(A + b.lse1), probs_self, nprob, max_len, max_len,
                               R, NH, E3, E3, E3, E, sc, 0, stream));
        CK(linear_fwd(p, y, A, A + b.o1, pk(lin_index(l, LIN_SO), 0), p->val[o + SOB], xin, A + b.xa, R, E, E, 0, 1, stream));
        // cross attention: q from a set, k = v from the pair's other set (transformer.py:252-262)
        CK(dreg_layernorm_fwd((const float*)(A + b.xa), p->val[o + N2W], p->val[o + N2B], pe, A + b.h2, (float*)(A + b.st2), R, E, 1e-5f, 0, stream));
        CK(linear_fwd(p, y, A, A + b.h2, pk(lin_index(l, LIN_CI), 0), p->val[o + CIB], nullptr, A + b.qkv2, R, E, E3, 0, 0, stream));
        CK(dreg_mha_varlen_fwd(A + b.qkv2, A + b.qkv2 + E * 2, A + b.qkv2 + 2 * E * 2, A + b.o2, (float*)(A + b.lse2), probs_cross, nprob, max_len, max_len,
                               R, NH, E3, E3, E3, E, sc, 0, stream));
        CK(linear_fwd(p, y, A, A + b.o2, pk(lin_index(l, LIN_CO), 0), p->val[o + COB], A + b.xa, A + b.xb, R, E, E, 0, 1, stream));
        // feed-forward (transformer.py:283-293)
        CK(dreg_layernorm_fwd((const float*)(A + b.xb), p->val[o + N3W], p->val[o + N3B], nullptr, A + b.h3, (float*)(A + b.st3), R, E, 1e-5f, 0, stream));
        CK(linear_fwd(p, y, A, A + b.h3, pk(lin_index(l, LIN_1), 0), p->val[o + L1B], nullptr, A + b.f, R, E, FF, 1, 0, stream));
        CK(linear_fwd(p, y, A, A + b.f, pk(lin_index(l, LIN_2), 0), p->val[o + L2B], A + b.xb, xout, R, FF, E, 0, 1, stream));
        xin = xout;
    }
    // the shared final norm of the six layer outputs: cond (fp32) and LN(x) + pe (compute dtype) from one read (nerf_regtr.py:170-206)
    const int R6 = NL * R;
    CK(dreg_layernorm_fwd2((const float*)(A + y.allx), p->val[FNW], p->val[FNB], pe, R, cond, A + y.dec_in, (float*)(A + y.stf), R6, E, 1e-5f, stream));
    CK(linear_fwd(p, y, A, A + y.dec_in, pk(NLIN - 2, 0), p->val[QB], nullptr, A + y.q, R6, E, E, 0, 0, stream));
    CK(linear_fwd(p, y, A, A + y.dec_in, pk(NLIN - 1, 0), p->val[KB], nullptr, A + y.k, R6, E, E, 0, 0, stream));
    CK(dreg_corr_attention_varlen_fwd(A + y.q, A + y.k, xyz, corr, (float*)(A + y.corr_lse), probs_cross, nprob, max_len, max_len, NL, R, 0.0625f, 0, stream));
    CK(dreg_overlap_fwd(cond, p->val[CW], p->val[CB], ov, R6, stream));
    return DREG_OK;
}

// cond / corr / ov: the forward outputs.  g_cond fp32 [6,R,256], g_corr fp32 [6,R,3], g_ov fp32 [6,R]: each may be null (at least one
// of them must reach the network).  d_feats fp32 [R,256] receives the gradient of `feats`.
// Parameter gradients are accumulated (+=) into the grad pointers given at creation (all of them must be non-null).  aux_stream
// (optional): the caller's second stream for the weight / bias gradient launches; `stream` is NOT joined with it here — the caller
// joins the two before it reads the gradients (train_step does, once for this executor and the trunk's).
int dreg_ps_backward(void* h, void* arena, size_t arena_bytes, const int64_t* packs, const float* feats, const float* xyz, const float* pe,
                     const int* probs_self, const int* probs_cross, int nprob, int max_len, int R,
                     const float* cond, const float* corr, const float* ov, const float* g_cond, const float* g_corr, const float* g_ov,
                     float* d_feats, void* stream, void* aux_stream, int last_only)
{
    Ps* p = (Ps*)h;
    if (R <= 0 || R != p->lay.R) return DREG_EINVAL;     // the arena holds the forward pass of exactly this row space
    const Layout& y = p->lay;
    if (arena_bytes < y.total) return DREG_EINVAL;
    for (int i = 0; i < PS_NPARAM; ++i) if (!p->grad[i]) return DREG_EINVAL;
    char* A = (char*)arena;
    hipStream_t st = (hipStream_t)stream;
    hipStream_t ax = aux_stream && aux_stream != stream ? (hipStream_t)aux_stream : st;
    const bool two = ax != st;
    const float sc = 0.17677669529663687f;
    // last_only: the three gradients belong to the LAST layer's outputs only ([R,256] / [R,3] / [R]) — the training losses read nothing else
    // (train_nerf_regtr.py:178,195,205-206,214,220) — so the heads, the decoder and the final norm are differentiated for that layer's R
    // rows instead of 6R rows of which five sixths carry a zero gradient (the reference's per-layer autograd nodes are never reached either)
    const int LB = last_only ? NL - 1 : 0, nlb = last_only ? 1 : NL;
    const int R6 = nlb * R;
    const size_t rb = (size_t)LB * R;                    // first row of the differentiated block in the [6R, ...] tensors
    const int fuse = p->fuse;
    auto pk = [&](int li, int t) { return (const void*)packs[2 * li + t]; };
    cond += rb * E; corr += rb * 3; ov += rb;
    if (p->ev.size() < (size_t)NLIN + 2) {
        p->ev.resize(NLIN + 2, nullptr);
        for (auto& e : p->ev) if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return DREG_ELAUNCH;
        if (hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming) != hipSuccess) return DREG_ELAUNCH;
    }
    std::vector<ReduceRec> rd;
    std::vector<ColsumRec> cs;
    std::vector<LnRec> ln;
    int rd_blocks = 0, cs_p = 0, cs_f = 0;
    // parameter gradients of linear layer `li` from its output gradient g (bf16 [rows, cout]) and its input x (bf16 [rows, cin]): split
    // partials now (second stream, behind an event: g is complete on the caller's stream here), sums at the end of the pass
    // grouped form (dreg_ps_set_group_wgrad, default with fuse): the split partials of ALL linear layers of the pass by one launch per tile
    // shape at the end of the pass (every operand lives in the arena until then) instead of 36 launches of 50 - 200 workgroups each
    struct GroupRec { char d[160]; int variant, nblocks; };
    std::vector<GroupRec> grp;
    const bool group = fuse && p->group_wgrad && !p->timing;
    auto param_grads = [&](int li, const void* g, const void* x, int rows, size_t wg_off, size_t cs_off) -> int {
        const Lin& l = p->lin[li];
        bool grouped = false;
        if (group) {
            GroupRec gr{};
            if (dreg_linear_wgrad_group_fill(gr.d, g, x, A + wg_off, y.wg_bytes[li], rows, l.cin, l.cout, &gr.variant, &gr.nblocks) == DREG_OK) { grp.push_back(gr); grouped = true; }
        }
        if (!grouped) {
            if (two) { if (hipEventRecord(p->ev[li], st) != hipSuccess || hipStreamWaitEvent(ax, p->ev[li], 0) != hipSuccess) return DREG_ELAUNCH; }
            PsScope sc(p, ax, 2, rows, l.cin, l.cout, 0);
            CK(dreg_conv3d_wgrad_partials(g, x, A + wg_off, y.wg_bytes[li], nullptr, 0, rows, 1, 1, 1, l.cin, l.cin, 1, 1, 1, l.cout, 1, 1, 0, nullptr, ax));
        }
        ReduceRec r{};
        r.part = (const float*)(A + wg_off); r.dw = p->grad[l.w];
        r.nsplit = dreg_conv3d_wgrad_splits(rows, 1, 1, 1, l.cin, l.cout, 1, 0);
        r.Cout = l.cout; r.Kpad = dreg_conv3d_kpad(1, l.cin, 0); r.ntaps = 1; r.Cin = l.cin; r.Cin_real = l.cin; r.accumulate = 1; r.block0 = rd_blocks;
        rd_blocks += dreg_wgrad_reduce_blocks(l.cout, l.cin, 1, r.nsplit);
        rd.push_back(r);
        if (!fuse) return dreg_colsum(g, p->grad[l.b], (float*)(A + y.cs_ws), (size_t)rows, l.cout, 1, 0, ax);
        ColsumRec c{};
        c.g = g; c.out = p->grad[l.b]; c.partial = (float*)(A + cs_off); c.M = rows; c.C = l.cout;
        c.rpc = dreg_colsum_rows_per_chunk((size_t)rows); c.nch = (rows + c.rpc - 1) / c.rpc; c.pblock0 = cs_p; c.fblock0 = cs_f; c.accumulate = 1;
        cs_p += c.nch; cs_f += (l.cout + 3) / 4;
        cs.push_back(c);
        return DREG_OK;
    };
    auto ln_rec = [&](size_t part_off, int wi, int bi, int rows) {
        LnRec r{(const float*)(A + part_off), p->grad[wi], p->grad[bi], dreg_layernorm_bwd_blocks(rows), 1};
        ln.push_back(r);
    };

    // ---- heads.  The overlap head's gradient joins the one `cond` receives from the losses (nerf_regtr.py:384-387)
    const float* Gcond = g_cond;
    if (g_ov) {
        // cond's gradient = what the losses sent (g_cond) + the head's, formed in dallx (free until the final norm's backward writes it)
        CK(dreg_overlap_bwd_acc(cond, p->val[CW], ov, g_ov, (float*)(A + y.dallx) + rb * E, g_cond, p->grad[CW], p->grad[CB], 1, (float*)(A + y.ov_ws), R6, stream));
        Gcond = (const float*)(A + y.dallx) + rb * E;
    }
    // ---- correspondence decoder (nerf_regtr.py:273-308,350-394)
    bool have_dec = false;
    if (g_corr) {
        const size_t ob = rb * E * 2;                     // byte offset of the block in the bf16 [6R,256] tensors
        CK(dreg_corr_attention_varlen_bwd(A + y.q + ob, A + y.k + ob, xyz, corr, g_corr, (const float*)(A + y.corr_lse) + rb, (float*)(A + y.dvec), A + y.dq + ob, A + y.dk + ob,
                                          probs_cross, nprob, max_len, max_len, nlb, R, 0.0625f, 0, stream));
        CK(param_grads(NLIN - 1, A + y.dk + ob, A + y.dec_in + ob, R6, y.wgk, y.csk));
        CK(param_grads(NLIN - 2, A + y.dq + ob, A + y.dec_in + ob, R6, y.wgq, y.csq));
        CK(linear_dgrad(p, y, A, A + y.dk + ob, pk(NLIN - 1, 1), A + y.ddec + ob, nullptr, nullptr, R6, E, E, stream));
        if (fuse) CK(linear_dgrad(p, y, A, A + y.dq + ob, pk(NLIN - 2, 1), A + y.ddec + ob, nullptr, A + y.ddec + ob, R6, E, E, stream));
        else {
            CK(linear_dgrad(p, y, A, A + y.dq + ob, pk(NLIN - 2, 1), A + y.ddec2 + ob, nullptr, nullptr, R6, E, E, stream));
            CK(dreg_add_inplace(A + y.ddec + ob, A + y.ddec2 + ob, (size_t)R6 * E, 0, stream));
        }
        have_dec = true;
    }
    // ---- the final norm's two applications: dallx = LN'(Gcond) + LN'(ddec), with a bf16 copy for the first GEMMs below
    float* dallx = (float*)(A + y.dallx);
    const float* allx = (const float*)(A + y.allx);
    {
        // the differentiated block of the final norm: rows [rb, rb + R6) of allx / its statistics / dallx / the bf16 copy
        const float* ax = allx + rb * E;
        const float* stf = (const float*)(A + y.stf) + rb * 2;
        float* dg = dallx + rb * E;
        char* dgb = A + y.dallx_bf + rb * E * 2;
        const char* dd = A + y.ddec + rb * E * 2;
    if (Gcond && have_dec && fuse) {
        CK(dreg_layernorm_bwd_parts(ax, Gcond, dd, p->val[FNW], stf, dg, nullptr, nullptr, dgb, (float*)(A + y.lnpf[0]), R6, E, 1, stream));
        ln_rec(y.lnpf[0], FNW, FNB, R6);
    } else if (Gcond || have_dec) {
        bool first = true;
        if (Gcond) {
            // Gcond may BE dallx (head-only gradient): the row pass reads a row's dy before it writes the row's dx, each row by one wave
            CK(dreg_layernorm_bwd_parts(ax, Gcond, nullptr, p->val[FNW], stf, dg, nullptr, nullptr, have_dec ? nullptr : dgb, (float*)(A + y.lnpf[0]), R6, E, 1, stream));
            ln_rec(y.lnpf[0], FNW, FNB, R6);
            first = false;
        }
        if (have_dec) {
            CK(dreg_layernorm_bwd_parts(ax, dd, nullptr, p->val[FNW], stf, dg, first ? nullptr : dg, nullptr, dgb, (float*)(A + y.lnpf[1]), R6, E, 0, stream));
            ln_rec(y.lnpf[1], FNW, FNB, R6);
        }
    } else {
        return DREG_EINVAL;     // no gradient reaches the network
    }
    }

    // ---- encoder layers, last to first.  G: fp32 gradient of the layer's output; its bf16 copy is the operand of the GEMMs
    const float* G = dallx + (size_t)(NL - 1) * R * E;
    const void* Gbf = A + y.dallx_bf + (size_t)(NL - 1) * R * E * 2;
    for (int l = NL - 1; l >= 0; --l) {
        const LayerBuf& b = y.L[l];
        const int o = l * PL_COUNT;
        const float* xin = l == 0 ? feats : allx + (size_t)(l - 1) * R * E;
        // Gbf = slice l of dallx_bf: written once with its final value (by the final norm's backward for the last layer, by layer l+1's LN1
        // backward otherwise) and never again — the second stream reads it until the end of the pass
        // feed-forward: x_out = linear2(relu(linear1(LN3(xb)))) + xb
        if (fuse) CK(linear_dgrad(p, y, A, Gbf, pk(lin_index(l, LIN_2), 1), A + b.gF, A + b.f, nullptr, R, FF, E, stream));
        else {
            CK(linear_dgrad(p, y, A, Gbf, pk(lin_index(l, LIN_2), 1), A + y.dO, nullptr, nullptr, R, FF, E, stream));
            CK(dreg_relu_bwd(A + b.f, A + y.dO, A + b.gF, (size_t)R * FF, 0, 0, 0, stream));
        }
        CK(param_grads(lin_index(l, LIN_2), Gbf, A + b.f, R, b.wg[LIN_2], b.cs[LIN_2]));
        CK(linear_dgrad(p, y, A, A + b.gF, pk(lin_index(l, LIN_1), 1), A + y.dH, nullptr, nullptr, R, E, FF, stream));
        CK(param_grads(lin_index(l, LIN_1), A + b.gF, A + b.h3, R, b.wg[LIN_1], b.cs[LIN_1]));
        float* Gb = (float*)(A + y.gx[0]);
        CK(dreg_layernorm_bwd_parts((const float*)(A + b.xb), A + y.dH, nullptr, p->val[o + N3W], (const float*)(A + b.st3), Gb, G, nullptr, A + b.gB,
                                    (float*)(A + b.lnp[2]), R, E, 0, stream));
        ln_rec(b.lnp[2], o + N3W, o + N3B, R);
        // cross attention: xb = out_proj(mha(in_proj(LN2(xa) + pe))) + xa
        CK(linear_dgrad(p, y, A, A + b.gB, pk(lin_index(l, LIN_CO), 1), A + y.dO, nullptr, nullptr, R, E, E, stream));
        CK(param_grads(lin_index(l, LIN_CO), A + b.gB, A + b.o2, R, b.wg[LIN_CO], b.cs[LIN_CO]));
        CK(dreg_mha_varlen_bwd(A + b.qkv2, A + b.qkv2 + E * 2, A + b.qkv2 + 2 * E * 2, A + b.o2, A + y.dO, (const float*)(A + b.lse2), (float*)(A + y.dvec),
                               A + b.gQ2, A + b.gQ2 + E * 2, A + b.gQ2 + 2 * E * 2, probs_cross, nprob, max_len, max_len, R, NH, E3, E3, E3, E, sc, 0, stream));
        CK(linear_dgrad(p, y, A, A + b.gQ2, pk(lin_index(l, LIN_CI), 1), A + y.dH, nullptr, nullptr, R, E, E3, stream));
        CK(param_grads(lin_index(l, LIN_CI), A + b.gQ2, A + b.h2, R, b.wg[LIN_CI], b.cs[LIN_CI]));
        float* Ga = (float*)(A + y.gx[1]);
        CK(dreg_layernorm_bwd_parts((const float*)(A + b.xa), A + y.dH, nullptr, p->val[o + N2W], (const float*)(A + b.st2), Ga, Gb, nullptr, A + b.gC,
                                    (float*)(A + b.lnp[1]), R, E, 0, stream));
        ln_rec(b.lnp[1], o + N2W, o + N2B, R);
        // self attention: xa = out_proj(mha(in_proj(LN1(x) + pe))) + x
        CK(linear_dgrad(p, y, A, A + b.gC, pk(lin_index(l, LIN_SO), 1), A + y.dO, nullptr, nullptr, R, E, E, stream));
        CK(param_grads(lin_index(l, LIN_SO), A + b.gC, A + b.o1, R, b.wg[LIN_SO], b.cs[LIN_SO]));
        CK(dreg_mha_varlen_bwd(A + b.qkv1, A + b.qkv1 + E * 2, A + b.qkv1 + 2 * E * 2, A + b.o1, A + y.dO, (const float*)(A + b.lse1), (float*)(A + y.dvec),
                               A + b.gQ1, A + b.gQ1 + E * 2, A + b.gQ1 + 2 * E * 2, probs_self, nprob, max_len, max_len, R, NH, E3, E3, E3, E, sc, 0, stream));
        CK(linear_dgrad(p, y, A, A + b.gQ1, pk(lin_index(l, LIN_SI), 1), A + y.dH, nullptr, nullptr, R, E, E3, stream));
        CK(param_grads(lin_index(l, LIN_SI), A + b.gQ1, A + b.h1, R, b.wg[LIN_SI], b.cs[LIN_SI]));
        // the layer input's gradient: through LN1, the by-passing residual (Ga) and — it is the previous layer's output — the final norm
        if (l == 0) {
            CK(dreg_layernorm_bwd_parts(xin, A + y.dH, nullptr, p->val[o + N1W], (const float*)(A + b.st1), d_feats, Ga, nullptr, nullptr,
                                        (float*)(A + b.lnp[0]), R, E, 0, stream));
        } else {
            float* Gprev = dallx + (size_t)(l - 1) * R * E;             // in place: dallx[l-1] becomes the previous layer's output gradient
            void* Gprev_bf = A + y.dallx_bf + (size_t)(l - 1) * R * E * 2;
            // (last_only: dallx[l-1] was never produced — the final norm sent nothing to the earlier layers' outputs)
            CK(dreg_layernorm_bwd_parts(xin, A + y.dH, nullptr, p->val[o + N1W], (const float*)(A + b.st1), Gprev, Ga, last_only ? nullptr : Gprev, Gprev_bf,
                                        (float*)(A + b.lnp[0]), R, E, 0, stream));
            G = Gprev; Gbf = Gprev_bf;
        }
        ln_rec(b.lnp[0], o + N1W, o + N1B, R);
    }

    // ---- the pass's tails: descriptor tables through pinned staging, then one launch per kind
    const size_t nb_rd = rd.size() * sizeof(ReduceRec), nb_cs = cs.size() * sizeof(ColsumRec), nb_ln = ln.size() * sizeof(LnRec);
    const size_t o_cs = al(nb_rd), o_ln = o_cs + al(nb_cs);
    // grouped weight-gradient descriptors: one table per tile shape, block0 = exclusive prefix of the workgroup counts inside it
    const int gd = dreg_wgrad_group_desc_bytes();
    std::stable_sort(grp.begin(), grp.end(), [](const GroupRec& a, const GroupRec& b) { return a.variant < b.variant; });
    const size_t o_grp = al(o_ln + nb_ln), nb_grp = grp.size() * (size_t)gd;
    if (gd > 160 || o_grp + nb_grp > Ps::TABLE_BYTES) return DREG_EINVAL;
    if (o_ln + nb_ln > Ps::TABLE_BYTES) return DREG_EINVAL;
    const unsigned slot = p->ring_next++ % Ps::RING;
    if (!p->ring_ev[slot]) { if (hipEventCreateWithFlags(&p->ring_ev[slot], hipEventDisableTiming) != hipSuccess) return DREG_ELAUNCH; }
    else if (hipEventSynchronize(p->ring_ev[slot]) != hipSuccess) return DREG_ELAUNCH;     // the copy that used this slot eight passes ago
    char* host = p->pinned + (size_t)slot * Ps::TABLE_BYTES;
    std::memcpy(host, rd.data(), nb_rd);
    std::memcpy(host + o_cs, cs.data(), nb_cs);
    std::memcpy(host + o_ln, ln.data(), nb_ln);
    struct GroupLaunch { size_t off; int n, variant, blocks; };
    std::vector<GroupLaunch> gl;
    for (size_t i = 0; i < grp.size(); ++i) {
        if (gl.empty() || gl.back().variant != grp[i].variant) gl.push_back(GroupLaunch{o_grp + i * (size_t)gd, 0, grp[i].variant, 0});
        const int b0 = gl.back().blocks;
        std::memcpy(grp[i].d + gd - (int)sizeof(int), &b0, sizeof(int));        // block0 is the descriptor's last field
        std::memcpy(host + o_grp + i * (size_t)gd, grp[i].d, (size_t)gd);
        gl.back().n += 1; gl.back().blocks += grp[i].nblocks;
    }
    // the tables are read on both streams: copy on the caller's stream, the second stream waits for it
    if (hipMemcpyAsync(A + y.tables, host, grp.empty() ? o_ln + nb_ln : o_grp + nb_grp, hipMemcpyHostToDevice, st) != hipSuccess) return DREG_ELAUNCH;
    if (hipEventRecord(p->ring_ev[slot], st) != hipSuccess) return DREG_ELAUNCH;
    if (two && hipStreamWaitEvent(ax, p->ring_ev[slot], 0) != hipSuccess) return DREG_ELAUNCH;
    // records 0 and (without fuse) 1 are the final norm's two applications: same destination, so the second one is a launch of its own
    const int ln_dup = (ln.size() >= 2 && ln[0].dg == ln[1].dg) ? 1 : 0;
    if (ln_dup) CK(dreg_layernorm_bwd_final_batched(A + y.tables + o_ln, 1, stream));
    CK(dreg_layernorm_bwd_final_batched(A + y.tables + o_ln + ln_dup * sizeof(LnRec), (int)ln.size() - ln_dup, stream));
    for (const GroupLaunch& q : gl) CK(dreg_wgrad_group_launch(A + y.tables + q.off, q.n, q.variant, q.blocks, ax));   // (every output gradient is complete: the event above follows the whole pass)
    CK(dreg_wgrad_reduce_batched(A + y.tables, (int)rd.size(), 0, rd_blocks, ax));
    if (fuse) CK(dreg_colsum_batched(A + y.tables + o_cs, (int)cs.size(), cs_p, cs_f, ax));
    return DREG_OK;
}

}  // extern "C"
