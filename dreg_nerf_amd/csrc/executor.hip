// Native executor of the FPN3D trunk (forward + backward) on gfx950: a recorded op program — convolutions, per-grid
// BatchNorm (+residual, +ReLU), max-pool and the active-set head convolutions — is issued from C++ in one call per pass,
// so a training step spends ~1 ms of host time on the ~900 kernel launches of the feature network instead of one Python
// autograd node per layer.  The program is recorded once per (batch, resolution) by dreg_nerf_amd/trunk_exec.py from the
// same Python description of the network that drives the per-op path (regtr.NeRFRegTr.fpn), and both paths call the same
// kernels in the same order: forward results are bit-identical.
//
// Reference call sites of the network this executes: conerf/model/resnet3d.py:86-161 (Bottleneck / ResNet forward),
// conerf/model/feature_pyramid_net.py:97-127 (FeaturePyramid.forward), conerf/register/nerf_regtr.py:131-137.
//
// Memory: the caller owns one arena (dreg_exec_arena_bytes) holding every activation (kept for the backward pass), the
// BatchNorm statistics, the activation gradients and the kernels' scratch; weight gradients are accumulated in place into
// the caller's fp32 gradient buffers (FlatAdamW's flat buffer), packed bf16 weights live in a second caller-owned buffer
// refreshed by dreg_exec_repack (one launch) after every optimizer update.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>
#include <string>
#include <cstdio>
#include "../../include/dreg_nerf.h"
#ifndef DREG_ELAUNCH
#define DREG_ELAUNCH (-2)
#endif

namespace {

enum { OP_CONV = 0, OP_BN = 1, OP_MAXPOOL = 2, OP_CONV_ROWS = 3 };
constexpr int OP_INTS = 16;      // ints per op record
constexpr int TENSOR_INTS = 5;   // B, D, H, W, C
constexpr int RL = 8;           // int64 fields per row list: rows, count, brick tiles, ntiles, halo_vox, nbr, rows_sorted, reserved
constexpr int PARAM_I64 = 5;     // value ptr, grad ptr, d0, d1, ksz  (conv weight: [d0=Cout][d1=Cin][ksz^3]; vectors: d0 = length)

struct Tensor { int B, D, H, W, C; size_t bytes, off, goff; };
struct Param {
    float* val; float* grad; int d0, d1, ksz;
    size_t pk_fwd = SIZE_MAX, pk_dgrad = SIZE_MAX, pk_cls = SIZE_MAX;   // offsets in the pack buffer
    size_t pk_halo_fwd = SIZE_MAX, pk_halo_dgrad = SIZE_MAX;            // halo-kernel packs (conv_halo.hip)
    size_t pk_brick_fwd = SIZE_MAX, pk_brick_dgrad = SIZE_MAX;          // staged-neighbourhood packs of the active-set 3^3 layers (conv_brick.hip)
    int cin_pad = 0;
};
struct Op {
    int kind, in, out, in2, w, b, p2, p3, p4, ksz, stride, pad, relu, add_same, rows_out, rows_in;
    size_t aux0 = 0, aux1 = 0;   // BN: scale_shift / mean_rstd; max-pool: argmax
    size_t wg_off = 0, wg_bytes = 0;   // OP_CONV*: this layer's weight-gradient split partials (kept until the batched reduce)
    int rd = -1;                 // index of its record in the reduce table
    int ds_rows = -1;            // OP_CONV_ROWS with an upsampled addend produced by another active-set convolution: that one's output row list
    int sparse_gx = 0;           // OP_CONV_ROWS: the input's gradient buffer has no other writer: kept zero outside the rows of the last step
    size_t cl_off = 0;           //   arena copy of the row list that was written (cleared at the start of the next backward)
    int cl_count = 0;
    int ds_sparse = 0;           // OP_CONV_ROWS with ds_rows >= 0 that is the only writer of its addend's gradient: that buffer is kept zero outside the rows of the last step as well
    size_t cl2_off = 0;
    int cl2_count = 0;
    int pool = -1;               // OP_BN: index of the max-pool op fused behind it; OP_MAXPOOL: index of the BatchNorm it is fused into
    int sstem = 0;               // OP_BN fused with its max-pool behind a row-list convolution over a sparse volume (fpn_ops.hip "stem over a sparse volume")
    int sstem_conv = -1, sstem_lat = -1;   //   that convolution; the active-set convolution that also reads the activation (-1: none)
    size_t xam_off = 0, pmask_off = 0;     //   raw x of the arg-max voxels [B,Do,Ho,Wo,C] bf16; pooled-window flags
    int halo = 0;                // OP_CONV: bit 0 = forward, bit 1 = data gradient run on the halo kernel
    int stats_bn = -1;           // OP_CONV whose bf16 output feeds a large-path BatchNorm: that op's index (its chunk sums come from this convolution's epilogue)
    size_t stats_off = 0;        // OP_BN with such a producer: its own chunk-sum buffer [B][V / 128][C][2] (the shared workspace may be used in between)
    int stats_conv = -1;
    int grp = 0;                 // OP_CONV of the 16^3 / 8^3 / 4^3 levels whose weight-gradient partials are written by the pass's grouped launch (dreg_wgrad_group_launch)
    int fold_sk_fwd = 0, fold_sk_bwd = 0;   // OP_CONV: a split-K launch of its forward / data gradient leaves the slices to the one-launch BatchNorm right behind it (forward: op + 1, backward: op - 1)
    int fold_into = -1, res_from = -1;   // OP_BN (ReLU-free, large path) whose output only feeds BatchNorm `fold_into` as its residual: not applied, that layer applies it on the fly (res_from = this op)
    int bt = -1;                 // OP_BN on the one-launch small path: index of its record in the two BatchNorm tail tables
    size_t keep_var = 0, keep_sums = 0;   //   per-grid variances [B][C] / gradient sums [B][C][2] kept until the batched tail launch
};
struct HaloPack { const float* w; size_t off; int Cout, Cin, transposed, brick; };
struct PackRec { const float* w; void* out; int Cout, Cin_real, inner, ntaps, for_dgrad, Kpad, dtype, row0; };
static_assert(sizeof(PackRec) == 48, "matches PackDesc of conv.hip");

struct ReduceRec { const float* part; float* dw; int nsplit, Cout, Kpad, ntaps, Cin, Cin_real, accumulate, block0; };
static_assert(sizeof(ReduceRec) == 48, "matches WgradReduceDesc of conv.hip");

struct BnTailRec { const float* a; const float* b; float* o0; float* o1; int B, V, C, block0; };
static_assert(sizeof(BnTailRec) == 48, "matches BnTailDesc of fpn_ops.hip");

struct TimedLaunch { hipEvent_t e0, e1; int op, kind, variant; };   // variant: 0 = implicit-GEMM dispatch, 1 = the same with an in-place addend (accumulating data gradient), 2 = conv_brick.hip (the halo kernel is dreg_exec_op_halo)

struct Exec {
    std::vector<Tensor> t;
    std::vector<Op> ops;
    std::vector<Param> prm;
    std::vector<char> needs_grad;          // per tensor: some parameter lies upstream
    size_t act_bytes = 0, arena_bytes = 0, pack_bytes = 0;
    size_t off_bn_ws = 0, off_coef = 0, off_ks = 0, off_rd = 0, off_cs = 0, off_tmp = 0;
    size_t sz_ks = 0;
    std::vector<ReduceRec> reduce;         // one per convolution with a weight gradient, in op order; part = arena offset
    const void* sparse_arena = nullptr;     // arena whose sparse gradient buffers are in the all-zero state
    std::vector<ReduceRec> reduce_abs;     // the same with absolute addresses for the arena it was last uploaded to
    int reduce_blocks = 0;
    const void* reduce_arena = nullptr;
    // small BatchNorms: running-statistics update (forward) and dgamma / dbeta (backward) of all layers of a pass in one launch each
    std::vector<BnTailRec> bn_fwd, bn_bwd; // a / b = arena offsets until uploaded
    size_t off_bnf = 0, off_bnb = 0;
    int bn_blocks = 0;
    std::vector<PackRec> packs;            // with out = offset (patched on export)
    std::vector<HaloPack> halo_packs;      // refreshed by dreg_exec_repack next to the batched pack launch
    int pack_rows = 0, pack_max_floats = 0;
    int out_slot = -1;
    bool timing = false;
    std::vector<TimedLaunch> timed;
    size_t timed_used = 0;
    // weight / bias gradients run on a second stream next to the data-gradient chain (they only share their input gy)
    hipStream_t aux = nullptr;
    std::vector<hipEvent_t> ev;            // per op: "gy of this op is complete" (recorded on the caller's stream)
    hipEvent_t ev_done = nullptr;
    bool use_aux = true;
    const uint8_t* in_rowocc = nullptr;   // output-row occupancy of the convolution that reads the network input (may be null)
    char* pack_base = nullptr;            // device address of the pack buffer (dreg_exec_export_pack_table)
    std::vector<char> written;            // backward pass state, kept across the segments of dreg_exec_backward_range
    bool aux_used = false;
    // grouped weight gradients: the eligible ops (program order), their descriptor tables per tile shape (built per arena), the launch point
    struct GrpLaunch { size_t off; int n, variant, blocks; };
    std::vector<int> grp_ops;
    std::vector<GrpLaunch> grp_launch;
    size_t off_grp = 0;
    bool grp_ready = false;
    int pend_sk_bn = -1, pend_sk_n = 0;    // a BatchNorm op whose dy lies in the split-K workspace as pend_sk_n slices (set by the data gradient of the op behind it; survives a segment boundary)
    size_t pend_sk_slice = 0;
    const void* fwd_rows_arena = nullptr;  // the arena whose row-list convolution outputs (rows_out >= 0) are known to be zero outside their last lists
    // guard mode (opt.guard): a poisoned band behind every region of the arena
    struct Band { size_t off; const char* tag; int idx; };
    std::vector<Band> bands;
    size_t off_guard_tab = 0, off_guard_res = 0;
    const void* guard_arena = nullptr;     // the arena whose bands are filled
    int guard_op = -1, guard_pass = 0; long long guard_band = -1;
    dreg_exec_opts opt;                    // creation options (include/dreg_nerf.h): per handle — the library has no process-global switches
};

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
constexpr size_t GUARD_BYTES = 64 * 1024;
// The arena's bump pointer.  In guard mode every `off += bytes` is followed by a poisoned band (named by the tag / index in force).
struct Bump {
    size_t v = 0;
    std::vector<Exec::Band>* bands = nullptr;
    const char* tag = "?"; int idx = -1;
    operator size_t() const { return v; }
    Bump& operator+=(size_t n) { v += n; if (bands && n) { bands->push_back(Exec::Band{v, tag, idx}); v += GUARD_BYTES; } return *this; }
    void name(const char* t, int i) { tag = t; idx = i; }
};
inline int out_dim(int i, int k, int s, int p) { return (i + 2 * p - k) / s + 1; }

struct Scope {   // optional HIP-event bracket of one launch group
    Exec* e; hipStream_t st; TimedLaunch* tl = nullptr;
    Scope(Exec* ex, hipStream_t s, int op, int kind) : e(ex), st(s) {
        if (!e->timing) return;
        if (e->timed_used == e->timed.size()) {
            TimedLaunch n{}; n.op = op; n.kind = kind;
            if (hipEventCreate(&n.e0) != hipSuccess || hipEventCreate(&n.e1) != hipSuccess) return;
            e->timed.push_back(n);
        }
        tl = &e->timed[e->timed_used++];
        tl->op = op; tl->kind = kind; tl->variant = 0;
        (void)hipEventRecord(tl->e0, st);
    }
    ~Scope() { if (tl) (void)hipEventRecord(tl->e1, st); }
    void variant(int v) { if (tl) tl->variant = v; }
};

#define CK(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

bool s2_class_ok(const Param& p, int ksz, int stride, int pad)
{
    return stride == 2 && ((ksz == 3 && pad == 1) || (ksz == 1 && pad == 0)) && p.d0 % 64 == 0 && p.d1 % 64 == 0 && dreg_conv_get_glds();
}

}  // namespace

extern "C" {

// tensors: int32 [nt][5] (B,D,H,W,C; bf16 NDHWC; tensor 0 is the external input), ops: int32 [nops][16]
//   { kind, in, out, in2 (addend / residual, -1), w, b, p2, p3, p4 (BN: gamma=w, beta=b, running_mean, running_var), ksz, stride,
//     pad, relu, add_same, rows_out, rows_in (row-list ids of OP_CONV_ROWS) },
// params: int64 [np][5] { fp32 value ptr, fp32 grad ptr (0 = frozen), d0, d1, ksz }.  The last op's output is the result.
void dreg_exec_default_opts(dreg_exec_opts* o)
{
    std::memset(o, 0, sizeof(*o));
    o->sparse_grads = 1; o->bn_batch_tails = 1; o->fuse_stem = 1; o->sparse_stem = 1; o->fold_res_bn = 1; o->fold_splitk = 1;
    o->group_wgrad = 1; o->s2_accumulate = 1; o->fuse_bn_stats = 1; o->brick = 1; o->defer_head_pg = 0;
}
void* dreg_exec_create(const int* tensors, int nt, const int* ops, int nops, const int64_t* params, int np)
{
    return dreg_exec_create_opts(tensors, nt, ops, nops, params, np, nullptr);
}
void* dreg_exec_create_opts(const int* tensors, int nt, const int* ops, int nops, const int64_t* params, int np, const dreg_exec_opts* opts)
{
    Exec* e = new Exec();
    if (opts) e->opt = *opts; else dreg_exec_default_opts(&e->opt);
    e->opt.brick &= 3;
    const int g_sparse_grads = e->opt.sparse_grads, g_bn_batch_tails = e->opt.bn_batch_tails, g_fuse_stem = e->opt.fuse_stem, g_sparse_stem = e->opt.sparse_stem,
              g_fold_res_bn = e->opt.fold_res_bn, g_fold_splitk = e->opt.fold_splitk, g_group_wgrad = e->opt.group_wgrad, g_fuse_bn_stats = e->opt.fuse_bn_stats,
              g_brick = e->opt.brick;
    e->t.resize(nt);
    for (int i = 0; i < nt; ++i) {
        Tensor& t = e->t[i];
        t.B = tensors[i * TENSOR_INTS]; t.D = tensors[i * TENSOR_INTS + 1]; t.H = tensors[i * TENSOR_INTS + 2];
        t.W = tensors[i * TENSOR_INTS + 3]; t.C = tensors[i * TENSOR_INTS + 4];
        t.bytes = align256((size_t)t.B * t.D * t.H * t.W * t.C * 2);
    }
    e->prm.resize(np);
    for (int i = 0; i < np; ++i) {
        Param& p = e->prm[i];
        p.val = (float*)params[i * PARAM_I64]; p.grad = (float*)params[i * PARAM_I64 + 1];
        p.d0 = (int)params[i * PARAM_I64 + 2]; p.d1 = (int)params[i * PARAM_I64 + 3]; p.ksz = (int)params[i * PARAM_I64 + 4];
    }
    e->ops.resize(nops);
    for (int i = 0; i < nops; ++i) {
        const int* r = ops + i * OP_INTS;
        Op& o = e->ops[i];
        o.kind = r[0]; o.in = r[1]; o.out = r[2]; o.in2 = r[3]; o.w = r[4]; o.b = r[5]; o.p2 = r[6]; o.p3 = r[7]; o.p4 = r[8];
        o.ksz = r[9]; o.stride = r[10]; o.pad = r[11]; o.relu = r[12]; o.add_same = r[13]; o.rows_out = r[14]; o.rows_in = r[15];
        if (o.in < 0 || o.in >= nt || o.out <= 0 || o.out >= nt || o.in2 >= nt) { delete e; return nullptr; }
    }
    e->out_slot = nops ? e->ops.back().out : -1;

    // which tensors carry a gradient: everything downstream of a trainable parameter
    e->needs_grad.assign(nt, 0);
    for (const Op& o : e->ops) {
        bool g = e->needs_grad[o.in] || (o.in2 >= 0 && e->needs_grad[o.in2]);
        if (o.kind == OP_CONV || o.kind == OP_CONV_ROWS) g = g || e->prm[o.w].grad || (o.b >= 0 && e->prm[o.b].grad);
        if (o.kind == OP_BN) g = g || e->prm[o.w].grad || e->prm[o.b].grad;
        e->needs_grad[o.out] = g;
    }

    // stem: a ReLU BatchNorm without residual whose output only feeds a 3^3/2 max-pool runs fused with it (fpn_ops.hip): the
    // full-resolution activation between them is never written
    for (size_t i = 0; i + 1 < e->ops.size(); ++i) {
        Op& o = e->ops[i];
        if (o.kind != OP_BN || o.in2 >= 0 || o.out == e->out_slot) continue;
        int users = 0, pool = -1;
        for (size_t j = 0; j < e->ops.size(); ++j) {
            const Op& q = e->ops[j];
            if (q.in == o.out || q.in2 == o.out) { ++users; if (q.kind == OP_MAXPOOL && q.in == o.out && j > i) pool = (int)j; }
        }
        if (users == 1 && pool >= 0 && e->t[o.out].C % 8 == 0 && g_fuse_stem) { o.pool = pool; e->ops[pool].pool = (int)i; }
        // ... or, behind a convolution computed on a row list (its output is zero elsewhere), with at most one more reader that is an
        // active-set convolution (the FPN's finest lateral): everything runs from the two row lists
        if (o.pool < 0 && pool >= 0 && o.relu && e->t[o.out].C % 8 == 0 && g_sparse_stem && g_fuse_stem) {
            int prod = -1, writers = 0, lat = -1, others = 0;
            for (size_t j = 0; j < e->ops.size(); ++j) {
                const Op& q = e->ops[j];
                if (q.out == o.in) { prod = (int)j; ++writers; }
                if ((int)j == pool) continue;
                if (q.in2 == o.out) ++others;
                else if (q.in == o.out) { if (q.kind == OP_CONV_ROWS && q.rows_in >= 0 && lat < 0) lat = (int)j; else ++others; }
            }
            if (writers == 1 && others == 0 && e->ops[prod].kind == OP_CONV && e->ops[prod].rows_out >= 0 && o.b >= 0) {
                o.pool = pool; e->ops[pool].pool = (int)i; o.sstem = 1; o.sstem_conv = prod; o.sstem_lat = lat;
            }
        }
    }

    // the downsample branch of a stage's first bottleneck (resnet3d.py:104-110: conv -> BatchNorm, no ReLU) is only ever added inside the
    // block's last BatchNorm: on the large path that BatchNorm applies the branch's scale / shift while it reads the branch's raw
    // convolution output, and the branch's own apply pass (read + write of a [B,V,C] tensor) is not launched
    if (g_fold_res_bn)
        for (size_t j = 0; j < e->ops.size(); ++j) {
            Op& bj = e->ops[j];
            const Tensor& xj = e->t[bj.in];
            if (bj.kind != OP_BN || bj.in2 >= 0 || bj.relu || bj.pool >= 0 || bj.out == e->out_slot || dreg_bn_small(xj.B, xj.D * xj.H * xj.W, xj.C, 0)) continue;
            int users = 0, k = -1;
            for (size_t q = 0; q < e->ops.size(); ++q) {
                const Op& o = e->ops[q];
                if (o.in == bj.out) ++users;
                if (o.in2 == bj.out) { ++users; if (o.kind == OP_BN && q > j && o.pool < 0) k = (int)q; }
            }
            if (users != 1 || k < 0 || e->ops[k].res_from >= 0) continue;
            bj.fold_into = k; e->ops[k].res_from = (int)j;
        }

    // split-K convolutions of the 8^3 / 4^3 levels (resnet3d.py:95-113 conv2 of layer3 / layer4) directly in front of (forward) / behind (backward)
    // a register-resident one-launch BatchNorm: the BatchNorm kernel sums the fp32 slices itself, the convolution's reduce launch is dropped
    if (g_fold_splitk)
        for (size_t i = 0; i < e->ops.size(); ++i) {
            Op& o = e->ops[i];
            if (o.kind != OP_CONV || o.b >= 0 || o.in2 >= 0 || o.relu || o.rows_out >= 0) continue;
            auto users = [&](int slot) { int u = 0; for (const Op& q : e->ops) u += (q.in == slot) + (q.in2 == slot); return u; };
            const Tensor& y = e->t[o.out];
            if (i + 1 < e->ops.size() && o.out != e->out_slot) {
                const Op& bn = e->ops[i + 1];
                if (bn.kind == OP_BN && bn.in == o.out && bn.pool < 0 && users(o.out) == 1 && dreg_bn_small_in_regs(y.B, y.D * y.H * y.W, y.C, 0)) o.fold_sk_fwd = 1;
            }
            const Tensor& x = e->t[o.in];
            if (i >= 1 && o.in != 0) {
                const Op& bn = e->ops[i - 1];
                if (bn.kind == OP_BN && bn.out == o.in && bn.pool < 0 && users(o.in) == 1 && dreg_bn_small_in_regs(x.B, x.D * x.H * x.W, x.C, 0) && o.stride == 1) o.fold_sk_bwd = 1;
            }
        }

    // arena: activations | BN statistics / argmax | gradients | scratch
    Bump off;
    if (e->opt.guard) { off.bands = &e->bands; off.name("arena start", 0); off += 256; }
    for (int i = 1; i < nt; ++i) { off.name("activation", i); e->t[i].off = off; off += e->t[i].bytes; }
    size_t max_tensor = 0, bn_ws = 0, coef = 0, cs = 0;
    for (Op& o : e->ops) {
        const Tensor& x = e->t[o.in];
        off.name(o.kind == OP_BN ? "BatchNorm statistics / stem scratch of op" : (o.kind == OP_MAXPOOL ? "max-pool argmax of op" : "conv op"), (int)(&o - e->ops.data()));
        if (o.kind == OP_BN) {
            const size_t sb = align256((size_t)x.B * x.C * 2 * sizeof(float));
            o.aux0 = off; off += sb; o.aux1 = off; off += sb;
            const size_t V = (size_t)x.D * x.H * x.W;
            size_t w = (size_t)x.B * dreg_bn_num_chunks((int)V) * x.C * 2 * sizeof(float);
            if (o.sstem) {
                const Tensor& p = e->t[e->ops[o.pool].out];
                const size_t w2 = dreg_sparse_stem_workspace_floats(p.B, p.D, p.H, p.W, p.C) * sizeof(float);
                if (w2 > w) w = w2;
                o.xam_off = off; off += align256((size_t)p.B * p.D * p.H * p.W * p.C * 2);
                o.pmask_off = off; off += align256((size_t)p.B * p.D * p.H * p.W);
            }
            if (w > bn_ws) bn_ws = w;
            if (sb > coef) coef = sb;
            bool shared = false;          // parameters used by two layers: both keep their own tail launches
            for (const Op& q : e->ops) shared = shared || (&q != &o && q.kind == OP_BN && (q.w == o.w || q.b == o.b || q.p2 == o.p2 || q.p3 == o.p3));
            if (g_bn_batch_tails && o.pool < 0 && !shared && dreg_bn_small(x.B, (int)V, x.C, 0)) {
                o.keep_var = off; off += align256((size_t)x.B * x.C * sizeof(float));
                o.keep_sums = off; off += sb;
                BnTailRec f{}, b{};
                f.a = (const float*)o.aux1; f.b = (const float*)o.keep_var; f.o0 = e->prm[o.p2].val; f.o1 = e->prm[o.p3].val;
                b.a = (const float*)o.keep_sums; b.b = nullptr; b.o0 = e->prm[o.w].grad; b.o1 = e->prm[o.b].grad;
                f.B = b.B = x.B; f.V = b.V = (int)V; f.C = b.C = x.C; f.block0 = b.block0 = e->bn_blocks;
                e->bn_blocks += (x.C + 255) / 256;
                o.bt = (int)e->bn_fwd.size();
                e->bn_fwd.push_back(f); e->bn_bwd.push_back(b);
            }
        } else if (o.kind == OP_MAXPOOL) {
            const Tensor& y = e->t[o.out];
            o.aux0 = off; off += align256((size_t)y.B * y.D * y.H * y.W * y.C);
        } else {
            const Param& w = e->prm[o.w];
            const Tensor& y = e->t[o.out];
            const size_t k1 = dreg_conv3d_igemm_workspace_bytes(x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0, o.ksz, o.stride, o.pad, 0, o.in2 >= 0, 0);
            const size_t k2 = dreg_conv3d_igemm_workspace_bytes(x.B, y.D, y.H, y.W, w.d0, x.D, x.H, x.W, w.d1, o.ksz, o.stride, o.pad, 1, 0, 0);
            if (k1 > e->sz_ks) e->sz_ks = k1;
            if (k2 > e->sz_ks) e->sz_ks = k2;
            if (w.grad) {
                // every layer keeps its own split partials: the splits of a whole backward segment are summed by ONE launch
                o.wg_bytes = dreg_conv3d_wgrad_workspace_bytes(x.B, y.D, y.H, y.W, x.C, w.d0, o.ksz, 0);
                bool shared = false;      // a parameter used by two layers: the later one sums its splits with its own launch
                for (const ReduceRec& q : e->reduce) shared = shared || q.dw == w.grad;
                if (shared) o.rd = -2;
                else {
                    ReduceRec r{};
                    r.dw = w.grad; r.nsplit = dreg_conv3d_wgrad_splits(x.B, y.D, y.H, y.W, x.C, w.d0, o.ksz, 0);
                    r.Cout = w.d0; r.Kpad = dreg_conv3d_kpad(o.ksz, x.C, 0); r.ntaps = o.ksz * o.ksz * o.ksz; r.Cin = x.C; r.Cin_real = w.d1;
                    r.accumulate = (o.kind == OP_CONV_ROWS || o.rows_out >= 0) ? 3 : 1;     // row-list launches write fewer slices than the dense rule sizes (bit 1: count behind the slices)
                    r.block0 = e->reduce_blocks;
                    e->reduce_blocks += dreg_wgrad_reduce_blocks(w.d0, w.d1, o.ksz, r.nsplit);
                    o.rd = (int)e->reduce.size();
                    e->reduce.push_back(r);
                }
            }
            const size_t c = dreg_colsum_workspace_bytes((size_t)y.B * y.D * y.H * y.W, w.d0);
            if (c > cs) cs = c;
        }
    }
    // BatchNorm statistics from the producing convolution's epilogue (training forward): large-path layers whose input is written by
    // exactly one plain convolution of this program
    if (g_fuse_bn_stats) {
        for (size_t j = 0; j < e->ops.size(); ++j) {
            Op& bn = e->ops[j];
            if (bn.kind != OP_BN || bn.pool >= 0) continue;
            const Tensor& x = e->t[bn.in];
            const int V = x.D * x.H * x.W;
            if (V % 128 != 0 || dreg_bn_small(x.B, V, x.C, 0)) continue;
            int prod = -1, writers = 0;
            for (size_t i = 0; i < j; ++i) if (e->ops[i].out == bn.in) { prod = (int)i; ++writers; }
            if (writers != 1 || e->ops[prod].kind != OP_CONV || e->ops[prod].stats_bn >= 0) continue;
            {   // a forward on the 256-channel halo kernel (decided below by the same predicate) has no statistics epilogue: that BatchNorm runs its own
                // pass; the 64-channel kernel has one (dreg_conv3_halo_n_bnstats)
                const Op& pc = e->ops[prod];
                const Tensor& px = e->t[pc.in];
                const Param& pp = e->prm[pc.w];
                if (!pc.relu && pp.d1 == px.C && pp.d0 != 64 && dreg_conv3_halo_use(px.B, px.D, px.H, px.W, px.C, pp.d0, pc.ksz, pc.stride, pc.pad)) continue;
            }
            e->ops[prod].stats_bn = (int)j;
            bn.stats_conv = prod;
            off.name("epilogue chunk sums of BatchNorm op", (int)j);
            bn.stats_off = off; off += align256((size_t)x.B * (V / 128) * x.C * 2 * sizeof(float));
        }
    }
    e->act_bytes = off;
    for (int i = 1; i < nt; ++i) {
        if (e->t[i].bytes > max_tensor) max_tensor = e->t[i].bytes;
        off.name("gradient of activation", i);
        if (e->needs_grad[i] && i != e->out_slot) { e->t[i].goff = off; off += e->t[i].bytes; } else e->t[i].goff = SIZE_MAX;
    }
    // the gradient of an upsample-add inside the active-set head is zero off the parents of the fine active set, which is the output
    // row list of the coarse convolution that produced the addend: sum the children there only
    for (Op& o : e->ops) {
        if (o.kind != OP_CONV_ROWS || o.in2 < 0 || o.add_same || !e->needs_grad[o.in2] || !g_sparse_grads) continue;
        for (const Op& q : e->ops) if (q.out == o.in2 && q.kind == OP_CONV_ROWS) o.ds_rows = q.rows_out;
    }
    // gradient buffers written by exactly one active-set convolution (the lateral sums in front of the head's 3^3 convolutions,
    // 1 GB at 8 x 64^3 x 256): instead of a dense memset per step they are kept zero outside the rows of the step before, and
    // those rows (a copy of the row list lives in the arena) are cleared when the next backward starts
    for (Op& o : e->ops) {
        if (o.kind != OP_CONV_ROWS || !e->needs_grad[o.in] || o.in == 0 || e->t[o.in].goff == SIZE_MAX) continue;
        int writers = 0;
        for (const Op& q : e->ops) {
            if (q.kind == OP_MAXPOOL && q.pool >= 0 && e->ops[q.pool].sstem) continue;    // its gradient never passes through the dense buffer
            if ((q.in == o.in || q.in2 == o.in) && e->needs_grad[q.out]) ++writers;
        }
        if (writers != 1 || !g_sparse_grads) continue;
        const Tensor& x = e->t[o.in];
        off.name("row-list copy (sparse_gx) of op", (int)(&o - e->ops.data()));
        o.sparse_gx = 1; o.cl_off = off; off += align256((size_t)x.B * x.D * x.H * x.W * sizeof(int));
    }
    for (Op& o : e->ops) {
        if (o.kind != OP_CONV_ROWS || o.ds_rows < 0 || e->t[o.in2].goff == SIZE_MAX || !g_sparse_grads) continue;
        int writers = 0;
        for (const Op& q : e->ops) if ((q.in == o.in2 || q.in2 == o.in2) && e->needs_grad[q.out]) ++writers;
        if (writers != 1) continue;
        const Tensor& ta = e->t[o.in2];
        off.name("row-list copy (ds_sparse) of op", (int)(&o - e->ops.data()));
        o.ds_sparse = 1; o.cl2_off = off; off += align256((size_t)ta.B * ta.D * ta.H * ta.W * sizeof(int));
    }
    // a dense-layout convolution computed on an output row list (the stem over a sparse volume: rows_out >= 0): its output is kept zero
    // outside the list the same way — the rows of the step before are cleared, a copy of the list lives in the arena
    for (Op& o : e->ops) {
        if (o.kind != OP_CONV || o.rows_out < 0) continue;
        const Tensor& y = e->t[o.out];
        off.name("row-list copy (stem output) of op", (int)(&o - e->ops.data()));
        o.cl_off = off; off += align256((size_t)y.B * y.D * y.H * y.W * sizeof(int));
    }
    off.name("BatchNorm workspace", 0);
    e->off_bn_ws = off; off += align256(bn_ws);
    off.name("BatchNorm coefficients", 0);
    e->off_coef = off; off += align256(coef);
    off.name("split-K workspace", 0);
    e->off_ks = off; off += align256(e->sz_ks);
    for (Op& o : e->ops) if (o.rd != -1) { off.name("weight-gradient partials of op", (int)(&o - e->ops.data())); o.wg_off = off; off += align256(o.wg_bytes); if (o.rd >= 0) e->reduce[o.rd].part = (const float*)o.wg_off; }
    // weight gradients of the small-volume ResNet levels (layer2-4: 42 bias-free convolutions at 16^3 / 8^3 / 4^3, 12-50 us launches of 50-400
    // workgroups each): their split partials are written by ONE launch per tile shape once the pass has produced all their output gradients
    if (g_group_wgrad)
        for (size_t i = 0; i < e->ops.size(); ++i) {
            Op& o = e->ops[i];
            const Tensor& y = e->t[o.out];
            if (o.kind != OP_CONV || o.rd < 0 || o.rows_out >= 0 || o.in == 0 || o.b >= 0 || o.out == e->out_slot || !e->prm[o.w].grad) continue;
            if ((long)y.D * y.H * y.W > 4096) continue;
            o.grp = 1; e->grp_ops.push_back((int)i);
        }
    if (e->grp_ops.size() < 4) { for (int i : e->grp_ops) e->ops[i].grp = 0; e->grp_ops.clear(); }
    off.name("descriptor tables", 0);
    e->off_grp = off; off += align256(e->grp_ops.size() * (size_t)dreg_wgrad_group_desc_bytes() + 16);
    e->off_rd = off; off += align256(e->reduce.size() * sizeof(ReduceRec) + 16);
    e->off_bnf = off; off += align256(e->bn_fwd.size() * sizeof(BnTailRec) + 16);
    e->off_bnb = off; off += align256(e->bn_bwd.size() * sizeof(BnTailRec) + 16);
    off.name("column-sum workspace", 0);
    e->off_cs = off; off += align256(cs);
    off.name("gradient temporary", 0);
    e->off_tmp = off; off += max_tensor;
    if (e->opt.guard) {       // the band table + the scan's result words live behind the last band (no band of their own)
        off.bands = nullptr;
        e->off_guard_tab = off; off += align256((e->bands.size() + 1) * sizeof(unsigned long long));
        e->off_guard_res = off; off += 256;
    }
    e->arena_bytes = off + 256;

    // weight packs: forward for every convolution; data-gradient (gather or stride-2 class form) where the input carries a gradient
    size_t poff = 0;
    auto add_pack = [&](Param& p, int kind, int cin_pad) {
        PackRec r{};
        const int ntaps = p.ksz * p.ksz * p.ksz;
        r.w = p.val; r.Cout = p.d0; r.Cin_real = p.d1; r.ntaps = ntaps; r.for_dgrad = kind; r.dtype = 0; r.row0 = e->pack_rows;
        int rows;
        if (kind == 0) { r.inner = cin_pad; r.Kpad = dreg_conv3d_kpad(p.ksz, cin_pad, 0); rows = p.d0; }
        else if (kind == 1) { r.inner = p.d0; r.Kpad = dreg_conv3d_kpad(p.ksz, p.d0, 0); rows = p.d1; }
        else { r.inner = p.d0; r.Kpad = dreg_conv3d_kpad(p.ksz == 1 ? 1 : 2, p.d0, 0); rows = (p.ksz == 1 ? 1 : 8) * p.d1; }
        r.out = (void*)poff;
        const size_t o = poff;
        poff += align256((size_t)rows * r.Kpad * 2);
        e->pack_rows += rows;
        const int ch = kind ? p.d0 : p.d1;
        const int fl = ntaps > 1 ? (ch < 64 ? ch : 64) * ntaps : 0;   // LDS staging of the k^3 route of the batched pack kernel
        if (fl > e->pack_max_floats) e->pack_max_floats = fl;
        e->packs.push_back(r);
        return o;
    };
    auto add_halo_pack = [&](Param& p, int transposed) {
        HaloPack hp{p.val, poff, p.d0, p.d1, transposed, 0};
        poff += align256(dreg_conv3_halo_pack_bytes_n(transposed ? p.d0 : p.d1, transposed ? p.d1 : p.d0));
        e->halo_packs.push_back(hp);
        return hp.off;
    };
    auto add_brick_pack = [&](Param& p, int transposed) {
        HaloPack hp{p.val, poff, p.d0, p.d1, transposed, 1};
        poff += align256(dreg_conv3_brick_pack_bytes(transposed ? p.d1 : p.d0, transposed ? p.d0 : p.d1));
        e->halo_packs.push_back(hp);
        return hp.off;
    };
    for (Op& o : e->ops) {
        if (o.kind != OP_CONV && o.kind != OP_CONV_ROWS) continue;
        Param& p = e->prm[o.w];
        const Tensor& x = e->t[o.in];
        const Tensor& y = e->t[o.out];
        if (o.kind == OP_CONV && !o.relu) {
            // dense 3^3 convolutions with 256 (or 64) output channels on large volumes: forward x -> y, data gradient gy -> gx (needs 256 / 64 INPUT channels)
            if (dreg_conv3_halo_use(x.B, x.D, x.H, x.W, x.C, p.d0, o.ksz, o.stride, o.pad) && p.d1 == x.C) o.halo |= 1;
            if (e->needs_grad[o.in] && dreg_conv3_halo_use(y.B, y.D, y.H, y.W, p.d0, p.d1, o.ksz, o.stride, o.pad) && p.d0 % 32 == 0) o.halo |= 2;
        }
        if (o.kind == OP_CONV_ROWS && o.ksz == 3 && o.pad == 1 && p.d1 == x.C && dreg_brick_supported(x.B, x.D, x.H, x.W, 16, 256)) {
            // the same layer in the staged-neighbourhood form (used when the step's row sets come with tile tables)
            // (packs only for the launches the mask in force at creation enables: dreg_exec_opts.brick of dreg_exec_create_opts)
            auto on = [&](int nout) { return (nout == 64 && (g_brick & 1)) || (nout == 256 && (g_brick & 2)); };
            if (on(p.d0) && p.d1 % 16 == 0 && p.pk_brick_fwd == SIZE_MAX) p.pk_brick_fwd = add_brick_pack(p, 0);
            if (e->needs_grad[o.in] && on(p.d1) && p.d0 % 16 == 0 && p.pk_brick_dgrad == SIZE_MAX) p.pk_brick_dgrad = add_brick_pack(p, 1);
        }
        if (o.halo & 1) { if (p.pk_halo_fwd == SIZE_MAX) p.pk_halo_fwd = add_halo_pack(p, 0); }
        else if (p.pk_fwd == SIZE_MAX) { p.cin_pad = x.C; p.pk_fwd = add_pack(p, 0, x.C); }
        if (e->needs_grad[o.in]) {
            if (o.halo & 2) { if (p.pk_halo_dgrad == SIZE_MAX) p.pk_halo_dgrad = add_halo_pack(p, 1); }
            else if (o.kind == OP_CONV && s2_class_ok(p, o.ksz, o.stride, o.pad)) { if (p.pk_cls == SIZE_MAX) p.pk_cls = add_pack(p, 2, x.C); }
            else if (p.pk_dgrad == SIZE_MAX) p.pk_dgrad = add_pack(p, 1, x.C);
        }
    }
    e->pack_bytes = poff + 256;
    return e;
}

void dreg_exec_destroy(void* h)
{
    Exec* e = (Exec*)h;
    if (!e) return;
    for (auto& t : e->timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    for (auto& v : e->ev) if (v) (void)hipEventDestroy(v);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    delete e;
}

size_t dreg_exec_arena_bytes(void* h) { return ((Exec*)h)->arena_bytes; }
size_t dreg_exec_pack_bytes(void* h) { return ((Exec*)h)->pack_bytes; }
int dreg_exec_num_packs(void* h) { return (int)((Exec*)h)->packs.size(); }
// byte offset of tensor `slot` in the arena (activation) — slot 0 is external
size_t dreg_exec_tensor_offset(void* h, int slot) { return ((Exec*)h)->t[slot].off; }
int dreg_exec_output_slot(void* h) { return ((Exec*)h)->out_slot; }

// Write the 48-byte pack descriptors (dreg_pack_conv_weights_batched) for a pack buffer at device address pack_base into host memory.
int dreg_exec_pack_rows(void* h) { return ((Exec*)h)->pack_rows; }
// host_row_desc: int32 [dreg_exec_pack_rows] receives the record index of every packed row
int dreg_exec_export_pack_table(void* h, void* host_out, int* host_row_desc, void* pack_base)
{
    Exec* e = (Exec*)h;
    e->pack_base = (char*)pack_base;
    PackRec* o = (PackRec*)host_out;
    for (size_t i = 0; i < e->packs.size(); ++i) {
        o[i] = e->packs[i]; o[i].out = (char*)pack_base + (size_t)e->packs[i].out;
        const int end = i + 1 < e->packs.size() ? e->packs[i + 1].row0 : e->pack_rows;
        for (int r = e->packs[i].row0; r < end; ++r) host_row_desc[r] = (int)i;
    }
    return DREG_OK;
}
// descs_dev / row_desc_dev: the exported tables copied to the device by the caller
int dreg_exec_repack(void* h, const void* descs_dev, const int* row_desc_dev, void* stream)
{
    Exec* e = (Exec*)h;
    CK(dreg_pack_conv_weights_batched(descs_dev, (int)e->packs.size(), e->pack_rows, e->pack_max_floats, row_desc_dev, stream));
    for (const HaloPack& hp : e->halo_packs) {
        if (hp.brick) CK(dreg_pack_conv_weight_brick(hp.w, e->pack_base + hp.off, hp.Cout, hp.Cin, hp.transposed, stream));
        else CK(dreg_pack_conv_weight_halo(hp.w, e->pack_base + hp.off, hp.Cout, hp.Cin, hp.transposed, stream));
    }
    return DREG_OK;
}
// bit 0 / bit 1: the forward / data gradient of op `op` runs on the halo kernel (labels of the HIP-event timing records)
int dreg_exec_op_halo(void* h, int op) { Exec* e = (Exec*)h; return op >= 0 && op < (int)e->ops.size() ? e->ops[op].halo : 0; }

// 1 (default): weight / bias gradients on the executor's own second stream, overlapping the data-gradient chain; 0: one stream
void dreg_exec_set_overlap(void* h, int enable) { ((Exec*)h)->use_aux = enable != 0; }
// Output-row occupancy flags (dreg_conv_row_occupancy) of the convolution that reads the network input x_in — the stem: byte
// [B, Do, Ho], 0 = the row's receptive field in x_in is all zero.  Used by the next forward / backward calls; null = none.
void dreg_exec_set_input_row_occupancy(void* h, const uint8_t* rowocc) { ((Exec*)h)->in_rowocc = rowocc; }
void dreg_exec_set_timing(void* h, int enable) { ((Exec*)h)->timing = enable != 0; }   // records are kept until read
// After a stream synchronisation: elapsed ms of the bracketed launches since the last set_timing; records are (op, kind, variant, ms)
// with kind 0 forward, 1 data gradient, 2 weight gradient (+reduce); variant 2 = the launch ran on csrc/conv_brick.hip; op_kind holds 3 ints per record.  Returns the number written (<= max) and restarts.
int dreg_exec_read_timings(void* h, int* op_kind, float* ms, int max)
{
    Exec* e = (Exec*)h;
    int n = 0;
    for (size_t i = 0; i < e->timed_used && n < max; ++i) {
        float v = 0.f;
        if (hipEventElapsedTime(&v, e->timed[i].e0, e->timed[i].e1) != hipSuccess) continue;
        op_kind[3 * n] = e->timed[i].op; op_kind[3 * n + 1] = e->timed[i].kind; op_kind[3 * n + 2] = e->timed[i].variant; ms[n] = v; ++n;
    }
    e->timed_used = 0;
    return n;
}

// The descriptor tables of the batched launches live in the arena (they hold addresses of this arena's buffers): uploaded once per arena.
static int guard_setup(Exec* e, char* A, hipStream_t st)
{
    if (!e->opt.guard || e->guard_arena == (const void*)A) return DREG_OK;
    std::vector<unsigned long long> offs(e->bands.size());
    for (size_t i = 0; i < offs.size(); ++i) offs[i] = e->bands[i].off;
    if (!offs.empty() && hipMemcpyAsync(A + e->off_guard_tab, offs.data(), offs.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st) != hipSuccess) return DREG_ELAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return DREG_ELAUNCH;
    int rc = dreg_guard_fill(A, A + e->off_guard_tab, (int)offs.size(), (int)GUARD_BYTES, (void*)st);
    if (rc) return rc;
    e->guard_arena = A;
    return DREG_OK;
}
static int guard_scan(Exec* e, char* A, long long* out4, hipStream_t st)
{
    int rc = dreg_guard_scan(A, A + e->off_guard_tab, (int)e->bands.size(), (int)GUARD_BYTES, A + e->off_guard_res, (void*)st);
    if (rc) return rc;
    if (hipMemcpyAsync(out4, A + e->off_guard_res, 4 * sizeof(long long), hipMemcpyDeviceToHost, st) != hipSuccess) return DREG_ELAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return DREG_ELAUNCH;
    return DREG_OK;
}
// guard = 2: scan behind op `op` of pass `pass`; DREG_EGUARD (and the place remembered) when a band has changed
static int guard_after_op(Exec* e, char* A, int op, int pass, hipStream_t st, hipStream_t aux)
{
    if (e->opt.guard < 2) return DREG_OK;
    if (aux && hipStreamSynchronize(aux) != hipSuccess) return DREG_ELAUNCH;
    long long r[4];
    int rc = guard_scan(e, A, r, st);
    if (rc) return rc;
    if (r[0] > 0) { e->guard_op = op; e->guard_pass = pass; e->guard_band = r[1]; return DREG_EGUARD; }
    return DREG_OK;
}
static int upload_tables(Exec* e, char* A, hipStream_t st)
{
    { int rc = guard_setup(e, A, st); if (rc) return rc; }
    if (e->reduce_arena == (const void*)A) return DREG_OK;
    e->reduce_abs = e->reduce;
    for (ReduceRec& r : e->reduce_abs) r.part = (const float*)(A + (size_t)r.part);
    std::vector<BnTailRec> f = e->bn_fwd, b = e->bn_bwd;
    for (BnTailRec& r : f) { r.a = (const float*)(A + (size_t)r.a); r.b = (const float*)(A + (size_t)r.b); }
    for (BnTailRec& r : b) r.a = (const float*)(A + (size_t)r.a);
    if (!e->reduce_abs.empty() && hipMemcpyAsync(A + e->off_rd, e->reduce_abs.data(), e->reduce_abs.size() * sizeof(ReduceRec), hipMemcpyHostToDevice, st) != hipSuccess) return DREG_ELAUNCH;
    if (!f.empty() && (hipMemcpyAsync(A + e->off_bnf, f.data(), f.size() * sizeof(BnTailRec), hipMemcpyHostToDevice, st) != hipSuccess ||
                       hipMemcpyAsync(A + e->off_bnb, b.data(), b.size() * sizeof(BnTailRec), hipMemcpyHostToDevice, st) != hipSuccess)) return DREG_ELAUNCH;
    // grouped weight-gradient descriptors (absolute addresses of this arena): one table per tile shape, block0 = exclusive prefix inside it
    const int gd = dreg_wgrad_group_desc_bytes();
    struct Rec { std::vector<char> d; int variant, nblocks, op; };
    std::vector<Rec> recs;
    e->grp_launch.clear(); e->grp_ready = false;
    bool all_ok = !e->grp_ops.empty();
    for (int i : e->grp_ops) {
        const Op& o = e->ops[i];
        const Tensor& x = e->t[o.in];
        const Tensor& y = e->t[o.out];
        const Param& w = e->prm[o.w];
        Rec r; r.d.resize(gd); r.op = i;
        if (e->t[o.out].goff == SIZE_MAX ||
            dreg_conv3d_wgrad_group_fill(r.d.data(), A + y.goff, A + x.off, A + o.wg_off, o.wg_bytes, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0,
                                         o.ksz, o.stride, o.pad, &r.variant, &r.nblocks) != DREG_OK || w.d1 != x.C) { all_ok = false; break; }
        recs.push_back(std::move(r));
    }
    std::vector<char> gtab;
    if (all_ok) {
        std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.variant < b.variant; });
        gtab.resize(recs.size() * (size_t)gd);
        for (size_t k = 0; k < recs.size(); ++k) {
            if (e->grp_launch.empty() || e->grp_launch.back().variant != recs[k].variant) e->grp_launch.push_back(Exec::GrpLaunch{e->off_grp + k * (size_t)gd, 0, recs[k].variant, 0});
            const int b0 = e->grp_launch.back().blocks;
            std::memcpy(recs[k].d.data() + gd - (int)sizeof(int), &b0, sizeof(int));
            std::memcpy(gtab.data() + k * (size_t)gd, recs[k].d.data(), (size_t)gd);
            e->grp_launch.back().n += 1; e->grp_launch.back().blocks += recs[k].nblocks;
        }
        if (hipMemcpyAsync(A + e->off_grp, gtab.data(), gtab.size(), hipMemcpyHostToDevice, st) != hipSuccess) return DREG_ELAUNCH;
        e->grp_ready = true;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return DREG_ELAUNCH;   // the host copies above are locals
    e->reduce_arena = A;
    return DREG_OK;
}
// One launch per run of consecutive flagged records of a BatchNorm tail table (what: 0 running statistics, 1 parameter gradients).
static int flush_bn_tails(Exec* e, char* A, std::vector<char>& done, int what, hipStream_t st)
{
    for (int lo = 0, nrec = (int)done.size(); lo < nrec;) {
        if (!done[lo]) { ++lo; continue; }
        int hi = lo;
        while (hi + 1 < nrec && done[hi + 1]) ++hi;
        Scope sc(e, st, -1, 4 + what);
        const int b0 = e->bn_fwd[lo].block0;
        const int b1 = hi + 1 < nrec ? e->bn_fwd[hi + 1].block0 : e->bn_blocks;
        if (what == 0) CK(dreg_bn_running_update_batched((const BnTailRec*)(A + e->off_bnf) + lo, hi - lo + 1, b0, b1 - b0, 0.1f, (void*)st));
        else CK(dreg_bn_param_grad_batched((const BnTailRec*)(A + e->off_bnb) + lo, hi - lo + 1, b0, b1 - b0, 1, (void*)st));
        for (int q = lo; q <= hi; ++q) done[q] = 0;
        lo = hi + 1;
    }
    return DREG_OK;
}

// rowlists: int64 [nlists][RL] = (device int32* rows, count, brick tile tables or zeros: include/dreg_nerf.h).  x: tensor 0.  The result lands at dreg_exec_tensor_offset(output slot).
int dreg_exec_forward(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in,
                      const int64_t* rowlists, int nlists, int train, void* stream)
{
    Exec* e = (Exec*)h;
    if (arena_bytes < e->arena_bytes) return DREG_EINVAL;
    char* A = (char*)arena;
    const char* PK = (const char*)pack_base;
    hipStream_t st = (hipStream_t)stream;
    CK(upload_tables(e, (char*)arena, st));
    std::vector<char> bn_done(e->bn_fwd.size(), 0);
    std::vector<int> sums_rpc(e->ops.size(), 0);    // per BatchNorm op: rows per chunk of the sums its producer left (0 = none)
    int fwd_sk_bn = -1, fwd_sk_n = 0;               // the BatchNorm whose input lies in the split-K workspace as fwd_sk_n slices
    size_t fwd_sk_slice = 0;
    // After a COMPLETE pass the row-list outputs of `arena` are in the known state (zero outside the copied lists).  Any early error return leaves them
    // unknown: the next forward then clears them densely and forgets the copied lists.
    struct ArenaMark { Exec* e; const void* a; bool ok = false;
                       ~ArenaMark() { if (ok) e->fwd_rows_arena = a; else { e->fwd_rows_arena = nullptr; for (Op& o : e->ops) if (o.kind == OP_CONV && o.rows_out >= 0) o.cl_count = 0; } } } arena_mark{e, arena};
    auto act = [&](int s) -> void* { return s == 0 ? (void*)x_in : (void*)(A + e->t[s].off); };
    for (size_t i = 0; i < e->ops.size(); ++i) {
        if (i > 0) CK(guard_after_op(e, A, (int)i - 1, 0, st, nullptr));
        const Op& o = e->ops[i];
        const Tensor& x = e->t[o.in];
        const Tensor& y = e->t[o.out];
        if (o.kind == OP_CONV) {
            const Param& w = e->prm[o.w];
            const float* bias = o.b >= 0 ? e->prm[o.b].val : nullptr;
            const void* add = o.in2 >= 0 ? act(o.in2) : nullptr;
            const Tensor* ta = o.in2 >= 0 ? &e->t[o.in2] : nullptr;
            if (o.rows_out >= 0) {
                // only the listed output voxels can be non-zero (bias-free layer over a sparse volume): compute those, keep the rest zero
                if (o.rows_out >= nlists || bias || add) return DREG_EINVAL;
                const int64_t* rl = rowlists + RL * o.rows_out;
                const int* r = (const int*)rl[0];
                const int n = (int)rl[1];
                Op& om = const_cast<Op&>(o);
                if (e->fwd_rows_arena != arena) CK(dreg_fill_zero(act(o.out), y.bytes, stream));
                else if (om.cl_count > 0) CK(dreg_zero_rows(act(o.out), (const int*)(A + o.cl_off), om.cl_count, y.C, 0, stream));
                if (n > 0 && hipMemcpyAsync(A + o.cl_off, r, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return DREG_ELAUNCH;
                om.cl_count = n;
                if (n > 0) {
                    Scope sc(e, st, (int)i, 0);
                    CK(dreg_conv3d_igemm_rows(act(o.in), PK + w.pk_fwd, act(o.out), nullptr, nullptr, r, n, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0,
                                              o.ksz, o.stride, o.pad, 0, o.relu, 0, 0, 0, 0, 0, stream));
                }
                continue;
            }
            Scope sc(e, st, (int)i, 0);
            if (o.halo & 1) {
                if (train && o.stats_bn >= 0)
                    CK(dreg_conv3_halo_n_bnstats(act(o.in), PK + w.pk_halo_fwd, act(o.out), bias, add, x.B, x.D, x.H, x.W, x.C, w.d0, ta ? ta->D : 0, ta ? ta->H : 0,
                                                 ta ? ta->W : 0, o.add_same, (float*)(A + e->ops[o.stats_bn].stats_off), &sums_rpc[o.stats_bn], stream));
                else
                    CK(dreg_conv3_halo_n(act(o.in), PK + w.pk_halo_fwd, act(o.out), bias, add, x.B, x.D, x.H, x.W, x.C, w.d0, ta ? ta->D : 0, ta ? ta->H : 0, ta ? ta->W : 0,
                                         o.add_same, 0, stream));
                continue;
            }
            if (train && o.stats_bn >= 0 && !(o.in == 0 && e->in_rowocc)) {
                CK(dreg_conv3d_igemm_bnstats(act(o.in), PK + w.pk_fwd, act(o.out), bias, add, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0,
                                             o.ksz, o.stride, o.pad, o.relu, ta ? ta->D : 0, ta ? ta->H : 0, ta ? ta->W : 0, o.add_same,
                                             A + e->off_ks, e->sz_ks, (float*)(A + e->ops[o.stats_bn].stats_off), &sums_rpc[o.stats_bn], stream));
                continue;
            }
            if (train && o.fold_sk_fwd && !(o.halo & 1)) {
                int ns = 0; size_t sl = 0;
                CK(dreg_conv3d_igemm_defer(act(o.in), PK + w.pk_fwd, act(o.out), bias, add, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0,
                                           o.ksz, o.stride, o.pad, 0, o.relu, ta ? ta->D : 0, ta ? ta->H : 0, ta ? ta->W : 0, o.add_same,
                                           A + e->off_ks, e->sz_ks, o.in == 0 ? e->in_rowocc : nullptr, &ns, &sl, stream));
                if (ns > 0) { fwd_sk_bn = (int)i + 1; fwd_sk_n = ns; fwd_sk_slice = sl; }
                continue;
            }
            CK(dreg_conv3d_igemm_occ(act(o.in), PK + w.pk_fwd, act(o.out), bias, add, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0,
                                     o.ksz, o.stride, o.pad, 0, o.relu, ta ? ta->D : 0, ta ? ta->H : 0, ta ? ta->W : 0, o.add_same, 0, 0,
                                     A + e->off_ks, e->sz_ks, o.in == 0 ? e->in_rowocc : nullptr, stream));
        } else if (o.kind == OP_CONV_ROWS) {
            const Param& w = e->prm[o.w];
            if (o.rows_out < 0 || o.rows_out >= nlists) return DREG_EINVAL;
            const float* bias = o.b >= 0 ? e->prm[o.b].val : nullptr;
            const void* add = o.in2 >= 0 ? act(o.in2) : nullptr;
            const Tensor* ta = o.in2 >= 0 ? &e->t[o.in2] : nullptr;
            Scope sc(e, st, (int)i, 0);
            const int64_t* rl = rowlists + RL * o.rows_out;
            if ((e->opt.brick & (w.d0 == 64 ? 1 : 2)) && w.pk_brick_fwd != SIZE_MAX && rl[2] && rl[3] > 0) {
                sc.variant(2);
                CK(dreg_conv3_brick(act(o.in), PK + w.pk_brick_fwd, act(o.out), bias, add, (const void*)rl[2], (int)rl[3], (const int*)rl[4], (const void*)rl[5],
                                    (const int*)rl[6], x.B, x.D, x.H, x.W, x.C, w.d0, ta ? ta->D : 0, ta ? ta->H : 0, ta ? ta->W : 0, 0, 0, stream));
                continue;
            }
            CK(dreg_conv3d_igemm_rows(act(o.in), PK + w.pk_fwd, act(o.out), bias, add, (const int*)rl[0], (int)rl[1],
                                      x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0, o.ksz, 1, o.pad, 0, 0, ta ? ta->D : 0, ta ? ta->H : 0, ta ? ta->W : 0,
                                      0, 0, stream));
        } else if (o.kind == OP_BN && o.pool >= 0 && o.sstem) {
            const Op& q = e->ops[o.pool];
            const Tensor& p = e->t[q.out];
            const Op& cv = e->ops[o.sstem_conv];
            if (cv.rows_out >= nlists) return DREG_EINVAL;
            const int64_t* rl = rowlists + RL * cv.rows_out;
            const int* ra = nullptr; int na = 0;
            if (o.sstem_lat >= 0) {
                const int li = e->ops[o.sstem_lat].rows_in;
                if (li < 0 || li >= nlists) return DREG_EINVAL;
                ra = (const int*)rowlists[RL * li]; na = (int)rowlists[RL * li + 1];
            }
            CK(dreg_sparse_stem_fwd(act(o.in), (const int*)rl[0], (int)rl[1], ra, na, act(o.out), act(q.out), (uint8_t*)(A + q.aux0), A + o.xam_off, (uint8_t*)(A + o.pmask_off),
                                    e->prm[o.w].val, e->prm[o.b].val, e->prm[o.p2].val, e->prm[o.p3].val, (float*)(A + o.aux0), (float*)(A + o.aux1),
                                    (float*)(A + e->off_bn_ws), x.B, x.D, x.H, x.W, p.D, p.H, p.W, x.C, 1e-5f, 0.1f, train, o.relu, stream));
        } else if (o.kind == OP_BN && o.pool >= 0) {
            const Op& q = e->ops[o.pool];
            const Tensor& p = e->t[q.out];
            CK(dreg_bn_relu_maxpool_fwd(act(o.in), act(q.out), (uint8_t*)(A + q.aux0), e->prm[o.w].val, e->prm[o.b].val, e->prm[o.p2].val, e->prm[o.p3].val,
                                        (float*)(A + o.aux0), (float*)(A + o.aux1), (float*)(A + e->off_bn_ws), x.B, x.D, x.H, x.W, p.D, p.H, p.W, x.C,
                                        1e-5f, 0.1f, train, o.relu, stream));
        } else if (o.kind == OP_MAXPOOL && o.pool >= 0) {
            // done by the BatchNorm in front of it
        } else if (o.kind == OP_BN) {
            const int V = x.D * x.H * x.W;
            const void* res_p = o.in2 >= 0 ? act(o.in2) : nullptr;
            void* out_p = o.fold_into >= 0 ? nullptr : act(o.out);       // folded branch: statistics, scale / shift only
            dreg_bn_extra ex{};
            if (o.res_from >= 0) { res_p = act(e->ops[o.res_from].in); ex.res_scale_shift = (const float*)(A + e->ops[o.res_from].aux0); }
            if (train && sums_rpc[i] > 0) {
                CK(dreg_bn3d_fwd_ex(act(o.in), res_p, out_p, e->prm[o.w].val, e->prm[o.b].val, e->prm[o.p2].val, e->prm[o.p3].val,
                                    (float*)(A + o.aux0), (float*)(A + o.aux1), (float*)(A + o.stats_off), x.B, V, x.C, 1e-5f, 0.1f, 1, o.relu, 0,
                                    nullptr, nullptr, sums_rpc[i], &ex, stream));
                continue;
            }
            int deferred = 0;
            if (fwd_sk_bn == (int)i) { ex.splitk_part = (const float*)(A + e->off_ks); ex.splitk_nsplit = fwd_sk_n; ex.splitk_slice = fwd_sk_slice; fwd_sk_bn = -1; }
            CK(dreg_bn3d_fwd_ex(act(o.in), res_p, out_p, e->prm[o.w].val, e->prm[o.b].val, e->prm[o.p2].val, e->prm[o.p3].val,
                                (float*)(A + o.aux0), (float*)(A + o.aux1), (float*)(A + e->off_bn_ws), x.B, V, x.C, 1e-5f, 0.1f, train, o.relu, 0,
                                o.bt >= 0 ? (float*)(A + o.keep_var) : nullptr, &deferred, 0, &ex, stream));
            if (deferred) bn_done[o.bt] = 1;
        } else if (o.kind == OP_MAXPOOL) {
            CK(dreg_maxpool3d_fwd(act(o.in), act(o.out), (uint8_t*)(A + o.aux0), x.B, x.D, x.H, x.W, y.D, y.H, y.W, x.C, 0, stream));
        } else return DREG_EINVAL;
    }
    CK(guard_after_op(e, A, (int)e->ops.size() - 1, 0, st, nullptr));
    CK(flush_bn_tails(e, A, bn_done, 0, st));
    CK(guard_after_op(e, A, (int)e->ops.size(), 0, st, nullptr));
    arena_mark.ok = true;
    return DREG_OK;
}
int dreg_exec_guard_bands(void* h) { return (int)((Exec*)h)->bands.size(); }
int dreg_exec_guard_check(void* h, void* arena, long long* out4, void* stream)
{
    Exec* e = (Exec*)h;
    if (!e->opt.guard || !out4) return DREG_EINVAL;
    if (e->guard_arena != arena) { out4[0] = 0; out4[1] = (long long)e->bands.size(); out4[2] = 0; out4[3] = 0; return DREG_OK; }   // never filled: nothing ran on it
    return guard_scan(e, (char*)arena, out4, (hipStream_t)stream);
}
int dreg_exec_guard_describe(void* h, int band, char* buf, int buf_bytes)
{
    Exec* e = (Exec*)h;
    if (band < 0 || band >= (int)e->bands.size() || !buf || buf_bytes <= 0) return DREG_EINVAL;
    const Exec::Band& b = e->bands[band];
    std::snprintf(buf, (size_t)buf_bytes, "band %d at arena offset %zu, behind: %s %d", band, b.off, b.tag, b.idx);
    return DREG_OK;
}
int dreg_exec_guard_last(void* h, int* op, int* pass, long long* band)
{
    Exec* e = (Exec*)h;
    if (op) *op = e->guard_op;
    if (pass) *pass = e->guard_pass;
    if (band) *band = e->guard_band;
    return DREG_OK;
}

// grad_out: gradient of the result tensor (bf16, same shape).  Weight / bias / BatchNorm gradients are ACCUMULATED into the
// parameters' grad pointers; nothing is returned for the input tensor 0.  aux_stream (optional): a second stream of the caller's
// for the weight / bias gradient launches; `stream` waits for it before this call's work is considered complete.
int dreg_exec_backward_range(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in, const void* grad_out,
                             const int64_t* rowlists, int nlists, void* stream, void* aux_stream, int op_begin, int op_end, int flags);
int dreg_exec_backward(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in, const void* grad_out,
                       const int64_t* rowlists, int nlists, void* stream, void* aux_stream)
{
    return dreg_exec_backward_range(h, arena, arena_bytes, pack_base, x_in, grad_out, rowlists, nlists, stream, aux_stream,
                                    0, (int)((Exec*)h)->ops.size(), 3);
}
// The same in segments: ops [op_begin, op_end) are processed in reverse order; flags bit 0 = first segment of a pass (the one that
// contains the last op), bit 1 = last segment (the caller's stream then joins the parameter-gradient stream).  Between two segments
// the caller may record events on both streams: every parameter gradient of the ops processed so far has been enqueued — the
// data-parallel step launches the all-reduce of finished gradient buckets there (dreg_nerf_amd/optim.py GradSync).
int dreg_exec_backward_range(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in, const void* grad_out,
                             const int64_t* rowlists, int nlists, void* stream, void* aux_stream, int op_begin, int op_end, int flags)
{
    Exec* e = (Exec*)h;
    if (arena_bytes < e->arena_bytes || op_begin < 0 || op_end > (int)e->ops.size() || op_begin > op_end) return DREG_EINVAL;
    char* A = (char*)arena;
    const char* PK = (const char*)pack_base;
    hipStream_t st = (hipStream_t)stream;
    auto act = [&](int s) -> void* { return s == 0 ? (void*)x_in : (void*)(A + e->t[s].off); };
    if (flags & 1) { e->written.assign(e->t.size(), 0); e->aux_used = false; e->pend_sk_bn = -1; }
    if (e->written.size() != e->t.size()) return DREG_EINVAL;
    std::vector<char>& written = e->written;
    auto grad = [&](int s) -> void* { return s == e->out_slot ? (void*)grad_out : (void*)(A + e->t[s].goff); };
    // destination for a new contribution to tensor s: its gradient buffer the first time, the temporary afterwards (then add)
    auto dst_for = [&](int s) -> void* { return written[s] ? (void*)(A + e->off_tmp) : grad(s); };
    auto commit = [&](int s) -> int {
        if (written[s]) {
            const Tensor& t = e->t[s];
            return dreg_add_inplace(grad(s), A + e->off_tmp, (size_t)t.B * t.D * t.H * t.W * t.C, 0, stream);
        }
        written[s] = 1;
        return DREG_OK;
    };
    written[e->out_slot] = 1;
    // the second stream is the caller's (one per process/device: HIP maps streams onto a handful of hardware queues, and a
    // stream created per executor can land on the queue of the main stream, which serialises instead of overlapping)
    bool aux_on = e->use_aux && aux_stream != nullptr && aux_stream != stream;
    e->aux = (hipStream_t)aux_stream;
    if (aux_on && e->ev.empty()) {
        e->ev.assign(e->ops.size(), nullptr);
        if (hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming) != hipSuccess) aux_on = false;
    }
    bool& aux_used = e->aux_used;
    CK(upload_tables(e, A, st));
    std::vector<char> bn_done(e->bn_bwd.size(), 0);
    if (flags & 1) {
        // sparse gradient buffers back to all-zero: a dense memset the first time this arena is seen, the rows of the last step after that
        const bool fresh = e->sparse_arena != arena;
        for (Op& o : e->ops) {
            if (!o.sparse_gx) continue;
            const Tensor& x = e->t[o.in];
            if (fresh) { if (hipMemsetAsync(A + x.goff, 0, (size_t)x.B * x.D * x.H * x.W * x.C * 2, st) != hipSuccess) return DREG_ELAUNCH; }
            else if (o.cl_count > 0) CK(dreg_zero_rows(A + x.goff, (const int*)(A + o.cl_off), o.cl_count, x.C, 0, stream));
            o.cl_count = 0;
        }
        for (Op& o : e->ops) {
            if (!o.ds_sparse) continue;
            const Tensor& ta = e->t[o.in2];
            if (fresh) { if (hipMemsetAsync(A + ta.goff, 0, (size_t)ta.B * ta.D * ta.H * ta.W * ta.C * 2, st) != hipSuccess) return DREG_ELAUNCH; }
            else if (o.cl2_count > 0) CK(dreg_zero_rows(A + ta.goff, (const int*)(A + o.cl2_off), o.cl2_count, ta.C, 0, stream));
            o.cl2_count = 0;
        }
        e->sparse_arena = arena;
    }
    const int g_s2_accumulate = e->opt.s2_accumulate, g_group_wgrad = e->opt.group_wgrad, g_brick = e->opt.brick, g_defer_head_pg = e->opt.defer_head_pg;
    std::vector<char> rd_done(e->reduce.size(), 0);   // records whose partials this call produced and nobody summed yet
    hipStream_t rd_stream = st;
    size_t rd_pending = 0;
    // One launch sums the splits of every pending layer into the torch-layout gradients (on the stream the partials were produced
    // on: all of them the second stream, or all of them the caller's); layers no gradient reached are left out, so a flush is one
    // launch per run of consecutive records.  Flushed every ~192 MB of partials: the deep layers' sums then run next to the rest
    // of the backward pass, and only the last few layers' are left for the end.
    auto flush_reduce = [&]() -> int {
        for (int lo = 0, nrec = (int)rd_done.size(); lo < nrec;) {
            if (!rd_done[lo]) { ++lo; continue; }
            int hi = lo;
            while (hi + 1 < nrec && rd_done[hi + 1]) ++hi;
            Scope sc(e, rd_stream, -1, 3);
            const int b0 = e->reduce[lo].block0;
            const int b1 = hi + 1 < nrec ? e->reduce[hi + 1].block0 : e->reduce_blocks;
            CK(dreg_wgrad_reduce_batched((const ReduceRec*)(A + e->off_rd) + lo, hi - lo + 1, b0, b1 - b0, (void*)rd_stream));
            for (int q = lo; q <= hi; ++q) rd_done[q] = 0;
            lo = hi + 1;
        }
        return DREG_OK;
    };
    // weight / bias gradient launches of convolution op i (second stream behind the event "its output gradient is complete")
    // grouped weight gradients: only in a whole-pass call with the second stream and no per-launch timing brackets
    const bool grp_active = g_group_wgrad && e->grp_ready && aux_on && !e->timing && op_begin == 0 && op_end == (int)e->ops.size() && (flags & 3) == 3;
    std::vector<int> grp_seen;
    auto param_grads = [&](int i, bool event_recorded) -> int {
        const Op& o = e->ops[i];
        const Tensor& x = e->t[o.in];
        const Tensor& y = e->t[o.out];
        const Param& w = e->prm[o.w];
        const void* gy = grad(o.out);
        const bool rows = o.kind == OP_CONV_ROWS;
        const bool lrows = rows || (o.kind == OP_CONV && o.rows_out >= 0 && o.rows_out < nlists);    // reduced over an output row list
        const int* r_out = lrows ? (const int*)rowlists[RL * o.rows_out] : nullptr;
        const int n_out = lrows ? (int)rowlists[RL * o.rows_out + 1] : 0;
        hipStream_t ws = st;
        if (aux_on && !(grp_active && o.grp && !(o.b >= 0 && e->prm[o.b].grad))) {
            if (!e->ev[i] && hipEventCreateWithFlags(&e->ev[i], hipEventDisableTiming) != hipSuccess) return DREG_ELAUNCH;
            if (!event_recorded && hipEventRecord(e->ev[i], st) != hipSuccess) return DREG_ELAUNCH;
            if (hipStreamWaitEvent(e->aux, e->ev[i], 0) != hipSuccess) return DREG_ELAUNCH;
            ws = e->aux;
            aux_used = true;
        }
        hipStream_t bs = ws;                  // bias sums share one scratch buffer: same stream as the weight gradients
        bool flush_now = false;
        if (w.grad && grp_active && o.grp) {
            // written by the grouped launch at the pass's launch point (below); nothing else to do for this layer (no bias)
            grp_seen.push_back(i);
        } else
        if (w.grad) {
            Scope sc(e, ws, i, 2);
            if (o.rd >= 0) {
                CK(dreg_conv3d_wgrad_partials(gy, act(o.in), A + o.wg_off, o.wg_bytes, lrows ? r_out : nullptr, n_out, x.B, x.D, x.H, x.W, x.C, w.d1,
                                              y.D, y.H, y.W, w.d0, o.ksz, rows ? 1 : o.stride, o.pad, (!lrows && o.in == 0) ? e->in_rowocc : nullptr, (void*)ws));
                rd_done[o.rd] = 1;
                rd_stream = ws;
                rd_pending += o.wg_bytes;
                flush_now = rd_pending >= ((size_t)192 << 20);
            } else if (lrows) {
                CK(dreg_conv3d_wgrad_rows(gy, act(o.in), w.grad, A + o.wg_off, o.wg_bytes, r_out, n_out, x.B, x.D, x.H, x.W, x.C, w.d1,
                                          y.D, y.H, y.W, w.d0, o.ksz, rows ? 1 : o.stride, o.pad, 1, (void*)ws));
            } else {
                CK(dreg_conv3d_wgrad_occ(gy, act(o.in), w.grad, A + o.wg_off, o.wg_bytes, x.B, x.D, x.H, x.W, x.C, w.d1, y.D, y.H, y.W, w.d0,
                                         o.ksz, o.stride, o.pad, 1, 0, 1, o.in == 0 ? e->in_rowocc : nullptr, (void*)ws));
            }
        }
        if (flush_now) { CK(flush_reduce()); rd_pending = 0; }   // outside the launch's timing bracket
        if (o.b >= 0 && e->prm[o.b].grad) {
            if (rows) CK(dreg_colsum_rows(gy, r_out, n_out, e->prm[o.b].grad, (float*)(A + e->off_cs), w.d0, 1, 0, (void*)bs));
            else CK(dreg_colsum(gy, e->prm[o.b].grad, (float*)(A + e->off_cs), (size_t)y.B * y.D * y.H * y.W, w.d0, 1, 0, (void*)bs));
        }
        return DREG_OK;
    };
    // the held-back weight gradients: one launch per tile shape when every eligible layer was met, else (a layer no gradient reached) one by one
    auto flush_group = [&](int ev_op) -> int {
        if (grp_seen.empty()) return DREG_OK;
        if (!e->ev[ev_op] && hipEventCreateWithFlags(&e->ev[ev_op], hipEventDisableTiming) != hipSuccess) return DREG_ELAUNCH;
        if (hipEventRecord(e->ev[ev_op], st) != hipSuccess || hipStreamWaitEvent(e->aux, e->ev[ev_op], 0) != hipSuccess) return DREG_ELAUNCH;
        aux_used = true;
        if (grp_seen.size() == e->grp_ops.size()) {
            for (const Exec::GrpLaunch& q : e->grp_launch) CK(dreg_wgrad_group_launch(A + q.off, q.n, q.variant, q.blocks, (void*)e->aux));
        } else {
            for (int j : grp_seen) {
                const Op& oj = e->ops[j]; const Tensor& xj = e->t[oj.in]; const Tensor& yj = e->t[oj.out]; const Param& wj = e->prm[oj.w];
                CK(dreg_conv3d_wgrad_partials(grad(oj.out), act(oj.in), A + oj.wg_off, oj.wg_bytes, nullptr, 0, xj.B, xj.D, xj.H, xj.W, xj.C, wj.d1,
                                              yj.D, yj.H, yj.W, wj.d0, oj.ksz, oj.stride, oj.pad, nullptr, (void*)e->aux));
            }
        }
        for (int j : grp_seen) { rd_done[e->ops[j].rd] = 1; rd_pending += e->ops[j].wg_bytes; }
        rd_stream = e->aux;
        grp_seen.clear();
        return DREG_OK;
    };
    std::vector<int> deferred_pg;            // ops whose parameter-gradient launches are held back (their events are recorded)
    bool deep_reached = false;
    auto run_deferred = [&]() -> int {
        for (int j : deferred_pg) CK(param_grads(j, true));
        deferred_pg.clear();
        return DREG_OK;
    };
    for (int i = op_end - 1; i >= op_begin; --i) {
        if (i < op_end - 1) CK(guard_after_op(e, A, i + 1, 1, st, aux_on ? e->aux : nullptr));
        const Op& o = e->ops[i];
        const Tensor& x = e->t[o.in];
        const Tensor& y = e->t[o.out];
        if (o.kind == OP_MAXPOOL && o.pool >= 0) continue;          // un-pooled inside the fused BatchNorm backward below
        if (o.kind == OP_BN && o.pool >= 0) {
            const Op& q = e->ops[o.pool];
            const Tensor& p = e->t[q.out];
            if (!e->needs_grad[q.out] || !written[q.out]) continue;
            if (!e->prm[o.w].grad || !e->prm[o.b].grad) return DREG_EINVAL;
            if (o.sstem) {
                const Op& cv = e->ops[o.sstem_conv];
                if (cv.rows_out >= nlists) return DREG_EINVAL;
                const int64_t* rl = rowlists + RL * cv.rows_out;
                const int* ra = nullptr; int na = 0;
                const bool lat = o.sstem_lat >= 0 && written[o.out];       // the lateral's data gradient reached the activation
                if (lat) { const int li = e->ops[o.sstem_lat].rows_in; ra = (const int*)rowlists[RL * li]; na = (int)rowlists[RL * li + 1]; }
                CK(dreg_sparse_stem_bwd(act(o.in), grad(q.out), (const uint8_t*)(A + q.aux0), A + o.xam_off, lat ? grad(o.out) : nullptr, ra, na,
                                        (const int*)rl[0], (int)rl[1], (float*)(A + o.aux0), (float*)(A + o.aux1), dst_for(o.in), e->prm[o.w].grad, e->prm[o.b].grad,
                                        (float*)(A + e->off_coef), (float*)(A + e->off_bn_ws), x.B, x.D, x.H, x.W, p.D, p.H, p.W, x.C, o.relu, 1, stream));
                CK(commit(o.in));
                continue;
            }
            CK(dreg_bn_relu_maxpool_bwd(act(o.in), grad(q.out), (const uint8_t*)(A + q.aux0), (float*)(A + o.aux0), (float*)(A + o.aux1), dst_for(o.in),
                                        e->prm[o.w].grad, e->prm[o.b].grad, (float*)(A + e->off_coef), (float*)(A + e->off_bn_ws),
                                        x.B, x.D, x.H, x.W, p.D, p.H, p.W, x.C, o.relu, 1, stream));
            CK(commit(o.in));
            continue;
        }
        if (!e->needs_grad[o.out] || !written[o.out]) continue;   // nothing flows back through this op
        const void* gy = grad(o.out);
        if (o.kind == OP_CONV || o.kind == OP_CONV_ROWS) {
            const Param& w = e->prm[o.w];
            const bool rows = o.kind == OP_CONV_ROWS;
            const int* r_out = rows ? (const int*)rowlists[RL * o.rows_out] : nullptr;
            const int n_out = rows ? (int)rowlists[RL * o.rows_out + 1] : 0;
            // parameter gradients first, on the second stream: gy is complete here (every consumer of this op's output has been
            // processed), and nothing below modifies gy or the op's input activation
            const bool pg = w.grad || (o.b >= 0 && e->prm[o.b].grad);
            // The FPN head's weight gradients (64^3 / 32^3 volumes: throughput-bound launches) are complete first, exactly while the main
            // stream runs the head's throughput-bound data gradients: side by side the two only slow each other down (~25 % per kernel).
            // They are held back until the pass reaches the 8^3 / 4^3 levels, whose launch-latency-bound chain leaves most CUs idle.
            const long vox_out = (long)y.D * y.H * y.W;
            if (pg && aux_on && g_defer_head_pg && !deep_reached && vox_out >= 32768) {
                if (!e->ev[i] && hipEventCreateWithFlags(&e->ev[i], hipEventDisableTiming) != hipSuccess) return DREG_ELAUNCH;
                if (hipEventRecord(e->ev[i], st) != hipSuccess) return DREG_ELAUNCH;
                deferred_pg.push_back(i);
            } else if (pg) {
                if (vox_out <= 512 && !deep_reached) { deep_reached = true; CK(run_deferred()); }
                CK(param_grads(i, false));
                // the first eligible op of the program = the last one the backward pass meets: every grouped layer's output gradient is complete
                if (grp_active && !e->grp_ops.empty() && i == e->grp_ops.front()) CK(flush_group(i));
            }
            if (o.in2 >= 0 && e->needs_grad[o.in2]) {
                const Tensor& ta = e->t[o.in2];
                if (o.add_same) { CK(hipMemcpyAsync(dst_for(o.in2), gy, (size_t)y.B * y.D * y.H * y.W * y.C * 2, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : DREG_ELAUNCH); }
                else if (rows && o.ds_rows >= 0 && o.ds_rows < nlists) {
                    void* dst = dst_for(o.in2);
                    const int nds = (int)rowlists[RL * o.ds_rows + 1];
                    if (o.ds_sparse && dst == grad(o.in2)) {
                        // zero everywhere (see the start of this call); remember which rows this step writes
                        if (nds > 0 && hipMemcpyAsync(A + o.cl2_off, (const int*)rowlists[RL * o.ds_rows], (size_t)nds * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return DREG_ELAUNCH;
                        const_cast<Op&>(o).cl2_count = nds;
                    } else
                    CK(dreg_fill_zero(dst, (size_t)ta.B * ta.D * ta.H * ta.W * ta.C * 2, stream));
                    CK(dreg_downsample_sum_rows(gy, dst, (const int*)rowlists[RL * o.ds_rows], (int)rowlists[RL * o.ds_rows + 1],
                                                y.D, y.H, y.W, ta.D, ta.H, ta.W, y.C, 0, stream));
                }
                else CK(dreg_downsample_sum(gy, dst_for(o.in2), y.B, y.D, y.H, y.W, ta.D, ta.H, ta.W, y.C, 0, stream));
                CK(commit(o.in2));
            }
            bool fused_add = false;
            if (e->needs_grad[o.in]) {
                // a second contribution to an existing gradient: the plain data-gradient convolution adds it in its epilogue (fp32,
                // in place, one rounding) instead of going through the temporary and a separate add
                const bool halo_d = (o.halo & 2) != 0;
                const bool s2_cls = w.pk_cls != SIZE_MAX && s2_class_ok(w, o.ksz, o.stride, o.pad);
                fused_add = written[o.in] && !rows && (halo_d || (s2_cls && g_s2_accumulate) || (!s2_cls && w.pk_dgrad != SIZE_MAX));
                void* gx = fused_add ? grad(o.in) : dst_for(o.in);
                Scope sc(e, st, i, 1);
                if (halo_d) {
                    // gx = [gx +] conv(gy, flipped-tap pack): the halo kernel's same-size addend is the in-place accumulation
                    CK(dreg_conv3_halo_n(gy, PK + w.pk_halo_dgrad, gx, nullptr, fused_add ? gx : nullptr, y.B, y.D, y.H, y.W, w.d0, w.d1,
                                         fused_add ? x.D : 0, fused_add ? x.H : 0, fused_add ? x.W : 0, 1, 0, stream));
                } else if (fused_add && s2_cls) {
                    CK(dreg_conv3d_dgrad_s2_acc(gy, PK + w.pk_cls, gx, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0, o.ksz, o.pad, stream));
                } else if (fused_add) {
                    sc.variant(1);        // dispatched WITH an addend (igemm_choose may pick another tile for it)
                    CK(dreg_conv3d_igemm_ws(gy, PK + w.pk_dgrad, gx, nullptr, gx, x.B, y.D, y.H, y.W, w.d0, x.D, x.H, x.W, w.d1,
                                            o.ksz, o.stride, o.pad, 1, 0, x.D, x.H, x.W, 1, 0, 0, A + e->off_ks, e->sz_ks, stream));
                } else if (rows) {
                    if (o.rows_in < 0 || o.rows_in >= nlists) return DREG_EINVAL;
                    const int64_t* rl = rowlists + RL * o.rows_in;
                    const int* r_in = (const int*)rl[0];
                    const int n_in = (int)rl[1];
                    if (o.sparse_gx && gx == grad(o.in)) {
                        // the buffer is zero everywhere (see the start of this call); remember which rows this step writes
                        if (n_in > 0 && hipMemcpyAsync(A + o.cl_off, r_in, (size_t)n_in * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return DREG_ELAUNCH;
                        const_cast<Op&>(o).cl_count = n_in;
                    } else CK(dreg_fill_zero(gx, (size_t)x.B * x.D * x.H * x.W * x.C * 2, stream));
                    if ((g_brick & (w.d1 == 64 ? 1 : 2)) && w.pk_brick_dgrad != SIZE_MAX && rl[2] && rl[3] > 0) {
                        sc.variant(2);
                        CK(dreg_conv3_brick(gy, PK + w.pk_brick_dgrad, gx, nullptr, nullptr, (const void*)rl[2], (int)rl[3], (const int*)rl[4], (const void*)rl[5],
                                            (const int*)rl[6], x.B, x.D, x.H, x.W, w.d0, w.d1, 0, 0, 0, 0, 0, stream));
                    } else
                        CK(dreg_conv3d_igemm_rows(gy, PK + w.pk_dgrad, gx, nullptr, nullptr, r_in, n_in,
                                                  x.B, y.D, y.H, y.W, w.d0, x.D, x.H, x.W, x.C, o.ksz, 1, o.pad, 1, 0, 0, 0, 0, 0, 0, stream));
                } else if (w.pk_cls != SIZE_MAX && s2_class_ok(w, o.ksz, o.stride, o.pad)) {
                    CK(dreg_conv3d_dgrad_s2(gy, PK + w.pk_cls, gx, x.B, x.D, x.H, x.W, x.C, y.D, y.H, y.W, w.d0, o.ksz, o.pad, stream));
                } else {
                    if (w.pk_dgrad == SIZE_MAX) return DREG_EINVAL;
                    if (o.fold_sk_bwd && gx == grad(o.in) && i - 1 >= 0) {
                        int ns = 0; size_t sl = 0;
                        CK(dreg_conv3d_igemm_defer(gy, PK + w.pk_dgrad, gx, nullptr, nullptr, x.B, y.D, y.H, y.W, w.d0, x.D, x.H, x.W, w.d1,
                                                   o.ksz, o.stride, o.pad, 1, 0, 0, 0, 0, 0, A + e->off_ks, e->sz_ks, nullptr, &ns, &sl, stream));
                        if (ns > 0) { e->pend_sk_bn = i - 1; e->pend_sk_n = ns; e->pend_sk_slice = sl; }
                    } else
                    CK(dreg_conv3d_igemm_ws(gy, PK + w.pk_dgrad, gx, nullptr, nullptr, x.B, y.D, y.H, y.W, w.d0, x.D, x.H, x.W, w.d1,
                                            o.ksz, o.stride, o.pad, 1, 0, 0, 0, 0, 0, 0, 0, A + e->off_ks, e->sz_ks, stream));
                }
            }
            if (e->needs_grad[o.in] && !fused_add) CK(commit(o.in));
        } else if (o.kind == OP_BN) {
            const int V = x.D * x.H * x.W;
            const bool res_g = o.in2 >= 0 && e->needs_grad[o.in2];
            // dx goes to its own buffer or the temporary; a second temporary is never needed: the residual's contribution is
            // written first (into a fresh buffer or, when one exists already, via the temporary + add) only if dx takes the direct path
            void* dx = dst_for(o.in);
            void* dres = nullptr;
            bool res_via_copy = false;
            if (res_g) {
                if (!written[o.in2]) dres = grad(o.in2);
                else if (dx != (void*)(A + e->off_tmp)) dres = A + e->off_tmp;
                else res_via_copy = true;   // both want the temporary: run dx first, then recompute-free path below
            }
            if (!e->prm[o.w].grad || !e->prm[o.b].grad) return DREG_EINVAL;
            int deferred = 0;
            if (res_via_copy) {
                // rare (never in ResNet-50/FPN): produce dres = masked gy through a second pass after dx has been folded in
                CK(dreg_bn3d_bwd_defer_params(act(o.in), gy, act(o.out), (float*)(A + o.aux0), (float*)(A + o.aux1), dx, nullptr, e->prm[o.w].grad, e->prm[o.b].grad,
                                              (float*)(A + e->off_coef), (float*)(A + e->off_bn_ws), x.B, V, x.C, o.relu, 1, 0,
                                              o.bt >= 0 ? (float*)(A + o.keep_sums) : nullptr, &deferred, stream));
                if (deferred) bn_done[o.bt] = 1;
                CK(commit(o.in));
                if (o.relu) CK(dreg_relu_bwd(act(o.out), gy, A + e->off_tmp, (size_t)x.B * V * x.C, 0, 0, 0, stream));
                else if (hipMemcpyAsync(A + e->off_tmp, gy, (size_t)x.B * V * x.C * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) return DREG_ELAUNCH;
                CK(commit(o.in2));
            } else {
                // without a residual the ReLU mask is recomputed from x: y is not read
                dreg_bn_extra ex{};
                if (e->pend_sk_bn == i) { ex.splitk_part = (const float*)(A + e->off_ks); ex.splitk_nsplit = e->pend_sk_n; ex.splitk_slice = e->pend_sk_slice; e->pend_sk_bn = -1; }
                CK(dreg_bn3d_bwd_ex(act(o.in), gy, o.in2 >= 0 ? act(o.out) : nullptr, (float*)(A + o.aux0), (float*)(A + o.aux1), dx, dres, e->prm[o.w].grad, e->prm[o.b].grad,
                                    (float*)(A + e->off_coef), (float*)(A + e->off_bn_ws), x.B, V, x.C, o.relu, 1, 0,
                                    o.bt >= 0 ? (float*)(A + o.keep_sums) : nullptr, &deferred, &ex, stream));
                if (deferred) bn_done[o.bt] = 1;
                if (res_g && dres == (void*)(A + e->off_tmp)) { CK(commit(o.in2)); CK(commit(o.in)); }
                else { CK(commit(o.in)); if (res_g) CK(commit(o.in2)); }
            }
        } else if (o.kind == OP_MAXPOOL) {
            if (!e->needs_grad[o.in]) continue;
            // a second contribution (the stem activation also feeds the FPN's finest lateral): added in place, no temporary + add
            CK(dreg_maxpool3d_bwd_acc(gy, (const uint8_t*)(A + o.aux0), grad(o.in), x.B, x.D, x.H, x.W, y.D, y.H, y.W, x.C, written[o.in] ? 1 : 0, 0, stream));
            written[o.in] = 1;
        }
    }
    CK(guard_after_op(e, A, op_begin, 1, st, aux_on ? e->aux : nullptr));
    CK(run_deferred());                      // (a segment that never reached the deep levels)
    if (!grp_seen.empty()) CK(flush_group(e->grp_ops.front()));      // (the launch point was never met)
    CK(flush_reduce());
    CK(flush_bn_tails(e, A, bn_done, 1, st));
    CK(guard_after_op(e, A, -1, 1, st, aux_on ? e->aux : nullptr));
    if ((flags & 2) && aux_used) {   // the caller's stream continues (optimizer) only after every parameter gradient has landed
        if (hipEventRecord(e->ev_done, e->aux) != hipSuccess || hipStreamWaitEvent(st, e->ev_done, 0) != hipSuccess) return DREG_ELAUNCH;
    }
    return DREG_OK;
}

}  // extern "C"
