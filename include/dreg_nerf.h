/* dreg_nerf.h — C ABI of libdreg_nerf_hip.so (gfx950 / MI355X).
 *
 * The reference (AIBluefisher/DReg-NeRF) has no FFI layer of its own: its hot path reaches native code only through
 * third-party Python packages (torch/cuDNN/cuBLAS, tiny-cuda-nn, MinkowskiEngine, nerfacc, torch_scatter).  Each entry
 * point below replaces one such call site; the citation names the reference line that makes the call.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer (inputs, outputs, workspace) is owned by the caller and must be
 *     device memory of the current HIP device; nothing is retained after return; no allocation inside.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no implicit synchronisation.
 *   - return 0 on success, DREG_EINVAL (-1) for unsupported shapes/arguments, otherwise the hipError_t of the launch.
 *   - `dtype`: 0 = bfloat16 storage (fp32 accumulate), 1 = float32 (exact-f32 MFMA) — applies to activations and
 *     packed weights; statistics, biases, gradients of parameters are always fp32.
 *   - activations are NDHWC: [B, D, H, W, C] with C contiguous and C*sizeof(elem) a multiple of 16 bytes.
 */
#ifndef DREG_NERF_H
#define DREG_NERF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DREG_OK 0
#define DREG_EINVAL (-1)

/* 1: this build stages the operands of bf16 stride-1 convolutions with buffer_load ... lds (always, in the product library) */
int dreg_conv_get_glds(void);

/* ---------------------------------------------------------------------------------------------- convolution / GEMM
 * Replaces torch.nn.Conv3d -> cuDNN (conerf/model/resnet3d.py:79-84,120,143-147;
 * conerf/model/feature_pyramid_net.py:21-36,47-56) and torch.nn.Linear -> cuBLAS
 * (conerf/register/transformer.py:127-133,242-293; conerf/register/nerf_regtr.py:268-270,289-290) with ksz = 1. */

/* padded K (elements) of one packed weight row for a ksz^3 kernel over Cin channels */
int dreg_conv3d_kpad(int ksz, int Cin, int dtype);

/* torch-layout fp32 weight [Cout][Cin_real][ksz^3] -> packed operand.
 * for_dgrad = 0: [Cout][kpad(ksz,Cin)] with K = tap*Cin + ci (Cin >= Cin_real = zero-padded channel count)
 * for_dgrad = 1: [Cin_real][kpad(ksz,Cout)] with K = tap*Cout + co (operand of the data-gradient pass) */
int dreg_pack_conv_weight(const float* w, void* out, int Cout, int Cin_real, int Cin, int ksz, int for_dgrad,
                          int dtype, void* stream);
/* Batched form (all packs of a step in one launch): descs = DEVICE array of n 48-byte records
 * { const float* w; void* out; int Cout, Cin_real, inner (Cout for dgrad packs, padded Cin otherwise), ksz^3, for_dgrad, Kpad,
 *   dtype, row0 } with row0 = exclusive prefix of the packed row counts (Cin_real [x8 for 3^3 class packs] for dgrad packs,
 * Cout otherwise), total_rows = its sum, stage_floats = max over bf16 records with ksz > 1 of min(channels, 64) * ksz^3,
 * row_desc (optional device int32 [total_rows]) = record index of every packed row. */
int dreg_pack_conv_weights_batched(const void* descs, int n, int total_rows, int stage_floats, const int* row_desc,
                                   void* stream);

/* Implicit-GEMM convolution on MFMA.
 * transposed = 0 (forward):        out[b,o,:] = sum_d in[b, o*stride - pad + d, :] . W[:, d, :]  (+bias) (+up2(addend)) (relu)
 * transposed = 1 (data gradient):  out[b,i,:] = sum_d in[b, (i + pad - d)/stride, :] . W'[:, d, :]   ("in" is dOut)
 * in [B,Di,Hi,Wi,Cin], out [B,Do,Ho,Wo,Cout]; Cout % 64 == 0; ksz in {1,3,5}; stride in {1,2}; Cin a power of two
 * unless ksz == 1.  addend (optional, dtype of out) is [B,Da,Ha,Wa,Cout], added with nearest x2 upsampling + crop
 * (FeaturePyramid_v1._upsample, feature_pyramid_net.py:58-61), or element-wise when add_same = 1 (the transformer's
 * residual connections, transformer.py:250,262,283-284,289,293).  out_f32 = 1: bf16 operands, fp32 output.
 * relu = 2 (with add_same = 1): the "addend" is NOT added but used as a mask — out = addend > 0 ? value : 0 — the ReLU backward of a
 * layer whose forward epilogue applied ReLU (addend = that layer's stored activation), fused into the data gradient that produces its
 * output gradient (transformer.py:291 linear1 -> relu -> linear2). */
int dreg_conv3d_igemm(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                      int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                      int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                      int dtype, int out_f32, void* stream);
/* The same with an fp32 scratch buffer (dreg_conv3d_igemm_workspace_bytes; 0 = none needed): small row spaces — the 8^3 / 4^3
 * levels of resnet3d.py's layer3/layer4 — then run split-K and are finished by a reduce + bias/ReLU/cast pass. */
size_t dreg_conv3d_igemm_workspace_bytes(int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                                         int ksz, int stride, int pad, int transposed, int has_addend, int dtype);
int dreg_conv3d_igemm_ws(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                         int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                         int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                         int dtype, int out_f32, void* workspace, size_t workspace_bytes, void* stream);
int dreg_conv3d_igemm_occ(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                          int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                          int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                          int dtype, int out_f32, void* workspace, size_t workspace_bytes, const uint8_t* rowocc, void* stream);

/* Every ResNet3D convolution is followed by a training-mode BatchNorm3d over its own output (resnet3d.py:95-113): the forward bf16
 * convolution can leave that layer's statistics behind — per (grid, chunk of *rows_per_chunk output voxels, channel) the sum and the
 * sum of squares of the STORED values, bn_partial fp32 [B][V / *rows_per_chunk][Cout][2] with V = Do*Ho*Wo — so that
 * dreg_bn3d_fwd_from_sums needs no statistics pass (one full read of the tensor less per layer).  *rows_per_chunk = 0 when this
 * launch could not do it (split-K, a tile that does not divide V, the register-staged kernel): run the ordinary dreg_bn3d_fwd then.
 * bn_partial must hold B * (V / 128) * Cout * 2 floats.  The output is the one of dreg_conv3d_igemm_ws, bit for bit. */
int dreg_conv3d_igemm_bnstats(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                              int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                              int ksz, int stride, int pad, int relu, int Da, int Ha, int Wa, int add_same,
                              void* workspace, size_t workspace_bytes, float* bn_partial, int* rows_per_chunk, void* stream);
int dreg_bn3d_fwd_from_sums(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* sums,
                            int rows_per_chunk, int B, int V, int C, float eps, float momentum, int relu, int dtype, void* stream);

/* ---- dense 3^3 / stride 1 / pad 1 convolution with 256 output channels on an LDS-resident input halo (csrc/conv_halo.hip):
 * the FeaturePyramid_v1 head's upsample_transform_* / pyramid_transformation_1 layers (conerf/model/feature_pyramid_net.py:47-56,
 * 97-103; cuDNN conv3d in the reference) and their data gradients.  A workgroup owns a 4 x 8 x 8 box of output voxels, stages its
 * 6 x 10 x 10 input halo once per 32-channel chunk and runs all 27 taps out of LDS; only the weights stream per tap.
 * A second kernel serves 64 output channels — conv2 of layer1's bottlenecks at 32^3 (conerf/model/resnet3d.py:76-113, Bottleneck) and its data
 * gradient: 8 x 8 x 8 boxes, four waves that each own two z-planes x all 64 channels, two workgroups per CU (dreg_conv3_halo_n).
 * dreg_conv3_halo_supported: 1 when the shape qualifies (Cin % 32 == 0, operands < 2 GiB; Cout == 256 with D % 4 == H % 8 == W % 8 == 0, or
 *   Cout == 64 with D % 8 == H % 8 == W % 8 == 0).
 * dreg_pack_conv_weight_halo: torch weight fp32 [Cout][Cin][27] -> bf16 [Cin'/32][27][256][32] (dreg_conv3_halo_pack_bytes(Cin') bytes);
 *   transposed = 0: forward pack (Cout must be 256, Cin' = Cin); 1: data-gradient pack (Cin must be 256, Cin' = Cout, taps flipped).
 * dreg_conv3_halo: out[b,v,:] = bias + addend + sum_d in[b, v - 1 + d, :] . W[:, d, :]; in [B,D,H,W,Cin] bf16, out [B,D,H,W,256] bf16
 *   (fp32 when out_f32), addend [B,Da,Ha,Wa,256] of out's dtype added with nearest x2 upsampling (add_same = 0) or element-wise (1). */
int dreg_conv3_halo_supported(int B, int D, int H, int W, int Cin, int Cout);
/* supported AND large enough per grid (>= 32^3) to beat the split-K implicit GEMM; independent of B */
int dreg_conv3_halo_use(int B, int D, int H, int W, int Cin, int Cout, int ksz, int stride, int pad);
size_t dreg_conv3_halo_pack_bytes(int Cin_reduced);
int dreg_pack_conv_weight_halo(const float* w, void* out, int Cout, int Cin, int transposed, void* stream);
int dreg_conv3_halo(const void* in, const void* wpk, void* out, const float* bias, const void* addend,
                    int B, int D, int H, int W, int Cin, int Da, int Ha, int Wa, int add_same, int out_f32, void* stream);
/* rows = 256 or 64 output channels: pack size ([Cin'/32][27][rows][32] bf16; dreg_pack_conv_weight_halo takes either) and the launch
 * (out / addend [..., Cout]; Cout == 64: bf16 output only) */
size_t dreg_conv3_halo_pack_bytes_n(int Cin_reduced, int rows);
int dreg_conv3_halo_n(const void* in, const void* wpk, void* out, const float* bias, const void* addend,
                      int B, int D, int H, int W, int Cin, int Cout, int Da, int Ha, int Wa, int add_same, int out_f32, void* stream);
/* the bf16 forward that also leaves the BatchNorm statistics of its OUTPUT behind, like dreg_conv3d_igemm_bnstats: bn_partial [B][V / *rows_per_chunk][Cout][2]
 * (sums of the stored values and of their squares); *rows_per_chunk = 128 for Cout == 64 (a chunk = two z-planes of an 8^3 box), 0 = nothing was written
 * (Cout == 256, or bn_partial == NULL: the BatchNorm runs its own statistics pass) */
int dreg_conv3_halo_n_bnstats(const void* in, const void* wpk, void* out, const float* bias, const void* addend,
                              int B, int D, int H, int W, int Cin, int Cout, int Da, int Ha, int Wa, int add_same, float* bn_partial, int* rows_per_chunk, void* stream);

/* Data gradient of a stride-2 convolution (ksz 3 / pad 1: resnet3d.py conv2 of the first block of layer2-4; ksz 1 / pad 0: the
 * downsample branch) without the 7/8 structurally-zero taps of the gather form: ONE 2^3-tap convolution over dOut whose
 * 8 x Cin output channels are the 8 parity classes of dIn (scattered in place by the epilogue).  bf16 only;
 * wt_class_packed = dreg_pack_conv_weight(..., for_dgrad = 2): rows (class, ci), K = tap*Cout + co. */
int dreg_conv3d_dgrad_s2(const void* gout, const void* wt_class_packed, void* din, int B, int Di, int Hi, int Wi, int Cin,
                         int Do, int Ho, int Wo, int Cout, int ksz, int pad, void* stream);
/* the same ADDED to an existing dIn (fp32 add in the epilogue, one rounding; a 1^3 layer touches the even-coordinate voxels only) */
int dreg_conv3d_dgrad_s2_acc(const void* gout, const void* wt_class_packed, void* din, int B, int Di, int Hi, int Wi, int Cin,
                             int Do, int Ho, int Wo, int Cout, int ksz, int pad, void* stream);

/* Weight gradient (split over voxels, deterministic two-stage reduction):
 * dw[Cout][Cin_real][ksz^3] (fp32, torch layout) (+)= sum_m gout[m,:]^T x in[gather(m, tap), :]. */
/* which bf16 weight-gradient kernel a launch of this shape runs: BM * 1000 + BNC (256256 = the 8-wave tile, 256128 = 4 waves / 32-voxel stages); for profiler labels */
/* which kernel a convolution launch of this shape runs: kind * 1e8 + BM * 1e5 + BN * 100 + AP * 10 + splitK (kind 0 conv_igemm_glds_kernel, 1 conv_igemm_kernel; < 0 unsupported); nrows 0 = dense */
int dreg_conv3d_igemm_variant(int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksz, int stride, int pad,
                              int transposed, int nrows, int has_ws, int has_addend, int dtype);
int dreg_conv3d_wgrad_variant(int B, int Do, int Ho, int Wo, int Cin, int Cout, int ksz, int rows, int nrows, int occ);
int dreg_conv3d_wgrad_splits(int B, int Do, int Ho, int Wo, int Cin, int Cout, int ksz, int dtype);
size_t dreg_conv3d_wgrad_workspace_bytes(int B, int Do, int Ho, int Wo, int Cin, int Cout, int ksz, int dtype);
int dreg_conv3d_wgrad(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                      int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                      int ksz, int stride, int pad, int accumulate, int dtype, int use_tr, void* stream);
int dreg_conv3d_wgrad_occ(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                          int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                          int ksz, int stride, int pad, int accumulate, int dtype, int use_tr, const uint8_t* rowocc, void* stream);

/* Active-set ("row list") forms, bf16, stride 1: only the output voxels rows[0..nrows) (ascending int32 flat indices into
 * B*Do*Ho*Wo, device memory) are computed / reduced over; other rows of `out` are left untouched.  Used for the two FPN head
 * convolutions, whose outputs feed only the trilinear gather at the occupied voxels (nerf_regtr.py:138-147) — identical
 * results, work proportional to the occupied surface instead of the 64^3 volume. */
int dreg_conv3d_igemm_rows(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                           const int* rows, int nrows,
                           int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                           int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                           int out_f32, void* stream);
int dreg_conv3d_wgrad_rows(const void* gout, const void* in, float* dw, void* workspace, size_t workspace_bytes,
                           const int* rows, int nrows,
                           int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                           int ksz, int stride, int pad, int accumulate, void* stream);
/* loss.backward() produces ~100 weight gradients per step (train_nerf_regtr.py:232): the deferred form writes only the split
 * partials of one layer ([dreg_conv3d_wgrad_splits][Cout][Kpad] fp32 into a workspace the caller keeps; rows / rowocc as in the
 * _rows / _occ forms, both may be null), and dreg_wgrad_reduce_batched sums the partials of many layers into their torch-layout
 * gradients [Cout][Cin_real][k^3] with ONE launch.  descs_dev: n records of 48 bytes in device memory
 *   { const float* part; float* dw; int nsplit, Cout, Kpad, ntaps, Cin, Cin_real, accumulate, block0; }
 * with block0 = sum of dreg_wgrad_reduce_blocks(...) of the records before it; workgroups [block_base, block_base + nblocks) run.
 * nsplit is always dreg_conv3d_wgrad_splits(...) (the dense volume's count, which also sizes the workspace).  accumulate: bit 0 = add
 * to dw; bit 1 MUST be set for a workspace written with a row list: such a launch fills only as many slices as its row count is
 * worth and stores that number (int) behind the nsplit-th slice, where the sum reads it. */
int dreg_conv3d_wgrad_partials(const void* gout, const void* in, void* workspace, size_t workspace_bytes, const int* rows, int nrows,
                               int B, int Di, int Hi, int Wi, int Cin, int Cin_real, int Do, int Ho, int Wo, int Cout,
                               int ksz, int stride, int pad, const uint8_t* rowocc, void* stream);
int dreg_wgrad_reduce_blocks(int Cout, int Cin_real, int ksz, int nsplit);
int dreg_wgrad_reduce_batched(const void* descs_dev, int n, int block_base, int nblocks, void* stream);
/* The split partials of MANY linear layers by one launch per tile shape (the 36 `nn.Linear` weight gradients of a backward pass through
 * transformer.py:225-299 / nerf_regtr.py:268-270: ~25 us per launch for ~4 GFLOP one at a time).  dreg_linear_wgrad_group_fill writes the
 * descriptor (dreg_wgrad_group_desc_bytes bytes, host memory) of one layer — x: bf16 [rows, Cin], gout: bf16 [rows, Cout], partials into
 * `workspace` exactly as dreg_conv3d_wgrad_partials(B = rows, 1^3) leaves them — and returns its tile shape (*variant) and workgroup count;
 * the caller sets block0 (the descriptor's last int: exclusive prefix of the workgroup counts inside one variant's table), copies the tables
 * to the device and calls dreg_wgrad_group_launch once per variant, then dreg_wgrad_reduce_batched as usual.  Bit-identical partials. */
int dreg_wgrad_group_desc_bytes(void);
int dreg_linear_wgrad_group_fill(void* desc_host, const void* gout, const void* in, void* workspace, size_t workspace_bytes,
                                 int rows, int Cin, int Cout, int* variant, int* nblocks);
/* the same for a dense convolution layer (no row list, no occupancy flags): gout bf16 [B,Do,Ho,Wo,Cout], in bf16 [B,Di,Hi,Wi,Cin] */
int dreg_conv3d_wgrad_group_fill(void* desc_host, const void* gout, const void* in, void* workspace, size_t workspace_bytes,
                                 int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksz, int stride, int pad,
                                 int* variant, int* nblocks);
int dreg_wgrad_group_launch(const void* descs_dev, int n, int variant, int total_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------- FPN3D companions
 * BatchNorm3d with the reference's one-grid-per-call statistics (resnet3d.py:121,159; nerf_regtr.py:135), fused
 * residual add + ReLU (resnet3d.py:95-113). x,res,y: [B,V,C]; scale_shift, mean_rstd: fp32 [B,C,2] (saved for bwd);
 * workspace fp32 [B * dreg_bn_num_chunks(V) * C * 2].  train = 0 uses the running statistics. */
int dreg_bn_num_chunks(int V);
int dreg_bn3d_fwd(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                  int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, void* stream);
/* y may be NULL when the forward had no residual: the ReLU mask is then recomputed as x*scale + shift > 0 (one read less). */
int dreg_bn3d_bwd(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                  void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                  int B, int V, int C, int relu, int accumulate, int dtype, void* stream);
/* The two launches of a SMALL BatchNorm (one grid's column slab per workgroup; dreg_bn_small says which shapes) with the tiny cross-grid
 * tail left to the caller, who batches it over the layers of a pass: the forward keeps the per-grid variances in var_keep [B][C], the
 * backward the per-grid sums in sums_keep [B][C][2], and *deferred = 1.  For other shapes (and eval mode) they run as dreg_bn3d_fwd /
 * dreg_bn3d_bwd and *deferred = 0.  The tables are n records of 48 bytes
 *   { const float* a; const float* b; float* o0; float* o1; int B, V, C, block0; }
 * (running update: a = mean_rstd, b = var_keep, o0 / o1 = running_mean / running_var;  parameter gradients: a = sums_keep, b unused,
 * o0 / o1 = dgamma / dbeta) in device memory, block0 = sum of ceil(C / 256) over the records before; workgroups
 * [block_base, block_base + nblocks) of the table run.  Same arithmetic per channel as the unbatched launches (results bit-identical). */
int dreg_bn_small(int B, int V, int C, int dtype);
int dreg_bn3d_fwd_defer_update(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                               int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, float* var_keep, int* deferred, void* stream);
int dreg_bn3d_bwd_defer_params(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                               void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                               int B, int V, int C, int relu, int accumulate, int dtype, float* sums_keep, int* deferred, void* stream);
/* Optional extras of ONE BatchNorm call, handed over as an argument (the library keeps no per-thread "next call" state):
 *  - res_scale_shift (forward, large path): `res` is the INPUT of a ReLU-free BatchNorm that was run with y == NULL (statistics, scale / shift
 *    only — the downsample branch of a bottleneck's first block, resnet3d.py:104-110); res_scale_shift = that layer's scale_shift.  The branch's
 *    output is formed on the fly, rounded as the separate apply pass stores it (bit-identical).
 *  - splitk_part / splitk_nsplit / splitk_slice (register-resident one-launch kernels only, dreg_bn_small_in_regs): the call's x (forward) / dy
 *    (backward) is the sum of the fp32 split-K slices [nsplit][slice] a convolution left un-summed (dreg_conv3d_igemm_defer: conv2 -> bn2 forward,
 *    conv2's data gradient -> bn1 backward at the 8^3 / 4^3 levels of resnet3d.py's layer3 / layer4): forward — x := the rounded sum (stored to x:
 *    the backward pass reads it); backward — dy := the rounded sum.  DREG_EINVAL when the call does not take that path. */
typedef struct dreg_bn_extra {
    const float* res_scale_shift;
    const float* splitk_part;
    int splitk_nsplit;
    size_t splitk_slice;
} dreg_bn_extra;
/* dreg_bn3d_fwd_defer_update (sums_rows_per_chunk = 0) / dreg_bn3d_fwd_from_sums (> 0: workspace = the chunk sums, training mode) with extras (may be NULL) */
int dreg_bn3d_fwd_ex(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                     int B, int V, int C, float eps, float momentum, int train, int relu, int dtype, float* var_keep, int* deferred,
                     int sums_rows_per_chunk, const dreg_bn_extra* ex, void* stream);
int dreg_bn3d_bwd_ex(const void* x, const void* dy, const void* y, const float* scale_shift, const float* mean_rstd,
                     void* dx, void* dres, float* dgamma, float* dbeta, float* coef, float* workspace,
                     int B, int V, int C, int relu, int accumulate, int dtype, float* sums_keep, int* deferred, const dreg_bn_extra* ex, void* stream);
/* dreg_conv3d_igemm_occ (bf16 in / out) that may leave a split-K launch's fp32 slices [*sk_nsplit][*sk_slice] un-summed in `workspace` for the
 * BatchNorm behind it: *sk_nsplit = 0 when the launch finished its output itself (not split-K, or a bias / ReLU epilogue). */
int dreg_conv3d_igemm_defer(const void* in, const void* wt_packed, void* out, const float* bias, const void* addend,
                            int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                            int ksz, int stride, int pad, int transposed, int relu, int Da, int Ha, int Wa, int add_same,
                            void* workspace, size_t workspace_bytes, const uint8_t* rowocc, int* sk_nsplit, size_t* sk_slice, void* stream);
int dreg_bn_small_in_regs(int B, int V, int C, int dtype);
int dreg_bn_running_update_batched(const void* descs_dev, int n, int block_base, int nblocks, float momentum, void* stream);
int dreg_bn_param_grad_batched(const void* descs_dev, int n, int block_base, int nblocks, int accumulate, void* stream);

/* Stem fused (resnet3d.py:118-123: conv1 -> bn1 -> relu -> maxpool, bf16): pooled = maxpool3(relu(bn(x))) and its backward without the
 * full-resolution activation / gradient in between; pooled values and arg-max taps are bit-identical to dreg_bn3d_fwd + dreg_maxpool3d_fwd. */
int dreg_bn_relu_maxpool_fwd(const void* x, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                             int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, float eps, float momentum, int train, int relu, void* stream);
int dreg_bn_relu_maxpool_bwd(const void* x, const void* dp, const uint8_t* argmax, const float* scale_shift, const float* mean_rstd,
                             void* dx, float* dgamma, float* dbeta, float* coef, float* workspace,
                             int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu, int accumulate, void* stream);
/* The same behind a stem computed on a row list over a sparse volume (dreg_conv_rows: x is zero outside `rows`, ascending flat indices into
 * [B,Di*Hi*Wi]): statistics over the listed rows, pooled windows without a listed row written as the constant relu(shift), the dense-layout
 * activation `act` (optional) written on rows_a only (the FPN's finest lateral reads nothing else: feature_pyramid_net.py:97-103), xam = raw
 * x of the arg-max voxels (bf16 [B,Do,Ho,Wo,C]); pmask: B*Do*Ho*Wo bytes of scratch.  Backward: the BatchNorm sums run over the POOLED
 * elements (+ dl, the activation's gradient from the lateral, non-zero on rows_a only); dx is written on `rows` only (all the stem's weight
 * gradient reads).  Pooled values / arg-max taps equal dreg_bn_relu_maxpool_fwd's given the same scale / shift; the statistics differ by
 * fp32 summation order.  workspace: dreg_sparse_stem_workspace_floats floats. */
size_t dreg_sparse_stem_workspace_floats(int B, int Do, int Ho, int Wo, int C);
int dreg_sparse_stem_fwd(const void* x, const int* rows, int n, const int* rows_a, int n_a, void* act, void* pooled, uint8_t* argmax, void* xam, uint8_t* pmask,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale_shift, float* mean_rstd, float* workspace,
                         int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, float eps, float momentum, int train, int relu, void* stream);
int dreg_sparse_stem_bwd(const void* x, const void* dp, const uint8_t* argmax, const void* xam, const void* dl, const int* rows_a, int n_a,
                         const int* rows, int n, const float* scale_shift, const float* mean_rstd, void* dx, float* dgamma, float* dbeta, float* coef, float* workspace,
                         int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int relu, int accumulate, void* stream);
/* nn.MaxPool3d(3, 2, 1) (resnet3d.py:123,161); argmax: uint8 [B,Do,Ho,Wo,C] tap index for the backward pass. */
int dreg_maxpool3d_fwd(const void* x, void* y, uint8_t* argmax, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                       int C, int dtype, void* stream);
int dreg_maxpool3d_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int Di, int Hi, int Wi, int Do, int Ho,
                       int Wo, int C, int dtype, void* stream);
/* the same adding into an existing dx (accumulate = 1): a tensor with a second consumer gets its gradient in one read-modify-write */
int dreg_maxpool3d_bwd_acc(const void* dy, const uint8_t* argmax, void* dx, int B, int Di, int Hi, int Wi, int Do, int Ho,
                           int Wo, int C, int accumulate, int dtype, void* stream);

/* backward of the nearest x2 upsample + crop: out[b,z,y,x,:] = sum of the in-range 2x2x2 children of g */
int dreg_downsample_sum(const void* g, void* out, int B, int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int C,
                        int dtype, void* stream);
/* the same over a list of coarse voxels only (out is left untouched elsewhere; used when the fine gradient lives on an active set) */
int dreg_downsample_sum_rows(const void* g, void* out, const int* rows, int nrows, int Df, int Hf, int Wf, int Dc, int Hc, int Wc,
                             int C, int dtype, void* stream);

/* out[c] (+)= sum_m g[m][c]  (bias gradients) */
size_t dreg_colsum_workspace_bytes(size_t M, int C);
int dreg_colsum(const void* g, float* out, float* workspace, size_t M, int C, int accumulate, int dtype, void* stream);

/* F.interpolate(trilinear, align_corners=True) + masked gather fused (nerf_regtr.py:138-147): p1 [B,d,h,w,C],
 * idx int64 [N] = (x*Yr + y)*Zr + z flat fine-grid index, pt_batch int32 [N] grid id; out [N,C]. */
int dreg_trilinear_gather_fwd(const void* p1, const int64_t* idx, const int* pt_batch, void* out, int N, int d, int h,
                              int w, int C, int Zr, int Xr, int Yr, int dtype, int out_f32, void* stream);
int dreg_trilinear_gather_bwd(const float* dfeat, const int64_t* idx, const int* pt_batch, float* dp1_f32, int N, int d,
                              int h, int w, int C, int Zr, int Xr, int Yr, void* stream);
int dreg_cast_from_f32(const float* in, void* out, size_t n, int dtype, void* stream);
/* zero fill at HBM rate (bytes % 16 == 0 and a 16-byte aligned pointer; otherwise hipMemsetAsync) */
int dreg_fill_zero(void* p, size_t bytes, void* stream);
/* Guard bands (the executors' debug mode, dreg_exec_opts.guard): n bands of band_bytes (a multiple of 16) at base + offsets[i] (device uint64 [n]) are
 * filled with the poison word 0xA5C3A5C3; dreg_guard_scan writes result (device int64 [4]) = { bands with a changed word, smallest such band index
 * (n if none), byte offset of its first changed 16-byte word, changed words in total }. */
int dreg_guard_fill(void* base, const void* offsets_dev, int n, int band_bytes, void* stream);
int dreg_guard_scan(const void* base, const void* offsets_dev, int n, int band_bytes, void* result_dev, void* stream);
/* Bias gradients of many linear layers in two launches.  descs_dev: n records of 56 bytes { const bf16* g [M][C]; float* out [C];
 * float* partial (dreg_colsum_workspace_bytes(M, C), one per record); int M, C, rpc (dreg_colsum_rows_per_chunk(M)), nch = ceil(M / rpc),
 * pblock0 (sum of nch of the records before), fblock0 (sum of ceil(C / 4) before), accumulate, pad }; total_pblocks / total_fblocks = the
 * sums over all records.  bf16, C % 8 == 0, C <= 2048.  Every sum is formed in dreg_colsum's order (bit-identical results). */
int dreg_colsum_rows_per_chunk(size_t M);
int dreg_colsum_batched(const void* descs_dev, int n, int total_pblocks, int total_fblocks, void* stream);
int dreg_add_inplace(void* dst, const void* src, size_t n, int dtype, void* stream);   /* dst += src */
/* Input staging (nerf_regtr.py:131-147): grids = DEVICE array of B pointers to fp32 [7,Z,X,Y] voxel grids (xyz | rgb | alpha).
 * pack: -> [B,Z,X,Y,8] (dtype) = rgba + 4 zero channels, the NDHWC stem input.  gather: xyz fp32 [N,3] of the occupied voxels
 * (idx int64 [N] = (x*Yr + y)*Zr + z, pt_batch int32 [N]). */
int dreg_pack_rgba_grids(const void* grids, void* out, int B, int Z, int X, int Y, int dtype, void* stream);
/* the same + inocc byte [B,Z,X] (null ok): 1 for every Y-row of the input volume that holds a non-zero value */
int dreg_pack_rgba_grids_occ(const void* grids, void* out, uint8_t* inocc, int B, int Z, int X, int Y, int dtype, void* stream);
/* sparse form (conerf/datasets/register/dataset.py:221-331 keeps dense 58.7 MB grids; they are zero outside voxel_mask.pt):
 * vals fp32 [N,7] of the occupied voxels idx[n] of grid pt_batch[n] -> zero-filled [B,Z,X,Y,8] + their rgba */
int dreg_pack_rgba_sparse(const float* vals, const int64_t* idx, const int* pt_batch, void* out, int N, int B, int Z, int X,
                          int Y, int dtype, void* stream);
int dreg_pack_rgba_sparse_occ(const float* vals, const int64_t* idx, const int* pt_batch, void* out, uint8_t* inocc, int N, int B, int Z, int X,
                              int Y, int dtype, void* stream);
/* rowocc byte [B,Do,Ho] = any inocc [B,Di,Hi] inside the window of output row (zo, ho, *) of a ksz^3 / stride / pad convolution; the _occ
   convolution entry points skip rows flagged 0 (their result is exactly zero).  Reference: conv1 of conerf/model/resnet3d.py:118-121 on the
   voxel grids of eval_ngp_nerf.py:397-405, which are zero outside voxel_mask. */
int dreg_conv_row_occupancy(const uint8_t* inocc, uint8_t* rowocc, int B, int Di, int Hi, int Do, int Ho, int ksz, int stride, int pad, void* stream);
int dreg_gather_grid_xyz(const void* grids, const int64_t* idx, const int* pt_batch, float* xyz, int N, int Zr, int Xr, int Yr,
                         void* stream);

/* Active sets of the FPN head (build-side; identical results to the dense evaluation of feature_pyramid_net.py:115-127 because
 * nerf_regtr.py:138-147 only consumes P1 at the trilinear corners of the occupied voxels): S1 = those corners, S2 = S1 dilated
 * by 3^3, S3 = S2 dilated.  rows int32 [3][V] (V = B*d*h*w, ascending flat indices, list k at rows + k*V), counts device
 * int32 [3], map1 int32 [V] = rank in S1 or -1 (may be NULL). */
size_t dreg_active_sets_workspace_bytes(int B, int d, int h, int w);
int dreg_active_sets(const int64_t* idx, const int* pt_batch, int N, int B, int Zr, int Xr, int Yr, int d, int h, int w,
                     int* rows, int* counts, int* map1, void* workspace, size_t workspace_bytes, void* stream);
/* Second pyramid level: child_flags = the S2 flags of dreg_active_sets (bytes [V, 2V) of its workspace); A = parents of S2 on
 * [B,d2,h2,w2] (where P2 is consumed by the nearest-x2 upsample-add of feature_pyramid_net.py:58-61), A2 = A dilated by 3^3.
 * rows2 int32 [2][V2], counts2 device int32 [2]. */
size_t dreg_active_sets_level2_workspace_bytes(int B, int d2, int h2, int w2);
int dreg_active_sets_level2(const uint8_t* child_flags, int B, int d, int h, int w, int d2, int h2, int w2, int* rows2,
                            int* counts2, void* workspace, size_t workspace_bytes, void* stream);
/* Output voxels of a strided convolution over a sparse volume whose receptive field holds an occupied input voxel (a bias-free
 * convolution is exactly zero everywhere else): the stem (conerf/model/resnet3d.py conv1: 5^3 taps, stride 2, pad 2) then runs on ~5 %
 * of its rows (dreg_conv3d_igemm_rows / dreg_conv3d_wgrad_partials take the list).  idx / pt_batch as in dreg_active_sets; rows: ascending
 * int32 flat indices into [B, d, h, w] (capacity B*d*h*w); count: device int32. */
size_t dreg_conv_rows_workspace_bytes(int B, int d, int h, int w);
int dreg_conv_rows(const int64_t* idx, const int* pt_batch, int N, int B, int Zr, int Xr, int Yr, int d, int h, int w, int ksz, int stride, int pad,
                   int* rows, int* count, void* workspace, size_t workspace_bytes, void* stream);
/* active-set forms of the gather backward (dp1 zero-filled here, gradient on the S1 rows only; comp: fp32 [n1,C] scratch) and
 * of the bias-gradient column sum (rows of g outside the list are known to be zero). */
int dreg_trilinear_gather_bwd_rows(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                   const int* map1, float* comp, void* dp1, int N, int B, int d, int h, int w, int C,
                                   int Zr, int Xr, int Yr, int dtype, void* stream);
/* deterministic (atomic-free) gather backward: one wave per S1 voxel collects the occupied fine voxels whose corner set
 * contains it; fine_map int32 [B*Zr*Xr*Yr] scratch; C in {64,128,192,256}. */
int dreg_trilinear_gather_bwd_gather(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                     int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr,
                                     int Yr, int dtype, void* stream);
/* The dense gradient buffers in front of the active-set convolutions only ever hold values on a row list.  Kept across steps they stay
 * zero elsewhere: _rows_only writes rows1 without the dense memset (dp1 must be zero outside rows1), and dreg_zero_rows returns the
 * rows written by the previous step to zero (buf: [*, C] in dtype, rows: int32 flat row indices). */
int dreg_trilinear_gather_bwd_gather_rows_only(const float* dfeat, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                               int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                                               int dtype, void* stream);
/* The same with d(feats) given through the first voxel-average round that consumed the gathered features (grid_downsample.py:6-94): g1 fp32
 * [rows, C] = gradient of that round's outputs, the pairs' blocks one after the other; seg_descs: device int64 [B][4] = per grid (inv_seg
 * pointer, inv_cnt pointer, index of its pair's first point, first g1 row of its pair); zero_dense as the two forms above (1 = memset dp1).
 * Bit-identical to dreg_voxel_downsample_bwd per pair + dreg_trilinear_gather_bwd_gather, without the [N, C] gradient in between. */
int dreg_trilinear_gather_bwd_gather_seg(const float* g1, const void* seg_descs, const int64_t* idx, const int* pt_batch, const int* rows1, int n1,
                                         int* fine_map, void* dp1, int N, int B, int d, int h, int w, int C, int Zr, int Xr, int Yr,
                                         int dtype, int zero_dense, void* stream);
/* nerf_regtr.py:138-168, forward: the trilinear gather fused with the first voxel-average round of ONE pair — out[seg] = mean over the round's
 * members of their trilinear samples (bit-identical to dreg_trilinear_gather_fwd + dreg_voxel_segment_mean; the [N, C] features are never
 * written).  idx / pt_batch: all points; point_start: the pair's first point; order / starts / n_out: the round's plan. */
int dreg_gather_segment_mean(const void* p1, const int64_t* idx, const int* pt_batch, int point_start, const uint32_t* order, const uint32_t* starts,
                             const int* n_out, float* out, int M, int d, int h, int w, int C, int Zr, int Xr, int Yr, int dtype, void* stream);
int dreg_zero_rows(void* buf, const int* rows, int nrows, int C, int dtype, void* stream);
int dreg_colsum_rows(const void* g, const int* rows, int nrows, float* out, float* workspace, int C, int accumulate,
                     int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------- trunk executor
 * The FPN3D forward/backward (resnet3d.py:86-161, feature_pyramid_net.py:97-127) as a recorded op program issued from C++:
 * one call per pass instead of one host round trip per layer.  tensors int32 [nt][5] (B,D,H,W,C; bf16 NDHWC; tensor 0 = the
 * external input), ops int32 [nops][16] { kind (0 conv, 1 batchnorm, 2 maxpool 3/2/1, 3 active-set conv), in, out,
 * in2 (addend / residual or -1), w, b, p2, p3, p4 (BN: gamma, beta, running_mean, running_var), ksz, stride, pad, relu,
 * add_same, rows_out, rows_in }, params int64 [np][5] { fp32 value ptr, fp32 grad ptr (0 = frozen), d0, d1, ksz }.
 * The caller owns the arena (activations kept for backward, statistics, activation gradients, scratch) and the packed-weight
 * buffer; weight / bias / BatchNorm gradients are ACCUMULATED into the grad pointers.  One forward may be outstanding. */
void* dreg_exec_create(const int* tensors, int nt, const int* ops, int nops, const int64_t* params, int np);
/* Creation options of ONE executor (the library has no process-global switches: two handles with different options may run side by side).
 * Every option only selects between forms that compute the same layer; "bit-identical" / "one rounding" as noted.  dreg_exec_default_opts fills the
 * product defaults; dreg_exec_create == dreg_exec_create_opts(..., NULL). */
typedef struct dreg_exec_opts {
    int sparse_grads;     /* 1: the single-writer gradient buffers of the active-set head are kept zero by clearing rows (0: dense memset per step) */
    int bn_batch_tails;   /* 1: the small BatchNorms' running-statistics / dgamma-dbeta launches batched per pass (bit-identical) */
    int fuse_stem;        /* 1: the stem's BatchNorm + ReLU + max-pool in one pass */
    int sparse_stem;      /* 1: ... and, behind a row-list stem, from the row lists (dreg_sparse_stem_fwd / _bwd); 0: the dense three-pass form (the per-op path's arithmetic) */
    int fold_res_bn;      /* 1: a downsample branch's BatchNorm applied inside the BatchNorm that adds it (large path; bit-identical) */
    int fold_splitk;      /* 1: split-K sums of the 8^3 / 4^3 convolutions folded into the one-launch BatchNorm next to them (bit-identical) */
    int group_wgrad;      /* 1: weight-gradient partials of the 16^3 / 8^3 / 4^3 levels by one launch per tile shape per backward pass (bit-identical) */
    int s2_accumulate;    /* 1: a stride-2 data gradient that is its input's second contribution adds in its own epilogue (one rounding instead of two) */
    int fuse_bn_stats;    /* 1: statistics of the large BatchNorm layers from the producing convolution's epilogue (dreg_conv3d_igemm_bnstats) */
    int brick;            /* mask: bit 0 (default) = active-set 3^3 launches with 64 output channels on csrc/conv_brick.hip, bit 1 = those with 256 as well; 0 = the per-op path's kernels */
    int defer_head_pg;    /* 0 (default; measured no gain): 1 = the head's weight / bias gradient launches held back until the backward pass reaches the 8^3 / 4^3 levels */
    int persistent_deep;  /* reserved (0) */
    int guard;            /* debug: 1 = every region of the arena (each activation, statistic, gradient, row-list copy, scratch buffer) is followed by a 64 KiB
                             poisoned guard band (dreg_exec_guard_check scans them); 2 = forward / backward additionally scan after EVERY op (stream
                             synchronisation per op) and return DREG_EGUARD at the first op behind which a band has changed (dreg_exec_guard_last) */
    int reserved[3];
} dreg_exec_opts;
void dreg_exec_default_opts(dreg_exec_opts* o);
void* dreg_exec_create_opts(const int* tensors, int nt, const int* ops, int nops, const int64_t* params, int np, const dreg_exec_opts* opts);
/* guard mode (opts.guard >= 1).  dreg_exec_guard_check: scan the bands of `arena` now (synchronises `stream`); out4 = { changed bands, first changed band,
 * byte offset of its first changed word inside the band, changed 16-byte words }.  dreg_exec_guard_describe: the region in FRONT of band `band`
 * ("act 17", "bn.aux0 op 5", ...) into buf.  dreg_exec_guard_last: the op index (negative: none) and band a guard = 2 pass stopped at, pass = 0 forward / 1 backward. */
#define DREG_EGUARD (-3)
int dreg_exec_guard_bands(void* h);
int dreg_exec_guard_check(void* h, void* arena, long long* out4, void* stream);
int dreg_exec_guard_describe(void* h, int band, char* buf, int buf_bytes);
int dreg_exec_guard_last(void* h, int* op, int* pass, long long* band);
void dreg_exec_destroy(void* h);
size_t dreg_exec_arena_bytes(void* h);
size_t dreg_exec_pack_bytes(void* h);
int dreg_exec_num_packs(void* h);
size_t dreg_exec_tensor_offset(void* h, int slot);
int dreg_exec_output_slot(void* h);
int dreg_exec_pack_rows(void* h);
int dreg_exec_export_pack_table(void* h, void* host_out, int* host_row_desc, void* pack_base);   /* 48-byte records + row->record map */
int dreg_exec_repack(void* h, const void* descs_dev, const int* row_desc_dev, void* stream);
int dreg_exec_op_halo(void* h, int op);   /* bit 0 / 1: forward / data gradient of convolution `op` runs on the halo kernel (dreg_conv3_halo_use) */
void dreg_exec_set_overlap(void* h, int enable);
void dreg_exec_set_input_row_occupancy(void* h, const uint8_t* rowocc);   /* flags for the convolution that reads x_in (the stem); null = none */                             /* 1 (default): weight gradients on aux_stream */
void dreg_exec_set_timing(void* h, int enable);                              /* HIP events around every convolution launch */
int dreg_exec_read_timings(void* h, int* op_kind, float* ms, int max);       /* op_kind: 3 ints per record (op, kind 0 fwd / 1 dgrad / 2 wgrad, variant: 2 = ran on conv_brick.hip), ms */
/* rowlists: int64 [nlists][8] per active-set row list = { device int32* rows, count, then (or zeros) the tile tables of
 * dreg_brick_tiles_build for the same set: tiles, ntiles, halo_vox, nbr, rows_sorted, 0 }.  A 3^3 active-set convolution whose row
 * list (forward: output rows; data gradient: input rows) comes with tables runs on dreg_conv3_brick, otherwise on dreg_conv3d_igemm_rows. */
int dreg_exec_forward(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in,
                      const int64_t* rowlists, int nlists, int train, void* stream);
int dreg_exec_backward(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in,
                       const void* grad_out, const int64_t* rowlists, int nlists, void* stream, void* aux_stream);
/* the same in segments (ops [op_begin, op_end) in reverse order; flags bit 0 = first segment of the pass, bit 1 = last): between two
 * segments every parameter gradient of the ops processed so far has been enqueued on `stream` / `aux_stream` */
int dreg_exec_backward_range(void* h, void* arena, size_t arena_bytes, const void* pack_base, const void* x_in, const void* grad_out,
                             const int64_t* rowlists, int nlists, void* stream, void* aux_stream, int op_begin, int op_end, int flags);

/* ---------------------------------------------------------------------------------------------- point-set half
 * Attention core of nn.MultiheadAttention (8 heads, d_head 32; transformer.py:242-281): q [Nq,ldq], k [Nk,ldk],
 * v [Nk,ldv], o [Nq,ldo], head h at column offset 32*h, lse fp32 [H,Nq].  Flash-style, nothing N x N is materialised. */
int dreg_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int Nq, int Nk, int H,
                 int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
int dreg_mha_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 float* dvec, void* dq, void* dk, void* dv, int Nq, int Nk, int H,
                 int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
/* CorrespondenceDecoder.simple_attention (nerf_regtr.py:273-308) for the L stacked layer outputs at once:
 * out[l] = softmax(scale * q[l] k[l]^T) xyz;  q [L,Nq,256], k [L,Nk,256] (dtype), xyz fp32 [Nk,3], out fp32 [L,Nq,3]. */
int dreg_corr_attention_fwd(const void* q, const void* k, const float* xyz, float* out, float* lse, int L, int Nq, int Nk,
                            float scale, int dtype, void* stream);
int dreg_corr_attention_bwd(const void* q, const void* k, const float* xyz, const float* out, const float* dout,
                            const float* lse, float* dvec, void* dq, void* dk, int L, int Nq, int Nk,
                            float scale, int dtype, void* stream);

/* Variable-length batched forms (all pairs of a step in one launch): `probs` is a device array of nprob x
 * {q_start, q_len, kv_start, kv_len} (int32) selecting row ranges of shared [R, ld] tensors; max_q / max_k bound the lengths. */
int dreg_mha_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* probs, int nprob,
                        int max_q, int max_k, int R, int H, int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
int dreg_mha_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                        float* dvec, void* dq, void* dk, void* dv, const int* probs, int nprob, int max_q, int max_k, int R, int H,
                        int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
int dreg_corr_attention_varlen_fwd(const void* q, const void* k, const float* xyz, float* out, float* lse, const int* probs, int nprob,
                                   int max_q, int max_k, int L, int R, float scale, int dtype, void* stream);
int dreg_corr_attention_varlen_bwd(const void* q, const void* k, const float* xyz, const float* out, const float* dout,
                                   const float* lse, float* dvec, void* dq, void* dk, const int* probs, int nprob,
                                   int max_q, int max_k, int L, int R, float scale, int dtype, void* stream);

/* nn.LayerNorm(256) fused with the position-embedding add (transformer.py:238-239,252-253,265-267): y = LN(x)*g + b (+pe).
 * x fp32 [N,256]; y in out_dtype; stats fp32 [N,2] = (mean, rstd). */
int dreg_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* pe, void* y, float* stats,
                       int N, int C, float eps, int out_dtype, void* stream);
size_t dreg_layernorm_bwd_workspace_bytes(int N);
int dreg_layernorm_bwd(const float* x, const void* dy, const float* gamma, const float* stats, float* dx, float* dgamma,
                       float* dbeta, float* workspace, int N, int C, int g_dtype, int accumulate_dx, int accumulate_w,
                       void* stream);
/* dx = LayerNorm-backward(dy) + dx_add (fp32 [N,256], null ok, may alias dx): the by-passing residual branch's gradient folded in */
int dreg_layernorm_bwd_add(const float* x, const void* dy, const float* gamma, const float* stats, float* dx, const float* dx_add, float* dgamma,
                           float* dbeta, float* workspace, int N, int C, int g_dtype, int accumulate_w, void* stream);

/* The same LayerNorm with two outputs from one read of x: y32 fp32 (no pe) and / or y16 bf16 (+ pe[row % pe_rows]); either may be null.
 * The encoder's shared final norm feeds the heads / losses in fp32 and the decoder's projections as LN(x) + pe (nerf_regtr.py:170-206). */
int dreg_layernorm_fwd2(const float* x, const float* gamma, const float* beta, const float* pe, int pe_rows, float* y32, void* y16, float* stats,
                        int N, int C, float eps, void* stream);
/* Row pass of the LayerNorm backward alone: dx = (LN-backward(dy [+ dy2_bf16]) + dx_add) + dx_add2 (addends optional, may alias dx),
 * optional bf16 copy of dx (the operand of the GEMMs that consume it), block partials of dgamma / dbeta into `part`
 * (dreg_layernorm_bwd_workspace_bytes(N), kept by the caller).  dreg_layernorm_bwd_final_batched sums the partials of many LayerNorm
 * applications with one launch: descs_dev = n records of 32 bytes { const float* part; float* dgamma; float* dbeta; int nblk
 * (dreg_layernorm_bwd_blocks(N)); int accumulate; }. */
int dreg_layernorm_bwd_parts(const float* x, const void* dy, const void* dy2_bf16, const float* gamma, const float* stats, float* dx, const float* dx_add,
                             const float* dx_add2, void* dx_bf16, float* part, int N, int C, int g_dtype, void* stream);
int dreg_layernorm_bwd_blocks(int N);
int dreg_layernorm_bwd_final_batched(const void* descs_dev, int n, void* stream);

/* PositionEmbeddingCoordsSine.forward (position_embedding.py:30-53): xyz fp32 [N,3] -> pe fp32 [N,256]. */
int dreg_posenc_sine(const float* xyz, float* pe, int N, float scale, float temperature, void* stream);

/* conf_logits_decoder + sigmoid (nerf_regtr.py:384-387): s[n] = sigmoid(f[n,:] . w + b). */
int dreg_overlap_fwd(const float* f, const float* w, const float* b, float* s, int N, void* stream);
size_t dreg_overlap_bwd_workspace_bytes(int N);
int dreg_overlap_bwd(const float* f, const float* w, const float* s, const float* gy, float* df, float* dw, float* db,
                     float* workspace, int N, void* stream);

/* the same with accumulation: df = df_add + dlogit * w (df_add: the gradient f receives from elsewhere — f also feeds the losses directly —
 * or null; may be df itself), dw / db += when accumulate_w */
int dreg_overlap_bwd_acc(const float* f, const float* w, const float* s, const float* gy, float* df, const float* df_add, float* dw, float* db,
                         int accumulate_w, float* workspace, int N, void* stream);

/* out = (y > 0 ? g : 0) cast to out_dtype (backward of the fused ReLU epilogue; also the fp32 -> bf16 gradient cast). */
int dreg_relu_bwd(const void* y, const void* g, void* out, size_t n, int y_dtype, int g_dtype, int out_dtype, void* stream);

/* compute_rigid_transform (se3.py:89-140): a,b fp32 [P,N,3], w fp32 [P,N] -> out fp32 [P,3,4]; 3x3 SVD on device. */
int dreg_weighted_kabsch(const float* a, const float* b, const float* w, float* out, int P, int N, float eps, void* stream);
/* every (pair, decoder layer) of a step in one launch, reading the shared row space (nerf_regtr.py:226-236 without the cats):
 * xyz [R,3], corr [L,R,3], ov [L,R], probs int32 [P][4] = (s0, ns, t0, nt) -> out [P,L,3,4] */
int dreg_weighted_kabsch_pairs(const float* xyz, const float* corr, const float* ov, const int* probs, float* out, int P, int L,
                               int R, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------- training losses
 * train_nerf_regtr.py:186-229 for all P pairs of a step (rows = every pair's (src | tgt) key points, probs as above):
 * overlap BCE (input = labels, target = prediction: the reference's argument order), label smooth-L1, correspondence L1
 * (conerf/loss/correspondence_loss.py:16-51, incl. its [nl,N,1] x [N] broadcast) and InfoNCE (conerf/loss/feature_loss.py:
 * 24-73) on precomputed fp32 logits.  d_ov / d_corr / the in-place logits gradient are those of the mean-over-pairs total. */
int dreg_reg_point_losses(const float* gt, const float* tilde, const float* ov, const float* corr, const float* xyz,
                          const float* pose, const int* probs, float* partial, float* d_ov, float* d_corr, int L, int R, int P,
                          int robust, float eps, float w_overlap, float w_corr, void* stream);
/* InfoNCE's matrix products (feature_loss.py:43-47: logits = f_a (triu(W) + triu(W)^T) f_p^T per pair; and their gradients) replace torch.mm ->
 * rocBLAS: batched exact-fp32 MFMA GEMMs over a descriptor table, C[M x N] = op(A)[M x K] . op(B)[K x N] row-major (tA = 1: A stored [K][M]; tB = 1: B
 * stored [N][K]).  descs_dev: n records of dreg_gemm_f32_desc_bytes() = 72 bytes
 *   { int a_id, b_id, c_id, tA, tB, M, N, K, lda, ldb, ldc, tile0; int64 a_off, b_off, c_off }     (offsets in elements)
 * tile0 = 64 x 64 output tiles of the records before, total_tiles = of all; bases8: 8 device pointers indexed by the ids (host array).
 * A pair's logits block is [ns][ld] with ld = nt rounded up to a multiple of 4 (dreg_infonce_rows reads / writes it with that stride).
 * dreg_infonce_wsym: out[E][E] = triu(W) + triu(W)^T. */
int dreg_gemm_f32_desc_bytes(void);
int dreg_gemm_f32_batched(const void* descs_dev, int n, int total_tiles, const float* const* bases8, void* stream);
int dreg_infonce_wsym(const float* W, float* out, int E, void* stream);
int dreg_infonce_nn(const float* xyz, const float* pose, const int* probs, const int* src_off, int* nn, float* mask,
                    float* count, int P, int total_src, float r_p, void* stream);
int dreg_infonce_rows(float* logits, const float* xyz, const float* pose, const int* probs, const int* src_off,
                      const long long* logit_off, const int* nn, const float* mask, const float* count, float* loss_row, int P,
                      int total_src, float r_n, float scale, int write_grad, void* stream);
int dreg_reg_losses_final(const float* partial, const float* loss_row, const float* count, const int* probs, const int* src_off,
                          float* out, int P, int L, float eps, float w_overlap, float w_cont, float w_feat, float w_corr,
                          void* stream);
/* dreg_reg_point_losses accepts tilde = NULL: the label-consistency term ('nerf_cont', train_nerf_regtr.py:198-201 — no gradient) is
 * then left out (partial[.][1] = 0, out[1] = 0, out[4] without it) and completed by this call once the 'tilde' labels exist (the training step marches them
 * on a side stream under backward): partial[.][1], out[1] and out[4] become bit-identical to the one-call form's. */
int dreg_nerf_cont_deferred(const float* gt, const float* tilde, const int* probs, float* partial, float* out, int P, int L, int R,
                            float w_overlap, float w_cont, float w_feat, float w_corr, void* stream);

/* Stand-in {0,1} visibility labels of data without NeRF blocks on disk (synthetic scenes; the reference always ray-marches them: train_nerf_regtr.py:186-201):
 * gt[l, r] = 1[x + 0.31 y - 0.17 z > 0.0123] of key point xyz[r] (fp32 [R,3]) for every layer l < L, tilde[l, r] the same test of corr[l, r] (fp32 [L,R,3]);
 * gt, tilde fp32 [L,R].  One launch for the dozen element-wise ones of the torch formula (dreg_nerf_amd/synth.py), same bits. */
int dreg_halfspace_labels(const float* xyz, const float* corr, float* gt, float* tilde, int L, int R, void* stream);

/* batched_grid_subsample (grid_downsample.py:6-44; MinkowskiEngine UNWEIGHTED_AVERAGE): mean of (xyz | feat) over rows
 * sharing (batch, floor(p/dl)); rows out ordered by (batch, ix, iy, iz).  Outputs sized for N rows; n_out / batch_counts
 * are device ints; inv_seg / inv_cnt (per input row) feed the backward pass; err is set when |p/dl| >= 32768. */
size_t dreg_voxel_downsample_workspace_bytes(int N);
int dreg_voxel_downsample_fwd(const float* pts, const float* feats, const int* pt_batch, float* out_pts, float* out_feats,
                              int* n_out, int* batch_counts, uint32_t* inv_seg, float* inv_cnt, int* err,
                              void* workspace, size_t workspace_bytes, int N, int C, int nbatch, float dl, void* stream);
/* The same downsample split in two, so that every data-dependent size can be resolved before the feature network runs:
 * plan (xyz only: cells, sort, segments, averaged points; order uint32 [N] and starts uint32 [N+1] are kept by the caller),
 * then segment means of a feature matrix over that plan (M = the plan's n_out). */
int dreg_voxel_downsample_plan(const float* pts, const int* pt_batch, float* out_pts, int* n_out, int* batch_counts,
                               uint32_t* inv_seg, float* inv_cnt, int* err, uint32_t* order, uint32_t* starts,
                               void* workspace, size_t workspace_bytes, int N, int nbatch, float dl, void* stream);
/* the same with pass-through flags per batch id (frozen uint8 [nbatch] on the device, NULL = none): the rows of a frozen batch come out unchanged and in
 * order (each row a cell of its own).  One set of launches then serves every pair of a step although a pair whose point count has dropped to <= 3,000
 * takes no further round (grid_downsample.py:83-94). */
int dreg_voxel_downsample_plan_frozen(const float* pts, const int* pt_batch, float* out_pts, int* n_out, int* batch_counts,
                                      uint32_t* inv_seg, float* inv_cnt, int* err, uint32_t* order, uint32_t* starts,
                                      void* workspace, size_t workspace_bytes, int N, int nbatch, float dl, const uint8_t* frozen, void* stream);
int dreg_voxel_segment_mean(const float* feats, const uint32_t* order, const uint32_t* starts, const int* n_out,
                            float* out_feats, int M, int C, void* stream);
int dreg_voxel_downsample_bwd(const float* gout, const uint32_t* inv_seg, const float* inv_cnt, float* gin, int N, int C,
                              void* stream);

/* clip_grad_norm_ + torch.optim.AdamW on flat fp32 buffers (train_nerf_regtr.py:96-102,232-237). */
int dreg_grad_norm(const float* g, float* norm_out, float* workspace, size_t n, void* stream);
int dreg_adamw_step(float* p, float* g, float* m, float* v, const float* norm, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float max_norm, void* stream);

/* ---------------------------------------------------------------------------------------------- NGP dense query
 * Replaces tiny-cuda-nn behind NGPradianceField (conerf/radiance_fields/ngp.py:92-146): HashGrid(L=16,F=2,T=2^19,
 * Nmin=16,b=1.4472692) + FullyFusedMLP 32->64->16 for density (ngp.py:148-176) and SH4 + FullyFusedMLP 32->64->64->16
 * (sigmoid) for colour (ngp.py:178-193), as used by SampleGrid.query_radiance_and_density_from_camera
 * (conerf/register/sample_grid.py:223-242,321-341) and Evaluator.sample_points (eval_ngp_nerf.py:383-405).
 * Level arrays and aabb are HOST pointers (16 entries each / 6 floats); everything else is device memory. */
uint32_t dreg_ngp_level_table(float per_level_scale, int log2_hashmap_size, int base_resolution,
                              uint32_t* offset, uint32_t* size, uint32_t* res, float* scale, uint32_t* hashed);
int dreg_f32_to_f16(const float* in, void* out, size_t n, void* stream);
/* x fp32 [Np,3] world -> density fp32 [Np] (= exp(h0-1) * inside-aabb), raw fp16 [Np,16] (h0 | 15 geometry features) */
int dreg_ngp_density_fwd(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                         const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale,
                         const uint32_t* hashed, const float* aabb, int Np, void* stream);
/* The per-block glue of the dense query as kernels: the direction half of the colour net's first layer for K shared directions
 * (out: 2 * K * 64 words — fp32 [K][64] biases, then uint32 [K][64] = the same values as two packed fp16 halves hi | lo << 16, the MFMA
 * operand of the colour kernel; dreg_ngp_rgb_mean_fwd takes the pointer to this buffer as `dirbias`), and alpha / density mask
 * (sample_grid.py:338-341: alpha = clip(1 - exp(-delta * density), 0, 1), keep = density > threshold). */
int dreg_ngp_dir_bias(const float* dirs, const void* w1_f16, float* out, int K, void* stream);
int dreg_ngp_alpha_keep(const float* density, float* alpha, uint8_t* keep, int N, float delta, float threshold, void* stream);
/* mean over ndir fixed viewing directions of the colour net.  dirbias = the buffer dreg_ngp_dir_bias wrote: 2 * ndir * 64 words —
 * fp32 [ndir][64] = W1[:, :16] . sh4(dir_k), FOLLOWED BY uint32 [ndir][64] (the same values as packed fp16 halves hi | lo << 16).
 * The kernel reads the second half at dirbias + ndir * 64: a caller that builds its own bias table must supply both halves. */
int dreg_ngp_rgb_mean_fwd(const void* raw, const void* w1, const void* w2, const void* w3, const float* dirbias, float* rgb,
                          int ndir, int Np, void* stream);
/* unbounded scenes: positions pass through contract_to_unisphere (conerf/radiance_fields/ngp.py:41-63) before the hash grid */
int dreg_ngp_density_fwd_contract(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                                  const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                  const float* aabb, int Np, int contract, void* stream);
/* the same with a caller-owned scratch buffer (dreg_ngp_density_workspace_bytes(Np) bytes, fp16 [16][Np][2] level features): the hash-grid
 * encoding (tcnn HashGrid of ngp.py:92-110) runs as its own launch in which every XCD looks up only two of the 16 levels — their tables
 * stay in that XCD's 4 MB L2 instead of streaming 128-byte lines from the Infinity Cache for 4-byte corners — and the density MLP reads
 * the features.  Same outputs, bit for bit.  workspace null / too small: the fused kernel of dreg_ngp_density_fwd_contract. */
size_t dreg_ngp_density_workspace_bytes(int Np);
int dreg_ngp_density_fwd_ws(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                            const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                            const float* aabb, int Np, int contract, void* workspace, size_t workspace_bytes, const int* order, int x_in_slot_order, void* stream);
/* order (optional; null = the points as given): a permutation (device int32 [Np]); slot j of the launch works on point order[j], outputs
 * stay at the point's own index.  Consecutive points of a block's query (sample_grid.py:223-231: ascending flat indices, z fastest) differ
 * along the SLOWEST axis of the hash grid's tables, so a wave's 64 corner reads hit 64 cache lines; dreg_grid_x_order builds the order in
 * which x runs fastest from the occupancy volume (byte [rx][ry][rz]) and the ascending index list. */
/* x_in_slot_order = 1 (with an order): x[j] already IS the position of point order[j] (dreg_grid_sample_points_ordered writes that array
 * next to the ordinary one), so the coordinates are read contiguously; outputs still go to the point's own index. */
int dreg_grid_sample_points_ordered(const int64_t* idx, const float* jitter, const int* order, float* world, float* world_slot,
                                    int rx, int ry, int rz, const float* aabb, int Np, void* stream);
size_t dreg_grid_x_order_workspace_bytes(int rx, int ry, int rz);
int dreg_grid_x_order(const uint8_t* binary, const int64_t* idx, int* order, void* workspace, size_t workspace_bytes, int rx, int ry, int rz, int Np, void* stream);
/* The dense query's cell lists without torch.nonzero (conerf/register/sample_grid.py:223-242: nonzero + jittered positions), plus the
 * x-fastest lane order, in four launches and ONE host readback:
 *   dreg_grid_occupied_count   x-prefix counts per (y, z) column, occupied cells per (x, y) row, both scanned; the int32 at
 *                              dreg_grid_occupied_totals(workspace)[0] then holds N = number of occupied cells (the caller reads it);
 *   dreg_grid_occupied_build   indices int64 [N] ascending, order int32 [N] (dreg_grid_x_order's), world / world_slot fp32 [N,3]
 *                              (dreg_grid_sample_points_ordered's arithmetic; jitter fp32 [N,3] indexed like indices);
 *   dreg_ngp_density_keep_fwd_ws = dreg_ngp_density_fwd_ws + alpha / keep (dreg_ngp_alpha_keep's arithmetic) from the same launch;
 *   dreg_grid_write_kept       voxel_mask int64 (capacity N, ascending kept indices) and voxel_grid[idx] = (xyz, rgb, alpha) of the kept
 *                              points (grid zeroed by the caller) in three launches (kept points per row, scan, write); totals[1] = number
 *                              of kept cells.
 * workspace: dreg_grid_occupied_workspace_bytes(rx, ry, rz) bytes, owned by the caller from _count to _write_kept. */
size_t dreg_grid_occupied_workspace_bytes(int rx, int ry, int rz);
void* dreg_grid_occupied_totals(void* workspace, int rx, int ry, int rz);
int dreg_grid_occupied_count(const uint8_t* binary, void* workspace, size_t workspace_bytes, int rx, int ry, int rz, void* stream);
int dreg_grid_occupied_build(const uint8_t* binary, void* workspace, const float* jitter, const float* aabb, int64_t* indices, int* order,
                             float* world, float* world_slot, int rx, int ry, int rz, int N, void* stream);
int dreg_ngp_density_keep_fwd_ws(const float* x, const void* table, const void* w1, const void* w2, float* density, void* raw,
                                 const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                 const float* aabb, int Np, int contract, void* workspace, size_t workspace_bytes, const int* order, int x_in_slot_order,
                                 float* alpha, uint8_t* keep, float delta, float threshold, void* stream);
int dreg_grid_write_kept(void* occ_workspace, const float* xyz, const float* rgb, const float* alpha, const int64_t* indices, const uint8_t* keep,
                         int64_t* mask, float* grid, int rx, int ry, int rz, int Np, void* stream);
/* one viewing direction per point: NGPradianceField.query_rgb(dir, embedding) / forward (conerf/radiance_fields/ngp.py:178-208) */
int dreg_ngp_rgb_dir_fwd(const void* raw, const void* w1, const void* w2, const void* w3, const float* dirs, float* rgb, int Np, void* stream);
/* jittered sample of every occupied cell mapped to world space (sample_grid.py:226-242, AABB contraction) */
int dreg_grid_sample_points(const int64_t* idx, const float* jitter, float* world, int rx, int ry, int rz, const float* aabb,
                            int Np, void* stream);
/* grid[idx[n]] = (xyz, rgb, alpha) for keep[n] != 0   (eval_ngp_nerf.py:397-405) */
int dreg_grid_scatter7(const float* xyz, const float* rgb, const float* alpha, const int64_t* idx, const uint8_t* keep,
                       float* grid, int Np, void* stream);

/* ---------------------------------------------------------------------------------------------- surface-field visibility (N1)
 * One fused kernel for nerfacc.ray_aabb_intersect + _C.ray_marching + tcnn density + transmittance scan + scatter_max + threshold
 * + max over cameras (conerf/utils/nerfacc_utils.py:84-222; conerf/loss/confidence_loss.py:56-160; conerf/register/sample_grid.py:244-318):
 * label[p] |= (max over samples of alpha*T along the ray camera c -> point p, t in [t_min(scene aabb), |p - o_c|)) >= cut_off.
 * cams fp32 [Nc,3], pts fp32 [Np,3], binary uint8 [rx,ry,rz] (occupancy grid), label int32 [Np] zeroed by the caller;
 * table/w1/w2 = fp16 copies of mlp_base.params; level arrays and the three aabbs (grid roi, scene, model) are HOST pointers. */
int dreg_surface_visibility(const float* cams, const float* pts, const uint8_t* binary, int* label,
                            const void* table, const void* w1, const void* w2,
                            const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale,
                            const uint32_t* hashed, const float* roi_aabb, const float* scene_aabb, const float* model_aabb,
                            int rx, int ry, int rz, int Nc, int Np, float render_step_size, float cut_off, float early_stop_eps,
                            float alpha_thre, void* stream);
/* the same labels from a persistent launch: a lane whose ray is decided takes the next ray from a queue, and rays of points that already
 * carry the label are skipped (the label is an OR over the cameras).  queue: 8 bytes of device memory, zeroed by the caller on `stream`. */
int dreg_surface_visibility_queue(const float* cams, const float* pts, const uint8_t* binary, int* label,
                                  const void* table, const void* w1, const void* w2,
                                  const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                  const float* roi_aabb, const float* scene_aabb, const float* model_aabb,
                                  int rx, int ry, int rz, int Nc, int Np, float render_step_size, float cut_off, float early_stop_eps,
                                  float alpha_thre, void* queue, const uint32_t* coarse_bits, void* stream);
/* coarse_bits (optional): one bit per 4^3 block of `binary` (dreg_occupancy_coarse_bits; the buffer zeroed first): empty space is then
 * walked in coarse cells looked up in LDS instead of fine cells looked up in global memory.  Same labels. */
int dreg_occupancy_coarse_bits(const uint8_t* binary, uint32_t* bits, int rx, int ry, int rz, void* stream);
/* Several blocks in one launch (a training step's labels: 8 blocks).  The caller owns a table of n records of
 * dreg_surface_visibility_desc_bytes() bytes: each filled on the HOST by dreg_surface_visibility_fill_desc (the arguments of
 * dreg_surface_visibility_queue for one block), the table copied to the device by the caller; every wave of the one persistent launch then
 * works through all blocks' ray queues (total_rays = sum of Nc * Np), so the blocks' tails overlap.  Same labels. */
size_t dreg_surface_visibility_desc_bytes(void);
int dreg_surface_visibility_fill_desc(void* host_desc, const float* cams, const float* pts, const uint8_t* binary, int* label,
                                      const void* table, const void* w1, const void* w2,
                                      const uint32_t* offset, const uint32_t* size, const uint32_t* res, const float* scale, const uint32_t* hashed,
                                      const float* roi_aabb, const float* scene_aabb, const float* model_aabb,
                                      int rx, int ry, int rz, int Nc, int Np, float render_step_size, float cut_off, float early_stop_eps,
                                      float alpha_thre, void* queue, const uint32_t* coarse_bits);
int dreg_surface_visibility_multi(const void* descs_dev, int n, long total_rays, void* stream);
/* the same launch with at most max_waves one-wave workgroups (0 = default 4096): a background launch next to LDS-hungry kernels on another stream
 * (each workgroup holds 18.7 KB of LDS while the launch runs); same labels, proportionally longer */
int dreg_surface_visibility_multi_waves(const void* descs_dev, int n, long total_rays, int max_waves, void* stream);

/* ---------------------------------------------------------------------------------------------- active-set 3^3 convolution with
 * staged-neighbourhood reuse (csrc/conv_brick.hip): the FPN head layers upsample_transform_{1,2} / pyramid_transformation_1 and their
 * data gradients on the voxels around the occupied surface (conerf/model/feature_pyramid_net.py:47-56,97-103; the reference runs
 * cuDNN conv3d over the whole volume).  Same results as dreg_conv3d_igemm_rows up to the order of the fp32 sums.
 * dreg_brick_tiles_build: flags uint8 [B,D,H,W] (1 = row is computed) -> rows in brick-major order + tile tables:
 *   meta int32 [4] = (rows, candidate tiles, tiles emitted, overflow flag); rows_sorted int32 [max_rows]; tiles [max_tiles] x 16 B
 *   { row0, nrows, nhalo, pad }; halo_vox int32 [max_tiles][1280] (voxel staged in each LDS slot); nbr uint16 [max_tiles][256][28]
 *   (LDS byte offset of every (row, tap) neighbour).  The caller reads meta[2] (= ntiles) / meta[3] on the host before it launches.
 * dreg_conv3_brick: out[rows] = bias + addend + conv3(in) at the tiled rows; Cout in {256, 64}; Cin % 16 == 0; wpk from
 * dreg_pack_conv_weight_brick (transposed = 1: the data-gradient pack, rows = Cin of the layer). */
int dreg_brick_supported(int B, int D, int H, int W, int Cin, int Cout);
size_t dreg_conv3_brick_pack_bytes(int rows, int red);
int dreg_pack_conv_weight_brick(const float* w, void* out, int Cout, int Cin, int transposed, void* stream);
size_t dreg_brick_tiles_workspace_bytes(int B, int D, int H, int W);
int dreg_brick_tiles_build(const uint8_t* flags, int B, int D, int H, int W, int max_rows, int max_tiles, void* workspace, size_t workspace_bytes,
                           int* meta, int* rows_sorted, void* tiles, int* halo_vox, void* nbr, void* stream);
int dreg_conv3_brick(const void* in, const void* wpk, void* out, const float* bias, const void* addend, const void* tiles, int ntiles,
                     const int* halo_vox, const void* nbr, const int* rows_sorted,
                     int B, int D, int H, int W, int Cin, int Cout, int Da, int Ha, int Wa, int add_same, int out_f32, void* stream);

/* ---------------------------------------------------------------------------------------------- native point-set executor
 * (csrc/pointset_exec.hip) The six encoder layers (conerf/register/transformer.py:50-86,225-299), the shared final norm
 * (nerf_regtr.py:170-206), the correspondence decoder and the overlap head (nerf_regtr.py:273-308,350-394,384-387): forward in one
 * call, backward in one call, for the key points of every pair of a step (R rows; `probs` = attention problem tables as in the
 * varlen attention calls).  params: int64 [dreg_ps_num_params()][2] = (fp32 value ptr, fp32 grad ptr or 0) in the order: per layer
 * norm1.{weight,bias}, self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}, norm2.*, cross_attn.{same four},
 * norm3.*, linear1.*, linear2.*; then transformer_encoder.norm.*, correspondence_decoder.{q_proj,k_proj,conf_logits_decoder}.*.
 * packs: int64 [dreg_ps_num_linears()][2] = (forward pack, data-gradient pack) device pointers of every linear layer (per layer
 * in_proj_s, out_proj_s, in_proj_c, out_proj_c, linear1, linear2; then q_proj, k_proj), bf16 from dreg_pack_conv_weight.
 * The arena (dreg_ps_arena_bytes(R), caller-owned) keeps the forward pass's activations for the backward call of the SAME R.
 * dreg_ps_backward: g_cond (fp32 [6,R,256]), g_corr, g_ov may each be null; parameter
 * gradients are accumulated into the grad pointers; weight / bias gradients run on aux_stream (optional) and the caller joins it with
 * `stream` before reading any gradient.  last_only = 1: g_cond [R,256] / g_corr [R,3] / g_ov [R] are the gradients of the LAST layer's
 * outputs only (what the training losses read: train_nerf_regtr.py:178,195,205-206,214,220); heads, decoder and final norm are then
 * differentiated for that layer's R rows instead of 6R.  dreg_ps_set_fuse(0): the arithmetic of the per-op path, bit for bit (tests). */
int dreg_ps_num_params(void);
int dreg_ps_num_linears(void);
void* dreg_ps_create(const int64_t* params);
void dreg_ps_destroy(void* h);
void dreg_ps_set_fuse(void* h, int fuse);
void dreg_ps_set_group_wgrad(void* h, int on);   /* 1 (default, with fuse): the split partials of all linear layers' weight gradients by one launch per tile shape at the end of the backward pass (dreg_wgrad_group_launch); 0: one launch per layer.  Bit-identical.  Per handle. */
void dreg_ps_set_timing(void* h, int enable);            /* HIP events around every linear-layer launch (forward, data gradient, weight gradient) */
int dreg_ps_read_timings(void* h, int* info, float* ms, int cap);   /* after a device sync: 5 ints per record (kind, rows, cin, cout, flags) + ms; returns the count */
size_t dreg_ps_arena_bytes(void* h, int R);
int dreg_ps_forward(void* h, void* arena, size_t arena_bytes, const int64_t* packs, const float* feats, const float* xyz, const float* pe,
                    const int* probs_self, const int* probs_cross, int nprob, int max_len, int R,
                    float* cond, float* corr, float* ov, void* stream);
int dreg_ps_backward(void* h, void* arena, size_t arena_bytes, const int64_t* packs, const float* feats, const float* xyz, const float* pe,
                     const int* probs_self, const int* probs_cross, int nprob, int max_len, int R,
                     const float* cond, const float* corr, const float* ov, const float* g_cond, const float* g_corr, const float* g_ov,
                     float* d_feats, void* stream, void* aux_stream, int last_only);

#ifdef __cplusplus
}
#endif
#endif /* DREG_NERF_H */
