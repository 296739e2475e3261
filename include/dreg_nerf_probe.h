/* Kernel-variant switches of libdreg_nerf_hip_probe.so — the MEASUREMENT build of the library, NOT the product.
 *
 * The product library (libdreg_nerf_hip.so, include/dreg_nerf.h) has no process-global mutable state: every knob below is a compile-time
 * constant there (csrc/common.h DREG_KNOB) and none of these symbols exists in it.  The same sources compiled with -DDREG_PROBE give
 * libdreg_nerf_hip_probe.so = every entry point of dreg_nerf.h + the process-global setters below.  It is loaded EXPLICITLY
 * (dreg_nerf_amd.lib.probe()) by the A/B tools under tools/ and by the tests that force a kernel variant on a small shape; with every knob at
 * its default it runs the product's code.  Variants differ in speed only, except those marked "wrong results" (timing ablations). */
#ifndef DREG_NERF_PROBE_H
#define DREG_NERF_PROBE_H
#ifdef __cplusplus
extern "C" {
#endif

/* experiments (tools/bench_conv_halo.py): 0 = anti-phase wave groups, weights 2 units ahead (default); 1 = lockstep; 2 = 3 units ahead */
void dreg_conv3_halo_set_variant(int variant);
void dreg_conv3_halo64_set(int v);   /* 64-output-channel halo kernel: 1 = on (default), 0 = off (implicit GEMM serves the shape), 2 = the other loop form (A-fragment prefetch on / off) */
void dreg_conv3_halo_set_prof(void* u64_buf_64x8x5);   /* variant 5: per-wave shader-clock breakdown of the first 64 workgroups */
/* 1 (default): bf16 stride-1 convolutions stage operands with buffer_load...lds; 0: register-staged kernel (A/B checks) */
void dreg_conv_set_glds(int enable);
/* MEASUREMENT (tools/bench_bn_conv_fuse.py, round-5 review item 1): 1^3 convolution whose A load applies a BatchNorm + ReLU, a = relu(x * scale[b,ci] + shift[b,ci]) (register-staged kernel) */
int dreg_conv1_bnrelu_a_probe(const void* in, const void* wt_packed, void* out, const float* a_scale_shift, int B, int D, int H, int W, int Cin, int Cout, void* stream);
/* tuning / test knob: force the number of voxel splits of the weight-gradient kernels (0 = automatic) */
void dreg_conv_set_wgrad_splits(int splits);
void dreg_conv_set_wgrad_rows_fast(int enable);      /* 1 (default): row-list weight gradients keep packed voxel coordinates in LDS (no decode per load) and take the 8-wave 256 x 256 tile for 256 -> 256 layers */
void dreg_conv_set_row_splits(int on);               /* 1 (default): a row-list weight gradient splits its rows by the LIST's length (256 x 256 tile: fullest last round of 256 workgroups at >= 2,048 rows per split); 0: by the dense volume's rule */
void dreg_conv_set_wgrad_ring(int mode);             /* dense 8-wave weight-gradient tile: 3 (default) anti-phase wave groups over a ring of four 32-voxel units (lean load half for stride-1 same-volume layers, general loop otherwise), 8 the general anti-phase loop, 0 lockstep over two 64-voxel stages, 1 / 2 lockstep over four / five 32-voxel stages (measured no gain), 4 = 3 with s_memtime stamps, 5..7 stamped ablations (wrong results) */
int dreg_conv_wgrad_probe_read(unsigned long long* out8); /* ring mode 4 (measurement only): { issue, fragment reads, wait for loads, barrier, MFMAs, barrier cycles; units x waves; waves } since the last read */
void dreg_conv_set_wgrad_pipe(int enable);           /* 1: the dense 8-wave weight-gradient tile reads the fragments of the next MFMA group while the current group runs (default 0: measured no gain) */
void dreg_conv_set_igemm_ap(int max_tiles);          /* bf16 implicit-GEMM launches of at most this many 128-row tiles run the eight-wave anti-phase form of the tile (default 256 = one workgroup per CU; 0: never); bit-identical results */
void dreg_conv_set_pointwise_rmw_cin(int max_cin);  /* 1^3 convolutions with an addend and <= max_cin input channels (default 128) take the 128-row tile on large launches too (read-modify-write passes: HBM-bound, the addend prefetched four rows ahead); 0: the 256 x 256 tile */
void dreg_conv_set_igemm_ap256(int on);              /* 1 (default): the 256 x 256 implicit-GEMM tile of large launches runs as two anti-phase wave groups over a ring of four 32-channel stages; 0: lockstep over two 64-channel stages; bit-identical results */
void dreg_conv_igemm_probe(int enable);               /* MEASUREMENT ONLY: bf16 launches with Cout % 128 == 0 run the 128 x 128 implicit-GEMM kernel with s_memtime stamps around the phases of a K step */
int dreg_conv_igemm_probe_read(unsigned long long* out6); /* { wait-for-loads, barrier, issue, compute cycles; K steps x waves; waves }, summed over the waves since the last read */
void dreg_conv_set_wgrad_big(int mode);             /* large dense layers: 3 (default) the 8-wave 256 x 256 tile, 1 four waves on 256 x 128 with 32-voxel stages, 11-13 ablations of the 8-wave tile, 0 neither */
void dreg_conv_set_glds_stages(int stages);          /* LDS pipeline stages of the direct-to-LDS convolution: 0 = default (2), 2..4 forces; results do not depend on it */
void dreg_conv_set_wgrad_target_blocks(int blocks);   /* workgroups the automatic split choice aims for (default 384; 3072 until round 6) */
/* largest per-grid volume (voxels) whose BatchNorm runs the fused statistics+apply kernels (default 512 = the 8^3 level; 16^3 measured slower fused); 0 = never */
void dreg_bn_set_small_max_voxels(int v);
void dreg_sstem_set_pool_blocks(int n);              /* measurement: workgroups of the sparse stem's pooling launch (default 8192 = one pooled granule per thread at 8 x 32^3 x 64) */
void dreg_voxel_set_own_sort(int on);                 /* experiment, default 0: 1 = the voxel keys of a downsample round (<= 131,072) are built, sorted (stable 4-bit LSD radix sort, ballot ranks) and segmented by ONE launch of one workgroup instead of rocPRIM radix_sort_pairs + inclusive_scan + three small kernels; identical results, but 0.45-1.45 ms per round on its single CU against ~0.07 ms (so the product path keeps rocPRIM) */
void dreg_conv_set_bn_stats_epilogue(int on);        /* 0: dreg_conv3d_igemm_bnstats never emits sums (callers fall back to the BatchNorm's own statistics pass) */
void dreg_bn_set_store_g(int enable);                /* 1 (default): the backward of a residual + ReLU BatchNorm stores the masked gradient (= the residual gradient) in its statistics pass; the apply pass reads it instead of dy and y */
void dreg_bn_set_small_regs(int enable);             /* 1 (default): the one-launch BatchNorms of the 8^3 / 4^3 volumes load their rows once and keep them in registers between statistics and apply */
void dreg_bn_set_debug_skip(int mask);                /* MEASUREMENT ONLY (wrong results): bit 0 / 1 leave out the forward / backward statistics pass of the large BatchNorms */
void dreg_ngp_set_rgb_chunks(int on);                 /* 1 (default): shared-direction colour queries run the persistent 16-point-chunk kernel; 0: the 64-point-per-wave kernel */
void dreg_ngp_set_density_unroll(int n);             /* hash-grid levels of the density kernel whose 8 corner gathers are issued together: 1, 2, 4 or 8 */
void dreg_ngp_set_xcd_levels(int on);                /* 1 (default): dreg_ngp_density_fwd_ws encodes with XCD-resident level pairs (two launches); 0: the fused kernel */
int dreg_visibility_set_pass_bound(long passes);   /* test hook: passes of the persistent visibility kernels' march loop per wave (0 = default 2^22); a launch that reaches it sets bit 63 of its ray counter words */
void dreg_visibility_set_waves(int n);               /* one-wave workgroups of dreg_surface_visibility_queue's persistent launch (default 4096 = 256 CUs x 16) */
void dreg_conv_set_narrow_small(int on);              /* 1 (default): launches of < 224 128 x 128 tiles use 128 x 64 tiles (twice the workgroups) */

#ifdef __cplusplus
}
#endif
#endif
