/*
 * isegmi.h -- C ABI of libisegmi.so: the MI355X (gfx950) RoI/mask inference hot path of
 * detectron.jittor's Mask R-CNN and Yolact.jittor.
 *
 * Drop-in boundary.  The reference defines NO native / FFI / plugin interface for this path:
 * /root/reference holds only README.md (the submodules are empty) and the
 * only boundary it shows is the Python call surface
 *     COCODemo(cfg, min_image_size=800, confidence_threshold=0.5)      README.md:320-324
 *     coco_demo.run_on_opencv_image(image)                             README.md:331
 *     python eval.py --trained_model=... --score_threshold=... --top_k=...  README.md:243-249
 *     python tools/test_net.py --config-file ...                       README.md:344-347
 * Each entry point below names the reference operator it stands in for (SURVEY.md section 8a
 * ids M1..M13 / Y1..Y8 and Appendix A item) and the README line that reaches it.  The Python
 * package `isegmi` binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative code on failure, message via isegmi_last_error() (thread-local).  Pointers named
 * d_* are DEVICE pointers obtained from isegmi_malloc; h_* are host pointers.  All activations
 * are fp32 NHWC.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * A handle is single-stream and not thread-safe: one handle per GPU per process.
 */
#ifndef ISEGMI_H
#define ISEGMI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define ISEGMI_ABI_VERSION 1

int isegmi_version(void);
const char* isegmi_last_error(void);

/* ---- device plumbing (replaces `jt.flags.use_cuda = 1`, README.md:311) ---- */
int isegmi_device_count(int* n);
int isegmi_set_device(int device_id);
int isegmi_malloc(void** d_ptr, int64_t bytes);
int isegmi_free(void* d_ptr);
int isegmi_malloc_host(void** h_ptr, int64_t bytes);   /* pinned host memory (source of isegmi_engine_upload_async) */
int isegmi_free_host(void* h_ptr);
int isegmi_h2d(void* d_dst, const void* h_src, int64_t bytes);
int isegmi_d2h(void* h_dst, const void* d_src, int64_t bytes);
int isegmi_memset(void* d_ptr, int value, int64_t bytes);
int isegmi_sync(void);
/* Box calibration (diagnostic, not on the product path): what this device sustains on three bare loops of about ms_per_leg milliseconds each --
 * dependent v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_f16 chains on random register operands (four waves per SIMD, no memory traffic) and a
 * float4 copy of 1 GiB (read + written bytes).  bench.py prints them as "box" so that a slow box can be told from a slow build (boxes of one pool
 * differ by 3-12 %: MI355X_MICROARCH.md "DVFS give-back" 5). */
int isegmi_box_calibrate(double ms_per_leg, double* mfma_f32_tflops, double* mfma_f16_tflops, double* hbm_copy_gbs);

/* ---- convolution family: M2 M3 M4 M8 M10 M11 Y2 Y3 Y4 Y5 ----
 * Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 with fused per-channel affine
 * (folded BN or bias), residual add and activation.  Per output element the accumulation is
 * ONE fmaf chain from +0 over the R*S*Cin products, walked in groups of 128 input channels
 * (outermost), then (r, s), then the group's channels -- plain (r, s, cin) for Cin <= 128 and for
 * 1x1 convolutions -- bit-identical to the oracle (oracle/ora_ops.c, ora_conv2d). */
typedef struct isegmi_conv_desc {
    int32_t N, H, W, Cin;            /* input NHWC; Cin % 32 == 0, or Cin == 4 with R==S==7 (stem) */
    int32_t Cout, R, S, stride, pad;
    int32_t act;                     /* 0 none, 1 relu, 2 tanh, 3 LeakyReLU(0.1), 4 LeakyReLU(0.1) then + residual (DarkNet block); fp32 only for 2-4 */
    int32_t tile;                    /* 0 auto; fp32: 1: 128x128  2: 128x64  3: 64x64 (round-1 schedule; the stem)  4: 32x32 on 16x16x4 MFMA  5: 32x32 with four loader waves, software-pipelined  6: 32x64, two 16x16 tiles per wave (block tile MxN)
                                        10 / 12: 64x64, v2 schedule, loads 2 / 4 chunks ahead (the default for large grids)  7 / 9: the same with the early LDS store (measured slower; kept for A/B)
                                        13 / 14: hybrid launch -- v2 tiles (loads 2 / 4 ahead) on the rows that fill the CUs a whole number of times, 32x32 blocks on the left-over rows (the default for 513-2600-tile grids with a small left-over);
                                        15 (fp32, NEVER chosen by tile 0): FIXED-TREE SPLIT-K -- a 32x32 block whose K chunks are cut over four sets of waves, out = ((p0 + p1) + p2) + p3 of four
                                        k-ordered partial chains over the chunk ranges [q L, (q + 1) L), L = ceil(K / 32 / 4): an opt-in numerics mode (bit-exact against the oracle's
                                        ora_conv2d_split, NOT against isegmi_op_conv2d's other tiles) for the small-M, large-K layers of a bs = 1 forward; the engines' `conv_split_k`;
                                        fp16: 1: 256x256  2: 256x128  3: 128x128  4: 64x64  5: 64x128  6: 64x256  7: 128x256  8: 128x64  9: 192x256  10: 192x128  11: 160x256;
                                        12/13/14/16: 192x256, 256x256, 256x128, 160x256 with 4 loader waves, 17: 192x256 as 12 MFMA + 4 loader waves, 19: 128x256 + 4, 20: 192x128 as 6 + 2;
                                        26/27/28/29: row-strip kernel for 3x3/1/1 (192x256, 256x128, 160x256, 192x256 on 12 MFMA waves) with 4 loader waves; 30/31: 29/26 with three B chunk buffers;
                                        32/34/37/39: persistent forms of 12/14/17/19 (a block walks several tiles, the loader waves stream the next tile during the epilogue);
                                        + 2048 (test hook): persistent kernels on an 8-block grid */
    int32_t out_div;                 /* output pixels per "image" for addressing; 0 -> Ho*Wo */
    int64_t out_img_stride;          /* floats; 0 -> out_div*out_pix_stride */
    int64_t out_pix_stride;          /* floats; 0 -> Cout */
} isegmi_conv_desc;

int isegmi_conv_out_hw(const isegmi_conv_desc* d, int32_t* Ho, int32_t* Wo);
/* 1 when the split-K mode's shape rule takes this fp32 layer (at most 176 tiles of 64 x 64 and K >= 1024; not the stem), else 0.  The engines apply it, under
 * `conv_split_k`, to the bottleneck convolutions of the backbone except a stage's first conv1 / projection; the oracle models mirror it (oracle/ora.py). */
int isegmi_conv_split_qualifies(const isegmi_conv_desc* d);
/* number of floats of the packed weight image */
int isegmi_conv_packed_floats(const isegmi_conv_desc* d, int64_t* n);
/* host: natural [Cout][R][S][Cin] -> packed image consumed by isegmi_op_conv2d */
int isegmi_pack_conv_weights(const isegmi_conv_desc* d, const float* h_w_krsc, float* h_packed);
/* d_scale / d_shift / d_residual may be NULL (1 / 0 / none). residual is contiguous [M][Cout]. */
int isegmi_op_conv2d(const isegmi_conv_desc* d, const float* d_in, const float* d_wpacked,
                     const float* d_scale, const float* d_shift, const float* d_residual,
                     float* d_out, void* stream);
/* n <= 10 independent fp32 convolutions (no stem, tile 0) as ONE launch: the shared-weight heads of all FPN levels, the FPN's lateral / output convs.
 * Member i = descs[i] with d_in[i], d_w[i] (packed), d_scale[i] / d_shift[i] / d_res[i] (each array, or an entry, may be NULL), d_out[i]; every member's
 * result is bit-identical to its own isegmi_op_conv2d call -- same results, tile form chosen per GROUP (ring depth, 64 x 64 tiles or 32 x 32 blocks): every
 * fp32 tile form keeps one k-ordered chain per output.  The pointer arrays are HOST arrays; a NULL d_in[i] / d_w[i] / d_out[i] is ISEGMI_ERR_ARG. */
int isegmi_op_conv2d_group(int n, const isegmi_conv_desc* descs, const float* const* d_in, const float* const* d_w, const float* const* d_scale,
                           const float* const* d_shift, const float* const* d_res, float* const* d_out, void* stream);

/* fp16 variant (BASELINE configs[4]: "fp16 MFMA conv"): fp16 storage, v_mfma_f32_32x32x16_f16, fp32 accumulate
 * and epilogue.  Cin % 64 == 0 (or the stem, below); act none/relu; d_in / d_wpacked / d_residual are fp16, d_out fp16 or (out_f32)
 * fp32.  The 16-term sum inside one MFMA is not an ordered chain: parity with the oracle is tolerance-based. */
int isegmi_conv_packed_halfs(const isegmi_conv_desc* d, int64_t* n);
int isegmi_pack_conv_weights_f16(const isegmi_conv_desc* d, const float* h_w_krsc, uint16_t* h_packed);
int isegmi_op_conv2d_f16(const isegmi_conv_desc* d, const void* d_in, const void* d_wpacked,
                         const float* d_scale, const float* d_shift, const void* d_residual, void* d_out,
                         int out_f32, void* stream);
/* Fused identity bottleneck, fp16 (M2 `BottleneckWithFixedBatchNorm`, blocks 1.. of res2 / res3 under configs[4]):
 * out = relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x)))))))) + x) in ONE launch; the two Cmid-channel intermediates stay in LDS as
 * fp16 (rounded where the three-launch path rounds them when it stores them: results are bit-identical to three isegmi_op_conv2d_f16
 * calls with (r, s, cin)-ordered tiles), x is read once, out written once.  (Cin, Cmid) = (256, 64) or (512, 128); Cout = Cin = 4 Cmid;
 * weights are the isegmi_pack_conv_weights_f16 images of the three layers, scale / shift the folded FrozenBN of each. */
typedef struct isegmi_bottleneck_desc {
    int32_t N, H, W, Cin, Cmid;
    int32_t flags;                   /* bit 0 (test hook): 8-block grid, so that small shapes exercise the multi-tile stream; any other bit is rejected (ISEGMI_ERR_ARG) -- the
                                        timing-only experiment bits of the development builds (-DISEGMI_EXPERIMENT_FLAGS) produce wrong results and do not exist in a release build */
} isegmi_bottleneck_desc;
int isegmi_op_bottleneck_f16(const isegmi_bottleneck_desc* d, const void* d_x, const void* d_w1, const float* d_s1, const float* d_b1,
                             const void* d_w2, const float* d_s2, const float* d_b2, const void* d_w3, const float* d_s3,
                             const float* d_b3, void* d_out, void* stream);
/* The FIRST block of res2 (Cin = Cmid = 64, stride 1) with its projection shortcut d (1x1 64 -> 256 + BN) in the same launch:
 * out = relu(bn3(conv3(...)) + fp16(bnd(conv1x1_d(x)))) -- the shortcut tensor is rounded to fp16 where the four-launch path stores it. */
int isegmi_op_bottleneck_ds_f16(const isegmi_bottleneck_desc* d, const void* d_x, const void* d_w1, const float* d_s1, const float* d_b1,
                                const void* d_w2, const float* d_s2, const float* d_b2, const void* d_w3, const float* d_s3,
                                const float* d_b3, const void* d_wd, const float* d_sd, const float* d_bd, void* d_out, void* stream);
/* FPN top-down step under configs[4] (M3: `last_inner = inner_block(feature) + interpolate(last_inner, scale_factor=2, mode="nearest")`) as one launch: desc = the
 * lateral 1x1 / 1 / 0 conv (Cout % 8 == 0, W >= 8, tile 0), d_coarse = the coarser level [N][Hc][Wc][Cout] fp16; out = fp16(fp16(conv) + coarse[n][y/2][x/2]),
 * bit-identical to isegmi_op_conv2d_f16 followed by the engine's nearest-2x add (the lateral result is rounded to fp16 where that path stores it). */
int isegmi_op_conv1x1_up2x_add_f16(const isegmi_conv_desc* d, const void* d_in, const void* d_wpacked, const float* d_scale, const float* d_shift,
                                   const void* d_coarse, int Hc, int Wc, void* d_out, void* stream);
/* RPNHead under configs[4] (M4: `t = relu(conv3x3(x)); logits, deltas = cls_logits(t), bbox_pred(t)`) as one launch: desc = the 3x3 / 1 / 1 conv (Cout 256,
 * act 1, tile 0), d_w2packed / scale2 / shift2 = the fused cls + bbox 1x1 (256 -> cout2 <= 32 outputs, isegmi_pack_conv_weights_f16 image), d_out2 = fp32
 * [M][cout2].  t is rounded to fp16 where the two-launch path stores it and never leaves the CU; results are bit-identical to isegmi_op_conv2d_f16 twice.
 * *fused = 0 and NOTHING is launched when the layer is too small for the 192 x 256 row-strip tile the fusion lives on: the caller then runs the two launches. */
int isegmi_op_conv3x3_head_f16(const isegmi_conv_desc* d, const void* d_in, const void* d_wpacked, const float* d_scale, const float* d_shift,
                               const void* d_w2packed, const float* d_scale2, const float* d_shift2, int cout2, float* d_out2, int* fused, void* stream);
/* fp16 stem (M2 `StemWithFixedBatchNorm` conv1 under configs[4]): desc Cin=4 R=S=7 stride=2 pad=3 with H, W the image
 * size; d_in of isegmi_op_conv2d_f16 is then the haloed fp16 image [N][H+6][(W+7)&~1][4] this op writes from the fp32
 * NHWC C=3 batch (3 zero pixels on every side, zero 4th channel); weights are given as [Cout][7][7][4]. */
int isegmi_op_pad_c3_to_f16_halo(const float* d_in_nhwc3, int N, int H, int W, void* d_out, void* stream);
/* The whole stem in one launch (csrc/stem_pool_f16.hip): out = maxpool3x3/2/1(relu(bn(conv7x7/2/3(image)))) from the haloed image above, the packed
 * stem weights (isegmi_pack_conv_weights_f16 of the Cin=4 R=S=7 desc, Cout = 64) and the folded FrozenBN; d_out = [N][Hp][Wp][64] fp16 with
 * Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1.  Bit-identical to isegmi_op_conv2d_f16 (act 1) followed by the engine's fp16 max-pool (3, 2, 1; a max of fp16 values is exact);
 * the conv output stays in LDS.  flags (test hooks): bit 0: 8-block grid (blocks walk many units); bit 1: the shortest units (every seam between row segments). */
int isegmi_op_stem_pool_f16(int N, int H, int W, const void* d_halo, const void* d_w, const float* d_scale, const float* d_shift, void* d_out,
                            int flags, void* stream);

/* MFMA shape of the fp16 backbone tiles (process-wide tuning knob): 0: v_mfma_f32_32x32x16_f16 everywhere; 1: the 192 x 256 row-strip tile of the 3x3
 * layers and the fused RPN head on v_mfma_f32_16x16x32_f16 (same output tile per wave, same LDS images; the chip holds a higher clock on it in power-bound
 * loops: +8 % on the 634-GF layer); 2: 1 + the persistent 192 x 256 / 256 x 128 / 128 x 256 tiles and the UP2X merge (memory-bound: no gain); 3 (default):
 * 1 + the 144-row forms (48-row wave tiles, possible with 16 x 16 blocks only) for Cout <= 256 layers whose 192-row tiles leave CUs idle in one round.
 * RESULTS DO NOT DEPEND ON THE SHAPE: one 16 x 16 x 32 instruction sums its 32 products bit for bit as two chained 32 x 32 x 16 instructions do
 * (tools/microbench/mfma_shape.hip), so every fused-vs-unfused bit identity of the fp16 family holds under any setting.
 * conv tile ids 40 / 44 / 47 / 49 force the 16 x 16 x 32 form of 30 / 34 / 37 / 39 for one launch, 41 / 46 the 144-row forms of 40 / 47. */
int isegmi_set_f16_mfma_shape(int shape);
int isegmi_get_f16_mfma_shape(int* shape);

/* device front end (Y1 FastBaseTransform, README.md:243-249 `--image=...`; M1 build_transform + to_image_list, README.md:320-331): a uint8
 * [N][Hin][Win][3] batch -> fp32 NHWC3 [N][Hpad][Wpad][3]: bilinear (align_corners = False) to Hout x Wout (identity when the sizes match),
 * out[.., swap_rb ? 2-c : c] = (v[c] - mean3[c]) / std3[c], zeros in the padding; bit-identical to the oracle's ora_fast_base_transform /
 * ora_build_transform (oracle/ora_ops.c: the rounding sequence is stated there) and to the numpy transforms of isegmi/transforms.py */
int isegmi_op_preprocess_u8(const uint8_t* d_in, int N, int Hin, int Win, float* d_out, int Hout, int Wout, int Hpad, int Wpad,
                            int64_t out_img_stride, const float* mean3, const float* std3, int swap_rb, void* stream);

/* max_pool2d(k,s,p), -inf padding (M2/Y2 stem; k=1,s=2 = LastLevelMaxPool M3) */
int isegmi_op_maxpool(const float* d_in, int N, int H, int W, int C, int k, int s, int p,
                      float* d_out, void* stream);
/* bilinear align_corners=False to (Ho,Wo), optional +add, optional relu (Y3, Y4) */
int isegmi_op_resize_bilinear(const float* d_in, int N, int H, int W, int C, int Ho, int Wo,
                              const float* d_add, int relu, float* d_out, void* stream);
/* modulated deformable im2col: the sampling stage of the DCNv2 3x3 convolutions of the YOLACT++ backbones
 * (the YOLACT++ rows of README.md:216-221).  d_offset_mask [N][Ho][Wo][3*R*S] is the raw conv_offset_mask
 * output (channel 2k = dy_k, 2k+1 = dx_k, 2*R*S+k = mask logit); d_out [N][Ho][Wo][R*S][C] feeds a 1x1
 * convolution over R*S*C channels with the layer's KRSC weights.  C % 4 == 0. */
int isegmi_op_deform_im2col(const float* d_x, int N, int H, int W, int C, const float* d_offset_mask,
                            int R, int S, int stride, int pad, int dil, float* d_out, void* stream);
/* out = lateral + nearest2x(coarse) (M3) */
int isegmi_op_upsample_nearest2x_add(const float* d_coarse, int N, int Hc, int Wc, int C,
                                     const float* d_lateral, int H, int W, float* d_out,
                                     void* stream);
/* NHWC3 -> NHWC4 zero pad (stem input) */
int isegmi_op_pad_c3_to_c4(const float* d_in, int64_t npix, float* d_out, void* stream);
/* fn: 0 exp 1 sigmoid 2 tanh 3 log2 -- exposes the deterministic math for parity tests */
int isegmi_op_map_f32(const float* d_x, float* d_y, int64_t n, int fn, void* stream);

/* ---- selection: torch.topk / sort stand-in (M6, M9, Y6) ----
 * rows independent problems; row r = d_keys + r*row_stride, n elements; output sorted by
 * (score desc, index asc); k <= 8192.  k_eff = min(k, n, d_limit[r / rows_per_limit]) when
 * d_limit != NULL.  d_vals/d_idx are [rows][k]; d_cnt [rows] (optional) receives k_eff. */
int isegmi_op_topk(const float* d_keys, int64_t row_stride, int rows, int n, int k,
                   const int32_t* d_limit, int rows_per_limit, float* d_vals, int32_t* d_idx,
                   int32_t* d_cnt, void* stream);

/* ---- Yolact Detect (Y6; App. A.6/A.9; README.md:243 --top_k / --score_threshold path) ----
 * softmax -> decode -> conf prefilter -> per-class top_k -> fast-NMS -> per-image top max_det.
 * d_conf holds LOGITS [N][P][ncls]; d_loc [N][P][4]; d_mask [N][P][mask_dim] (tanh applied);
 * d_priors [P][4] (cx,cy,w,h).  Outputs are fixed-capacity [N][max_det]. */
typedef struct isegmi_yolact_detect_args {
    int32_t N, P, ncls, mask_dim, top_k, max_det;
    float conf_thresh, nms_thresh;
    const float* d_conf;
    const float* d_loc;
    const float* d_mask;
    const float* d_priors;
    /* workspace (caller-allocated device memory) */
    float* d_ws_scoresT;     /* [N][ncls-1][P] */
    float* d_ws_boxes;       /* [N][P][4] decoded xyxy (relative) */
    int32_t* d_ws_counts;    /* [2N] */
    float* d_ws_tk_vals;     /* [N][ncls-1][top_k] */
    int32_t* d_ws_tk_idx;    /* [N][ncls-1][top_k] */
    int32_t* d_ws_tk_cnt;    /* [N][ncls-1] */
    float* d_ws_cand;        /* [N][ncls-1][top_k] */
    float* d_ws_fin_vals;    /* [N][max_det] */
    int32_t* d_ws_fin_idx;   /* [N][max_det] */
    int32_t* d_ws_fin_cnt;   /* [N] */
    /* outputs */
    int32_t* d_out_count;    /* [N] */
    float* d_out_boxes;      /* [N][max_det][4] relative xyxy */
    float* d_out_scores;     /* [N][max_det] */
    int32_t* d_out_classes;  /* [N][max_det] 0..ncls-2 */
    float* d_out_coeffs;     /* [N][max_det][mask_dim] */
    int32_t* d_out_prior;    /* [N][max_det] */
    /* optional fused-head layout (all zero = the three contiguous buffers above): the prediction head was run as ONE
     * convolution whose per-pixel output row holds [A x 4 loc | A x ncls conf | A x mask_dim mask]; then d_conf, d_loc
     * and d_mask all point at that buffer [N][P/A][pix_stride] and prior p = pix*A + a reads from row pix.
     * mask_tanh: the mask block holds pre-activation values; tanh is applied to the <= max_det gathered rows. */
    int32_t A;
    int32_t mask_tanh;
    int64_t pix_stride;
    int32_t off_loc, off_conf, off_mask;
    int32_t second_threshold;   /* App. A.6 fork, fast_nms(second_threshold=True): a box must also have its class score > conf_thresh (a prior passes
                                   the pre-filter on its BEST class and is ranked in every class); 0 = the default detect() call */
} isegmi_yolact_detect_args;
int isegmi_op_yolact_detect(const isegmi_yolact_detect_args* a, void* stream);

/* ---- Yolact postprocess masks (Y7; App. A.9) ----
 * d_proto [N][PH][PW][32]; d_coeffs [N][K][32]; d_boxes [N][K][4] relative; d_count [N].
 * d_ws_lo [N][K][PH][PW] fp32 workspace (holds the crop windows' pixels of the proto-resolution masks afterwards, the rest is
 * undefined); d_out_masks [N][K][h][w] uint8 {0,1} (only the first
 * count[n] masks of image n are written); d_out_boxes [N][K][4] int64 (may be NULL). */
int isegmi_op_yolact_masks(const float* d_proto, const float* d_coeffs, const float* d_boxes,
                           const int32_t* d_count, int N, int PH, int PW, int mask_dim, int K, int h,
                           int w, float* d_ws_lo, uint8_t* d_out_masks, int64_t* d_out_boxes,
                           void* stream);

/* ---- Mask R-CNN RoI ops (M6 M7 M9 M10 M11 M12; App. A.4-A.8; all reached from README.md:331) ---- */
/* Semantic forks of the NMS (which of them detectron.jittor takes is unknown -- the reference tree holds no code -- so every one is a
 * switch).  OR-ed into the `nms_flags` argument of isegmi_op_rpn_level / _rpn_levels / isegmi_box_post_args; 0 = maskrcnn-benchmark's CUDA path. */
#define ISEGMI_NMS_GE 1           /* suppress on iou >= thr (the CPU loop) instead of iou > thr (the CUDA kernel, jt.nms) */
#define ISEGMI_NMS_NO_PLUS_ONE 2  /* plain areas (x2-x1)*(y2-y1) in the IoU instead of the legacy +1 widths */
#define ISEGMI_NMS_INDEX_ORDER 4  /* box post-processing only: a class's kept detections in ascending proposal index (the CPU NMS returns nonzero(keep))
                                     instead of score order; the RPN truncates a score-sorted list, where both orders keep the same boxes in the same order */
/* greedy NMS (A.6): `problems` independent sets of n <= 6144 boxes; visiting order (score desc, index
 * asc); IoU with legacy +1 areas when plus_one; suppress on iou > thr (ge: >=).  d_keep [problems][n]
 * receives ORIGINAL indices in score order, d_cnt [problems] the count (<= max_keep when max_keep > 0). */
int isegmi_op_nms(const float* d_boxes, const float* d_scores, int problems, int n, float thr,
                  int plus_one, int ge, int max_keep, int32_t* d_keep, int32_t* d_cnt, void* stream);
/* LevelMapper + RoIAlign (A.7).  aligned 0: the legacy op (no half-pixel shift, RoI at least one pixel wide and high -- maskrcnn-benchmark); aligned 1:
 * ROIAlign(aligned=True) -- scaled corners minus 0.5, no minimum size (the other side of the App. A.7 fork).  d_feats: host array of nlevels device pointers
 * (NHWC, level k_min first); rois [N][K][4] image coords; counts [N]; out [N*K][PH][PW][C] (rows past
 * count zero-filled).  fixed_level >= 0 bypasses the LevelMapper.  d_out_level [N][K] optional.
 * sampling > 0: fixed sampling x sampling grid per bin; sampling <= 0: adaptive ceil(roi / pooled) (ROIAlign's
 * sampling_ratio = 0, used by the R-50-C4 config of README.md:263-273). */
int isegmi_op_roi_align(const float* const* d_feats, const int32_t* Hs, const int32_t* Ws,
                        const float* scales, int nlevels, const float* d_rois, const int32_t* d_counts,
                        int N, int K, int C, int PH, int PW, int sampling, int aligned, int k_min, int fixed_level,
                        float* d_out, int32_t* d_out_level, void* stream);
/* nn.AvgPool2d over the whole window of every RoI (FastRCNNPredictor of the C4 box head): x [R][HW][C] -> out [R][C] */
int isegmi_op_avgpool_full(const float* d_x, int64_t R, int HW, int C, float* d_out, void* stream);
/* fp16-storage LevelMapper + RoIAlign (configs[4]): d_feats / d_out are fp16, arithmetic is the fp32 op's. */
int isegmi_op_roi_align_f16(const void* const* d_feats, const int32_t* Hs, const int32_t* Ws,
                            const float* scales, int nlevels, const float* d_rois, const int32_t* d_counts,
                            int N, int K, int C, int PH, int PW, int sampling, int aligned, int k_min, void* d_out, void* stream);
/* The FPN heads' RoIAlign (sampling 2, LevelMapper) as two launches.  isegmi_op_roi_prep, one thread per RoI: d_table [N*K][2*(PH+PW)+1][4]
 * int32 -- for each of the RoI's 2*PH sample rows and 2*PW sample columns {byte offset of the low tap, of the high tap, weight of the low
 * tap, of the high tap (fp32 bits)} with the validity test and clamps of the scalar op applied (a sample outside the map: weights +0 and tap offsets
 * past the end of the map, which the pooling launch's range-checked loads read as 0), then {level index, H, W, signature of (C * elem_bytes, PH, PW)} -- and, when
 * d_order is not NULL, d_order [N][K] = every image's RoI rows n*K + k sorted by (level, Morton code of the RoI centre on that level's map),
 * rows past count last.  K <= 2048; elem_bytes 4 (fp32 features) or 2 (fp16); a level's map must stay under 512 MiB per image; `aligned` as in isegmi_op_roi_align.
 * isegmi_op_roi_align_ordered / _f16_ordered then run workgroup L on one 128-byte channel slice of RoI d_order[L / 8 ...] (d_order NULL: row
 * order) so that RoIs sharing pixels are in flight together and every XCD's L2 holds its own slice of them; 7x7 or 14x14 bins, C in
 * {32..256} (fp32) / {64..512} (fp16).  Outputs are those of isegmi_op_roi_align / _f16 bit for bit for ANY feature values (inf / NaN included), whatever
 * permutation d_order holds (entries outside [0, N*K) are skipped, rows it leaves out are not written).  d_rois / d_counts / scales / k_min must be the
 * ones the table was made from; a table made for another C * elem_bytes / PH / PW fills the RoI's output with NaN. */
int64_t isegmi_op_roi_table_bytes(int N, int K, int PH, int PW);
int isegmi_op_roi_prep(const float* d_rois, const int32_t* d_counts, int N, int K, const int32_t* Hs, const int32_t* Ws,
                       const float* scales, int nlevels, int k_min, int C, int PH, int PW, int elem_bytes, int aligned,
                       int32_t* d_order, void* d_table, void* stream);
int isegmi_op_roi_align_ordered(const float* const* d_feats, const int32_t* Hs, const int32_t* Ws,
                                const float* scales, int nlevels, const float* d_rois, const int32_t* d_counts,
                                const int32_t* d_order, const void* d_table, int N, int K, int C, int PH, int PW,
                                int k_min, float* d_out, void* stream);
int isegmi_op_roi_align_f16_ordered(const void* const* d_feats, const int32_t* Hs, const int32_t* Ws,
                                    const float* scales, int nlevels, const float* d_rois, const int32_t* d_counts,
                                    const int32_t* d_order, const void* d_table, int N, int K, int C, int PH, int PW,
                                    int k_min, void* d_out, void* stream);
/* PostProcessor.filter_results (A.5): softmax, per-class decode(10,10,5,5)+clip, score filter, NMS,
 * kth-value cut to det_per_img.  Output order: class ascending, NMS (score) order inside a class -- proposal-index order with
 * ISEGMI_NMS_INDEX_ORDER. */
typedef struct isegmi_box_post_args {
    int32_t N, R, ncls, det_per_img, cap, nms_flags;   /* nms_flags: OR of ISEGMI_NMS_* */
    float score_thresh, nms_thresh;
    int64_t logits_stride, regr_stride;   /* floats between consecutive rois */
    const float* d_logits;     /* [N*R] rows of ncls */
    const float* d_regr;       /* [N*R] rows; class j deltas at 4j..4j+3 */
    const float* d_props;      /* [N][R][4] */
    const int32_t* d_prop_cnt; /* [N] */
    const int32_t* d_image_hw; /* [N][2] unpadded (h, w) */
    float* d_ws_prob;          /* [N][R][ncls] */
    float* d_ws_cand_scores;   /* [N][ncls-1][R] */
    float* d_ws_cand_boxes;    /* [N][ncls-1][R][4] */
    int32_t* d_ws_kept_total;  /* [N] */
    float* d_ws_top_vals;      /* [N][det_per_img] */
    int32_t* d_ws_top_idx;     /* [N][det_per_img] */
    int32_t* d_out_count;      /* [N] */
    float* d_out_boxes;        /* [N][cap][4] */
    float* d_out_scores;       /* [N][cap] */
    int32_t* d_out_labels;     /* [N][cap] 1..ncls-1 (0 = empty) */
    /* optional (all four or none): classes with more than 128 candidates get their suppression matrix from the whole chip (three launches instead of
     * one block per class building it alone: ~85 us for a 1000-candidate class); same kept lists */
    void* d_ws_crowd_matrix;     /* [N][ncls-1] x 131072 bytes */
    void* d_ws_crowd_keys;       /* [N][ncls-1][R] 8-byte keys */
    float* d_ws_crowd_boxes;     /* [N][ncls-1][R][4] */
    int32_t* d_ws_crowd_m;       /* [N][ncls-1] */
} isegmi_box_post_args;
int isegmi_op_box_postprocess(const isegmi_box_post_args* a, void* stream);
/* mask predictor tail (A.8): out[r,p] = sigmoid(<feat[r,p,:], w[label_r,:]> + b[label_r]); label 0 -> zeros */
int isegmi_op_mask_logits_select(const float* d_feat, int R, int HW, int C, const float* d_w,
                                 const float* d_b, const int32_t* d_labels, float* d_out, void* stream);
/* Masker(threshold, padding=1) paste (A.8): masks [N][K][M][M], boxes [N][K][4] -> u8 [N][K][im_h][im_w] */
int isegmi_op_paste_masks(const float* d_masks, const float* d_boxes, const int32_t* d_counts, int N,
                          int K, int M, int im_h, int im_w, float thr, uint8_t* d_out, void* stream);
/* AnchorGenerator.grid_anchors (M5, A.3): d_out [(y*grid_w + x)*A + a][4] = d_base[a] + (x, y, x, y) * stride */
int isegmi_op_grid_anchors(const float* d_base, int A, int stride, int grid_h, int grid_w, float* d_out, void* stream);
/* one RPN level (A.4) for N images: fused head [N][HW][A*5] (A logits then A*4 deltas per pixel) */
int isegmi_op_rpn_level(const float* d_head, const float* d_anchors, const int32_t* d_image_hw, int N,
                        int HW, int A, int pre_nms, int post_nms, float nms_thr, float min_size,
                        int nms_flags /* ISEGMI_NMS_GE | ISEGMI_NMS_NO_PLUS_ONE */, float* d_ws_prob, float* d_ws_tk_vals, int32_t* d_ws_tk_idx,
                        int32_t* d_ws_tk_cnt, float* d_out_boxes, float* d_out_scores,
                        int32_t* d_out_cnt, void* d_ws_nms /* optional: N * 131072 bytes, the suppression
                        matrix of the chip-wide NMS used when pre_nms <= 1024; NULL = single-block NMS */,
                        void* stream);
/* the same for nl <= 5 FPN levels at once (SURVEY 2.1 "level x image as one batch dimension"): five launches -- sigmoid, two-level top-k, suppression
 * matrix, greedy scan -- over all (level, image) rows; 256 < pre_nms <= 1024.  d_heads / d_anchors: HOST arrays of nl DEVICE pointers ([N][HW_l][A*5] /
 * [HW_l*A][4]).  Outputs [N][nl][post_nms][4], [N][nl][post_nms] (-1 beyond the count), [N][nl]: level l's list is what isegmi_op_rpn_level gives for it,
 * bit for bit.  Workspaces (isegmi_op_rpn_levels_workspace): prob prob_elems floats, cand_vals / cand_idx cand_elems each, tk_vals / tk_idx
 * nl*N*pre_nms, tk_cnt nl*N, nms nl*N*131072 bytes. */
int isegmi_op_rpn_levels_workspace(int nl, int N, const int32_t* HW, int A, int pre_nms, int64_t* prob_elems, int64_t* cand_elems);
int isegmi_op_rpn_levels(int nl, const float* const* d_heads, const float* const* d_anchors, const int32_t* HW, const int32_t* d_image_hw, int N, int A,
                         int pre_nms, int post_nms, float nms_thr, float min_size, int nms_flags, float* d_ws_prob, float* d_ws_cand_vals,
                         int32_t* d_ws_cand_idx, float* d_ws_tk_vals, int32_t* d_ws_tk_idx, int32_t* d_ws_tk_cnt, void* d_ws_nms,
                         float* d_out_boxes, float* d_out_scores, int32_t* d_out_cnt, void* stream);

/* ---- COCO run-length encoding on the device (the on-disk format behind inference() / tools/test_net.py,
 * README.md:344-347, annotation layout README.md:55-66; Yolact eval.py Detections.add_mask / dump, README.md:243-249) ----
 * pycocotools rleEncode + rleToString restated: for every valid slot (n, k), k < d_count[n], of the uint8 planes d_masks [N][K][plane_h][plane_w]
 * the column-major run lengths of the top-left (h_n, w_n) window (d_image_hw [N][2], NULL = the whole plane), starting with the run of
 * zeros, and their compressed ASCII string.  Slot m = n*K + k owns runs d_out_counts[d_out_run_off[m] .. d_out_run_off[m+1]) and characters
 * d_out_chars[d_out_str_off[m] .. d_out_str_off[m+1]); invalid slots own nothing.  d_out_status [4] = {total runs, total chars,
 * overflow bits (1: runs > cap_runs, 2: chars > cap_chars; the outputs are then incomplete), 0}.  Bit-identical to isegmi/coco.py
 * rle_counts / rle_to_string.  Workspace sizes: isegmi_rle_workspace. */
typedef struct isegmi_rle_args {
    int32_t N, K, plane_h, plane_w;
    int32_t cap_runs;                 /* entries of d_out_counts / d_ws_starts / d_ws_len: a multiple of 1024 */
    int32_t cap_chars;                /* bytes of d_out_chars */
    const uint8_t* d_masks;
    const int32_t* d_count;           /* [N] or NULL (all K slots valid) */
    const int32_t* d_image_hw;        /* [N][2] or NULL */
    const int32_t* d_windows;         /* [N*K][4] (x0, y0, x1, y1) or NULL: every set pixel of slot m lies inside [x0, x1) x [y0, y1); only that part
                                         of a plane is then read (pixels outside a window need not even be initialised) */
    void* d_ws_trans;                 /* workspace, sizes from isegmi_rle_workspace */
    int32_t* d_ws_col;
    int32_t* d_ws_nruns;
    int32_t* d_ws_tile;
    uint8_t* d_ws_len;
    uint32_t* d_ws_starts;
    int32_t* d_out_run_off;           /* [N*K + 1] */
    uint32_t* d_out_counts;           /* [cap_runs] */
    int32_t* d_out_str_off;           /* [N*K + 1] */
    uint8_t* d_out_chars;             /* [cap_chars] */
    int32_t* d_out_status;            /* [4] */
} isegmi_rle_args;
int isegmi_rle_workspace(int N, int K, int plane_h, int plane_w, int cap_runs, int64_t* trans_bytes, int64_t* col_bytes,
                         int64_t* nruns_bytes, int64_t* tile_bytes, int64_t* len_bytes, int64_t* starts_bytes);
int isegmi_op_rle_encode(const isegmi_rle_args* a, void* stream);

/* ---- model engine ----
 * Replaces the model object the reference builds inside COCODemo(cfg, ...) (README.md:320-324)
 * and Yolact eval.py (README.md:243): weights are pushed per layer under their upstream
 * state-dict names with BN already folded to (scale, shift) by the Python host; activations,
 * workspaces and outputs live in named device buffers owned by the engine. */
typedef struct isegmi_engine isegmi_engine;
/* model_kind: 1 = Yolact R50-FPN, 2 = Mask R-CNN R50/R101-FPN.  H, W = network input size. */
int isegmi_engine_create(int model_kind, int max_batch, int H, int W, isegmi_engine** out);
int isegmi_engine_destroy(isegmi_engine* e);
int isegmi_engine_set_param(isegmi_engine* e, const char* name, float value);
/* "graph" param != 0: a forward (yolact / maskrcnn) runs eagerly once, is captured into a hipGraph on the next call with the
 * same (batch, input pointer) and replayed afterwards (latency mode: ends joined on the main stream; any set_param /
 * set_conv / set_tensor drops the captured graphs).  Counters for tests and diagnostics: */
int isegmi_engine_graph_stats(isegmi_engine* h, int64_t* captures, int64_t* replays, int64_t* failures);
/* h_w_krsc: natural [Cout][R][S][Cin]; h_scale / h_shift [Cout] or NULL */
int isegmi_engine_set_conv(isegmi_engine* e, const char* name, int Cout, int R, int S, int Cin,
                           const float* h_w_krsc, const float* h_scale, const float* h_shift);
/* constant tensors (priors, anchors, deconv weights ...) */
int isegmi_engine_set_tensor(isegmi_engine* e, const char* name, const void* h_data, int64_t bytes);
/* Yolact.forward (Y2-Y6): d_images NHWC3 fp32 [N][H][W][3] already normalised (Y1). Asynchronous
 * on the engine stream; results in buffers det.count/box/score/class/coeff/prior and proto. */
int isegmi_yolact_forward(isegmi_engine* e, const float* d_images_nhwc3, int N);
/* GeneralizedRCNN.forward (M2-M11): d_images NHWC3 fp32 [N][H][W][3], normalised and zero-padded to the
 * engine's (H,W); h_image_hw [N][2] = unpadded (h,w).  Results: det.count/box/score/label [N][cap],
 * det.mask28 [N][cap][28][28], proposals, rpn.* ... in named buffers. */
int isegmi_maskrcnn_forward(isegmi_engine* e, const float* d_images_nhwc3, const int32_t* h_image_hw, int N);
/* The same on a padded canvas (H, W) <= the engine's (H, W): d_images is [N][H][W][3] contiguous.  upstream's to_image_list pads every
 * BATCH to its own size (max over its images, rounded up to SIZE_DIVISIBILITY), so one engine -- weights packed once, buffers sized once
 * for the largest canvas -- serves every batch of a data set (COCODemo / inference(), README.md:320-331, 344-347).  The host sets the
 * per-level base anchors as tensors "anchor_base.<l>" [A][4] and the strides as params "anchor_stride<l>"; the grid is laid out on the
 * device for the current canvas (isegmi_op_grid_anchors). */
int isegmi_maskrcnn_forward_canvas(isegmi_engine* e, const float* d_images_nhwc3, const int32_t* h_image_hw, int N, int H, int W);
/* Masker paste of the last forward into (out_h,out_w) planes; boxes first scaled by h_ratios_wh [N][2] =
 * (out_w/w_i, out_h/h_i) like BoxList.resize -> det.masks u8 [N][cap][out_h][out_w], det.box_resized */
int isegmi_maskrcnn_paste(isegmi_engine* e, const float* h_ratios_wh, int out_h, int out_w);
/* postprocess (Y7): masks of the last forward at (out_h,out_w) -> det.masks u8, det.box_int i64 */
int isegmi_yolact_postprocess(isegmi_engine* e, int out_h, int out_w);
/* the same with image n assembled at ITS (h, w) = h_image_hw[n] inside a common plane of the batch's maximum size (a batch of images of
 * different original sizes; upstream's evalimage postprocesses one image at a time at its own size) */
int isegmi_yolact_postprocess_sizes(isegmi_engine* e, const int32_t* h_image_hw, int N);
int isegmi_engine_sync(isegmi_engine* e);
int isegmi_engine_stream(isegmi_engine* e, void** stream);
/* asynchronous H2D of an input batch from PINNED host memory on the engine's copy stream: ordered after the previous
 * forward consumed its input, and the next forward is ordered behind it (stands where the reference's
 * jt.array(image) host->device transfer sits, README.md:311/331) */
int isegmi_engine_upload_async(isegmi_engine* e, void* d_dst, const void* h_src_pinned, int64_t bytes);
/* isegmi_op_preprocess_u8 on the engine's main stream: behind any upload_async of the bytes, in front of the next forward */
int isegmi_engine_preprocess_u8(isegmi_engine* e, const uint8_t* d_u8, int N, int Hin, int Win, float* d_out, int Hout, int Wout,
                                int Hpad, int Wpad, int64_t out_img_stride, const float* mean3, const float* std3, int swap_rb);
/* per-step completion marks on the results stream; step_times returns the intervals between consecutive marks (ms) */
int isegmi_engine_mark_step(isegmi_engine* e);
int isegmi_engine_step_times(isegmi_engine* e, float* ms, int cap, int* count);
/* host wait for the mark `back` marks before the newest (0 = newest): a producer loop's bound on the steps it keeps in flight */
int isegmi_engine_wait_mark(isegmi_engine* e, int back);
/* dtype: 0 f32, 1 i32, 2 u8, 3 i64; shape4 receives up to 4 dims */
int isegmi_engine_buffer_info(isegmi_engine* e, const char* name, void** d_ptr, int64_t* bytes,
                              int32_t* dtype, int64_t* shape4, int32_t* ndim);
/* device memory held by the engine: packed weights + constant tensors, and activation / workspace / output buffers (bytes) */
int isegmi_engine_memory(isegmi_engine* e, int64_t* weight_bytes, int64_t* buffer_bytes);
/* Diagnostic: which of the engine's ten streams (0 main, 1-3 side, 4 tail, 5 heads, 6-8 heads-side, 9 copy) share an in-order hardware queue of the
 * runtime (equal class numbers = one queue; found by a spin-kernel probe, ~2 ms; call on an idle engine).  An engine's throughput depends on this
 * placement, which the runtime derives from the process's stream-creation history. */
int isegmi_engine_stream_layout(isegmi_engine* e, int32_t* queue_class, int n);
/* per-stage hipEvent timings of the last synchronised forward (set_param "timing" 1 first) */
int isegmi_engine_get_timings(isegmi_engine* e, char* names, int names_cap, float* ms, int ms_cap,
                              int* count);

/* ---- device-side COCO output (README.md:344-347, 243-249): what inference() / eval ship per batch ----
 * isegmi_engine_rle: isegmi_op_rle_encode over det.masks of the last postprocess / paste on the results stream; h_image_hw [N][2] = every
 *   image's own (h, w) inside the mask planes (NULL: the whole plane) -> buffers rle.run_off / rle.counts / rle.str_off / rle.chars /
 *   rle.status.  Capacities: params "rle_cap_runs" / "rle_cap_chars" (default 256 Ki entries / bytes per image of max_batch).
 * isegmi_engine_pack_coco_records: ONE block of fixed size for n_block image slots (>= the last forward's batch: a short last batch still
 *   fills a block of the per-step size; the extra slots carry count 0) (layout: csrc/results.cpp, mirrored by isegmi/dist.py):
 *   status, boxes in original-image coordinates, counts, scores, labels, [mask scores], string offsets, RLE strings.
 * isegmi_engine_download_*: the block's asynchronous D2H into pinned memory on the results stream, behind its producer (two slots). */
int isegmi_engine_rle(isegmi_engine* e, const int32_t* h_image_hw);
int isegmi_engine_coco_record_bytes(isegmi_engine* e, int N, int64_t* bytes, int64_t* chars_offset);
int isegmi_engine_pack_coco_records(isegmi_engine* e, void* d_dst, int64_t cap, int n_block, int64_t* bytes);
int isegmi_engine_download_async(isegmi_engine* e, int slot, void* h_dst_pinned, const void* d_src, int64_t bytes);
int isegmi_engine_download_fence(isegmi_engine* e, int slot);
int isegmi_engine_download_wait(isegmi_engine* e, int slot);

/* one contiguous record block of the last Yolact forward for the all-gather:
 * [count i32 N][box f32 N*K*4][score f32 N*K][class i32 N*K][coeff f32 N*K*32]([proto f32 N*PH*PW*32]) */
int isegmi_yolact_pack_records(isegmi_engine* e, void* d_dst, int64_t cap, int with_proto, int64_t* bytes);

/* Mask R-CNN record block for the all-gather:
 * [count i32 N][box f32 N*K*4][score f32 N*K][label i32 N*K][mask28 f32 N*K*784] */
int isegmi_maskrcnn_pack_records(isegmi_engine* e, void* d_dst, int64_t cap, int64_t* bytes);

/* conv-kernel statistics since the last call (set_param "conv_timing" 1): algorithmic FLOPs, summed
 * HIP-event time (ms) of the conv launches on the engine stream, launch count; resets them */
int isegmi_engine_conv_stats(isegmi_engine* e, double* flops, double* ms, int64_t* launches);
/* HBM-bound (non-conv) stages under set_param("op_timing", 1): per stage label the summed HIP-event time (us, on the stream the stage is
 * launched on), its ALGORITHMIC bytes and the number of timed scopes since the last call; labels joined by '\n' in `names`.
 * Call after isegmi_engine_sync.  Resets the accumulators.  (bench.py: roofline_hbm) */
int isegmi_engine_op_stats(isegmi_engine* e, char* names, int names_cap, double* us, double* bytes, int64_t* launches, int cap, int* count);

/* per-layer text report (label, GFLOP, ms, TFLOP/s per line) accumulated under conv_timing; clears it */
int isegmi_engine_conv_report(isegmi_engine* e, char* buf, int cap);

/* ---- multi-GPU: images shard by batch, one process per GPU; the only exchange is one
 * RCCL all-gather of fixed-size records per batch (upstream analogue: the pickle all_gather of
 * {image_id: BoxList} in maskrcnn-benchmark engine/inference.py, reached from README.md:344-347). */
typedef struct isegmi_comm isegmi_comm;
int isegmi_comm_unique_id(void* out128);                       /* rank 0; ship the 128 bytes to all ranks */
int isegmi_comm_create(const void* uid128, int rank, int world, isegmi_comm** out);
int isegmi_comm_destroy(isegmi_comm* c);
/* d_recv holds world*bytes; runs on producer_stream behind the work already queued there (NULL: the communicator's own stream) */
int isegmi_comm_allgather(isegmi_comm* c, const void* d_send, void* d_recv, int64_t bytes,
                          void* producer_stream);
int isegmi_comm_wait(isegmi_comm* c);
/* Two record slots (0, 1) so that step t+1 packs while step t gathers, without a host sync in between:
 *   fence_producer(slot, s): work enqueued on stream s afterwards runs after the slot's last all-gather finished (WAR);
 *   allgather_slot: as isegmi_comm_allgather (= slot 0) through the given slot; wait_slot: host wait for that slot. */
int isegmi_comm_fence_producer(isegmi_comm* c, int slot, void* producer_stream);
int isegmi_comm_allgather_slot(isegmi_comm* c, int slot, const void* d_send, void* d_recv, int64_t bytes,
                               void* producer_stream);
int isegmi_comm_wait_slot(isegmi_comm* c, int slot);
/* Collectives of one communicator run in the order the host issued them, whatever streams they were given: one that goes to another stream than its
 * predecessor first makes that stream wait for the predecessor's completion event.  out4 = {rank, world, collectives issued, how many of them changed
 * stream} (diagnostic: isegmi.dist keeps every data step, empty step and redo of a run on ONE stream, so the last number stays 0 between control words). */
int isegmi_comm_info(isegmi_comm* c, int64_t* out4);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
