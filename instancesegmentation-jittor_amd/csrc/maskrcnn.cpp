// maskrcnn.cpp -- Mask R-CNN R50/R101-FPN forward graph.
//
// GeneralizedRCNN: ResNet (stride in the first 1x1, FrozenBN folded by the host) -> FPN (nearest
// top-down, no activation, P6 = P5[::2,::2]) -> RPN head (shared 3x3 + fused 1x1 cls|bbox) ->
// per-level top-k/decode/clip/NMS -> per-image top-1000 -> RoIAlign 7x7 -> FC6/FC7/predictors ->
// per-class NMS + kth-value cut -> RoIAlign 14x14 on detections -> 4x conv3x3 -> deconv2x2 (as four
// strided 1x1 convolutions) -> class-selected 1x1 + sigmoid.  Reached from
// COCODemo.run_on_opencv_image (README.md:331) and tools/test_net.py (README.md:344-347).
#include <string.h>

#include "engine.h"

namespace isegmi {

int nms_launch(const float*, const float*, int, int, float, int, int, int, int*, int*, hipStream_t);
int rpn_sigmoid_launch(const float* head, int64_t total, int A, int CH, float* prob, hipStream_t st);
int rpn_decode_nms_launch(const float* head, const float* anchors, const float* tk_vals, const int* tk_idx, const int* tk_cnt,
                          const int* image_hw, int N, int HWA, int A, int CH, int pre_nms, int post_nms, float thr, float min_size,
                          int ge, int level, int L, int post_cap, float* out_boxes, float* out_scores, int* out_cnt, void* nms_ws,
                          hipStream_t st);
int grid_anchors_launch(const float* base, int A, int stride, int gh, int gw, float* out, hipStream_t st);
int sum_counts_launch(const int* cnt, int N, int L, int* total, hipStream_t st);
int gather_proposals_launch(const float* cand_boxes, const float* fin_vals, const int* fin_idx, const int* fin_cnt, int N,
                            int cand_per_img, int K, float* props, float* prop_scores, int* prop_cnt, hipStream_t st);
int roi_align_launch(const float* const* feats, const int* Hs, const int* Ws, const float* scales, int nlevels, const float* rois,
                     const int* counts, int N, int K, int C, int PH, int PW, int g, int k_min, int fixed_level, float* out,
                     int* out_level, hipStream_t st, const int* order = nullptr, const void* tab = nullptr, int aligned = 0);
int box_postprocess_launch(const isegmi_box_post_args* a, hipStream_t st);
int mask_logits_select_launch(const float* feat, int R, int HW, int C, const float* w, const float* b, const int* labels, float* out,
                              hipStream_t st);
int paste_masks_launch(const float* masks, const float* boxes, const int* counts, int N, int K, int M, int im_h, int im_w, float thr,
                       uint8_t* out, hipStream_t st, int* win, bool clear);
int scale_boxes_launch(const float* boxes, const float* ratios, int N, int K, float* out, hipStream_t st);
int topk_launch(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit, float* out_vals,
                int* out_idx, int* out_cnt, hipStream_t st);
int64_t topk_scratch_elems(int rows, int n, int k);
int topk_launch_ws(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit, float* out_vals,
                   int* out_idx, int* out_cnt, float* ws_vals, int* ws_idx, hipStream_t st);

// SURVEY 7.2 / App. A.6-A.7 semantic forks as engine parameters (defaults = the maskrcnn-benchmark CUDA path): "nms_ge" 1 suppress on iou >= thr;
// "nms_plus_one" 0 plain areas in the NMS IoU; "nms_index_order" 1 a class's detections in ascending proposal index (CPU nonzero order);
// "roi_aligned" 1 ROIAlign(aligned=True).
static int maskrcnn_nms_flags(Engine& e) {
    return ((int)e.param("nms_ge", 0) ? ISEGMI_NMS_GE : 0) | ((int)e.param("nms_plus_one", 1) ? 0 : ISEGMI_NMS_NO_PLUS_ONE) |
           ((int)e.param("nms_index_order", 0) ? ISEGMI_NMS_INDEX_ORDER : 0);
}

#define TRY(x)               \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

static int need_tensor(Engine& e, const std::string& name, int64_t bytes, const RawBuf** out) {
    auto it = e.tensors.find(name);
    if (it == e.tensors.end()) { set_error("tensor not set: " + name); return ISEGMI_ERR_STATE; }
    if (bytes > 0 && it->second.bytes != bytes) { set_error("tensor " + name + " has the wrong size"); return ISEGMI_ERR_STATE; }
    *out = &it->second;
    return ISEGMI_OK;
}

// image_hw is caller memory and changes from batch to batch on real data: it reaches the device through the pinned ring (asynchronous
// copy on the main stream) into one of TWO device buffers used alternately -- the previous forward's RoI heads (tail stream) may still
// read theirs; the one before that finished before the previous forward's WAR wait on tail_done, which precedes this copy in stream
// order.  Kept out of maskrcnn_forward so that the forward's enqueue code is capturable into a hipGraph.
int maskrcnn_set_image_hw(Engine& e, const int32_t* h_image_hw, int N) {
    std::vector<int32_t> now(h_image_hw, h_image_hw + 2 * N);
    if (now == e.last_hw && e.last_hw_ptr != nullptr) return ISEGMI_OK;  // steady-state batches: the current buffer already holds it
    e.hw_slot ^= 1;
    void* p;
    TRY(eng_buf(e, e.hw_slot ? "image_hw.1" : "image_hw.0", (int64_t)e.max_batch * 8, &p, 1, {N, 2}));
    TRY(eng_stage_small(e, h_image_hw, (size_t)N * 8, p, e.stream));
    e.last_hw = now;
    e.last_hw_ptr = p;
    return ISEGMI_OK;
}

static int maskrcnn_c4_forward(Engine& e, const float* d_images, int N);

// rows of every per-image detection buffer: DETECTIONS_PER_IMG, or more when "detections_cap" is set -- upstream's kth-value rule keeps every
// detection whose score ties with the 100th, so an image can return more than DETECTIONS_PER_IMG; the extra rows hold those ties
int maskrcnn_det_cap(Engine& e) {
    const int dpi = (int)e.param("detections_per_img", 100);
    const int cap = (int)e.param("detections_cap", 0.0f);
    return cap > dpi ? cap : dpi;
}

// Anchors of one level for the current canvas: the host sets the A base anchors ("anchor_base.<l>", generate_anchors) and the stride
// ("anchor_stride<l>"); the grid is laid out on the device whenever the canvas changed (always under graph capture, so that a replayed graph
// never depends on which canvas ran last).  Runs on the main stream, after the forward's WAR wait on the previous forward's tail.
static int level_anchors(Engine& e, int l, int A, int gh, int gw, bool regen, const float** out) {
    const std::string ls = std::to_string(l);
    const RawBuf* base;
    TRY(need_tensor(e, "anchor_base." + ls, (int64_t)A * 16, &base));
    const int stride = (int)e.param("anchor_stride" + ls, 0.0f);
    if (stride <= 0) { set_error("anchor_stride" + ls + " not set"); return ISEGMI_ERR_STATE; }
    void* q;
    TRY(eng_buf(e, "anchors." + ls, (int64_t)gh * gw * A * 16, &q, 0, {(int64_t)gh * gw * A, 4}));
    if (regen) TRY(grid_anchors_launch((const float*)base->d, A, stride, gh, gw, (float*)q, e.stream));
    *out = (const float*)q;
    return ISEGMI_OK;
}

int maskrcnn_forward(Engine& e, const float* d_images, int N) {
    if (e.param("arch_c4", 0.0f) != 0.0f) return maskrcnn_c4_forward(e, d_images, N);
    const int H = e.cur_H, W = e.cur_W;
    const bool regen_anchors = e.capturing || e.anchor_H != H || e.anchor_W != W;
    if (H % 32 || W % 32) { set_error("Mask R-CNN input must be padded to a multiple of 32"); return ISEGMI_ERR_ARG; }
    e.cur = e.stream;
#define st e.cur
    eng_mark(e, "start");
    void* p;
    int* d_hw = (int*)e.last_hw_ptr;  // maskrcnn_set_image_hw

    const int dt = e.fp16 ? 1 : 0;  // storage type of everything after the stem
    Tensor x4, s, x;
    bool stem_fused = false;
    if (dt) {  // fp16: images are rounded to fp16 into a zero-haloed 4-channel buffer the stem kernel reads without bounds tests
        TRY(eng_act(e, "input4h", N, H + 6, (W + 7) & ~1, 4, &x4, 1));
        TRY(pad_c3_to_f16_halo_launch(d_images, N, H, W, x4.d, st));
        TRY(eng_input_consumed(e));
        TRY(eng_stem_pool_f16(e, "backbone.body.stem.conv1", x4, H, W, "pool", &x, &stem_fused));   // conv + BN + ReLU + max-pool in one launch
        if (!stem_fused) TRY(eng_conv_stem_f16(e, "backbone.body.stem.conv1", x4, H, W, "stem", &s));
    } else {
        TRY(eng_act(e, "input4", N, H, W, 4, &x4));
        TRY(pad_c3_c4_launch(d_images, (int64_t)N * H * W, x4.d, st));
        TRY(eng_input_consumed(e));
        TRY(eng_conv(e, "backbone.body.stem.conv1", x4, 2, 3, 1, nullptr, "stem", &s));
    }
    if (!stem_fused) {
        const int Ho = (s.H + 2 - 3) / 2 + 1, Wo = (s.W + 2 - 3) / 2 + 1;
        TRY(eng_act(e, "pool", N, Ho, Wo, s.C, &x, dt));
        if (dt) TRY(maxpool_to_f16_launch(s.d, 1, N, s.H, s.W, s.C, 3, 2, 1, x.d, st));
        else TRY(maxpool_launch(s.d, N, s.H, s.W, s.C, 3, 2, 1, x.d, st));
    }
    eng_mark(e, "stem");
    const int depth = (int)e.param("resnet_depth", 50);
    const int blocks[4] = {3, 4, depth == 101 ? 23 : 6, 3};
    Tensor C[4];
    for (int li = 0; li < 4; ++li) {
        for (int b = 0; b < blocks[li]; ++b) {
            const std::string nm = "backbone.body.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int sd = (b == 0 && li > 0) ? 2 : 1;
            Tensor idt = x, t1, t2, y;
            // Buffers by LIVENESS, not by layer (round 3): a stage owns one t1, one t2, two alternating block outputs and its final output
            // C<l>.  Everything runs in order on the main stream (the shortcut of block 0 is joined before conv3), so a buffer's last reader
            // is always enqueued before its next writer.  Besides the memory (R101 bs=8: 33 x 3 buffers -> 4 x 5), a dead activation is now
            // overwritten while its lines still sit in the Infinity Cache instead of being written back to HBM behind the live traffic.
            const bool alias = e.param("alias_buffers", 1.0f) != 0.0f;  // 0: one buffer per layer output (rounds 1-2; kept for A/B)
            const std::string sg = alias ? "res" + std::to_string(li + 2) : nm;
            const std::string out_name = !alias ? nm + ".out" : b == blocks[li] - 1 ? sg + ".C" : sg + (b & 1 ? ".outB" : ".outA");
            bool fused = false;
            // fp16: the identity blocks of res2 / res3 and res2's first block (projection included) are ONE launch each, t1 / t2 stay in LDS
            if (b > 0 || sd == 1) TRY(eng_bottleneck_f16(e, nm, x, b == 0, out_name, &y, &fused));
            if (!fused) {
                const bool pair = b == 0 && !dt && e.param("conv_groups", 1.0f) != 0.0f && e.param("conv_tile", 0) == 0.0f;
                if (pair) {  // fp32: the projection shortcut and conv1 read the same x: one grouped launch (round 5) instead of a side stream
                    std::vector<ConvGroupItem> g(2);
                    g[0].layer = nm + ".conv1"; g[0].in = x; g[0].stride = sd; g[0].act = 1; g[0].out_name = sg + ".t1"; g[0].out = &t1;
                    g[1].layer = nm + ".downsample.0"; g[1].in = x; g[1].stride = sd; g[1].out_name = nm + ".ds"; g[1].out = &idt;
                    TRY(eng_conv_group(e, g));
                } else {
                if (b == 0) {  // projection shortcut on a side stream, concurrent with conv1 -> conv2
                    TRY(eng_fork(e, 0));
                    SideScope sc(e, 0);
                    TRY(eng_conv(e, nm + ".downsample.0", x, sd, 0, 0, nullptr, nm + ".ds", &idt));
                }
                TRY(eng_conv(e, nm + ".conv1", x, sd, 0, 1, nullptr, sg + ".t1", &t1, false, /*may_split=*/b > 0));  // STRIDE_IN_1X1
                }
                TRY(eng_conv(e, nm + ".conv2", t1, 1, 1, 1, nullptr, sg + ".t2", &t2, false, /*may_split=*/true));   // (`conv_split_k`: see eng_conv)
                if (b == 0 && !pair) TRY(eng_join(e, 0));
                TRY(eng_conv(e, nm + ".conv3", t2, 1, 0, 1, &idt, out_name, &y, false, /*may_split=*/true));
            }
            x = y;
        }
        C[li] = x;
        eng_mark(e, li == 0 ? "res2" : li == 1 ? "res3" : li == 2 ? "res4" : "res5");
    }

    // ---- FPN top-down chain (main stream); leaves (3x3 output convs) and the RPN follow per level
    // fp32 (round 5): the four lateral convs, the four output convs, the RPN head's 3x3 over the five levels and its cls + bbox 1x1 over the five levels are
    // ONE grouped launch each (eng_conv_group; bit-identical to the separate launches): 18 launches become 4, the small levels run inside the big levels'
    // launches.  Needs the batched RPN selection (which then takes all five levels at once); "conv_groups" 0 restores the per-level launches.
    const int pre_nms_g = (int)e.param("rpn_pre_nms_top_n", 1000);
    const bool grp = !dt && e.param("conv_groups", 1.0f) != 0.0f && e.param("conv_tile", 0) == 0.0f && (int)e.param("rpn_select_groups", -1.0f) != 0 &&
                     pre_nms_g > 256 && pre_nms_g <= 1024 && (int)e.param("rpn_post_nms_top_n", 1000) > 0;
    Tensor P[5], last[4], lat;
    if (grp) {
        Tensor lats[3];
        std::vector<ConvGroupItem> g(4);
        g[0].layer = "backbone.fpn.fpn_inner1"; g[0].in = C[0]; g[0].out_name = "fpn.lat1"; g[0].out = &lats[0];
        g[1].layer = "backbone.fpn.fpn_inner2"; g[1].in = C[1]; g[1].out_name = "fpn.lat2"; g[1].out = &lats[1];
        g[2].layer = "backbone.fpn.fpn_inner3"; g[2].in = C[2]; g[2].out_name = "fpn.lat3"; g[2].out = &lats[2];
        g[3].layer = "backbone.fpn.fpn_inner4"; g[3].in = C[3]; g[3].out_name = "fpn.last4"; g[3].out = &last[3];
        TRY(eng_conv_group(e, g));
        for (int l = 2; l >= 0; --l) {
            TRY(eng_act(e, "fpn.last" + std::to_string(l + 1), N, lats[l].H, lats[l].W, lats[l].C, &last[l], 0));
            TRY(nearest2x_add_launch(last[l + 1].d, N, last[l + 1].H, last[l + 1].W, last[l + 1].C, lats[l].d, lats[l].H, lats[l].W, last[l].d, st));
        }
    } else {
    TRY(eng_conv(e, "backbone.fpn.fpn_inner4", C[3], 1, 0, 0, nullptr, "fpn.last4", &last[3]));
    for (int l = 2; l >= 0; --l) {
        const std::string ls = std::to_string(l + 1);
        if (dt) {  // fp16: the merge with the coarser level in the lateral conv's own epilogue (one launch, the lateral tensor never exists)
            bool merged = false;
            TRY(eng_conv_up2x_f16(e, "backbone.fpn.fpn_inner" + ls, C[l], last[l + 1], "fpn.last" + ls, &last[l], &merged));
            if (merged) continue;
        }
        TRY(eng_conv(e, "backbone.fpn.fpn_inner" + ls, C[l], 1, 0, 0, nullptr, "fpn.lat" + ls, &lat));
        TRY(eng_act(e, "fpn.last" + ls, N, lat.H, lat.W, lat.C, &last[l], dt));
        if (dt) TRY(nearest2x_add_f16_launch(last[l + 1].d, N, last[l + 1].H, last[l + 1].W, last[l + 1].C, lat.d, lat.H, lat.W, last[l].d, st));
        else TRY(nearest2x_add_launch(last[l + 1].d, N, last[l + 1].H, last[l + 1].W, last[l + 1].C, lat.d, lat.H, lat.W, last[l].d, st));
    }
    }
    eng_mark(e, "fpn_topdown");

    // ---- RPN config
    const int A = 3, CH = 15, L = 5;
    const int pre_nms = (int)e.param("rpn_pre_nms_top_n", 1000), post_nms = (int)e.param("rpn_post_nms_top_n", 1000);
    const int fpn_post = (int)e.param("rpn_fpn_post_nms_top_n", 1000);
    const float rpn_thr = e.param("rpn_nms_thresh", 0.7f), rpn_min = e.param("rpn_min_size", 0.0f);
    const int ge = maskrcnn_nms_flags(e), roi_aligned = (int)e.param("roi_aligned", 0) ? 1 : 0;
    // per level up to 6144 (above 1024 the single-block NMS with 96 KB of boxes in LDS takes over from the chip-wide bitmask NMS); the merged
    // list feeds the per-class box NMS, whose suppression matrix holds 1024 proposals per image
    if (pre_nms > 6144 || post_nms > 6144 || fpn_post > 1024) { set_error("RPN: PRE / POST_NMS_TOP_N_TEST <= 6144 per level, FPN_POST_NMS_TOP_N_TEST <= 1024"); return ISEGMI_ERR_ARG; }
    float *cand_boxes, *cand_scores;
    int *cand_cnt, *cand_total;
    TRY(eng_buf(e, "rpn.cand_boxes", (int64_t)N * L * post_nms * 16, &p, 0, {N, L * post_nms, 4})); cand_boxes = (float*)p;
    TRY(eng_buf(e, "rpn.cand_scores", (int64_t)N * L * post_nms * 4, &p, 0, {N, L * post_nms})); cand_scores = (float*)p;
    TRY(eng_buf(e, "rpn.cand_cnt", (int64_t)N * L * 4, &p, 1, {N, L})); cand_cnt = (int*)p;
    TRY(eng_buf(e, "rpn.cand_total", (int64_t)N * 4, &p, 1, {N})); cand_total = (int*)p;

    // Per level, finest first: FPN output conv -> RPN head convs on the main stream, then that level's
    // selection (sigmoid, top-k, decode, NMS: small latency-bound grids) on a side stream so it hides under the
    // next level's convolutions.  P2's selection (201 600 anchors) gets all the remaining levels to hide under.
    // SELECTION BATCHED OVER (level, image) (round 5; SURVEY 2.1 "level x image as ONE batch dimension"): the levels' sigmoid, pre-NMS top-k, suppression
    // matrix and greedy scan run as five launches per GROUP of levels instead of four or five per level (22 launches, 520 us of latency-bound grids per
    // bs = 2 step).  Batches of two or more: one group of all five levels on the tail stream behind the last head conv (it hides under the next forward's
    // backbone anyway).  One image (latency): P2 -- 201 600 anchors, the longest chain -- goes first on a side stream under the other levels' convolutions,
    // P3..P6 as one group behind P6's head.  "rpn_select_groups": 0 = the per-level launches (A/B; also taken for pre_nms outside 257..1024), 1 / 2 force.
    const int sel_groups_param = (int)e.param("rpn_select_groups", -1.0f);
    const bool batched_sel = sel_groups_param != 0 && pre_nms > 256 && pre_nms <= 1024 && post_nms > 0;
    const int sel_groups = !batched_sel ? 0 : grp ? 1 : (sel_groups_param > 0 ? (sel_groups_param > 2 ? 2 : sel_groups_param) : (N >= 2 ? 1 : 2));
    const float* lvl_head[5]; const float* lvl_anc[5]; int lvl_hwa[5];
    auto select_group = [&](int l0, int l1) -> int {   // levels [l0, l1) on e.cur
        const int nl = l1 - l0;
        const std::string gs = std::to_string(l0) + "_" + std::to_string(l1);
        int slot[5];
        for (int l = l0; l < l1; ++l) slot[l - l0] = l;
        int64_t pe = 0, ce = 0;
        TRY(rpn_levels_workspace(nl, N, lvl_hwa + l0, pre_nms, &pe, &ce));
        void* q;
        float *prob, *cv, *tkv; int *ci, *tki, *tkc; void* nws;
        TRY(eng_buf(e, "rpn.g_prob" + gs, pe * 4, &q)); prob = (float*)q;
        TRY(eng_buf(e, "rpn.g_cand_vals" + gs, ce * 4, &q)); cv = (float*)q;
        TRY(eng_buf(e, "rpn.g_cand_idx" + gs, ce * 4, &q, 1)); ci = (int*)q;
        TRY(eng_buf(e, "rpn.g_tk_vals" + gs, (int64_t)nl * N * pre_nms * 4, &q)); tkv = (float*)q;
        TRY(eng_buf(e, "rpn.g_tk_idx" + gs, (int64_t)nl * N * pre_nms * 4, &q, 1)); tki = (int*)q;
        TRY(eng_buf(e, "rpn.g_tk_cnt" + gs, (int64_t)nl * N * 4, &q, 1)); tkc = (int*)q;
        TRY(eng_buf(e, "rpn.g_nms_ws" + gs, (int64_t)nl * N * 131072, &nws, 1));
        double bytes = 0;   // SURVEY 8d, as for the per-level form: logits once + deltas and anchors of the pre-NMS top-k + the kept boxes and scores
        for (int l = l0; l < l1; ++l) bytes += (double)N * ((double)lvl_hwa[l] * 4 + (double)(pre_nms < lvl_hwa[l] ? pre_nms : lvl_hwa[l]) * 32 + (double)post_nms * 20);
        OpScope op(e, st, "rpn_select (sigmoid + top-k + decode + NMS, per FPN level)", bytes);
        return rpn_levels_select_launch(nl, lvl_head + l0, lvl_anc + l0, lvl_hwa + l0, slot, d_hw, N, A, CH, pre_nms, post_nms, rpn_thr, rpn_min, ge, L, post_nms,
                                        prob, cv, ci, tkv, tki, tkc, nws, cand_boxes, cand_scores, cand_cnt, st);
    };
    auto rpn_level = [&](int l) -> int {
        const std::string ls = std::to_string(l);
        Tensor t, head;
        bool head_fused = false;   // fp16, big levels: the 3x3 and the fused cls + bbox 1x1 in one launch, t stays in LDS
        if (dt) TRY(eng_rpn_head_f16(e, "rpn.head.conv", "rpn.head.cls_bbox", P[l], "rpn.head" + ls, &head, &head_fused));
        if (!head_fused) {
            TRY(eng_conv(e, "rpn.head.conv", P[l], 1, 1, 1, nullptr, "rpn.t" + ls, &t));
            TRY(eng_conv(e, "rpn.head.cls_bbox", t, 1, 0, 0, nullptr, "rpn.head" + ls, &head, /*out_f32=*/true));
        }
        const int HW = head.H * head.W, HWA = HW * A;
        const float* anc;
        TRY(level_anchors(e, l, A, head.H, head.W, regen_anchors, &anc));
        lvl_head[l] = head.d; lvl_anc[l] = anc; lvl_hwa[l] = HWA;
        if (sel_groups == 1) return ISEGMI_OK;                 // all five levels: one group on the tail stream, below
        if (sel_groups == 2) {
            if (l != 0) return ISEGMI_OK;                      // P3..P6: one group behind P6's head, below
            if (e.multi_stream && !e.capturing) {              // P2: its own group, now, under the remaining levels' convolutions
                TRY(eng_fork(e, 0));
                SideScope sc(e, 0);
                return select_group(0, 1);
            }
            return select_group(0, 1);
        }
        float *prob, *tkv;
        int *tki, *tkc;
        void* q;
        TRY(eng_buf(e, "rpn.prob" + ls, (int64_t)N * HWA * 4, &q)); prob = (float*)q;
        TRY(eng_buf(e, "rpn.tk_vals" + ls, (int64_t)N * pre_nms * 4, &q)); tkv = (float*)q;
        TRY(eng_buf(e, "rpn.tk_idx" + ls, (int64_t)N * pre_nms * 4, &q, 1)); tki = (int*)q;
        TRY(eng_buf(e, "rpn.tk_cnt" + ls, (int64_t)N * 4, &q, 1)); tkc = (int*)q;
        void* nms_ws = nullptr;  // suppression matrix of the chip-wide NMS (one per level: the levels run concurrently)
        if (pre_nms <= 1024) TRY(eng_buf(e, "rpn.nms_ws" + ls, (int64_t)N * 131072, &nms_ws, 1));
        // candidate buffers of the two-level top-k (long rows): engine-owned, one pair per level (the levels run concurrently)
        float* tws_v = nullptr; int* tws_i = nullptr;
        if (const int64_t we = topk_scratch_elems(N, HWA, pre_nms)) {
            TRY(eng_buf(e, "rpn.tk_ws_vals" + ls, we * 4, &q)); tws_v = (float*)q;
            TRY(eng_buf(e, "rpn.tk_ws_idx" + ls, we * 4, &q, 1)); tws_i = (int*)q;
        }
        auto select = [&]() -> int {
            // logits once (4 B per anchor) + deltas and anchors of the pre-NMS top-k (16 B each) + the kept boxes and scores (20 B)
            OpScope op(e, st, "rpn_select (sigmoid + top-k + decode + NMS, per FPN level)",
                       (double)N * ((double)HWA * 4 + (double)(pre_nms < HWA ? pre_nms : HWA) * 32 + (double)post_nms * 20));
            TRY(rpn_sigmoid_launch(head.d, (int64_t)N * HWA, A, CH, prob, st));
            TRY(topk_launch_ws(prob, HWA, N, HWA, pre_nms, nullptr, 1, tkv, tki, tkc, tws_v, tws_i, st));
            TRY(rpn_decode_nms_launch(head.d, anc, tkv, tki, tkc, d_hw, N, HWA, A, CH, pre_nms, post_nms, rpn_thr, rpn_min,
                                      ge, l, L, post_nms, cand_boxes, cand_scores, cand_cnt, nms_ws, st));
            return ISEGMI_OK;
        };
        // Batches of two or more (throughput): the level's selection goes straight to the TAIL stream, which is idle here (the main stream waited for the
        // previous forward's RoI heads before the FPN output convs) and takes everything after the RPN convs anyway; the five levels' chains then run one
        // after the other under the remaining RPN convs and the next forward's backbone.  One image (latency): on three side streams, level l on stream
        // l % 3, in parallel.  Fewer busy streams = fewer collisions on the runtime's four in-order hardware queues: +0.8 % R50 fp32 bs=2, +1.1 % R101 fp16
        // bs=8, +2.1 % R50 fp16 bs=2; bs=1 p50 5.62 -> 5.79 ms if forced there (profiles/r03_experiments.txt 3e).  "rpn_select_on_tail": 1 / 0 force.
        const float sel_tail = e.param("rpn_select_on_tail", -1.0f);
        if (e.multi_stream && !e.capturing && (sel_tail < 0.0f ? N >= 2 : sel_tail != 0.0f)) {
            hipEvent_t ev;
            TRY(eng_next_event(e, &ev));
            HIP_TRY(hipEventRecord(ev, e.cur));
            HIP_TRY(hipStreamWaitEvent(e.tail, ev, 0));
            hipStream_t saved = e.cur;
            e.cur = e.tail;
            const int rc = select();
            e.cur = saved;
            return rc;
        }
        const int sk = l % 3;
        TRY(eng_fork(e, sk));
        SideScope sc(e, sk);
        return select();
    };
    // WAR: the previous forward's RoI heads (tail stream) still gather from P2..P5 and read proposals / det buffers;
    // everything up to here (backbone, top-down chain) was free to run underneath them.
    if (e.multi_stream && e.tail_pending && !e.capturing) HIP_TRY(hipStreamWaitEvent(e.stream, e.tail_done, 0));
    if (grp) {
        std::vector<ConvGroupItem> g(4), gt(5), gh(5);
        for (int l = 0; l < 4; ++l) {
            g[l].layer = "backbone.fpn.fpn_layer" + std::to_string(l + 1); g[l].in = last[l]; g[l].pad = 1; g[l].out_name = "P" + std::to_string(l + 2); g[l].out = &P[l];
        }
        TRY(eng_conv_group(e, g));
        const int Ho = (P[3].H - 1) / 2 + 1, Wo = (P[3].W - 1) / 2 + 1;
        TRY(eng_act(e, "P6", N, Ho, Wo, P[3].C, &P[4], 0));
        TRY(maxpool_launch(P[3].d, N, P[3].H, P[3].W, P[3].C, 1, 2, 0, P[4].d, st));
        Tensor t[5], head[5];
        for (int l = 0; l < 5; ++l) {
            gt[l].layer = "rpn.head.conv"; gt[l].in = P[l]; gt[l].pad = 1; gt[l].act = 1; gt[l].out_name = "rpn.t" + std::to_string(l); gt[l].out = &t[l];
        }
        TRY(eng_conv_group(e, gt));
        for (int l = 0; l < 5; ++l) {
            gh[l].layer = "rpn.head.cls_bbox"; gh[l].in = t[l]; gh[l].out_name = "rpn.head" + std::to_string(l); gh[l].out = &head[l]; gh[l].out_f32 = true;
        }
        TRY(eng_conv_group(e, gh));
        for (int l = 0; l < 5; ++l) {
            const float* anc;
            TRY(level_anchors(e, l, A, head[l].H, head[l].W, regen_anchors, &anc));
            lvl_head[l] = head[l].d; lvl_anc[l] = anc; lvl_hwa[l] = head[l].H * head[l].W * A;
        }
    } else {
    for (int l = 0; l < 4; ++l) {
        TRY(eng_conv(e, "backbone.fpn.fpn_layer" + std::to_string(l + 1), last[l], 1, 1, 0, nullptr, "P" + std::to_string(l + 2), &P[l]));
        TRY(rpn_level(l));
    }
    {
        const int Ho = (P[3].H - 1) / 2 + 1, Wo = (P[3].W - 1) / 2 + 1;
        TRY(eng_act(e, "P6", N, Ho, Wo, P[3].C, &P[4], dt));
        if (dt) TRY(maxpool_to_f16_launch(P[3].d, 1, N, P[3].H, P[3].W, P[3].C, 1, 2, 0, P[4].d, st));
        else TRY(maxpool_launch(P[3].d, N, P[3].H, P[3].W, P[3].C, 1, 2, 0, P[4].d, st));
        TRY(rpn_level(4));
    }
    }
    e.anchor_H = H; e.anchor_W = W;
    // From here on everything runs on the TAIL stream: it (not the main stream) joins the side streams, so the main
    // stream is free the moment its last RPN convolution is queued and the next forward's backbone starts underneath
    // the remaining per-level selection kernels, proposal merge and RoI heads (all latency-bound or small grids).
    if (e.multi_stream) {
        hipStream_t srcs[4] = {e.stream, e.side[0], e.side[1], e.side[2]};
        for (int i = 0; i < 4; ++i) {
            hipEvent_t ev;
            TRY(eng_next_event(e, &ev));
            HIP_TRY(hipEventRecord(ev, srcs[i]));
            HIP_TRY(hipStreamWaitEvent(e.tail, ev, 0));
        }
        e.cur = e.tail;
    }
    if (sel_groups == 1) TRY(select_group(0, L));
    else if (sel_groups == 2) TRY(select_group(1, L));
    eng_mark(e, "fpn_out+rpn");
    TRY(sum_counts_launch(cand_cnt, N, L, cand_total, st));
    float *fin_vals, *props, *prop_scores;
    int *fin_idx, *fin_cnt, *prop_cnt;
    const int R = fpn_post;
    TRY(eng_buf(e, "rpn.fin_vals", (int64_t)N * R * 4, &p)); fin_vals = (float*)p;
    TRY(eng_buf(e, "rpn.fin_idx", (int64_t)N * R * 4, &p, 1)); fin_idx = (int*)p;
    TRY(eng_buf(e, "rpn.fin_cnt", (int64_t)N * 4, &p, 1)); fin_cnt = (int*)p;
    TRY(eng_buf(e, "proposals", (int64_t)N * R * 16, &p, 0, {N, R, 4})); props = (float*)p;
    TRY(eng_buf(e, "proposal_scores", (int64_t)N * R * 4, &p, 0, {N, R})); prop_scores = (float*)p;
    TRY(eng_buf(e, "proposal_count", (int64_t)N * 4, &p, 1, {N})); prop_cnt = (int*)p;
    {
        OpScope op(e, st, "rpn_merge (top-k over the levels' candidates + gather)", (double)N * ((double)L * post_nms * 4 + (double)R * 40));
        TRY(topk_launch(cand_scores, (int64_t)L * post_nms, N, L * post_nms, R, cand_total, 1, fin_vals, fin_idx, fin_cnt, st));
        TRY(gather_proposals_launch(cand_boxes, fin_vals, fin_idx, fin_cnt, N, L * post_nms, R, props, prop_scores, prop_cnt, st));
    }
    eng_mark(e, "proposals");

    // ---- box head
    const float* feats[4] = {P[0].d, P[1].d, P[2].d, P[3].d};
    const int Hs[4] = {P[0].H, P[1].H, P[2].H, P[3].H}, Ws[4] = {P[0].W, P[1].W, P[2].W, P[3].W};
    const float scales[4] = {0.25f, 0.125f, 0.0625f, 0.03125f};
    Tensor roi7, f6, f7, cb;
    // "roi_table": bit 0 box head, bit 1 mask head -- RoIAlign as roi_prep (per-RoI sample table + launch order) and the table-driven channel-slice launch
    // instead of one workgroup per RoI in proposal order (same bits; A/B).  Same-box A/B, profiles/r05_experiments.txt 10: the box head gains in both
    // dtypes; the mask head's N x 100 RoIs gain in fp16 (125 -> 79 us) and lose the prep launch's time in fp32 (43 -> 52 us): default 3 fp16, 1 fp32.
    const int roi_table = (int)e.param("roi_table", dt ? 3.0f : 1.0f);
    TRY(eng_act(e, "box.roi_feat", N * R, 7, 7, 256, &roi7, dt));
    // SURVEY 8d "RoIAlign box: read rois + P2-P5 once (compulsory) + write the pooled features"
    double feat_bytes = 0;
    for (int l = 0; l < 4; ++l) feat_bytes += (double)N * Hs[l] * Ws[l] * 256 * (dt ? 2 : 4);
    {
        // (roi_prep is launched inside the stage's scope: its time counts against the same algorithmic bytes)
        OpScope op(e, st, "roi_align 7x7 (box head)", feat_bytes + (double)N * R * 20 + (double)N * R * 49 * 256 * (dt ? 2 : 4));
        int* order = nullptr;
        void* tab = nullptr;
        if ((roi_table & 1) && R <= 2048) {
            TRY(eng_buf(e, "roi_order", (int64_t)N * R * 4, &p, 1, {N, R}));
            order = (int*)p;
            TRY(eng_buf(e, "box.roi_table", (int64_t)N * R * 29 * 16, &tab, 1, {N * R, 29, 4}));
            TRY(roi_prep_launch(props, prop_cnt, N, R, Hs, Ws, scales, 4, 2, 256, 7, 7, dt ? 2 : 4, order, tab, st, roi_aligned));
        }
        if (dt) TRY(roi_align_f16_launch((const void* const*)feats, Hs, Ws, scales, 4, props, prop_cnt, N, R, 256, 7, 7, 2, 2, roi7.d, st, order, tab, roi_aligned));
        else TRY(roi_align_launch(feats, Hs, Ws, scales, 4, props, prop_cnt, N, R, 256, 7, 7, 2, 2, -1, roi7.d, nullptr, st, order, tab, roi_aligned));
    }
    TRY(eng_conv(e, "roi_heads.box.feature_extractor.fc6", roi7, 1, 0, 1, nullptr, "box.fc6", &f6));
    TRY(eng_conv(e, "roi_heads.box.feature_extractor.fc7", f6, 1, 0, 1, nullptr, "box.fc7", &f7));
    TRY(eng_conv(e, "roi_heads.box.predictor.cls_bbox", f7, 1, 0, 0, nullptr, "box.cls_bbox", &cb, /*out_f32=*/true));
    const int ncls = 81, dpi = (int)e.param("detections_per_img", 100), cap = maskrcnn_det_cap(e);
    if (cb.C != ncls * 5) { set_error("cls_bbox layer must have 81+324 outputs"); return ISEGMI_ERR_STATE; }
    isegmi_box_post_args a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.R = R; a.ncls = ncls; a.det_per_img = dpi; a.cap = cap; a.nms_flags = ge;
    a.score_thresh = e.param("roi_score_thresh", 0.05f);
    a.nms_thresh = e.param("roi_nms_thresh", 0.5f);
    a.logits_stride = cb.C; a.regr_stride = cb.C;
    a.d_logits = cb.d; a.d_regr = cb.d + ncls; a.d_props = props; a.d_prop_cnt = prop_cnt; a.d_image_hw = d_hw;
    TRY(eng_buf(e, "box.prob", (int64_t)N * R * ncls * 4, &p, 0, {N, R, ncls})); a.d_ws_prob = (float*)p;
    TRY(eng_buf(e, "box.cand_scores", (int64_t)N * (ncls - 1) * R * 4, &p)); a.d_ws_cand_scores = (float*)p;
    TRY(eng_buf(e, "box.cand_boxes", (int64_t)N * (ncls - 1) * R * 16, &p)); a.d_ws_cand_boxes = (float*)p;
    TRY(eng_buf(e, "box.kept_total", (int64_t)N * 4, &p, 1, {N})); a.d_ws_kept_total = (int*)p;
    TRY(eng_buf(e, "box.top_vals", (int64_t)N * dpi * 4, &p)); a.d_ws_top_vals = (float*)p;
    TRY(eng_buf(e, "box.top_idx", (int64_t)N * dpi * 4, &p, 1)); a.d_ws_top_idx = (int*)p;
    if ((int)e.param("box_nms_chip_wide", 1)) {   // crowded classes: suppression matrix on the whole chip (rcnn_ops.hip BoxCrowd); 0 = every class in its own block (A/B)
        TRY(eng_buf(e, "box.crowd_matrix", (int64_t)N * (ncls - 1) * 131072, &p, 1)); a.d_ws_crowd_matrix = p;
        TRY(eng_buf(e, "box.crowd_keys", (int64_t)N * (ncls - 1) * R * 8, &p, 1)); a.d_ws_crowd_keys = p;
        TRY(eng_buf(e, "box.crowd_boxes", (int64_t)N * (ncls - 1) * R * 16, &p)); a.d_ws_crowd_boxes = (float*)p;
        TRY(eng_buf(e, "box.crowd_m", (int64_t)N * (ncls - 1) * 4, &p, 1)); a.d_ws_crowd_m = (int*)p;
    }
    TRY(eng_buf(e, "det.count", (int64_t)N * 4, &p, 1, {N})); a.d_out_count = (int*)p;
    TRY(eng_buf(e, "det.box", (int64_t)N * cap * 16, &p, 0, {N, cap, 4})); a.d_out_boxes = (float*)p;
    TRY(eng_buf(e, "det.score", (int64_t)N * cap * 4, &p, 0, {N, cap})); a.d_out_scores = (float*)p;
    TRY(eng_buf(e, "det.label", (int64_t)N * cap * 4, &p, 1, {N, cap})); a.d_out_labels = (int*)p;
    {
        // logits + box regressions of every proposal once, the proposals, the detections out
        OpScope op(e, st, "box_postprocess (softmax + decode + per-class NMS + top-100)", (double)N * R * ((double)ncls * 5 * 4 + 16) + (double)N * cap * 24);
        TRY(box_postprocess_launch(&a, st));
    }
    eng_mark(e, "box_head");

    // ---- mask head
    Tensor m;
    TRY(eng_act(e, "mask.roi_feat", N * cap, 14, 14, 256, &m, dt));
    {
        OpScope op(e, st, "roi_align 14x14 (mask head)", feat_bytes + (double)N * cap * 20 + (double)N * cap * 196 * 256 * (dt ? 2 : 4));
        int* order = nullptr;
        void* tab = nullptr;
        if ((roi_table & 2) && cap <= 2048) {
            TRY(eng_buf(e, "mask.roi_order", (int64_t)N * cap * 4, &p, 1, {N, cap}));
            order = (int*)p;
            TRY(eng_buf(e, "mask.roi_table", (int64_t)N * cap * 57 * 16, &tab, 1, {N * cap, 57, 4}));
            TRY(roi_prep_launch(a.d_out_boxes, a.d_out_count, N, cap, Hs, Ws, scales, 4, 2, 256, 14, 14, dt ? 2 : 4, order, tab, st, roi_aligned));
        }
        if (dt) TRY(roi_align_f16_launch((const void* const*)feats, Hs, Ws, scales, 4, a.d_out_boxes, a.d_out_count, N, cap, 256, 14, 14, 2, 2, m.d, st, order, tab, roi_aligned));
        else TRY(roi_align_launch(feats, Hs, Ws, scales, 4, a.d_out_boxes, a.d_out_count, N, cap, 256, 14, 14, 2, 2, -1, m.d, nullptr, st, order, tab, roi_aligned));
    }
    for (int i = 1; i <= 4; ++i) {
        Tensor o;
        TRY(eng_conv(e, "roi_heads.mask.feature_extractor.mask_fcn" + std::to_string(i), m, 1, 1, 1, nullptr, "mask.fcn" + std::to_string(i), &o));
        m = o;
    }
    Tensor up;
    TRY(eng_act(e, "mask.deconv", N * cap, 28, 28, 256, &up, dt));
    {
        // ConvTranspose2d(2,2,s2): out[r, 2i+a, 2j+b, :] = W_ab * in[r, i, j, :] + bias  -> four strided 1x1 convs.
        Tensor rows;  // view: (r, i) as "images" of 1 x 14 pixels
        rows.d = m.d; rows.N = N * cap * 14; rows.H = 1; rows.W = 14; rows.C = 256; rows.dt = dt;
        std::vector<ConvGroupItem> g(4);   // the four parities are independent: one grouped launch under fp32 (eng_conv_group), four launches under fp16
        for (int ab = 0; ab < 4; ++ab) {
            const int aa = ab >> 1, bb = ab & 1;
            g[ab].layer = "roi_heads.mask.predictor.conv5_mask." + std::to_string(ab); g[ab].in = rows; g[ab].act = 1;
            g[ab].dst = (char*)up.d + (int64_t)(aa * 28 + bb) * 256 * (dt ? 2 : 4); g[ab].out_div = 14; g[ab].out_img_stride = (int64_t)2 * 28 * 256; g[ab].out_pix_stride = 2 * 256;
        }
        TRY(eng_conv_group(e, g));
    }
    const RawBuf *lw, *lb;
    TRY(need_tensor(e, "mask_logits.w", (int64_t)ncls * 256 * 4, &lw));
    TRY(need_tensor(e, "mask_logits.b", (int64_t)ncls * 4, &lb));
    TRY(eng_buf(e, "det.mask28", (int64_t)N * cap * 784 * 4, &p, 0, {N, cap, 28, 28}));
    {
        OpScope op(e, st, "mask_logits_select (1x1 -> the label's channel + sigmoid)", (double)N * cap * 784 * (256.0 * (dt ? 2 : 4) + 4));
        if (dt) TRY(mask_logits_select_f16_launch(up.d, N * cap, 784, 256, (const float*)lw->d, (const float*)lb->d, a.d_out_labels, (float*)p, st));
        else TRY(mask_logits_select_launch(up.d, N * cap, 784, 256, (const float*)lw->d, (const float*)lb->d, a.d_out_labels, (float*)p, st));
    }
    eng_mark(e, "mask_head");
    TRY(eng_tail_end(e));
    e.cur = e.stream;
    return ISEGMI_OK;
#undef st
}

// ---------------------------------------------------------------------------------------------------------------------
// e2e_mask_rcnn_R_50_C4_1x (the config README.md:263-273 prints): ResNet-50 conv1..conv4 -> one stride-16 map (1024 ch) ->
// RPNHead (3x3 1024->1024, 15 anchors: 5 sizes x 3 ratios, ratio-major) -> top PRE_NMS_TOP_N_TEST (6000) -> decode / clip /
// NMS 0.7 -> POST_NMS_TOP_N_TEST (1000), no cross-level merge -> ROIAlign 14x14, 1/16, sampling_ratio 0 (adaptive) ->
// ResNet conv5 head (3 bottlenecks, the first with stride 2) -> 7x7x2048 -> AvgPool 7 -> FastRCNNPredictor (81 | 324) ->
// box post-processing as in the FPN model -> the SAME extractor on the detections (SHARE_BOX_FEATURE_EXTRACTOR) ->
// MaskRCNNC4Predictor: ConvTranspose 2x2/2 2048->256 + ReLU -> 1x1 -> 81 -> sigmoid, class-selected: 14x14 masks.
// fp32 only; single stream (the RoI heads are ~1.5 TFLOP per image here and dominate).
static int res5_head(Engine& e, const std::string& prefix, const std::string& tag, const Tensor& in, Tensor* out) {
    Tensor x = in;
    for (int b = 0; b < 3; ++b) {
        const std::string nm = prefix + "." + std::to_string(b);
        const std::string bt = tag + "." + std::to_string(b);
        const int sd = b == 0 ? 2 : 1;
        Tensor idt = x, t1, t2, y;
        if (b == 0) TRY(eng_conv(e, nm + ".downsample.0", x, sd, 0, 0, nullptr, bt + ".ds", &idt));
        TRY(eng_conv(e, nm + ".conv1", x, sd, 0, 1, nullptr, bt + ".t1", &t1));  // STRIDE_IN_1X1
        TRY(eng_conv(e, nm + ".conv2", t1, 1, 1, 1, nullptr, bt + ".t2", &t2));
        TRY(eng_conv(e, nm + ".conv3", t2, 1, 0, 1, &idt, bt + ".out", &y));
        x = y;
    }
    *out = x;
    return ISEGMI_OK;
}

static int maskrcnn_c4_forward(Engine& e, const float* d_images, int N) {
    const int H = e.cur_H, W = e.cur_W;
    const bool regen_anchors = e.capturing || e.anchor_H != H || e.anchor_W != W;
    if (H % 16 || W % 16) { set_error("Mask R-CNN C4 input must be padded to a multiple of 16"); return ISEGMI_ERR_ARG; }
    if (e.fp16) { set_error("the C4 configuration is fp32 only"); return ISEGMI_ERR_STATE; }
    if (e.multi_stream && e.tail_pending && !e.capturing) HIP_TRY(hipStreamWaitEvent(e.stream, e.tail_done, 0));
    e.cur = e.stream;
    hipStream_t st = e.stream;
    eng_mark(e, "start");
    void* p;
    int* d_hw = (int*)e.last_hw_ptr;  // maskrcnn_set_image_hw
    Tensor x4, s, x;
    TRY(eng_act(e, "input4", N, H, W, 4, &x4));
    TRY(pad_c3_c4_launch(d_images, (int64_t)N * H * W, x4.d, st));
    TRY(eng_input_consumed(e));
    TRY(eng_conv(e, "backbone.body.stem.conv1", x4, 2, 3, 1, nullptr, "stem", &s));
    {
        const int Ho = (s.H + 2 - 3) / 2 + 1, Wo = (s.W + 2 - 3) / 2 + 1;
        TRY(eng_act(e, "pool", N, Ho, Wo, s.C, &x));
        TRY(maxpool_launch(s.d, N, s.H, s.W, s.C, 3, 2, 1, x.d, st));
    }
    const int blocks[3] = {3, 4, 6};
    for (int li = 0; li < 3; ++li)
        for (int b = 0; b < blocks[li]; ++b) {
            const std::string nm = "backbone.body.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int sd = (b == 0 && li > 0) ? 2 : 1;
            Tensor idt = x, t1, t2, y;
            if (b == 0) TRY(eng_conv(e, nm + ".downsample.0", x, sd, 0, 0, nullptr, nm + ".ds", &idt));
            TRY(eng_conv(e, nm + ".conv1", x, sd, 0, 1, nullptr, nm + ".t1", &t1));
            TRY(eng_conv(e, nm + ".conv2", t1, 1, 1, 1, nullptr, nm + ".t2", &t2));
            TRY(eng_conv(e, nm + ".conv3", t2, 1, 0, 1, &idt, nm + ".out", &y));
            x = y;
        }
    Tensor C4 = x;  // [N, H/16, W/16, 1024]
    {   // expose under a stable name for tests
        Tensor alias;
        TRY(eng_act(e, "C4", N, C4.H, C4.W, C4.C, &alias));
        HIP_TRY(hipMemcpyAsync(alias.d, C4.d, (size_t)C4.numel() * 4, hipMemcpyDeviceToDevice, st));
        C4 = alias;
    }
    eng_mark(e, "backbone");

    // ---- RPN on the single map
    const int A = 15, CH = 75;
    const int pre_nms = (int)e.param("rpn_pre_nms_top_n", 6000), post_nms = (int)e.param("rpn_post_nms_top_n", 1000);
    const float rpn_thr = e.param("rpn_nms_thresh", 0.7f), rpn_min = e.param("rpn_min_size", 0.0f);
    const int ge = maskrcnn_nms_flags(e), roi_aligned = (int)e.param("roi_aligned", 0) ? 1 : 0;
    if (pre_nms > 6144 || post_nms > 1024) { set_error("C4 RPN: pre_nms <= 6144, post_nms <= 1024"); return ISEGMI_ERR_ARG; }
    Tensor t, head;
    TRY(eng_conv(e, "rpn.head.conv", C4, 1, 1, 1, nullptr, "rpn.t", &t));
    TRY(eng_conv(e, "rpn.head.cls_bbox", t, 1, 0, 0, nullptr, "rpn.head", &head));
    if (head.C != CH) { set_error("C4 rpn.head.cls_bbox must have 15 + 60 outputs"); return ISEGMI_ERR_STATE; }
    const int HWA = head.H * head.W * A;
    const float* anc;
    TRY(level_anchors(e, 0, A, head.H, head.W, regen_anchors, &anc));
    e.anchor_H = H; e.anchor_W = W;
    const int R = post_nms;
    float *prob, *tkv, *props, *prop_scores;
    int *tki, *tkc, *prop_cnt;
    TRY(eng_buf(e, "rpn.prob", (int64_t)N * HWA * 4, &p)); prob = (float*)p;
    TRY(eng_buf(e, "rpn.tk_vals", (int64_t)N * pre_nms * 4, &p)); tkv = (float*)p;
    TRY(eng_buf(e, "rpn.tk_idx", (int64_t)N * pre_nms * 4, &p, 1)); tki = (int*)p;
    TRY(eng_buf(e, "rpn.tk_cnt", (int64_t)N * 4, &p, 1)); tkc = (int*)p;
    TRY(eng_buf(e, "proposals", (int64_t)N * R * 16, &p, 0, {N, R, 4})); props = (float*)p;
    TRY(eng_buf(e, "proposal_scores", (int64_t)N * R * 4, &p, 0, {N, R})); prop_scores = (float*)p;
    TRY(eng_buf(e, "proposal_count", (int64_t)N * 4, &p, 1, {N})); prop_cnt = (int*)p;
    TRY(rpn_sigmoid_launch(head.d, (int64_t)N * HWA, A, CH, prob, st));
    TRY(topk_launch(prob, HWA, N, HWA, pre_nms, nullptr, 1, tkv, tki, tkc, st));
    // one level: the NMS output (score order) IS the proposal list (select_over_all_levels only runs for > 1 level)
    void* nms_ws = nullptr;
    if (pre_nms <= 1024) TRY(eng_buf(e, "rpn.nms_ws", (int64_t)N * 131072, &nms_ws, 1));
    TRY(rpn_decode_nms_launch(head.d, anc, tkv, tki, tkc, d_hw, N, HWA, A, CH, pre_nms, post_nms, rpn_thr, rpn_min, ge, 0, 1,
                              R, props, prop_scores, prop_cnt, nms_ws, st));
    eng_mark(e, "rpn");

    // ---- box head: ROIAlign 14x14 (adaptive sampling) -> conv5 head -> avgpool -> predictors
    const float* feats[1] = {C4.d};
    const int Hs[1] = {C4.H}, Ws[1] = {C4.W};
    const float scales[1] = {0.0625f};
    Tensor roi, f5, cb;
    TRY(eng_act(e, "box.roi_feat", N * R, 14, 14, C4.C, &roi));
    TRY(roi_align_launch(feats, Hs, Ws, scales, 1, props, prop_cnt, N, R, C4.C, 14, 14, 0, 4, 0, roi.d, nullptr, st, nullptr, nullptr, roi_aligned));
    TRY(res5_head(e, "roi_heads.box.feature_extractor.head.layer4", "box.res5", roi, &f5));
    Tensor pooled;
    TRY(eng_act(e, "box.pooled", N * R, 1, 1, f5.C, &pooled));
    TRY(avgpool_full_launch(f5.d, (int64_t)N * R, f5.H * f5.W, f5.C, pooled.d, st));
    TRY(eng_conv(e, "roi_heads.box.predictor.cls_bbox", pooled, 1, 0, 0, nullptr, "box.cls_bbox", &cb));
    const int ncls = 81, dpi = (int)e.param("detections_per_img", 100), cap = maskrcnn_det_cap(e);
    if (cb.C != ncls * 5) { set_error("cls_bbox layer must have 81+324 outputs"); return ISEGMI_ERR_STATE; }
    isegmi_box_post_args a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.R = R; a.ncls = ncls; a.det_per_img = dpi; a.cap = cap; a.nms_flags = ge;
    a.score_thresh = e.param("roi_score_thresh", 0.05f);
    a.nms_thresh = e.param("roi_nms_thresh", 0.5f);
    a.logits_stride = cb.C; a.regr_stride = cb.C;
    a.d_logits = cb.d; a.d_regr = cb.d + ncls; a.d_props = props; a.d_prop_cnt = prop_cnt; a.d_image_hw = d_hw;
    TRY(eng_buf(e, "box.prob", (int64_t)N * R * ncls * 4, &p, 0, {N, R, ncls})); a.d_ws_prob = (float*)p;
    TRY(eng_buf(e, "box.cand_scores", (int64_t)N * (ncls - 1) * R * 4, &p)); a.d_ws_cand_scores = (float*)p;
    TRY(eng_buf(e, "box.cand_boxes", (int64_t)N * (ncls - 1) * R * 16, &p)); a.d_ws_cand_boxes = (float*)p;
    TRY(eng_buf(e, "box.kept_total", (int64_t)N * 4, &p, 1, {N})); a.d_ws_kept_total = (int*)p;
    TRY(eng_buf(e, "box.top_vals", (int64_t)N * dpi * 4, &p)); a.d_ws_top_vals = (float*)p;
    TRY(eng_buf(e, "box.top_idx", (int64_t)N * dpi * 4, &p, 1)); a.d_ws_top_idx = (int*)p;
    if ((int)e.param("box_nms_chip_wide", 1)) {   // crowded classes: suppression matrix on the whole chip (rcnn_ops.hip BoxCrowd); 0 = every class in its own block (A/B)
        TRY(eng_buf(e, "box.crowd_matrix", (int64_t)N * (ncls - 1) * 131072, &p, 1)); a.d_ws_crowd_matrix = p;
        TRY(eng_buf(e, "box.crowd_keys", (int64_t)N * (ncls - 1) * R * 8, &p, 1)); a.d_ws_crowd_keys = p;
        TRY(eng_buf(e, "box.crowd_boxes", (int64_t)N * (ncls - 1) * R * 16, &p)); a.d_ws_crowd_boxes = (float*)p;
        TRY(eng_buf(e, "box.crowd_m", (int64_t)N * (ncls - 1) * 4, &p, 1)); a.d_ws_crowd_m = (int*)p;
    }
    TRY(eng_buf(e, "det.count", (int64_t)N * 4, &p, 1, {N})); a.d_out_count = (int*)p;
    TRY(eng_buf(e, "det.box", (int64_t)N * cap * 16, &p, 0, {N, cap, 4})); a.d_out_boxes = (float*)p;
    TRY(eng_buf(e, "det.score", (int64_t)N * cap * 4, &p, 0, {N, cap})); a.d_out_scores = (float*)p;
    TRY(eng_buf(e, "det.label", (int64_t)N * cap * 4, &p, 1, {N, cap})); a.d_out_labels = (int*)p;
    TRY(box_postprocess_launch(&a, st));
    eng_mark(e, "box_head");

    // ---- mask head: the shared extractor on the detections, then MaskRCNNC4Predictor
    Tensor mroi, m5, up;
    TRY(eng_act(e, "mask.roi_feat", N * cap, 14, 14, C4.C, &mroi));
    TRY(roi_align_launch(feats, Hs, Ws, scales, 1, a.d_out_boxes, a.d_out_count, N, cap, C4.C, 14, 14, 0, 4, 0, mroi.d, nullptr, st, nullptr, nullptr, roi_aligned));
    TRY(res5_head(e, "roi_heads.box.feature_extractor.head.layer4", "mask.res5", mroi, &m5));
    TRY(eng_act(e, "mask.deconv", N * cap, 14, 14, 256, &up));
    {
        Tensor rows;  // view: (roi, i) as "images" of 1 x 7 pixels
        rows.d = m5.d; rows.N = N * cap * 7; rows.H = 1; rows.W = 7; rows.C = m5.C; rows.dt = 0;
        for (int ab = 0; ab < 4; ++ab) {
            const int aa = ab >> 1, bb = ab & 1;
            TRY(eng_conv_into(e, "roi_heads.mask.predictor.conv5_mask." + std::to_string(ab), rows, 1, 0, 1,
                              (char*)up.d + (int64_t)(aa * 14 + bb) * 256 * 4, 7, (int64_t)2 * 14 * 256, 2 * 256));
        }
    }
    const RawBuf *lw, *lb;
    TRY(need_tensor(e, "mask_logits.w", (int64_t)ncls * 256 * 4, &lw));
    TRY(need_tensor(e, "mask_logits.b", (int64_t)ncls * 4, &lb));
    TRY(eng_buf(e, "det.mask14", (int64_t)N * cap * 196 * 4, &p, 0, {N, cap, 14, 14}));
    TRY(mask_logits_select_launch(up.d, N * cap, 196, 256, (const float*)lw->d, (const float*)lb->d, a.d_out_labels, (float*)p, st));
    eng_mark(e, "mask_head");
    e.cur = e.stream;
    return ISEGMI_OK;
}

int maskrcnn_paste(Engine& e, const float* h_ratios_wh, int out_h, int out_w) {
    const int N = e.last_N;
    if (N <= 0) { set_error("paste before forward"); return ISEGMI_ERR_STATE; }
    const int cap = maskrcnn_det_cap(e);
    void *p, *rb, *rt;
    TRY(eng_buf(e, "ws.ratios", (int64_t)e.max_batch * 8, &rt));
    hipStream_t ps = (e.multi_stream && e.tail_pending) ? e.tail : e.stream;  // results stream of the last forward
    e.cur = ps;
    {
        // in stream order behind every earlier reader of ws.ratios (the previous paste ran on a results stream this one is ordered behind)
        std::vector<float> now(h_ratios_wh, h_ratios_wh + 2 * N);
        if (now != e.last_ratios || e.last_ratios_ptr != (const void*)rt) {
            TRY(eng_stage_small(e, h_ratios_wh, (size_t)N * 8, rt, ps));
            e.last_ratios = now;
            e.last_ratios_ptr = rt;
        }
    }
    TRY(eng_buf(e, "det.box_resized", (int64_t)N * cap * 16, &rb, 0, {N, cap, 4}));
    TRY(scale_boxes_launch((const float*)e.bufs["det.box"].d, (const float*)rt, N, cap, (float*)rb, ps));
    TRY(eng_buf(e, "det.masks", (int64_t)N * cap * out_h * out_w, &p, 2, {N, cap, out_h, out_w}));
    const bool c4 = e.param("arch_c4", 0.0f) != 0.0f;  // MaskRCNNC4Predictor emits 14x14 masks
    void* wq;
    TRY(eng_buf(e, "det.mask_window", (int64_t)e.max_batch * cap * 16, &wq, 1, {N, cap, 4}));
    // "sparse_masks": the planes are read through their windows only (isegmi_engine_rle: the device-side COCO output), so the 107 MB per image of
    // zero background need not be written; det.masks is then NOT a full binary plane (pixels outside a window are undefined)
    {
        // SURVEY 8d "Mask paste: read 0.31 MB, write <= 106 MB u8 per image": the planes are written whole unless sparse_masks (windows only: the
        // window bytes are data dependent and not counted here)
        const bool whole = e.param("sparse_masks", 0.0f) == 0.0f;
        OpScope op(e, ps, whole ? "paste_masks (Masker: resize + threshold + paste, whole uint8 planes)" : "paste_masks (sparse: box windows only)",
                   (double)N * cap * (c4 ? 196 : 784) * 4 + (whole ? (double)N * cap * out_h * out_w : 0.0));
        TRY(paste_masks_launch((const float*)e.bufs[c4 ? "det.mask14" : "det.mask28"].d, (const float*)rb, (const int*)e.bufs["det.count"].d, N,
                               cap, c4 ? 14 : 28, out_h, out_w, e.param("mask_threshold", 0.5f), (uint8_t*)p, ps, (int*)wq, whole));
    }
    if (ps == e.tail) HIP_TRY(hipEventRecord(e.tail_done, e.tail));
    e.cur = e.stream;
    eng_mark(e, "paste");
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_maskrcnn_forward_canvas(isegmi_engine* h, const float* d_images, const int32_t* h_image_hw, int N, int H, int W) {
    ARG_CHECK(h && d_images && h_image_hw, "null");
    ARG_CHECK(h->e.kind == 2, "engine is not a Mask R-CNN engine");
    ARG_CHECK(N > 0 && N <= h->e.max_batch, "batch size");
    ARG_CHECK(H > 0 && W > 0 && H <= h->e.H && W <= h->e.W, "canvas must fit inside the engine's maximum input size");
    for (int i = 0; i < N; ++i)
        ARG_CHECK(h_image_hw[2 * i] > 0 && h_image_hw[2 * i] <= H && h_image_hw[2 * i + 1] > 0 && h_image_hw[2 * i + 1] <= W,
                  "image_hw must fit inside the padded canvas");
    Engine& e = h->e;
    TRY(eng_wait_upload(e, d_images, (int64_t)N * H * W * 3 * 4, e.stream));
    TRY(maskrcnn_set_image_hw(e, h_image_hw, N));
    e.cur_H = H; e.cur_W = W;
    char key[160];  // everything a captured graph bakes in: batch, canvas, input pointer, and which of the two image_hw buffers is current
    snprintf(key, sizeof(key), "maskrcnn:%d:%dx%d:%p:%p", N, H, W, (const void*)d_images, e.last_hw_ptr);
    const bool graph_on = e.param("graph", 0.0f) != 0.0f;
    const int rc = eng_graph_run(e, key, [&]() { return maskrcnn_forward(e, d_images, N); });
    if (graph_on) { e.anchor_H = -1; e.anchor_W = -1; }  // a replay leaves ITS canvas's anchors behind, whatever the bookkeeping says
    e.cur = e.stream;
    if (rc == ISEGMI_OK) e.last_N = N;
    return rc;
}

extern "C" int isegmi_maskrcnn_forward(isegmi_engine* h, const float* d_images, const int32_t* h_image_hw, int N) {
    ARG_CHECK(h, "null");
    return isegmi_maskrcnn_forward_canvas(h, d_images, h_image_hw, N, h->e.H, h->e.W);
}

// h_ratios_wh [N][2] = (out_w / w_i, out_h / h_i) as float, computed by the host like BoxList.resize
extern "C" int isegmi_maskrcnn_paste(isegmi_engine* h, const float* h_ratios_wh, int out_h, int out_w) {
    ARG_CHECK(h && h_ratios_wh && out_h > 0 && out_w > 0, "paste args");
    ARG_CHECK(h->e.kind == 2, "engine is not a Mask R-CNN engine");
    return maskrcnn_paste(h->e, h_ratios_wh, out_h, out_w);
}

// One contiguous record block of the last Mask R-CNN forward for the all-gather:
//   [count i32 x N][box f32 x N*K*4][score f32 x N*K][label i32 x N*K][mask f32 x N*K*M*M]   (M = 28; 14 for the C4 predictor)
// (the 28x28 masks travel; the consumer pastes them).  D2D copies on the engine stream.
extern "C" int isegmi_maskrcnn_pack_records(isegmi_engine* h, void* d_dst, int64_t cap, int64_t* bytes) {
    ARG_CHECK(h && d_dst && bytes, "null");
    Engine& e = h->e;
    ARG_CHECK(e.kind == 2, "engine is not a Mask R-CNN engine");
    const int N = e.last_N;
    ARG_CHECK(N > 0, "pack before forward");
    const int K = maskrcnn_det_cap(e);
    hipStream_t rs = (e.multi_stream && e.tail_pending) ? e.tail : e.stream;  // results stream
    const bool c4 = e.param("arch_c4", 0.0f) != 0.0f;  // MaskRCNNC4Predictor: 14x14 masks
    const int64_t msz = c4 ? 196 : 784;
    const char* names[5] = {"det.count", "det.box", "det.score", "det.label", c4 ? "det.mask14" : "det.mask28"};
    const int64_t sizes[5] = {(int64_t)N * 4, (int64_t)N * K * 16, (int64_t)N * K * 4, (int64_t)N * K * 4, (int64_t)N * K * msz * 4};
    int64_t off = 0;
    for (int i = 0; i < 5; ++i) {
        ARG_CHECK(off + sizes[i] <= cap, "record buffer too small");
        HIP_TRY(hipMemcpyAsync((char*)d_dst + off, e.bufs[names[i]].d, (size_t)sizes[i], hipMemcpyDeviceToDevice, rs));
        off += sizes[i];
    }
    if (rs == e.tail) HIP_TRY(hipEventRecord(e.tail_done, e.tail));
    *bytes = off;
    return ISEGMI_OK;
}
