// engine.h -- model engine: packed-weight registry, named activation buffers, per-model forward graphs.
#pragma once
#include "rpn_levels.h"
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/isegmi.h"
#include "common.h"

namespace isegmi {

struct Tensor {
    float* d = nullptr;
    int N = 0, H = 0, W = 0, C = 0;
    int dt = 0;  // 0 fp32, 1 fp16 storage
    int64_t numel() const { return (int64_t)N * H * W * C; }
};

struct ConvLayer {
    int Cout = 0, R = 0, S = 0, Cin = 0;
    float* d_w = nullptr;      // packed (fp32 image, or fp16 image when f16)
    bool f16 = false;
    float* d_scale = nullptr;  // may be null (=1)
    float* d_shift = nullptr;  // may be null (=0)
};

struct RawBuf {
    void* d = nullptr;
    int64_t bytes = 0;
    std::vector<int64_t> shape;
    int dtype = 0;  // 0 f32, 1 i32, 2 u8, 3 i64
};

struct StageTime {
    std::string name;
    hipEvent_t ev;
};

struct Engine {
    int kind = 0;  // 1 yolact, 2 maskrcnn
    int max_batch = 0, H = 0, W = 0;       // H, W: the LARGEST network input (padded canvas) this engine serves
    int cur_H = 0, cur_W = 0;              // Mask R-CNN: padded canvas of the current forward (<= H, W; to_image_list pads each batch to its own size)
    int anchor_H = -1, anchor_W = -1;      // canvas the anchors.* buffers were generated for
    hipStream_t stream = nullptr;          // main stream (all results are complete on it when a forward returns)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};  // side streams for independent branches
    hipStream_t cur = nullptr;             // stream the next launch goes to
    hipStream_t tail = nullptr;            // Yolact Detect + postprocess: latency-bound tail, overlapped with the NEXT forward's backbone
    hipEvent_t tail_done = nullptr;        // recorded after the last tail launch; the next forward's head/proto writers wait on it
    bool tail_pending = false;
    // Yolact cross-step pipelining: FPN + protonet + prediction heads of step i run on their own stream group while step i+1's
    // backbone already runs on the main stream (the backbone chain of mid-size layers leaves CUs idle that the heads' large
    // layers fill).  lat_done fences the only backbone buffers the heads phase reads (C3-C5, by the lateral convs).
    hipStream_t heads = nullptr;
    hipStream_t hside[3] = {nullptr, nullptr, nullptr};
    hipEvent_t lat_done = nullptr;
    bool lat_pending = false;
    hipEvent_t heads_done = nullptr;       // end of the last pipelined heads phase (a non-pipelined forward's heads phase waits on it)
    bool heads_pending = false;
    bool multi_stream = true;
    // hipGraph replay of a forward ("graph" param): the ~130-150 launches of one forward are captured once per
    // (entry, batch, input pointer) and replayed with one hipGraphLaunch -- removes the per-launch gaps that bound bs=1
    // latency.  A replay ends joined on the main stream (no cross-step tail overlap), so it is a latency mode.
    bool capturing = false;
    std::map<std::string, hipGraphExec_t> graphs;
    std::map<std::string, int> graph_warm;
    int64_t graph_captures = 0, graph_replays = 0, graph_failures = 0;
    // asynchronous input upload (isegmi_engine_upload_async): pinned host -> device on a copy stream; in_done marks the point
    // where the last forward has consumed its input buffer (WAR for the next upload; recorded at the end of a hipGraph replay)
    hipStream_t copy = nullptr;
    hipEvent_t in_done = nullptr;
    bool in_pending = false;
    // one completion event per upload destination (input slot / uint8 staging buffer): the consumer of a buffer waits for ITS upload
    // only, so the copy of batch i+1 really overlaps forward i
    struct Upload { const char* dst = nullptr; int64_t bytes = 0; hipEvent_t done = nullptr; bool waited = true; };
    std::vector<Upload> uploads;
    // per-step completion marks on the results stream (isegmi_engine_mark_step / _step_times): true per-step latency samples
    std::vector<hipEvent_t> step_marks;
    // small host arrays that change from step to step (image sizes, resize ratios) reach the device through a ring of pinned slots: a
    // truly asynchronous copy on the consumer's own stream instead of a synchronous hipMemcpy of pageable memory
    char* pin_ring = nullptr;
    static constexpr int PIN_SLOTS = 32, PIN_SLOT_BYTES = 4096;
    hipEvent_t pin_ev[PIN_SLOTS] = {};
    bool pin_used[PIN_SLOTS] = {};
    int pin_next = 0;
    int hw_slot = 0;                       // Mask R-CNN image_hw lives in two device buffers used alternately (WAR against the previous forward's tail)
    // asynchronous download of a step's record block (device-side COCO output) on its own stream: two slots, like the RCCL records
    void* stream_set = nullptr;  // the pooled set the ten streams below belong to (engine.cpp: acquire_streams / release_streams)
    hipEvent_t dl_done[2] = {nullptr, nullptr};
    bool dl_used[2] = {false, false};
    bool fp16 = false;                     // fp16 storage + f16 MFMA convs (BASELINE configs[4]); set before loading weights
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;
    std::map<std::string, ConvLayer> convs;
    std::map<std::string, RawBuf> tensors;   // user-set constant tensors (priors, anchors, deconv weights ...)
    std::map<std::string, RawBuf> bufs;      // activations / workspaces / outputs, allocated on first use
    std::map<std::string, float> params;
    bool finalized = false;
    std::vector<int32_t> last_hw;      // host copies of the small per-batch inputs already resident on the device
    const void* last_hw_ptr = nullptr;
    std::vector<float> last_ratios;
    const void* last_ratios_ptr = nullptr;
    int last_N = 0;
    // timing
    bool timing = false;
    std::vector<StageTime> marks;
    std::vector<std::pair<std::string, float>> last_times;
    // per-conv-launch HIP-event timing (roofline measurement): accumulated until read
    bool conv_timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> conv_evs;
    std::vector<std::pair<std::string, double>> conv_ev_info;  // (label, flops) per pending event pair
    bool conv_trace = false;  // param "conv_trace": one stderr line per conv launch
    std::map<std::string, std::pair<double, double>> conv_layers;  // label -> (flops, ms) accumulated
    double conv_flops_pending = 0, conv_flops = 0, conv_ms = 0;
    int64_t conv_launches = 0;
    // HIP-event timing of the HBM-bound (non-conv) stages with their ALGORITHMIC bytes (SURVEY 8d "Algorithmic bytes"): param "op_timing";
    // isegmi_engine_op_stats accumulates and returns (label, us, bytes, launches) -- bench.py's roofline_hbm
    bool op_timing = false;
    struct OpEv { hipEvent_t a, b; const char* label; double bytes; };
    std::vector<OpEv> op_evs;
    struct OpStat { double us = 0, bytes = 0; int64_t launches = 0; };
    std::map<std::string, OpStat> op_stats;

    float param(const std::string& k, float def) const {
        auto it = params.find(k);
        return it == params.end() ? def : it->second;
    }
};

// engine.cpp helpers
int eng_buf(Engine& e, const std::string& name, int64_t bytes, void** out, int dtype = 0,
            std::vector<int64_t> shape = {});
int eng_act(Engine& e, const std::string& name, int N, int H, int W, int C, Tensor* t, int dt = 0);
int eng_conv(Engine& e, const std::string& layer, const Tensor& in, int stride, int pad, int act, const Tensor* residual,
             const std::string& out_name, Tensor* out, bool out_f32 = false, bool may_split = false);
// conv writing into a caller-provided strided destination (heads -> concatenated buffers, deconv parities)
int eng_graph_run(Engine& e, const std::string& key, const std::function<int()>& body);
void eng_graph_reset(Engine& e);
int eng_tail_end(Engine& e);
int eng_input_consumed(Engine& e);  // call right after the last kernel that reads the caller's input buffer
int eng_wait_upload(Engine& e, const void* d_ptr, int64_t bytes, hipStream_t st);  // st waits for every pending upload / front-end write into [d_ptr, d_ptr + bytes)
int eng_stage_small(Engine& e, const void* h_src, size_t bytes, void* d_dst, hipStream_t st);  // host array -> pinned ring slot -> async H2D on st
hipStream_t eng_results_stream(Engine& e);  // the stream the last forward's results complete on  // st waits for a pending upload_async whose destination holds d_ptr
int eng_bottleneck_f16(Engine& e, const std::string& block, const Tensor& x, bool first, const std::string& out_name, Tensor* out, bool* fused);
int eng_conv_up2x_f16(Engine& e, const std::string& layer, const Tensor& x, const Tensor& coarse, const std::string& out_name, Tensor* out, bool* merged);
// several independent convolutions of one point of the graph as ONE launch (fp32: conv2d_group_launch; fp16, a forced tile, a single member or
// "conv_groups" 0: one launch per member, as eng_conv / eng_conv_into would issue them)
struct ConvGroupItem {
    std::string layer;
    Tensor in;
    int stride = 1, pad = 0, act = 0;
    const Tensor* residual = nullptr;
    std::string out_name;            // named activation (eng_conv) ...
    Tensor* out = nullptr;
    void* dst = nullptr;             // ... or an explicit strided destination (eng_conv_into)
    int out_div = 0;
    int64_t out_img_stride = 0, out_pix_stride = 0;
    bool out_f32 = false;
};
int eng_conv_group(Engine& e, std::vector<ConvGroupItem>& items);
int conv2d_group_launch(int n, const isegmi_conv_desc* const* d, const float* const* in, const float* const* w, const float* const* scale,
                        const float* const* shift, const float* const* res, float* const* out, hipStream_t st);
int eng_rpn_head_f16(Engine& e, const std::string& conv, const std::string& headl, const Tensor& x, const std::string& out_name, Tensor* out, bool* fused);
int eng_stem_pool_f16(Engine& e, const std::string& layer, const Tensor& halo, int H, int W, const std::string& out_name, Tensor* out, bool* fused);
int eng_conv_stem_f16(Engine& e, const std::string& layer, const Tensor& halo, int H, int W, const std::string& out_name, Tensor* out);
int eng_conv_into(Engine& e, const std::string& layer, const Tensor& in, int stride, int pad, int act, void* dst, int out_div,
                  int64_t out_img_stride, int64_t out_pix_stride, bool out_f32 = false);
void eng_mark(Engine& e, const char* name);
// fork(k): side stream k waits for everything queued on the main stream so far; join(k): main waits for side k.
int eng_fork(Engine& e, int k);
int eng_join(Engine& e, int k);
// RAII: launches inside the scope go to side stream k (no-op when multi_stream is off)
struct SideScope {
    Engine& e;
    hipStream_t saved;
    SideScope(Engine& e_, int k) : e(e_), saved(e_.cur) { if (e.multi_stream) e.cur = e.side[k]; }
    ~SideScope() { e.cur = saved; }
};

// RAII: brackets the launches of one HBM-bound stage with HIP events on the stream they go to (no-op unless "op_timing" is set)
struct OpScope {
    Engine& e;
    hipStream_t st;
    const char* label;
    double bytes;
    hipEvent_t a = nullptr;
    OpScope(Engine& e_, hipStream_t st_, const char* label_, double bytes_) : e(e_), st(st_), label(label_), bytes(bytes_) {
        if (e.op_timing && !e.capturing && hipEventCreate(&a) == hipSuccess) (void)hipEventRecord(a, st);
    }
    ~OpScope() {
        if (!a) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) == hipSuccess) { (void)hipEventRecord(b, st); e.op_evs.push_back({a, b, label, bytes}); }
        else (void)hipEventDestroy(a);
    }
};

int yolact_forward(Engine& e, const float* d_images, int N);
int yolact_postprocess(Engine& e, int h, int w, const int32_t* h_image_hw = nullptr);
int maskrcnn_forward(Engine& e, const float* d_images, int N);
int eng_next_event(Engine& e, hipEvent_t* ev);
int maskrcnn_paste(Engine& e, const float* h_ratios_wh, int out_h, int out_w);

// kernels implemented in other translation units
int conv2d_launch(const isegmi_conv_desc* d, const float* in, const float* w, const float* scale, const float* shift,
                  const float* res, float* out, hipStream_t st);
int conv2d_f16_launch(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* res,
                      void* out, int out_f32, hipStream_t st);
int conv2d_f16_up2x_launch(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* coarse, int Hc, int Wc,
                           void* out, hipStream_t st);
int conv2d_f16_head_launch(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* w2,
                           const float* scale2, const float* shift2, int cout2, float* out2, bool* fused, hipStream_t st);
bool stem_pool_f16_supported(int Cout);
int stem_pool_f16_launch(int N, int H, int W, const void* halo, const void* w, const float* scale, const float* shift, void* out, int flags, hipStream_t st);
bool bottleneck_f16_supported(int Cin, int Cmid);
bool bottleneck_f16_ds_supported(int Cin, int Cmid);
int bottleneck_f16_launch(const isegmi_bottleneck_desc* d, const void* x, const void* w1, const float* s1, const float* b1, const void* w2,
                          const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* wd, const float* sd,
                          const float* bd, void* out, hipStream_t st);
int pad_c3_to_f16_halo_launch(const float* in, int N, int H, int W, void* out, hipStream_t st);
int avgpool_full_launch(const float* x, int64_t R, int HW, int C, float* out, hipStream_t st);
int resize_bilinear_f16_launch(const void* in, int N, int H, int W, int C, int Ho, int Wo, const void* add, int relu, void* out, hipStream_t st);
int maxpool_to_f16_launch(const void* in, int in_f16, int N, int H, int W, int C, int k, int s, int p, void* out, hipStream_t st);
int nearest2x_add_f16_launch(const void* coarse, int N, int Hc, int Wc, int C, const void* lat, int H, int W, void* out, hipStream_t st);
int roi_align_f16_launch(const void* const* feats, const int* Hs, const int* Ws, const float* scales, int nlevels, const float* rois,
                         const int* counts, int N, int K, int C, int PH, int PW, int g, int k_min, void* out, hipStream_t st,
                         const int* order = nullptr, const void* tab = nullptr, int aligned = 0);
int roi_prep_launch(const float* rois, const int* counts, int N, int K, const int* Hs, const int* Ws, const float* scales, int nlevels, int k_min, int C,
                    int PH, int PW, int esize, int* order, void* tab, hipStream_t st, int aligned = 0);
int mask_logits_select_f16_launch(const void* feat, int R, int HW, int C, const float* w, const float* b, const int* labels, float* out,
                                  hipStream_t st);
int maxpool_launch(const float* in, int N, int H, int W, int C, int k, int s, int p, float* out, hipStream_t st);
int resize_bilinear_launch(const float* in, int N, int H, int W, int C, int Ho, int Wo, const float* add, int relu, float* out,
                           hipStream_t st);
int nearest2x_add_launch(const float* coarse, int N, int Hc, int Wc, int C, const float* lat, int H, int W, float* out,
                         hipStream_t st);
int maskiou_conv1_launch(const float* lo, int NK, int PH, int PW, const float* w, const float* b, float* out, hipStream_t st);
int maskiou_rescore_launch(const float* feat, int N, int K, int HW, int C, const int* cls, const float* score, const int* count,
                           float* out, hipStream_t st);
int deform_im2col_launch(const float* x, int N, int H, int W, int C, const float* om, int R, int S, int stride, int pad, int dil,
                         float* out, hipStream_t st);
int pad_c3_c4_launch(const float* in, int64_t npix, float* out, hipStream_t st);
int preprocess_u8_launch(const uint8_t* in, int N, int Hin, int Win, float* out, int Hout, int Wout, int Hpad, int Wpad, int64_t out_img_stride,
                         const float* mean3, const float* std3, int swap_rb, hipStream_t st);
int pad_c3_c32_launch(const float* in, int64_t npix, float* out, hipStream_t st);
int topk_launch(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit,
                float* out_vals, int* out_idx, int* out_cnt, hipStream_t st);
int yolact_detect_launch(const isegmi_yolact_detect_args* a, hipStream_t st);
int yolact_masks_launch(const float* proto, const float* coeffs, const float* boxes, const int* count, int N, int PH, int PW,
                        int mask_dim, int K, int h, int w, float* ws_lo, uint8_t* out_masks, int64_t* out_boxes, hipStream_t st,
                        const int* image_hw = nullptr, int* win = nullptr, bool clear = true, bool dense_lo = false);

}  // namespace isegmi

struct isegmi_engine {
    isegmi::Engine e;
};
