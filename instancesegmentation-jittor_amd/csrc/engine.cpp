// engine.cpp -- engine object behind the C ABI: weights, buffers, timings, and the Yolact graph.
//
// Yolact graph = SURVEY.md 8a Y2..Y7 (App. A.9): ResNet-50 (stride on the 3x3) -> FPN (bilinear
// top-down, relu'd 3x3 preds, two stride-2 downsamples) -> protonet on P3 -> shared prediction head
// on P3..P7 -> Detect -> postprocess.  BN is folded into the conv epilogue (scale, shift) by the
// Python host (isegmi/yolact.py) exactly once, in fp32.
#include <mutex>
#include "engine.h"

#include <string.h>

namespace isegmi {

int eng_buf(Engine& e, const std::string& name, int64_t bytes, void** out, int dtype, std::vector<int64_t> shape) {
    RawBuf& b = e.bufs[name];
    if (b.bytes < bytes) {
        if (e.capturing) { set_error("buffer " + name + " would be (re)allocated during graph capture"); return ISEGMI_ERR_STATE; }
        // a captured graph holds the old device pointers of every buffer it touches: a grown buffer (a larger canvas or batch after a smaller one;
        // the liveness-aliased res<l>.* buffers are shared by many layers) invalidates all of them -- they are re-captured on their next use
        if (b.d) {
            HIP_TRY(hipFree(b.d));  // (synchronises the device: no replay is in flight when the graphs go)
            if (!e.graphs.empty()) eng_graph_reset(e);
        }
        b.d = nullptr;
        HIP_TRY(hipMalloc(&b.d, (size_t)(bytes > 0 ? bytes : 16)));
        b.bytes = bytes;
    }
    b.dtype = dtype;
    b.shape = shape;
    *out = b.d;
    return ISEGMI_OK;
}

int eng_act(Engine& e, const std::string& name, int N, int H, int W, int C, Tensor* t, int dt) {
    void* p = nullptr;
    int rc = eng_buf(e, name, (int64_t)N * H * W * C * (dt ? 2 : 4), &p, dt ? 4 : 0, {N, H, W, C});
    if (rc) return rc;
    t->d = (float*)p; t->N = N; t->H = H; t->W = W; t->C = C; t->dt = dt;
    return ISEGMI_OK;
}

static int find_conv(Engine& e, const std::string& layer, const ConvLayer** out) {
    auto it = e.convs.find(layer);
    if (it == e.convs.end()) { set_error("conv layer not set: " + layer); return ISEGMI_ERR_STATE; }
    *out = &it->second;
    return ISEGMI_OK;
}

static int timed_conv(Engine& e, const std::string& label, const isegmi_conv_desc* d, const float* in, const ConvLayer* L, const float* res, void* out,
                      bool out_f32 = false) {
    hipEvent_t a = nullptr, b = nullptr;
    if (e.conv_timing) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, e.cur));
    }
    if (e.conv_trace) {  // dev tools (tools/conv_traffic.py): the launch order, to join per-dispatch counters with layers
        const int Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
        fprintf(stderr, "convlaunch\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", label.c_str(), d->N, d->H, d->W, d->Cin, d->Cout, d->R, d->stride, Ho * Wo * d->N,
                res ? 1 : 0);
    }
    int rc = L->f16 ? conv2d_f16_launch(d, in, L->d_w, L->d_scale, L->d_shift, res, out, out_f32 ? 1 : 0, e.cur)
                    : conv2d_launch(d, in, L->d_w, L->d_scale, L->d_shift, res, (float*)out, e.cur);
    if (e.conv_timing) {
        HIP_TRY(hipEventRecord(b, e.cur));
        e.conv_evs.push_back({a, b});
        const int Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
        const int cin_true = (d->Cin == 4 && d->R == 7) ? 3 : d->Cin;
        const double fl = 2.0 * d->N * Ho * Wo * (double)d->Cout * d->R * d->S * cin_true;
        e.conv_flops_pending += fl;
        char geo[160];
        snprintf(geo, sizeof(geo), "%s [M=%d K=%d Cout=%d %dx%d/%d]", label.c_str(), d->N * Ho * Wo, d->R * d->S * d->Cin, d->Cout, d->R, d->S, d->stride);
        e.conv_ev_info.push_back({geo, fl});
    }
    return rc;
}

int eng_conv_into(Engine& e, const std::string& layer, const Tensor& in, int stride, int pad, int act, void* dst, int out_div,
                  int64_t out_img_stride, int64_t out_pix_stride, bool out_f32) {
    const ConvLayer* L;
    int rc = find_conv(e, layer, &L);
    if (rc) return rc;
    if (L->Cin != in.C) { set_error("conv " + layer + ": Cin mismatch"); return ISEGMI_ERR_ARG; }
    if ((in.dt == 1) != L->f16) { set_error("conv " + layer + ": activation / weight precision mismatch"); return ISEGMI_ERR_STATE; }
    isegmi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = in.N; d.H = in.H; d.W = in.W; d.Cin = in.C; d.Cout = L->Cout; d.R = L->R; d.S = L->S; d.stride = stride; d.pad = pad;
    d.act = act; d.tile = (int)e.param("conv_tile", 0); d.out_div = out_div; d.out_img_stride = out_img_stride;
    d.out_pix_stride = out_pix_stride;
    return timed_conv(e, layer, &d, in.d, L, nullptr, dst, out_f32);
}

int eng_conv(Engine& e, const std::string& layer, const Tensor& in, int stride, int pad, int act, const Tensor* residual,
             const std::string& out_name, Tensor* out, bool out_f32, bool may_split) {
    const ConvLayer* L;
    int rc = find_conv(e, layer, &L);
    if (rc) return rc;
    if (L->Cin != in.C) { set_error("conv " + layer + ": Cin mismatch"); return ISEGMI_ERR_ARG; }
    if ((in.dt == 1) != L->f16) { set_error("conv " + layer + ": activation / weight precision mismatch"); return ISEGMI_ERR_STATE; }
    if (residual && residual->dt != in.dt) { set_error("conv " + layer + ": residual precision mismatch"); return ISEGMI_ERR_STATE; }
    const int Ho = (in.H + 2 * pad - L->R) / stride + 1, Wo = (in.W + 2 * pad - L->S) / stride + 1;
    rc = eng_act(e, out_name, in.N, Ho, Wo, L->Cout, out, (L->f16 && !out_f32) ? 1 : 0);
    if (rc) return rc;
    isegmi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = in.N; d.H = in.H; d.W = in.W; d.Cin = in.C; d.Cout = L->Cout; d.R = L->R; d.S = L->S; d.stride = stride; d.pad = pad;
    d.act = act; d.tile = (int)e.param("conv_tile", 0);
    // `conv_split_k` (default 0): the fixed-tree split-K evaluation (conv tile 15) for the backbone's bottleneck convolutions the caller
    // marks -- every conv2 / conv3 and the conv1 of a stage's later blocks -- where the shape rule takes them (small M, K >= 1024: a bs = 1 forward).  An
    // OPT-IN NUMERICS MODE: results are bit-exact against the oracle models run with conv_split_k (the same layers, the same rule), not against the default.
    if (may_split && !L->f16 && d.tile == 0 && (int)e.param("conv_split_k", 0) != 0 && isegmi_conv_split_qualifies(&d)) d.tile = 15;
    return timed_conv(e, layer, &d, in.d, L, residual ? residual->d : nullptr, out->d, out_f32);
}

int eng_conv_group(Engine& e, std::vector<ConvGroupItem>& items) {
    const int n = (int)items.size();
    if (n == 0) return ISEGMI_OK;
    bool grouped = n > 1 && n <= 10 && e.param("conv_groups", 1.0f) != 0.0f && e.param("conv_tile", 0) == 0.0f;
    std::vector<const ConvLayer*> Ls(n);
    for (int i = 0; i < n; ++i) {
        int rc = find_conv(e, items[i].layer, &Ls[i]);
        if (rc) return rc;
        if (Ls[i]->f16 || items[i].in.dt == 1 || (Ls[i]->Cin == 4 && Ls[i]->R == 7)) grouped = false;
    }
    if (!grouped) {
        for (auto& it : items) {
            int rc = it.dst ? eng_conv_into(e, it.layer, it.in, it.stride, it.pad, it.act, it.dst, it.out_div, it.out_img_stride, it.out_pix_stride, it.out_f32)
                            : eng_conv(e, it.layer, it.in, it.stride, it.pad, it.act, it.residual, it.out_name, it.out, it.out_f32);
            if (rc) return rc;
        }
        return ISEGMI_OK;
    }
    std::vector<isegmi_conv_desc> ds(n);
    std::vector<const isegmi_conv_desc*> dp(n);
    std::vector<const float*> in(n), w(n), sc(n), sh(n), rs(n);
    std::vector<float*> out(n);
    double fl = 0;
    std::string label = "group[";
    for (int i = 0; i < n; ++i) {
        ConvGroupItem& it = items[i];
        const ConvLayer* L = Ls[i];
        if (L->Cin != it.in.C) { set_error("conv " + it.layer + ": Cin mismatch"); return ISEGMI_ERR_ARG; }
        if (it.residual && it.residual->dt != it.in.dt) { set_error("conv " + it.layer + ": residual precision mismatch"); return ISEGMI_ERR_STATE; }
        isegmi_conv_desc& d = ds[i];
        memset(&d, 0, sizeof(d));
        d.N = it.in.N; d.H = it.in.H; d.W = it.in.W; d.Cin = it.in.C; d.Cout = L->Cout; d.R = L->R; d.S = L->S; d.stride = it.stride; d.pad = it.pad; d.act = it.act;
        const int Ho = (d.H + 2 * d.pad - d.R) / d.stride + 1, Wo = (d.W + 2 * d.pad - d.S) / d.stride + 1;
        if (it.dst) {
            d.out_div = it.out_div; d.out_img_stride = it.out_img_stride; d.out_pix_stride = it.out_pix_stride;
            out[i] = (float*)it.dst;
        } else {
            int rc = eng_act(e, it.out_name, d.N, Ho, Wo, L->Cout, it.out, 0);
            if (rc) return rc;
            out[i] = it.out->d;
        }
        dp[i] = &d; in[i] = it.in.d; w[i] = (const float*)L->d_w; sc[i] = L->d_scale; sh[i] = L->d_shift; rs[i] = it.residual ? it.residual->d : nullptr;
        fl += 2.0 * d.N * Ho * Wo * (double)d.Cout * d.R * d.S * d.Cin;
        if (e.conv_trace) fprintf(stderr, "convlaunch\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", (it.layer + (i == 0 ? ".group" : ".grouped")).c_str(), d.N, d.H, d.W, d.Cin,
                                  d.Cout, d.R, d.stride, Ho * Wo * d.N, rs[i] ? 1 : 0);
        if (i < 3) label += (i ? " " : "") + it.layer;
    }
    label += n > 3 ? " ... x" + std::to_string(n) + "]" : "]";
    hipEvent_t a = nullptr, b = nullptr;
    if (e.conv_timing) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, e.cur));
    }
    int rc = conv2d_group_launch(n, dp.data(), in.data(), w.data(), sc.data(), sh.data(), rs.data(), out.data(), e.cur);
    if (rc) return rc;
    if (e.conv_timing) {
        HIP_TRY(hipEventRecord(b, e.cur));
        e.conv_evs.push_back({a, b});
        e.conv_flops_pending += fl;
        e.conv_ev_info.push_back({label + " [" + std::to_string(n) + " convolutions, one launch]", fl});
    }
    return ISEGMI_OK;
}

// fp16 bottleneck `block` (conv1 / conv2 / conv3 [/ downsample.0] of an upstream-named ResNet block, stride 1) as ONE launch of the fused kernel
// (csrc/bottleneck_f16.hip) when the block's shape has one: identity blocks of res2 / res3, and -- `first` -- the first block of res2 with its
// projection shortcut.  *fused tells the caller whether it ran (else: three or four eng_conv calls).
int eng_bottleneck_f16(Engine& e, const std::string& block, const Tensor& x, bool first, const std::string& out_name, Tensor* out, bool* fused) {
    *fused = false;
    const float mode = e.param("fused_bottleneck", 1.0f);  // 1: every block that has a fused kernel (default); 2: identity blocks only; 0: none (A/B)
    if (x.dt != 1 || mode == 0.0f || (first && mode == 2.0f)) return ISEGMI_OK;
    auto c1 = e.convs.find(block + ".conv1"), c2 = e.convs.find(block + ".conv2"), c3 = e.convs.find(block + ".conv3"), cd = e.convs.find(block + ".downsample.0");
    if (c1 == e.convs.end() || c2 == e.convs.end() || c3 == e.convs.end() || (first && cd == e.convs.end())) return ISEGMI_OK;
    const ConvLayer &L1 = c1->second, &L2 = c2->second, &L3 = c3->second;
    const ConvLayer* LD = first ? &cd->second : nullptr;
    const int Cmid = L1.Cout, Cout = L3.Cout;
    if (!(L1.f16 && L2.f16 && L3.f16) || L1.Cin != x.C || Cout != 4 * Cmid || L2.Cin != Cmid || L2.Cout != Cmid || L3.Cin != Cmid || L1.R != 1 || L2.R != 3 ||
        L2.S != 3 || L3.R != 1 || !L1.d_scale || !L1.d_shift || !L2.d_scale || !L2.d_shift || !L3.d_scale || !L3.d_shift)
        return ISEGMI_OK;
    if (first) {
        if (!LD->f16 || LD->Cin != x.C || LD->Cout != Cout || LD->R != 1 || !LD->d_scale || !LD->d_shift || !bottleneck_f16_ds_supported(x.C, Cmid)) return ISEGMI_OK;
    } else if (Cout != x.C || !bottleneck_f16_supported(x.C, Cmid)) return ISEGMI_OK;
    if ((int64_t)x.N * x.H * x.W * Cout * 2 >= (1ll << 31)) return ISEGMI_OK;
    int rc = eng_act(e, out_name, x.N, x.H, x.W, Cout, out, 1);
    if (rc) return rc;
    if ((const void*)out->d == (const void*)x.d) { set_error("bottleneck " + block + ": in-place"); return ISEGMI_ERR_STATE; }
    isegmi_bottleneck_desc d;
    memset(&d, 0, sizeof(d));
    d.N = x.N; d.H = x.H; d.W = x.W; d.Cin = x.C; d.Cmid = Cmid;
    hipEvent_t a = nullptr, b = nullptr;
    if (e.conv_timing) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, e.cur));
    }
    const int M = x.N * x.H * x.W;
    if (e.conv_trace) fprintf(stderr, "convlaunch\t%s.fused\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", block.c_str(), x.N, x.H, x.W, x.C, Cout, 3, 1, M, 1);
    rc = bottleneck_f16_launch(&d, x.d, L1.d_w, L1.d_scale, L1.d_shift, L2.d_w, L2.d_scale, L2.d_shift, L3.d_w, L3.d_scale, L3.d_shift,
                               first ? LD->d_w : nullptr, first ? LD->d_scale : nullptr, first ? LD->d_shift : nullptr, out->d, e.cur);
    if (rc) return rc;
    if (e.conv_timing) {
        HIP_TRY(hipEventRecord(b, e.cur));
        e.conv_evs.push_back({a, b});
        // algorithmic FLOPs of the convolutions the launch stands for (the halo the fused kernel recomputes for conv1 is not counted)
        const double fl = 2.0 * M * ((double)x.C * Cmid + 9.0 * Cmid * Cmid + (double)Cmid * Cout + (first ? (double)x.C * Cout : 0.0));
        e.conv_flops_pending += fl;
        char geo[160];
        snprintf(geo, sizeof(geo), "%s.fused [M=%d Cin=%d Cmid=%d Cout=%d 1x1+3x3+1x1%s]", block.c_str(), M, x.C, Cmid, Cout, first ? "+proj" : "");
        e.conv_ev_info.push_back({geo, fl});
    }
    *fused = true;
    return ISEGMI_OK;
}

// FPN top-down step under fp16: out = lateral1x1(x) + nearest2x(coarse) as ONE launch (UP2X residual mode of the persistent conv kernel); bit-identical to the
// lateral conv followed by nearest2x_add_f16.  *merged = false (nothing launched) with `fused_fpn_merge` 0 or a forced conv tile.
int eng_conv_up2x_f16(Engine& e, const std::string& layer, const Tensor& x, const Tensor& coarse, const std::string& out_name, Tensor* out, bool* merged) {
    *merged = false;
    if (x.dt != 1 || coarse.dt != 1 || e.param("fused_fpn_merge", 1.0f) == 0.0f || e.param("conv_tile", 0) != 0.0f) return ISEGMI_OK;
    auto it = e.convs.find(layer);
    if (it == e.convs.end()) return ISEGMI_OK;
    const ConvLayer& L = it->second;
    if (!L.f16 || L.Cin != x.C || L.R != 1 || L.S != 1 || L.Cout != coarse.C || L.Cout % 8 != 0 || x.W < 8 || coarse.N != x.N) return ISEGMI_OK;
    int rc = eng_act(e, out_name, x.N, x.H, x.W, L.Cout, out, 1);
    if (rc) return rc;
    isegmi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = x.N; d.H = x.H; d.W = x.W; d.Cin = x.C; d.Cout = L.Cout; d.R = 1; d.S = 1; d.stride = 1; d.pad = 0; d.act = 0;
    hipEvent_t a = nullptr, b = nullptr;
    if (e.conv_timing) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, e.cur));
    }
    const int M = x.N * x.H * x.W;
    if (e.conv_trace) fprintf(stderr, "convlaunch\t%s.up2x\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", layer.c_str(), x.N, x.H, x.W, x.C, L.Cout, 1, 1, M, 1);
    rc = conv2d_f16_up2x_launch(&d, x.d, L.d_w, L.d_scale, L.d_shift, coarse.d, coarse.H, coarse.W, out->d, e.cur);
    if (rc) return rc;
    if (e.conv_timing) {
        HIP_TRY(hipEventRecord(b, e.cur));
        e.conv_evs.push_back({a, b});
        const double fl = 2.0 * M * (double)x.C * L.Cout;
        e.conv_flops_pending += fl;
        char geo[200];
        snprintf(geo, sizeof(geo), "%s.up2x [M=%d K=%d Cout=%d 1x1/1 + nearest-2x top-down add]", layer.c_str(), M, x.C, L.Cout);
        e.conv_ev_info.push_back({geo, fl});
    }
    *merged = true;
    return ISEGMI_OK;
}

// RPNHead under fp16 (`t = relu(conv3x3(x)); head = cls_bbox(t)`) as ONE launch where the 3x3 runs on the 192 x 256 row-strip tile (csrc/conv_mfma_f16.hip,
// conv_f16_epilogue_head): t stays in LDS.  *fused = false (nothing launched) for levels too small for that tile, or with `fused_rpn_head` 0.
int eng_rpn_head_f16(Engine& e, const std::string& conv, const std::string& headl, const Tensor& x, const std::string& out_name, Tensor* out, bool* fused) {
    *fused = false;
    if (x.dt != 1 || e.param("fused_rpn_head", 1.0f) == 0.0f || e.param("conv_tile", 0) != 0.0f) return ISEGMI_OK;
    auto ic = e.convs.find(conv), ih = e.convs.find(headl);
    if (ic == e.convs.end() || ih == e.convs.end()) return ISEGMI_OK;
    const ConvLayer &L = ic->second, &H = ih->second;
    if (!L.f16 || !H.f16 || L.Cin != x.C || L.R != 3 || L.S != 3 || L.Cout != 256 || H.Cin != 256 || H.R != 1 || H.S != 1 || H.Cout > 32) return ISEGMI_OK;
    int rc = eng_act(e, out_name, x.N, x.H, x.W, H.Cout, out, 0);
    if (rc) return rc;
    isegmi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = x.N; d.H = x.H; d.W = x.W; d.Cin = x.C; d.Cout = 256; d.R = 3; d.S = 3; d.stride = 1; d.pad = 1; d.act = 1;
    hipEvent_t a = nullptr, b = nullptr;
    if (e.conv_timing) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, e.cur));
    }
    rc = conv2d_f16_head_launch(&d, x.d, L.d_w, L.d_scale, L.d_shift, H.d_w, H.d_scale, H.d_shift, H.Cout, out->d, fused, e.cur);
    if (rc) return rc;
    if (!*fused) {
        if (a) { HIP_TRY(hipEventDestroy(a)); HIP_TRY(hipEventDestroy(b)); }
        return ISEGMI_OK;
    }
    const int M = x.N * x.H * x.W;
    if (e.conv_trace) fprintf(stderr, "convlaunch\t%s+%s.fused\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", conv.c_str(), headl.c_str(), x.N, x.H, x.W, x.C, H.Cout, 3, 1, M, 0);
    if (e.conv_timing) {
        HIP_TRY(hipEventRecord(b, e.cur));
        e.conv_evs.push_back({a, b});
        const double fl = 2.0 * M * (9.0 * x.C * 256 + 256.0 * H.Cout);   // both convolutions' algorithmic FLOPs
        e.conv_flops_pending += fl;
        char geo[200];
        snprintf(geo, sizeof(geo), "%s+%s.fused [M=%d K=%d Cout=256 3x3/1 + 1x1 -> %d]", conv.c_str(), headl.c_str(), M, 9 * x.C, H.Cout);
        e.conv_ev_info.push_back({geo, fl});
    }
    return ISEGMI_OK;
}

// fp16 stem conv + BN + ReLU + 3x3/2 max-pool as ONE launch (csrc/stem_pool_f16.hip); *fused = false (nothing launched) when the layer is not the 64-channel
// stem or `fused_stem` is 0 (A/B against the two launches)
int eng_stem_pool_f16(Engine& e, const std::string& layer, const Tensor& halo, int H, int W, const std::string& out_name, Tensor* out, bool* fused) {
    *fused = false;
    if (e.param("fused_stem", 1.0f) == 0.0f || halo.dt != 1) return ISEGMI_OK;
    auto it = e.convs.find(layer);
    if (it == e.convs.end()) return ISEGMI_OK;
    const ConvLayer& L = it->second;
    if (!L.f16 || L.Cin != 4 || L.R != 7 || L.S != 7 || !stem_pool_f16_supported(L.Cout) || !L.d_scale || !L.d_shift) return ISEGMI_OK;
    const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1, Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
    int rc = eng_act(e, out_name, halo.N, Hp, Wp, L.Cout, out, 1);
    if (rc) return rc;
    hipEvent_t a = nullptr, b = nullptr;
    if (e.conv_timing) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, e.cur));
    }
    const int M = halo.N * Hc * Wc;
    if (e.conv_trace) fprintf(stderr, "convlaunch\t%s.fused\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", layer.c_str(), halo.N, H, W, 4, L.Cout, 7, 2, M, 1);
    rc = stem_pool_f16_launch(halo.N, H, W, halo.d, L.d_w, L.d_scale, L.d_shift, out->d, 0, e.cur);
    if (rc) return rc;
    if (e.conv_timing) {
        HIP_TRY(hipEventRecord(b, e.cur));
        e.conv_evs.push_back({a, b});
        const double fl = 2.0 * M * 147.0 * L.Cout;   // the convolution's algorithmic FLOPs (7 x 7 x 3 taps), as the unfused layer counts them
        e.conv_flops_pending += fl;
        char geo[160];
        snprintf(geo, sizeof(geo), "%s.fused [M=%d K=196 Cout=%d 7x7/2 + maxpool 3x3/2]", layer.c_str(), M, L.Cout);
        e.conv_ev_info.push_back({geo, fl});
    }
    *fused = true;
    return ISEGMI_OK;
}

// fp16 stem: `halo` is the [N][H+6][(W+7)&~1][4] fp16 image of pad_c3_to_f16_halo; H, W the image size
int eng_conv_stem_f16(Engine& e, const std::string& layer, const Tensor& halo, int H, int W, const std::string& out_name, Tensor* out) {
    const ConvLayer* L;
    int rc = find_conv(e, layer, &L);
    if (rc) return rc;
    if (!L->f16 || L->Cin != 4 || L->R != 7 || L->S != 7 || halo.dt != 1) { set_error("conv " + layer + ": not an fp16 stem"); return ISEGMI_ERR_STATE; }
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    rc = eng_act(e, out_name, halo.N, Ho, Wo, L->Cout, out, 1);
    if (rc) return rc;
    isegmi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = halo.N; d.H = H; d.W = W; d.Cin = 4; d.Cout = L->Cout; d.R = 7; d.S = 7; d.stride = 2; d.pad = 3; d.act = 1;
    return timed_conv(e, layer, &d, halo.d, L, nullptr, out->d, false);
}

int eng_next_event(Engine& e, hipEvent_t* ev) {
    if (e.ev_pool.empty()) {
        e.ev_pool.resize(128);
        for (auto& x : e.ev_pool) HIP_TRY(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    }
    *ev = e.ev_pool[e.ev_next++ % e.ev_pool.size()];
    return ISEGMI_OK;
}
int eng_fork(Engine& e, int k) {
    if (!e.multi_stream) return ISEGMI_OK;
    hipEvent_t ev;
    int rc = eng_next_event(e, &ev);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev, e.stream));
    HIP_TRY(hipStreamWaitEvent(e.side[k], ev, 0));
    return ISEGMI_OK;
}
int eng_join(Engine& e, int k) {
    if (!e.multi_stream) return ISEGMI_OK;
    hipEvent_t ev;
    int rc = eng_next_event(e, &ev);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev, e.side[k]));
    HIP_TRY(hipStreamWaitEvent(e.stream, ev, 0));
    return ISEGMI_OK;
}

// End of a forward's tail section.  Eager: leave the tail stream running (the next forward overlaps with it) and record
// the WAR fence.  Under graph capture: join the tail back into the main stream so the capture ends on one stream.
int eng_tail_end(Engine& e) {
    if (!e.multi_stream) return ISEGMI_OK;
    if (e.capturing) {
        hipEvent_t ev;
        int rc = eng_next_event(e, &ev);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ev, e.tail));
        HIP_TRY(hipStreamWaitEvent(e.stream, ev, 0));
        return ISEGMI_OK;
    }
    HIP_TRY(hipEventRecord(e.tail_done, e.tail));
    e.tail_pending = true;
    return ISEGMI_OK;
}

int eng_input_consumed(Engine& e) {
    if (e.capturing || e.in_done == nullptr) return ISEGMI_OK;
    HIP_TRY(hipEventRecord(e.in_done, e.cur));
    e.in_pending = true;
    return ISEGMI_OK;
}

// The consumer of an uploaded buffer (the front-end kernel for a uint8 staging buffer, the forward for an input slot) waits for the
// upload_async / front-end launches that filled [d_ptr, d_ptr + bytes) -- and only for those.
int eng_wait_upload(Engine& e, const void* d_ptr, int64_t bytes, hipStream_t st) {
    const char* p = (const char*)d_ptr;
    for (auto& u : e.uploads)
        if (!u.waited && p < u.dst + u.bytes && u.dst < p + bytes) { HIP_TRY(hipStreamWaitEvent(st, u.done, 0)); u.waited = true; }
    return ISEGMI_OK;
}

// Registers "the copy stream has, up to here, written [d_dst, d_dst + bytes)": one completion event per destination.
static int eng_note_upload(Engine& e, const void* d_dst, int64_t bytes) {
    Engine::Upload* u = nullptr;
    for (auto& x : e.uploads) if (x.dst == (const char*)d_dst) u = &x;
    if (!u) {
        if (e.uploads.size() >= 64) {  // destinations come and go (staging buffers are re-allocated when they grow): recycle a consumed entry
            for (auto& x : e.uploads) if (x.waited) { u = &x; break; }
            if (u == nullptr) { set_error("more than 64 upload destinations with unconsumed uploads"); return ISEGMI_ERR_STATE; }
        } else {
            e.uploads.emplace_back();
            u = &e.uploads.back();
            HIP_TRY(hipEventCreateWithFlags(&u->done, hipEventDisableTiming));
        }
        u->dst = (const char*)d_dst;
    }
    u->bytes = bytes;
    HIP_TRY(hipEventRecord(u->done, e.copy));
    u->waited = false;
    return ISEGMI_OK;
}

hipStream_t eng_results_stream(Engine& e) { return (e.multi_stream && e.tail_pending) ? e.tail : e.stream; }

int eng_stage_small(Engine& e, const void* h_src, size_t bytes, void* d_dst, hipStream_t st) {
    if (bytes > (size_t)Engine::PIN_SLOT_BYTES) { set_error("eng_stage_small: array larger than a pinned ring slot"); return ISEGMI_ERR_ARG; }
    if (e.capturing) { set_error("host array staged during graph capture"); return ISEGMI_ERR_STATE; }
    if (e.pin_ring == nullptr) HIP_TRY(hipHostMalloc((void**)&e.pin_ring, (size_t)Engine::PIN_SLOTS * Engine::PIN_SLOT_BYTES, hipHostMallocDefault));
    const int s = e.pin_next++ % Engine::PIN_SLOTS;
    if (e.pin_ev[s] == nullptr) HIP_TRY(hipEventCreateWithFlags(&e.pin_ev[s], hipEventDisableTiming));
    if (e.pin_used[s]) HIP_TRY(hipEventSynchronize(e.pin_ev[s]));  // the copy that last used this slot (PIN_SLOTS stagings ago) has long finished
    char* slot = e.pin_ring + (size_t)s * Engine::PIN_SLOT_BYTES;
    memcpy(slot, h_src, bytes);
    HIP_TRY(hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(e.pin_ev[s], st));
    e.pin_used[s] = true;
    return ISEGMI_OK;
}

void eng_graph_reset(Engine& e) {
    for (auto& kv : e.graphs) (void)hipGraphExecDestroy(kv.second);
    e.graphs.clear();
    e.graph_warm.clear();
}

// Run `body` (a forward's enqueue code) eagerly, or -- with the "graph" param set -- capture it on the second call with the
// same key and replay the instantiated graph from then on.  The first call stays eager so that every buffer exists.
int eng_graph_run(Engine& e, const std::string& key, const std::function<int()>& body) {
    const bool want = e.param("graph", 0.0f) != 0.0f && e.multi_stream && !e.timing && !e.conv_timing;
    if (!want) return body();
    auto it = e.graphs.find(key);
    if (it == e.graphs.end()) {
        int& warm = e.graph_warm[key];
        if (warm == 0) { warm = 1; return body(); }
        if (warm < 0) return body();  // capture failed before: stay eager
        if (e.tail_pending) { HIP_TRY(hipStreamWaitEvent(e.stream, e.tail_done, 0)); e.tail_pending = false; }
        HIP_TRY(hipStreamBeginCapture(e.stream, hipStreamCaptureModeRelaxed));
        e.capturing = true;
        const int rc = body();
        e.capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t er = hipStreamEndCapture(e.stream, &g);
        if (rc != ISEGMI_OK || er != hipSuccess || g == nullptr) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            warm = -1;
            ++e.graph_failures;
            for (int k = 0; k < 3; ++k) (void)hipStreamSynchronize(e.side[k]);
            (void)hipStreamSynchronize(e.tail);
            return body();
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) { (void)hipGetLastError(); warm = -1; ++e.graph_failures; return body(); }
        it = e.graphs.emplace(key, exec).first;
        ++e.graph_captures;
    }
    if (e.tail_pending) { HIP_TRY(hipStreamWaitEvent(e.stream, e.tail_done, 0)); e.tail_pending = false; }
    HIP_TRY(hipGraphLaunch(it->second, e.stream));
    ++e.graph_replays;
    // eng_input_consumed() is skipped under capture and never runs on a replay: the end of the graph is the (conservative) point
    // after which the next upload_async may overwrite the input / staging buffer
    if (e.in_done) { HIP_TRY(hipEventRecord(e.in_done, e.stream)); e.in_pending = true; }
    return ISEGMI_OK;
}

void eng_mark(Engine& e, const char* name) {
    if (!e.timing) return;
    StageTime s;
    s.name = name;
    if (hipEventCreate(&s.ev) != hipSuccess) return;
    (void)hipEventRecord(s.ev, e.cur ? e.cur : e.stream);
    e.marks.push_back(s);
}

static void collect_times(Engine& e) {
    for (size_t i = 0; i < e.conv_evs.size(); ++i) {
        auto& pr = e.conv_evs[i];
        float ms = 0;
        (void)hipEventElapsedTime(&ms, pr.first, pr.second);
        auto& acc = e.conv_layers[e.conv_ev_info[i].first];
        acc.first += e.conv_ev_info[i].second;
        acc.second += ms;
        e.conv_ms += ms;
        e.conv_launches += 1;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    e.conv_evs.clear();
    e.conv_ev_info.clear();
    e.conv_flops += e.conv_flops_pending;
    e.conv_flops_pending = 0;
    e.last_times.clear();
    if (!e.timing || e.marks.empty()) return;
    (void)hipEventSynchronize(e.marks.back().ev);
    for (size_t i = 1; i < e.marks.size(); ++i) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e.marks[i - 1].ev, e.marks[i].ev);
        e.last_times.push_back({e.marks[i].name, ms});
    }
    for (auto& m : e.marks) (void)hipEventDestroy(m.ev);
    e.marks.clear();
}

#define TRY(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) return _rc; \
    } while (0)

// Redirects everything the rest of a forward launches (main stream, side streams, current stream) to the heads stream group.
struct HeadsScope {
    Engine& e;
    hipStream_t s, sd[3];
    bool on = false;
    explicit HeadsScope(Engine& e_) : e(e_), s(e_.stream) { for (int k = 0; k < 3; ++k) sd[k] = e.side[k]; }
    // wide = the group's branches (three laterals, protonet || prediction heads per level) on the group's own three side streams; otherwise they
    // queue on the heads stream one after the other (see yolact_forward)
    void enter(bool wide) { on = true; e.stream = e.heads; for (int k = 0; k < 3; ++k) e.side[k] = wide ? e.hside[k] : e.heads; e.cur = e.heads; }
    ~HeadsScope() { if (on) { e.stream = s; for (int k = 0; k < 3; ++k) e.side[k] = sd[k]; e.cur = s; } }
};

int yolact_forward(Engine& e, const float* d_images, int N) {
    const int H = e.H, W = e.W;
    e.cur = e.stream;
    // cross-step pipelining of the heads phase (eager multi-stream throughput mode only)
    const bool pipe = e.multi_stream && !e.capturing && !e.timing && !e.conv_timing && e.heads != nullptr &&
                      e.param("graph", 0.0f) == 0.0f && e.param("pipeline_heads", 1.0f) != 0.0f;
    HeadsScope hscope(e);
    eng_mark(e, "start");
    const int dt = e.fp16 ? 1 : 0;  // fp16 storage + f16 MFMA convolutions (optional mode; heads / prototypes / Detect stay fp32)
    if (dt && e.convs.count("prediction_layers.0.head_cat") == 0) { set_error("fp16 Yolact needs the fused prediction head"); return ISEGMI_ERR_STATE; }
    Tensor x4;
    Tensor s, x;
    bool stem_fused = false;
    Tensor outs[5];
    const bool darknet = e.param("darknet", 0.0f) != 0.0f;  // yolact_darknet53_config: DarkNetBackbone([1, 2, 8, 8, 4]), selected layers 2-4
    if (darknet) {
        if (dt) { set_error("the Darknet53 backbone runs in fp32 only"); return ISEGMI_ERR_STATE; }
        // _preconv: 3x3 on the 3-channel image, via a zero-padded 32-channel copy; every conv is Conv + BN + LeakyReLU(0.1), a
        // block is 1x1 (C -> C/2) then 3x3 (C/2 -> C) with the shortcut added AFTER the activation (act 4)
        TRY(eng_act(e, "input32", N, H, W, 32, &x4));
        TRY(pad_c3_c32_launch(d_images, (int64_t)N * H * W, x4.d, e.cur));
        TRY(eng_input_consumed(e));
        TRY(eng_conv(e, "backbone._preconv.0", x4, 1, 1, 3, nullptr, "stem", &x));
        eng_mark(e, "stem");
        const int nblk[5] = {1, 2, 8, 8, 4};
        for (int li = 0; li < 5; ++li) {
            const std::string ln = "backbone.layers." + std::to_string(li);
            // C3 (layer 2's output, then C4, C5) is about to be overwritten: the previous step's lateral convs must have read them
            if (li == 2 && e.lat_pending) HIP_TRY(hipStreamWaitEvent(e.stream, e.lat_done, 0));
            Tensor y;
            TRY(eng_conv(e, ln + ".0.0", x, 2, 1, 3, nullptr, ln + ".down", &y));
            x = y;
            for (int b = 1; b <= nblk[li]; ++b) {
                const std::string nm = ln + "." + std::to_string(b);
                Tensor t1;
                TRY(eng_conv(e, nm + ".conv1", x, 1, 0, 3, nullptr, nm + ".t1", &t1));
                TRY(eng_conv(e, nm + ".conv2", t1, 1, 1, 4, &x, nm + ".out", &y));
                x = y;
            }
            outs[li] = x;
            if (li >= 1) eng_mark(e, li == 1 ? "layer1" : li == 2 ? "layer2" : li == 3 ? "layer3" : "layer4");
        }
    } else {
    if (dt) {
        TRY(eng_act(e, "input4h", N, H + 6, (W + 7) & ~1, 4, &x4, 1));
        TRY(pad_c3_to_f16_halo_launch(d_images, N, H, W, x4.d, e.cur));
        TRY(eng_input_consumed(e));
        TRY(eng_stem_pool_f16(e, "backbone.conv1", x4, H, W, "pool", &x, &stem_fused));
        if (!stem_fused) TRY(eng_conv_stem_f16(e, "backbone.conv1", x4, H, W, "stem", &s));
    } else {
        TRY(eng_act(e, "input4", N, H, W, 4, &x4));
        TRY(pad_c3_c4_launch(d_images, (int64_t)N * H * W, x4.d, e.cur));
        TRY(eng_input_consumed(e));
        TRY(eng_conv(e, "backbone.conv1", x4, 2, 3, 1, nullptr, "stem", &s));
    }
    if (!stem_fused) {
        const int Ho = (s.H + 2 - 3) / 2 + 1, Wo = (s.W + 2 - 3) / 2 + 1;
        TRY(eng_act(e, "pool", N, Ho, Wo, s.C, &x, dt));
        if (dt) TRY(maxpool_to_f16_launch(s.d, 1, N, s.H, s.W, s.C, 3, 2, 1, x.d, e.cur));
        else TRY(maxpool_launch(s.d, N, s.H, s.W, s.C, 3, 2, 1, x.d, e.cur));
    }
    eng_mark(e, "stem");
    const int blocks[4] = {3, 4, (int)e.param("resnet_depth", 50) == 101 ? 23 : 6, 3};
    for (int li = 0; li < 4; ++li) {
        for (int b = 0; b < blocks[li]; ++b) {
            const std::string nm = "backbone.layers." + std::to_string(li) + "." + std::to_string(b);
            const int st = (b == 0 && li > 0) ? 2 : 1;
            Tensor idt = x, t1, t2, y;
            // buffers by liveness (see maskrcnn.cpp): one t1 / t2 per stage, two alternating block outputs, the stage's final output on its own
            // (C3-C5 are read by the lateral convs of the pipelined heads phase; the lat_done fence below guards exactly those)
            const bool alias = e.param("alias_buffers", 1.0f) != 0.0f;  // 0: one buffer per layer output (rounds 1-2; kept for A/B)
            const std::string sg = alias ? "res" + std::to_string(li + 2) : nm;
            const std::string out_name = !alias ? nm + ".out" : b == blocks[li] - 1 ? sg + ".C" : sg + (b & 1 ? ".outB" : ".outA");
            if (dt && (b > 0 || st == 1)) {  // fp16 identity blocks of res2 / res3 and res2's first block: one launch, t1 / t2 stay in LDS (csrc/bottleneck_f16.hip)
                if (li == 1 && b == blocks[1] - 1 && e.lat_pending) HIP_TRY(hipStreamWaitEvent(e.stream, e.lat_done, 0));  // (see below)
                bool fused = false;
                TRY(eng_bottleneck_f16(e, nm, x, b == 0, out_name, &y, &fused));
                if (fused) { x = y; continue; }
            }
            const bool pair = b == 0 && !dt && e.param("conv_groups", 1.0f) != 0.0f && e.param("conv_tile", 0) == 0.0f;
            if (pair) {  // fp32: the projection shortcut and conv1 read the same x: one grouped launch (round 5) instead of a side stream
                std::vector<ConvGroupItem> g(2);
                g[0].layer = nm + ".conv1"; g[0].in = x; g[0].act = 1; g[0].out_name = sg + ".t1"; g[0].out = &t1;
                g[1].layer = nm + ".downsample.0"; g[1].in = x; g[1].stride = st; g[1].out_name = nm + ".ds"; g[1].out = &idt;
                TRY(eng_conv_group(e, g));
            } else {
            if (b == 0) {  // the projection shortcut is independent of conv1 -> conv2: side stream
                TRY(eng_fork(e, 0));
                SideScope sc(e, 0);
                TRY(eng_conv(e, nm + ".downsample.0", x, st, 0, 0, nullptr, nm + ".ds", &idt));
            }
            TRY(eng_conv(e, nm + ".conv1", x, 1, 0, 1, nullptr, sg + ".t1", &t1, false, /*may_split=*/b > 0));
            }
            if (e.convs.count(nm + ".conv2.conv_offset_mask")) {
                // DCNv2 3x3 (YOLACT++ backbones): offsets + mask logits from a plain 3x3 -> the nine taps sampled into columns ->
                // the deformable conv proper as a 1x1 over 9*C channels (weights handed over in KRSC order, bias folded into BN)
                if (dt) { set_error("the DCNv2 backbones run in fp32 only"); return ISEGMI_ERR_STATE; }
                Tensor om, col;
                TRY(eng_conv(e, nm + ".conv2.conv_offset_mask", t1, st, 1, 0, nullptr, nm + ".om", &om));
                TRY(eng_act(e, nm + ".col", N, om.H, om.W, 9 * t1.C, &col));
                TRY(deform_im2col_launch((const float*)t1.d, N, t1.H, t1.W, t1.C, (const float*)om.d, 3, 3, st, 1, 1, (float*)col.d, e.cur));
                TRY(eng_conv(e, nm + ".conv2", col, 1, 0, 1, nullptr, sg + ".t2", &t2));
            } else {
                TRY(eng_conv(e, nm + ".conv2", t1, st, 1, 1, nullptr, sg + ".t2", &t2, false, /*may_split=*/true));
            }
            if (b == 0 && !pair) TRY(eng_join(e, 0));
            // C3 (then C4, C5) is about to be overwritten: the previous step's lateral convs, running on the heads streams, must
            // have read them (they are the first thing of that phase, so this wait practically never blocks)
            if (li == 1 && b == blocks[1] - 1 && e.lat_pending) HIP_TRY(hipStreamWaitEvent(e.stream, e.lat_done, 0));
            TRY(eng_conv(e, nm + ".conv3", t2, 1, 0, 1, &idt, out_name, &y, false, /*may_split=*/true));
            x = y;
        }
        outs[li] = x;
        eng_mark(e, li == 0 ? "layer1" : li == 1 ? "layer2" : li == 2 ? "layer3" : "layer4");
    }
    }
    const Tensor C3 = outs[darknet ? 2 : 1], C4 = outs[darknet ? 3 : 2], C5 = outs[darknet ? 4 : 3];
    if (pipe) {  // hand the rest of this forward to the heads stream group; the caller's next forward starts its backbone at once
        hipEvent_t ev;
        TRY(eng_next_event(e, &ev));
        HIP_TRY(hipEventRecord(ev, e.stream));
        HIP_TRY(hipStreamWaitEvent(e.heads, ev, 0));
        // Side streams for the heads group only where a forward is latency-bound (small batches: bs=1 p50 1.98 vs 2.24 ms).  At the bench batch two
        // concurrent streams of chip-filling convolutions (backbone i+1 || heads i) leave nothing for more streams to fill, and every extra stream
        // is one more for the runtime to fold onto its four in-order hardware queues, where a branch then waits behind kernels of the other group
        // it does not depend on: without them +1-4 % (Yolact fp32 / yolact_base / fp16 bs=8; profiles/r03_experiments.txt 3c).  Parameter
        // "heads_side_streams": 1 always, 0 never, default by batch size.
        const float hs = e.param("heads_side_streams", -1.0f);
        hscope.enter(hs < 0.0f ? N <= 2 : hs != 0.0f);
    } else if (e.heads_pending) {  // mode switch without a sync in between: an earlier pipelined heads phase writes the same buffers
        HIP_TRY(hipStreamWaitEvent(e.stream, e.heads_done, 0));
        e.heads_pending = false;
    }
    // FPN: the three laterals are independent; so are the three prediction convs.  fp32 (round 5): each trio is ONE grouped launch (eng_conv_group), and
    // so are the five levels' upfeature and head_cat convs below: 16 launches become 4, and the small levels run inside the big level's launch instead of
    // as 23-us launches of their own.  "conv_groups" 0 restores the per-layer launches (A/B; fp16 and the unfused head layout keep them anyway).
    const bool grp = !dt && e.param("conv_groups", 1.0f) != 0.0f && e.param("conv_tile", 0) == 0.0f && e.convs.count("prediction_layers.0.head_cat") != 0;
    Tensor l5, l4, l3, x4f, x3f, P[5];
    if (grp) {
        std::vector<ConvGroupItem> g(3);
        g[0].layer = "fpn.lat_layers.2"; g[0].in = C3; g[0].out_name = "fpn.lat3"; g[0].out = &l3;
        g[1].layer = "fpn.lat_layers.1"; g[1].in = C4; g[1].out_name = "fpn.lat4"; g[1].out = &l4;
        g[2].layer = "fpn.lat_layers.0"; g[2].in = C5; g[2].out_name = "fpn.lat5"; g[2].out = &l5;
        TRY(eng_conv_group(e, g));
    } else {
    TRY(eng_fork(e, 0));
    TRY(eng_fork(e, 1));
    { SideScope sc(e, 0); TRY(eng_conv(e, "fpn.lat_layers.1", C4, 1, 0, 0, nullptr, "fpn.lat4", &l4)); }
    { SideScope sc(e, 1); TRY(eng_conv(e, "fpn.lat_layers.2", C3, 1, 0, 0, nullptr, "fpn.lat3", &l3)); }
    TRY(eng_conv(e, "fpn.lat_layers.0", C5, 1, 0, 0, nullptr, "fpn.lat5", &l5));
    TRY(eng_join(e, 0));
    TRY(eng_join(e, 1));
    }
    if (pipe) { HIP_TRY(hipEventRecord(e.lat_done, e.stream)); e.lat_pending = true; }
    TRY(eng_act(e, "fpn.x4", N, l4.H, l4.W, l4.C, &x4f, dt));
    if (dt) TRY(resize_bilinear_f16_launch(l5.d, N, l5.H, l5.W, l5.C, l4.H, l4.W, l4.d, 0, x4f.d, e.cur));
    else TRY(resize_bilinear_launch(l5.d, N, l5.H, l5.W, l5.C, l4.H, l4.W, l4.d, 0, x4f.d, e.cur));
    TRY(eng_act(e, "fpn.x3", N, l3.H, l3.W, l3.C, &x3f, dt));
    if (dt) TRY(resize_bilinear_f16_launch(x4f.d, N, x4f.H, x4f.W, x4f.C, l3.H, l3.W, l3.d, 0, x3f.d, e.cur));
    else TRY(resize_bilinear_launch(x4f.d, N, x4f.H, x4f.W, x4f.C, l3.H, l3.W, l3.d, 0, x3f.d, e.cur));
    if (grp) {
        std::vector<ConvGroupItem> g(3);
        g[0].layer = "fpn.pred_layers.2"; g[0].in = x3f; g[0].out_name = "P3"; g[0].out = &P[0];
        g[1].layer = "fpn.pred_layers.1"; g[1].in = x4f; g[1].out_name = "P4"; g[1].out = &P[1];
        g[2].layer = "fpn.pred_layers.0"; g[2].in = l5; g[2].out_name = "P5"; g[2].out = &P[2];
        for (auto& it : g) { it.pad = 1; it.act = 1; }
        TRY(eng_conv_group(e, g));
        TRY(eng_conv(e, "fpn.downsample_layers.0", P[2], 2, 1, 0, nullptr, "P6", &P[3]));
        TRY(eng_conv(e, "fpn.downsample_layers.1", P[3], 2, 1, 0, nullptr, "P7", &P[4]));
    } else {
    TRY(eng_fork(e, 0));
    TRY(eng_fork(e, 1));
    {
        SideScope sc(e, 0);  // P5 -> P6 -> P7 chain
        TRY(eng_conv(e, "fpn.pred_layers.0", l5, 1, 1, 1, nullptr, "P5", &P[2]));
        TRY(eng_conv(e, "fpn.downsample_layers.0", P[2], 2, 1, 0, nullptr, "P6", &P[3]));
        TRY(eng_conv(e, "fpn.downsample_layers.1", P[3], 2, 1, 0, nullptr, "P7", &P[4]));
    }
    { SideScope sc(e, 1); TRY(eng_conv(e, "fpn.pred_layers.1", x4f, 1, 1, 1, nullptr, "P4", &P[1])); }
    TRY(eng_conv(e, "fpn.pred_layers.2", x3f, 1, 1, 1, nullptr, "P3", &P[0]));
    TRY(eng_join(e, 0));
    TRY(eng_join(e, 1));
    }
    eng_mark(e, "fpn");
    // shared prediction head geometry
    const int A = (int)e.param("num_priors", 3), ncls = 81, md = 32;  // 9 for YOLACT++ (3 scales x 3 aspect ratios per cell)
    int Ptot = 0, off[5];
    for (int l = 0; l < 5; ++l) { off[l] = Ptot; Ptot += P[l].H * P[l].W * A; }
    {
        auto it = e.tensors.find("priors");
        if (it == e.tensors.end() || it->second.bytes != (int64_t)Ptot * 16) { set_error("priors tensor missing or wrong size"); return ISEGMI_ERR_STATE; }
    }
    // The three prediction convs (bbox 12, conf 243, mask 96) run as ONE 351-wide convolution when the host supplied
    // the fused layer: 6 instead of 1+4+2 64-wide column tiles per pixel tile, 5 launches instead of 15.  Its output
    // row per pixel is [A x 4 loc | A x 81 conf | A x 32 mask(pre-tanh)]; Detect reads it in place (HeadLayout).
    const bool fused = e.convs.count("prediction_layers.0.head_cat") != 0;
    const int CH = A * (4 + ncls + md);
    void *loc = nullptr, *conf = nullptr, *mask = nullptr, *headcat = nullptr;
    if (fused) {
        TRY(eng_buf(e, "headcat", (int64_t)N * (Ptot / A) * CH * 4, &headcat, 0, {N, Ptot / A, CH}));
    } else {
        TRY(eng_buf(e, "loc", (int64_t)N * Ptot * 4 * 4, &loc, 0, {N, Ptot, 4}));
        TRY(eng_buf(e, "conf", (int64_t)N * Ptot * ncls * 4, &conf, 0, {N, Ptot, ncls}));
        TRY(eng_buf(e, "mask", (int64_t)N * Ptot * md * 4, &mask, 0, {N, Ptot, md}));
    }
    auto head_level = [&](int l) -> int {
        Tensor uf;
        const std::string ln = "head.up" + std::to_string(l);
        TRY(eng_conv(e, "prediction_layers.0.upfeature.0", P[l], 1, 1, 1, nullptr, ln, &uf));
        const int hw = uf.H * uf.W;
        if (fused) {
            TRY(eng_conv_into(e, "prediction_layers.0.head_cat", uf, 1, 1, 0, (float*)headcat + (int64_t)(off[l] / A) * CH, hw,
                              (int64_t)(Ptot / A) * CH, CH, /*out_f32=*/true));
            return ISEGMI_OK;
        }
        TRY(eng_conv_into(e, "prediction_layers.0.bbox_layer", uf, 1, 1, 0, (float*)loc + (int64_t)off[l] * 4, hw, (int64_t)Ptot * 4, A * 4));
        TRY(eng_conv_into(e, "prediction_layers.0.conf_layer", uf, 1, 1, 0, (float*)conf + (int64_t)off[l] * ncls, hw, (int64_t)Ptot * ncls, A * ncls));
        TRY(eng_conv_into(e, "prediction_layers.0.mask_layer", uf, 1, 1, 2, (float*)mask + (int64_t)off[l] * md, hw, (int64_t)Ptot * md, A * md));
        return ISEGMI_OK;
    };
    // WAR: the previous forward's Detect / postprocess (tail stream) still reads loc/conf/mask/proto and the det.*
    // buffers; everything before this point touched only backbone/FPN buffers and was free to overlap with it.
    if (e.multi_stream && e.tail_pending && !e.capturing) HIP_TRY(hipStreamWaitEvent(e.stream, e.tail_done, 0));
    // protonet (side 0) || heads on P3 (main) || heads on P4,P6 (side 1) || heads on P5,P7 (side 2)
    Tensor proto;
    TRY(eng_fork(e, 0));
    TRY(eng_fork(e, 1));
    TRY(eng_fork(e, 2));
    {
        SideScope sc(e, 0);
        Tensor t, u;
        TRY(eng_conv(e, "proto_net.0", P[0], 1, 1, 1, nullptr, "proto.t0", &t));
        TRY(eng_conv(e, "proto_net.2", t, 1, 1, 1, nullptr, "proto.t1", &u));
        TRY(eng_conv(e, "proto_net.4", u, 1, 1, 1, nullptr, "proto.t2", &t));
        TRY(eng_act(e, "proto.up", N, t.H * 2, t.W * 2, t.C, &u, dt));
        if (dt) TRY(resize_bilinear_f16_launch(t.d, N, t.H, t.W, t.C, t.H * 2, t.W * 2, nullptr, 1, u.d, e.cur));
        else TRY(resize_bilinear_launch(t.d, N, t.H, t.W, t.C, t.H * 2, t.W * 2, nullptr, 1, u.d, e.cur));
        TRY(eng_conv(e, "proto_net.8", u, 1, 1, 1, nullptr, "proto.t3", &t));
        TRY(eng_conv(e, "proto_net.10", t, 1, 0, 1, nullptr, "proto", &proto, /*out_f32=*/true));
    }
    if (grp) {   // the shared head over all five levels: upfeature x 5 as one launch, head_cat x 5 as one launch (main stream; the protonet on side 0)
        Tensor uf[5];
        std::vector<ConvGroupItem> gu(5), gh(5);
        for (int l = 0; l < 5; ++l) {
            gu[l].layer = "prediction_layers.0.upfeature.0"; gu[l].in = P[l]; gu[l].pad = 1; gu[l].act = 1; gu[l].out_name = "head.up" + std::to_string(l); gu[l].out = &uf[l];
        }
        TRY(eng_conv_group(e, gu));
        for (int l = 0; l < 5; ++l) {
            gh[l].layer = "prediction_layers.0.head_cat"; gh[l].in = uf[l]; gh[l].pad = 1; gh[l].act = 0;
            gh[l].dst = (float*)headcat + (int64_t)(off[l] / A) * CH; gh[l].out_div = uf[l].H * uf[l].W;
            gh[l].out_img_stride = (int64_t)(Ptot / A) * CH; gh[l].out_pix_stride = CH; gh[l].out_f32 = true;
        }
        TRY(eng_conv_group(e, gh));
    } else {
    { SideScope sc(e, 1); TRY(head_level(1)); TRY(head_level(3)); }
    { SideScope sc(e, 2); TRY(head_level(2)); TRY(head_level(4)); }
    TRY(head_level(0));
    }
    TRY(eng_join(e, 0));
    TRY(eng_join(e, 1));
    TRY(eng_join(e, 2));
    eng_mark(e, "proto+heads");
    // Detect
    const int top_k = (int)e.param("nms_top_k", 200), max_det = (int)e.param("max_num_detections", 100);
    const int nc = ncls - 1;
    isegmi_yolact_detect_args a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.P = Ptot; a.ncls = ncls; a.mask_dim = md; a.top_k = top_k; a.max_det = max_det;
    a.conf_thresh = e.param("nms_conf_thresh", 0.05f);
    a.nms_thresh = e.param("nms_thresh", 0.5f);
    a.second_threshold = (int)e.param("nms_second_threshold", 0) ? 1 : 0;   // App. A.6 fork (fast_nms(second_threshold=...)): default off
    if (fused) {
        a.d_conf = a.d_loc = a.d_mask = (const float*)headcat;
        a.A = A; a.pix_stride = CH; a.off_loc = 0; a.off_conf = A * 4; a.off_mask = A * 4 + A * ncls; a.mask_tanh = 1;
    } else {
        a.d_conf = (const float*)conf; a.d_loc = (const float*)loc; a.d_mask = (const float*)mask;
    }
    a.d_priors = (const float*)e.tensors["priors"].d;
    void* p;
    TRY(eng_buf(e, "ws.scoresT", (int64_t)N * nc * Ptot * 4, &p)); a.d_ws_scoresT = (float*)p;
    TRY(eng_buf(e, "boxes_all", (int64_t)N * Ptot * 16, &p, 0, {N, Ptot, 4})); a.d_ws_boxes = (float*)p;
    TRY(eng_buf(e, "ws.counts", (int64_t)2 * N * 4, &p, 1)); a.d_ws_counts = (int32_t*)p;
    TRY(eng_buf(e, "ws.tk_vals", (int64_t)N * nc * top_k * 4, &p)); a.d_ws_tk_vals = (float*)p;
    TRY(eng_buf(e, "ws.tk_idx", (int64_t)N * nc * top_k * 4, &p, 1)); a.d_ws_tk_idx = (int32_t*)p;
    TRY(eng_buf(e, "ws.tk_cnt", (int64_t)N * nc * 4, &p, 1)); a.d_ws_tk_cnt = (int32_t*)p;
    TRY(eng_buf(e, "ws.cand", (int64_t)N * nc * top_k * 4, &p)); a.d_ws_cand = (float*)p;
    TRY(eng_buf(e, "ws.fin_vals", (int64_t)N * max_det * 4, &p)); a.d_ws_fin_vals = (float*)p;
    TRY(eng_buf(e, "ws.fin_idx", (int64_t)N * max_det * 4, &p, 1)); a.d_ws_fin_idx = (int32_t*)p;
    TRY(eng_buf(e, "ws.fin_cnt", (int64_t)N * 4, &p, 1)); a.d_ws_fin_cnt = (int32_t*)p;
    TRY(eng_buf(e, "det.count", (int64_t)N * 4, &p, 1, {N})); a.d_out_count = (int32_t*)p;
    TRY(eng_buf(e, "det.box", (int64_t)N * max_det * 16, &p, 0, {N, max_det, 4})); a.d_out_boxes = (float*)p;
    TRY(eng_buf(e, "det.score", (int64_t)N * max_det * 4, &p, 0, {N, max_det})); a.d_out_scores = (float*)p;
    TRY(eng_buf(e, "det.class", (int64_t)N * max_det * 4, &p, 1, {N, max_det})); a.d_out_classes = (int32_t*)p;
    TRY(eng_buf(e, "det.coeff", (int64_t)N * max_det * md * 4, &p, 0, {N, max_det, md})); a.d_out_coeffs = (float*)p;
    TRY(eng_buf(e, "det.prior", (int64_t)N * max_det * 4, &p, 1, {N, max_det})); a.d_out_prior = (int32_t*)p;
    // Detect is a chain of small latency-bound grids: run it (and postprocess) on the tail stream so the NEXT
    // forward's MFMA-bound backbone can start underneath it.
    hipStream_t ds = e.stream;
    if (e.multi_stream) {
        hipEvent_t ev;
        TRY(eng_next_event(e, &ev));
        HIP_TRY(hipEventRecord(ev, e.stream));
        HIP_TRY(hipStreamWaitEvent(e.tail, ev, 0));
        ds = e.tail;
    }
    {
        // confidences, box regressions and mask coefficients of every prior once (the fused head's [N][P][4 + 81 + 32] rows) + priors
        OpScope op(e, ds, "yolact_detect (softmax + decode + per-class top-k + fast-NMS + gather)", (double)N * Ptot * ((double)(4 + nc + md) * 4) + (double)Ptot * 16);
        TRY(yolact_detect_launch(&a, ds));
    }
    if (pipe) { HIP_TRY(hipEventRecord(e.heads_done, e.stream)); e.heads_pending = true; }
    TRY(eng_tail_end(e));
    eng_mark(e, "detect");
    return ISEGMI_OK;
}

// h_image_hw (optional, [N][2]): image n is assembled at its own (h_n, w_n) inside the common (h, w) plane
int yolact_postprocess(Engine& e, int h, int w, const int32_t* h_image_hw) {
    const int N = e.last_N;
    if (N <= 0) { set_error("postprocess before forward"); return ISEGMI_ERR_STATE; }
    const int K = (int)e.param("max_num_detections", 100);
    RawBuf& proto = e.bufs["proto"];
    const int PH = (int)proto.shape[1], PW = (int)proto.shape[2], md = (int)proto.shape[3];
    void *lo, *masks, *ib;
    hipStream_t rs = (e.multi_stream && e.tail_pending) ? e.tail : e.stream;  // results stream of the last forward
    int* d_ihw = nullptr;
    if (h_image_hw) {
        for (int i = 0; i < N; ++i)
            if (h_image_hw[2 * i] <= 0 || h_image_hw[2 * i] > h || h_image_hw[2 * i + 1] <= 0 || h_image_hw[2 * i + 1] > w) { set_error("postprocess: image size outside the plane"); return ISEGMI_ERR_ARG; }
        void* q;
        TRY(eng_buf(e, "pp.image_hw", (int64_t)e.max_batch * 8, &q, 1, {N, 2}));
        d_ihw = (int*)q;
        TRY(eng_stage_small(e, h_image_hw, (size_t)N * 8, d_ihw, rs));
    }
    TRY(eng_buf(e, "ws.lo", (int64_t)N * K * PH * PW * 4, &lo));
    TRY(eng_buf(e, "det.masks", (int64_t)N * K * h * w, &masks, 2, {N, K, h, w}));
    TRY(eng_buf(e, "det.box_int", (int64_t)N * K * 4 * 8, &ib, 3, {N, K, 4}));
    void* wq;
    TRY(eng_buf(e, "det.mask_window", (int64_t)e.max_batch * K * 16, &wq, 1, {N, K, 4}));
    {
        // SURVEY 8d "Yolact assembly: read the prototypes + coefficients, write n x h x w" (uint8 planes; whole unless sparse_masks)
        const bool whole = e.param("sparse_masks", 0.0f) == 0.0f;
        OpScope op(e, rs, whole ? "yolact_masks (proto @ coeff -> sigmoid -> crop -> upsample -> threshold, whole uint8 planes)" : "yolact_masks (sparse: box windows only)",
                   (double)N * PH * PW * md * 4 + (double)N * K * md * 4 + (whole ? (double)N * K * h * w : 0.0));
        TRY(yolact_masks_launch((const float*)proto.d, (const float*)e.bufs["det.coeff"].d, (const float*)e.bufs["det.box"].d,
                                (const int*)e.bufs["det.count"].d, N, PH, PW, md, K, h, w, (float*)lo, (uint8_t*)masks, (int64_t*)ib,
                                rs, d_ihw, (int*)wq, whole, /*dense_lo=*/e.convs.count("maskiou_net.2") != 0));
    }
    if (e.convs.count("maskiou_net.2")) {
        // YOLACT++ fast mask re-scoring on the proto-resolution masks just written to ws.lo: first layer (1 input channel) and
        // the global-max / class pick as small dedicated kernels, the rest on the MFMA conv kernels over all N*K slots
        auto w0 = e.tensors.find("maskiou.w0"), b0 = e.tensors.find("maskiou.b0");
        if (w0 == e.tensors.end() || b0 == e.tensors.end()) { set_error("maskiou_net.0 weights missing"); return ISEGMI_ERR_STATE; }
        hipStream_t saved = e.cur;
        e.cur = rs;
        Tensor t, u;
        TRY(eng_act(e, "maskiou.t0", N * K, (PH - 3) / 2 + 1, (PW - 3) / 2 + 1, 32, &t));
        TRY(maskiou_conv1_launch((const float*)lo, N * K, PH, PW, (const float*)w0->second.d, (const float*)b0->second.d, t.d, rs));
        for (int i = 2; i <= 8; i += 2) {
            if (t.H < 3 || t.W < 3) { e.cur = saved; set_error("input too small for the mask-IoU net (five stride-2 3x3 convs)"); return ISEGMI_ERR_ARG; }
            TRY(eng_conv(e, "maskiou_net." + std::to_string(i), t, 2, 0, 1, nullptr, "maskiou.t" + std::to_string(i), &u));
            t = u;
        }
        TRY(eng_conv(e, "maskiou_net.10", t, 1, 0, 1, nullptr, "maskiou.cls", &u));
        void* ms;
        TRY(eng_buf(e, "det.mask_score", (int64_t)N * K * 4, &ms, 0, {N, K}));
        TRY(maskiou_rescore_launch(u.d, N, K, u.H * u.W, u.C, (const int*)e.bufs["det.class"].d, (const float*)e.bufs["det.score"].d,
                                   (const int*)e.bufs["det.count"].d, (float*)ms, rs));
        e.cur = saved;
    }
    if (rs == e.tail) HIP_TRY(hipEventRecord(e.tail_done, e.tail));
    eng_mark(e, "masks");
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

// ---- which of the engine's streams share an in-order hardware queue?  The runtime folds a process's streams onto GPU_MAX_HW_QUEUES (4) queues by
// what the process has created so far; two streams of one queue serialise although the stream-level program says they may overlap (DESIGN section 4,
// "Streams against four in-order hardware queues").  Probe: a one-wave kernel spins ~40 us on stream a, an empty kernel follows on stream b; b's
// kernel ends before a's only if the two sit on different queues.  Streams are classed against one representative per class found so far.
namespace {
__global__ void probe_spin_kernel(long long cycles, int* sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (sink && cycles < 0) *sink = 1;
}
__global__ void probe_nop_kernel(int* sink) { if (sink && sink[0] == 0x7fffffff) sink[0] = 0; }

int same_queue_once(hipStream_t a, hipStream_t b, bool* same) {
    hipEvent_t s, ea, eb;
    HIP_TRY(hipEventCreate(&s)); HIP_TRY(hipEventCreate(&ea)); HIP_TRY(hipEventCreate(&eb));
    HIP_TRY(hipStreamSynchronize(a)); HIP_TRY(hipStreamSynchronize(b));
    HIP_TRY(hipEventRecord(s, a));
    hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, a, 100000LL, (int*)nullptr);   // ~40-50 us at 2.1-2.4 GHz
    HIP_TRY(hipEventRecord(ea, a));
    hipLaunchKernelGGL(probe_nop_kernel, dim3(1), dim3(64), 0, b, (int*)nullptr);
    HIP_TRY(hipEventRecord(eb, b));
    HIP_TRY(hipStreamSynchronize(a)); HIP_TRY(hipStreamSynchronize(b));
    float ta = 0.0f, tb = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ta, s, ea));
    HIP_TRY(hipEventElapsedTime(&tb, s, eb));
    *same = tb > 0.7f * ta;   // b's empty kernel finished only when the spin had: it queued behind it
    (void)hipEventDestroy(s); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    return ISEGMI_OK;
}

// A stream's FIRST launch pays for the lazy creation of its queue state (longer than the spin: a fresh stream then looks as if it had waited), and a
// launch can be late for other reasons: both streams are warmed up first, and "same queue" needs two probes out of two to say so.
int same_queue(hipStream_t a, hipStream_t b, bool* same) {
    hipLaunchKernelGGL(probe_nop_kernel, dim3(1), dim3(64), 0, a, (int*)nullptr);
    hipLaunchKernelGGL(probe_nop_kernel, dim3(1), dim3(64), 0, b, (int*)nullptr);
    HIP_TRY(hipStreamSynchronize(a)); HIP_TRY(hipStreamSynchronize(b));
    bool s1 = false, s2 = false;
    int rc = same_queue_once(a, b, &s1);
    if (rc) return rc;
    if (s1) { rc = same_queue_once(a, b, &s2); if (rc) return rc; }
    *same = s1 && s2;
    return ISEGMI_OK;
}
}  // namespace

// ---- process-wide pool of stream sets.  An engine BORROWS its ten streams (main, 3 side, tail, heads, 3 heads-side, copy) and hands them back
// when it is destroyed; the next engine of the process gets the very same streams.  Why not create / destroy per engine: the runtime places a
// new stream on one of its four in-order hardware queues by what the process has created (and destroyed) so far, and an engine's speed depends on
// which of its streams end up sharing a queue -- Mask R-CNN bs=2 ran 220.0 img/s as the first engine of a process and 204.8 as the second, after a
// Yolact engine had been created and closed (tools/second_engine_probe.py; the default bench.py line's embedded configs[2] figure was such a second
// engine).  With the pool every engine that follows a closed one sees the first one's placement.  Engines alive at the same time get sets of their own.
// Round 4: the ten streams are no longer "the next ten the runtime hands out".  Which hardware queue a new stream lands on depends on every
// stream the process created before (a foreign HIP library, another engine), and the placement decides the throughput: measured with the probe
// above (tools/stream_layout_probe.py, Yolact bs 8): 986 img/s whenever the MAIN stream's queue carries neither the tail, nor the heads, nor the
// copy stream and the tail's queue carries neither of the other two; 922 with heads + copy on main's queue (one or three foreign streams created
// first), 913 with the tail on it (four or five).  So a set is built from up to 24 CANDIDATE streams whose queues are probed, and the roles are
// dealt by queue class to reproduce the layout a fresh process gets (classes A main / side0 / heads-side1, B side2 / tail / heads-side2,
// C side1 / heads / copy, D heads-side0); candidates not needed stay idle in the set (destroying them would shift the next set's placement).
namespace {
struct StreamSet { int dev = 0; bool used = false; hipStream_t s[10] = {nullptr}; std::vector<hipStream_t> spare; int placed = 0; };
std::mutex g_sets_mu;
std::vector<StreamSet*> g_sets;

int acquire_streams(StreamSet** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_sets_mu);
    for (StreamSet* ss : g_sets)
        if (!ss->used && ss->dev == dev) { ss->used = true; *out = ss; return ISEGMI_OK; }
    StreamSet* ss = new StreamSet();
    ss->dev = dev;
    std::vector<hipStream_t> cand;
    std::vector<int> cls;            // queue class of each candidate (0 = the first candidate's queue)
    std::vector<int> rep;            // one candidate per class
    auto fail = [&](hipError_t er) {
        for (hipStream_t c : cand) (void)hipStreamDestroy(c);
        delete ss;
        set_error(std::string("hipStreamCreate: ") + hipGetErrorString(er));
        return ISEGMI_ERR_HIP;
    };
    // classes in order of discovery: A = the main stream's queue, B, C, D the next three.  Role -> preferred classes (first available wins):
    //   main A | tail B | heads C, copy C (never A, never the tail's) | side0 -- the backbone's projection shortcuts -- D or A, never behind the tail
    //   or the heads phase (942 / 920 img/s when it was; 988 on D, 979 on A) | side1 C, side2 B, heads-side D / A / B as in a fresh process
    static const int pref[10][3] = {{0, 0, 0}, {3, 0, -1}, {2, 3, -1}, {1, 3, -1}, {1, -1, -1}, {2, -1, -1}, {3, 0, -1}, {0, 3, -1}, {1, 3, -1}, {2, -1, -1}};
    int prefv[10][3];
    memcpy(prefv, pref, sizeof(prefv));
    if (const char* ov = getenv("ISEGMI_STREAM_LAYOUT")) {   // dev: ten class digits (0 = A .. 3 = D), role order main side0-2 tail heads hs0-2 copy
        for (int i = 0; i < 10 && ov[i] >= '0' && ov[i] <= '3'; ++i) { prefv[i][0] = ov[i] - '0'; prefv[i][1] = prefv[i][2] = -1; }
    }
    const bool probe = getenv("ISEGMI_STREAM_PLACEMENT") == nullptr || atoi(getenv("ISEGMI_STREAM_PLACEMENT")) != 0;  // 0: the round-3 behaviour (A/B)
    const int need[4] = {1, 4, 4, 4};   // (A often gets no second stream from the runtime: side0 / heads-side1 then take D)
    int have[4] = {0, 0, 0, 0};
    const int max_cand = probe ? 24 : 10;
    while ((int)cand.size() < max_cand) {
        hipStream_t st = nullptr;
        hipError_t er = cand.empty() ? hipStreamCreate(&st) : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (er != hipSuccess) return fail(er);
        cand.push_back(st);
        int c = -1;
        if (probe) {
            for (size_t r = 0; r < rep.size() && c < 0; ++r) {
                bool same = false;
                if (same_queue(cand[rep[r]], st, &same) != ISEGMI_OK) return fail(hipErrorUnknown);
                if (same) c = (int)r;
            }
            if (c < 0) { c = (int)rep.size(); rep.push_back((int)cand.size() - 1); }
        } else {
            c = 0;
        }
        cls.push_back(c);
        if (c < 4) ++have[c];
        if (cand.size() >= 10 && (!probe || (have[0] >= need[0] && have[1] >= need[1] && have[2] >= need[2] && have[3] >= need[3]))) break;
    }
    // deal the roles: a candidate of a preferred class if there is one left, else any left-over candidate (fewer than four queues, or an odd
    // distribution: correct either way, only the overlap differs)
    std::vector<bool> taken(cand.size(), false);
    ss->placed = 1;
    for (int i = 0; i < 10; ++i) {
        int pick = -1;
        if (i == 0) pick = 0;       // the one blocking stream is the main stream
        for (int q = 0; q < 3 && pick < 0 && probe; ++q)
            for (size_t j = 1; j < cand.size() && pick < 0; ++j) if (!taken[j] && prefv[i][q] >= 0 && cls[j] == prefv[i][q]) pick = (int)j;
        for (size_t j = 1; j < cand.size() && pick < 0; ++j) if (!taken[j]) { pick = (int)j; if (probe) ss->placed = 0; }
        taken[pick] = true;
        ss->s[i] = cand[pick];
    }
    for (size_t j = 0; j < cand.size(); ++j) if (!taken[j]) ss->spare.push_back(cand[j]);
    ss->used = true;
    g_sets.push_back(ss);
    *out = ss;
    return ISEGMI_OK;
}

void release_streams(StreamSet* ss) {
    if (!ss) return;
    for (int i = 0; i < 10; ++i) (void)hipStreamSynchronize(ss->s[i]);
    std::lock_guard<std::mutex> lk(g_sets_mu);
    ss->used = false;
}
}  // namespace


// queue_class[i] for the engine's ten streams (0 main, 1-3 side, 4 tail, 5 heads, 6-8 heads-side, 9 copy): equal numbers share a hardware queue.
// Call on an idle engine (it synchronises the streams).
extern "C" int isegmi_engine_stream_layout(isegmi_engine* h, int32_t* queue_class, int n) {
    ARG_CHECK(h && queue_class && n >= 10, "stream_layout args");
    StreamSet* ss = (StreamSet*)h->e.stream_set;
    ARG_CHECK(ss, "engine has no streams");
    int reps[10], nrep = 0;
    for (int i = 0; i < 10; ++i) {
        int cls = -1;
        for (int r = 0; r < nrep && cls < 0; ++r) {
            bool same = false;
            int rc = same_queue(ss->s[reps[r]], ss->s[i], &same);
            if (rc) return rc;
            if (same) cls = r;
        }
        if (cls < 0) { cls = nrep; reps[nrep++] = i; }
        queue_class[i] = cls;
    }
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_create(int model_kind, int max_batch, int H, int W, isegmi_engine** out) {
    ARG_CHECK(out, "null out");
    ARG_CHECK(model_kind == 1 || model_kind == 2, "model_kind: 1 yolact, 2 maskrcnn");
    ARG_CHECK(max_batch > 0 && H > 0 && W > 0, "sizes");
    isegmi_engine* h = new isegmi_engine();
    h->e.kind = model_kind; h->e.max_batch = max_batch; h->e.H = H; h->e.W = W;
    StreamSet* ss = nullptr;
    if (acquire_streams(&ss) != ISEGMI_OK) { delete h; return ISEGMI_ERR_HIP; }
    h->e.stream_set = ss;
    h->e.stream = ss->s[0];
    for (int i = 0; i < 3; ++i) h->e.side[i] = ss->s[1 + i];
    h->e.tail = ss->s[4];
    h->e.heads = ss->s[5];
    for (int i = 0; i < 3; ++i) h->e.hside[i] = ss->s[6 + i];
    h->e.copy = ss->s[9];
    hipError_t er = hipEventCreateWithFlags(&h->e.tail_done, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&h->e.lat_done, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&h->e.heads_done, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&h->e.in_done, hipEventDisableTiming);
    if (er != hipSuccess) { set_error(std::string("engine events: ") + hipGetErrorString(er)); release_streams(ss); delete h; return ISEGMI_ERR_HIP; }
    h->e.cur = h->e.stream;
    *out = h;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_destroy(isegmi_engine* h) {
    if (!h) return ISEGMI_OK;
    Engine& e = h->e;
    if (e.stream_set) for (int i = 0; i < 10; ++i) (void)hipStreamSynchronize(((StreamSet*)e.stream_set)->s[i]);  // nothing of this engine is in flight any more
    eng_graph_reset(e);
    if (e.heads) { (void)hipStreamSynchronize(e.heads); }
    if (e.tail) (void)hipStreamSynchronize(e.tail);
    if (e.tail_done) (void)hipEventDestroy(e.tail_done);
    if (e.lat_done) (void)hipEventDestroy(e.lat_done);
    if (e.heads_done) (void)hipEventDestroy(e.heads_done);
    if (e.copy) (void)hipStreamSynchronize(e.copy);
    if (e.in_done) (void)hipEventDestroy(e.in_done);
    for (auto& u : e.uploads) if (u.done) (void)hipEventDestroy(u.done);
    for (int i = 0; i < 2; ++i) if (e.dl_done[i]) (void)hipEventDestroy(e.dl_done[i]);
    for (int i = 0; i < Engine::PIN_SLOTS; ++i) if (e.pin_ev[i]) (void)hipEventDestroy(e.pin_ev[i]);
    if (e.pin_ring) (void)hipHostFree(e.pin_ring);
    for (auto& ev : e.step_marks) (void)hipEventDestroy(ev);
    for (auto& kv : e.convs) { (void)hipFree(kv.second.d_w); if (kv.second.d_scale) (void)hipFree(kv.second.d_scale); if (kv.second.d_shift) (void)hipFree(kv.second.d_shift); }
    for (auto& kv : e.tensors) (void)hipFree(kv.second.d);
    for (auto& kv : e.bufs) (void)hipFree(kv.second.d);
    for (auto& ev : e.ev_pool) (void)hipEventDestroy(ev);
    release_streams((StreamSet*)e.stream_set);  // back to the process-wide pool (synchronised there), not destroyed
    delete h;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_set_param(isegmi_engine* h, const char* name, float value) {
    ARG_CHECK(h && name, "null");
    if (std::string(name) != "graph") eng_graph_reset(h->e);  // kernel arguments are baked into captured graphs
    h->e.params[name] = value;
    if (std::string(name) == "timing") h->e.timing = value != 0.0f;
    if (std::string(name) == "conv_timing") h->e.conv_timing = value != 0.0f;
    if (std::string(name) == "op_timing") h->e.op_timing = value != 0.0f;
    if (std::string(name) == "conv_trace") h->e.conv_trace = value != 0.0f;
    if (std::string(name) == "multi_stream") h->e.multi_stream = value != 0.0f;
    if (std::string(name) == "fp16") h->e.fp16 = value != 0.0f;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_graph_stats(isegmi_engine* h, int64_t* captures, int64_t* replays, int64_t* failures) {
    ARG_CHECK(h, "null");
    if (captures) *captures = h->e.graph_captures;
    if (replays) *replays = h->e.graph_replays;
    if (failures) *failures = h->e.graph_failures;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_set_conv(isegmi_engine* h, const char* name, int Cout, int R, int S, int Cin, const float* h_w_krsc,
                                      const float* h_scale, const float* h_shift) {
    ARG_CHECK(h && name && h_w_krsc, "null");
    isegmi_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = 1; d.H = R; d.W = S; d.Cin = Cin; d.Cout = Cout; d.R = R; d.S = S; d.stride = 1; d.pad = 0;
    const bool f16 = h->e.fp16 && (Cin % 64 == 0 || (Cin == 4 && R == 7 && S == 7));
    eng_graph_reset(h->e);
    ConvLayer& L = h->e.convs[name];
    if (L.d_w) { (void)hipFree(L.d_w); L.d_w = nullptr; }
    if (L.d_scale) { (void)hipFree(L.d_scale); L.d_scale = nullptr; }
    if (L.d_shift) { (void)hipFree(L.d_shift); L.d_shift = nullptr; }
    L.Cout = Cout; L.R = R; L.S = S; L.Cin = Cin; L.f16 = f16;
    if (f16) {
        int64_t nh = 0;
        TRY(isegmi_conv_packed_halfs(&d, &nh));
        std::vector<uint16_t> packed((size_t)nh);
        TRY(isegmi_pack_conv_weights_f16(&d, h_w_krsc, packed.data()));
        HIP_TRY(hipMalloc((void**)&L.d_w, (size_t)nh * 2));
        HIP_TRY(hipMemcpy(L.d_w, packed.data(), (size_t)nh * 2, hipMemcpyHostToDevice));
    } else {
        int64_t nf = 0;
        TRY(isegmi_conv_packed_floats(&d, &nf));
        std::vector<float> packed((size_t)nf);
        TRY(isegmi_pack_conv_weights(&d, h_w_krsc, packed.data()));
        HIP_TRY(hipMalloc((void**)&L.d_w, (size_t)nf * 4));
        HIP_TRY(hipMemcpy(L.d_w, packed.data(), (size_t)nf * 4, hipMemcpyHostToDevice));
    }
    if (h_scale) { HIP_TRY(hipMalloc((void**)&L.d_scale, (size_t)Cout * 4)); HIP_TRY(hipMemcpy(L.d_scale, h_scale, (size_t)Cout * 4, hipMemcpyHostToDevice)); }
    if (h_shift) { HIP_TRY(hipMalloc((void**)&L.d_shift, (size_t)Cout * 4)); HIP_TRY(hipMemcpy(L.d_shift, h_shift, (size_t)Cout * 4, hipMemcpyHostToDevice)); }
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_set_tensor(isegmi_engine* h, const char* name, const void* h_data, int64_t bytes) {
    ARG_CHECK(h && name && h_data && bytes > 0, "args");
    eng_graph_reset(h->e);
    RawBuf& b = h->e.tensors[name];
    if (b.d) { (void)hipFree(b.d); b.d = nullptr; }
    HIP_TRY(hipMalloc(&b.d, (size_t)bytes));
    HIP_TRY(hipMemcpy(b.d, h_data, (size_t)bytes, hipMemcpyHostToDevice));
    b.bytes = bytes;
    return ISEGMI_OK;
}

extern "C" int isegmi_yolact_forward(isegmi_engine* h, const float* d_images_nhwc3, int N) {
    ARG_CHECK(h && d_images_nhwc3, "null");
    ARG_CHECK(h->e.kind == 1, "engine is not a yolact engine");
    ARG_CHECK(N > 0 && N <= h->e.max_batch, "batch size");
    Engine& e = h->e;
    TRY(eng_wait_upload(e, d_images_nhwc3, (int64_t)N * e.H * e.W * 3 * 4, e.stream));
    char key[96];
    snprintf(key, sizeof(key), "yolact:%d:%p", N, (const void*)d_images_nhwc3);
    const int rc = eng_graph_run(e, key, [&]() { return yolact_forward(e, d_images_nhwc3, N); });
    if (rc == ISEGMI_OK) e.last_N = N;
    return rc;
}

extern "C" int isegmi_yolact_postprocess(isegmi_engine* h, int out_h, int out_w) {
    ARG_CHECK(h, "null");
    ARG_CHECK(h->e.kind == 1, "engine is not a yolact engine");
    ARG_CHECK(out_h > 0 && out_w > 0, "output size");
    return yolact_postprocess(h->e, out_h, out_w);
}

extern "C" int isegmi_yolact_postprocess_sizes(isegmi_engine* h, const int32_t* h_image_hw, int N) {
    ARG_CHECK(h && h_image_hw, "null");
    ARG_CHECK(h->e.kind == 1, "engine is not a yolact engine");
    ARG_CHECK(N == h->e.last_N, "postprocess_sizes: N must be the last forward's batch");
    int ph = 0, pw = 0;
    for (int i = 0; i < N; ++i) { ph = h_image_hw[2 * i] > ph ? h_image_hw[2 * i] : ph; pw = h_image_hw[2 * i + 1] > pw ? h_image_hw[2 * i + 1] : pw; }
    ARG_CHECK(ph > 0 && pw > 0, "image sizes");
    return yolact_postprocess(h->e, ph, pw, h_image_hw);
}

extern "C" int isegmi_engine_sync(isegmi_engine* h) {
    ARG_CHECK(h, "null");
    if (h->e.copy) HIP_TRY(hipStreamSynchronize(h->e.copy));  // an upload_async without a consumer yet: the host may reuse its pinned source after sync()
    HIP_TRY(hipStreamSynchronize(h->e.stream));
    if (h->e.heads) HIP_TRY(hipStreamSynchronize(h->e.heads));
    if (h->e.tail) HIP_TRY(hipStreamSynchronize(h->e.tail));
    h->e.in_pending = false;
    h->e.tail_pending = false;
    h->e.lat_pending = false;
    h->e.heads_pending = false;
    for (auto& u : h->e.uploads) u.waited = true;  // the copy stream has drained: nothing is left to wait for
    collect_times(h->e);
    return ISEGMI_OK;
}

// Asynchronous input upload: `h_src` must be pinned host memory (isegmi_malloc_host) that stays untouched until the forward
// consuming `d_dst` has been enqueued and the engine synchronised (or a later upload_async to the same d_dst returned and was
// synchronised).  The copy runs on the engine's copy stream, after the last enqueued forward has consumed ITS input (so one device
// buffer may be reused every step, two alternate without any wait).  Each destination has its own completion event and only the
// consumer of that destination (isegmi_engine_preprocess_u8 reading it, or the forward taking it as its input) waits on it: an upload
// of batch i+1 enqueued BEFORE forward i does not hold forward i back.
extern "C" int isegmi_engine_upload_async(isegmi_engine* h, void* d_dst, const void* h_src, int64_t bytes) {
    ARG_CHECK(h && d_dst && h_src && bytes > 0, "upload args");
    Engine& e = h->e;
    if (e.in_pending) { HIP_TRY(hipStreamWaitEvent(e.copy, e.in_done, 0)); e.in_pending = false; }
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, e.copy));
    return eng_note_upload(e, d_dst, bytes);
}

// Device front end (see preprocess_u8_kernel).  When its uint8 source is the destination of a pending upload_async the kernel goes to the
// COPY stream, right behind that copy: batch i+1's transform then runs under forward i instead of sitting in the main stream in front of
// it (where it would hold forward i back until upload i+1 has crossed PCIe), and the forward that takes d_out as its input waits for it
// like for an upload.  Otherwise (synchronous isegmi_h2d of the bytes) it runs on the main stream, in front of the next forward.
extern "C" int isegmi_engine_preprocess_u8(isegmi_engine* h, const uint8_t* d_u8, int N, int Hin, int Win, float* d_out, int Hout, int Wout,
                                           int Hpad, int Wpad, int64_t out_img_stride, const float* mean3, const float* std3, int swap_rb) {
    ARG_CHECK(h && N > 0, "preprocess args");
    Engine& e = h->e;
    const int64_t src_bytes = (int64_t)N * Hin * Win * 3, dst_bytes = (int64_t)N * out_img_stride * 4;
    bool on_copy = false;  // the source holds an upload the main stream has not been ordered behind: stay on the copy stream, behind it
    for (auto& u : e.uploads) {
        const char* p = (const char*)d_u8;
        if (!u.waited && p < u.dst + u.bytes && u.dst < p + src_bytes) on_copy = true;
    }
    if (on_copy) {
        // d_out's previous reader is fenced by upload_async's wait on in_done, which precedes the copy in this stream
        {
            OpScope op(e, e.copy, "front end (uint8 -> resize / normalise / pad -> fp32 input)", (double)src_bytes + (double)dst_bytes);
            TRY(preprocess_u8_launch(d_u8, N, Hin, Win, d_out, Hout, Wout, Hpad, Wpad, out_img_stride, mean3, std3, swap_rb, e.copy));
        }
        // (The source's upload entry stays UNCONSUMED: a batch is several launches over one staging buffer -- one per image of a Mask R-CNN batch --
        // and every one of them must find it and stay on the copy stream; marking it consumed here sent image 2 of a batch to the main stream,
        // ahead of its H2D copy.  A staging buffer that is re-allocated goes through sync(), which clears every entry: nothing piles up.)
        return eng_note_upload(e, d_out, dst_bytes);
    }
    TRY(eng_wait_upload(e, d_u8, src_bytes, e.stream));
    OpScope op(e, e.stream, "front end (uint8 -> resize / normalise / pad -> fp32 input)", (double)src_bytes + (double)dst_bytes);
    return preprocess_u8_launch(d_u8, N, Hin, Win, d_out, Hout, Wout, Hpad, Wpad, out_img_stride, mean3, std3, swap_rb, e.stream);
}

// Records a completion mark for the step just enqueued on the stream its results finish on; isegmi_engine_step_times returns
// the intervals between consecutive marks (ms) -- per-step latency samples of a pipelined run -- and clears them.
extern "C" int isegmi_engine_mark_step(isegmi_engine* h) {
    ARG_CHECK(h, "null");
    Engine& e = h->e;
    ARG_CHECK(e.step_marks.size() < 65536, "too many pending step marks");
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipEventRecord(ev, (e.multi_stream && e.tail_pending) ? e.tail : e.stream));
    e.step_marks.push_back(ev);
    return ISEGMI_OK;
}

// Host wait for the completion mark `back` marks before the newest one (0 = the newest): bounds how many steps a producer loop keeps in flight
// without synchronising the whole engine.
extern "C" int isegmi_engine_wait_mark(isegmi_engine* h, int back) {
    ARG_CHECK(h && back >= 0, "wait_mark args");
    Engine& e = h->e;
    if ((size_t)back >= e.step_marks.size()) return ISEGMI_OK;
    HIP_TRY(hipEventSynchronize(e.step_marks[e.step_marks.size() - 1 - (size_t)back]));
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_step_times(isegmi_engine* h, float* ms, int cap, int* count) {
    ARG_CHECK(h && ms && count && cap >= 0, "step_times args");
    Engine& e = h->e;
    int c = 0;
    if (!e.step_marks.empty()) HIP_TRY(hipEventSynchronize(e.step_marks.back()));
    for (size_t i = 1; i < e.step_marks.size() && c < cap; ++i) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e.step_marks[i - 1], e.step_marks[i]));
        ms[c++] = t;
    }
    for (auto& ev : e.step_marks) (void)hipEventDestroy(ev);
    e.step_marks.clear();
    *count = c;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_stream(isegmi_engine* h, void** stream) {
    ARG_CHECK(h && stream, "null");
    // the stream on which the last forward's RESULTS complete (Yolact: the tail stream)
    *stream = (void*)((h->e.multi_stream && h->e.tail_pending) ? h->e.tail : h->e.stream);
    return ISEGMI_OK;
}

// Query a named engine buffer: device pointer, byte size, dtype (0 f32 1 i32 2 u8 3 i64), shape (up to 4 dims).
extern "C" int isegmi_engine_buffer_info(isegmi_engine* h, const char* name, void** d_ptr, int64_t* bytes, int32_t* dtype,
                                         int64_t* shape4, int32_t* ndim) {
    ARG_CHECK(h && name, "null");
    auto it = h->e.bufs.find(name);
    if (it == h->e.bufs.end()) { set_error(std::string("no such buffer: ") + name); return ISEGMI_ERR_ARG; }
    const RawBuf& b = it->second;
    if (d_ptr) *d_ptr = b.d;
    if (bytes) *bytes = b.bytes;
    if (dtype) *dtype = b.dtype;
    if (ndim) *ndim = (int32_t)b.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < b.shape.size() ? b.shape[i] : 1;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_memory(isegmi_engine* h, int64_t* weight_bytes, int64_t* buffer_bytes) {
    ARG_CHECK(h, "null");
    int64_t wb = 0, bb = 0;
    for (auto& kv : h->e.convs) {
        const ConvLayer& L = kv.second;
        isegmi_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.N = 1; d.H = L.R; d.W = L.S; d.Cin = L.Cin; d.Cout = L.Cout; d.R = L.R; d.S = L.S; d.stride = 1;
        int64_t n = 0;
        if (L.f16) { if (isegmi_conv_packed_halfs(&d, &n) == 0) wb += n * 2; }
        else if (isegmi_conv_packed_floats(&d, &n) == 0) wb += n * 4;
        wb += (L.d_scale ? L.Cout * 4 : 0) + (L.d_shift ? L.Cout * 4 : 0);
    }
    for (auto& kv : h->e.tensors) wb += kv.second.bytes;
    for (auto& kv : h->e.bufs) bb += kv.second.bytes;
    if (weight_bytes) *weight_bytes = wb;
    if (buffer_bytes) *buffer_bytes = bb;
    return ISEGMI_OK;
}

// Copies the stage timings of the last synchronised forward: names joined by ';' and ms values.
extern "C" int isegmi_engine_get_timings(isegmi_engine* h, char* names, int names_cap, float* ms, int ms_cap, int* count) {
    ARG_CHECK(h && names && ms && count, "null");
    std::string s;
    int c = 0;
    for (auto& kv : h->e.last_times) {
        if (c >= ms_cap) break;
        if (c) s += ";";
        s += kv.first;
        ms[c++] = kv.second;
    }
    ARG_CHECK((int)s.size() + 1 <= names_cap, "names buffer too small");
    memcpy(names, s.c_str(), s.size() + 1);
    *count = c;
    return ISEGMI_OK;
}

// Conv-kernel statistics accumulated over synchronised forwards since the last call (set_param
// "conv_timing" 1): algorithmic FLOPs (2*M*Cout*R*S*Cin, stem Cin=3), summed HIP-event time of the
// conv launches on the engine stream, and the number of launches.  Resets the accumulators.
extern "C" int isegmi_engine_conv_stats(isegmi_engine* h, double* flops, double* ms, int64_t* launches) {
    ARG_CHECK(h && flops && ms && launches, "null");
    *flops = h->e.conv_flops; *ms = h->e.conv_ms; *launches = h->e.conv_launches;
    h->e.conv_flops = 0; h->e.conv_ms = 0; h->e.conv_launches = 0;
    return ISEGMI_OK;
}

// HBM-bound stages timed under "op_timing" (call after a sync): accumulates the pending event pairs, then returns per label the summed
// HIP-event time (us), the summed ALGORITHMIC bytes and the number of timed scopes since the last call; `names` = labels joined
// by '\n'.  Resets the accumulators.
extern "C" int isegmi_engine_op_stats(isegmi_engine* h, char* names, int names_cap, double* us, double* bytes, int64_t* launches, int cap, int* count) {
    ARG_CHECK(h && names && us && bytes && launches && count && names_cap > 0 && cap > 0, "op_stats args");
    Engine& e = h->e;
    for (auto& ev : e.op_evs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
            auto& st = e.op_stats[ev.label];
            st.us += (double)ms * 1e3; st.bytes += ev.bytes; st.launches += 1;
        }
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    e.op_evs.clear();
    std::string s;
    int c = 0;
    for (auto& kv : e.op_stats) {
        if (c >= cap || (int)(s.size() + kv.first.size() + 2) > names_cap) break;
        if (c) s += "\n";
        s += kv.first;
        us[c] = kv.second.us; bytes[c] = kv.second.bytes; launches[c] = kv.second.launches;
        ++c;
    }
    memcpy(names, s.c_str(), s.size() + 1);
    *count = c;
    e.op_stats.clear();
    return ISEGMI_OK;
}

// Packs the last forward's detections into ONE contiguous record block for the all-gather:
//   [count i32 x N][box f32 x N*K*4][score f32 x N*K][class i32 x N*K][coeff f32 x N*K*32]
// (+ [proto f32 x N*PH*PW*32] when with_proto).  D2D copies on the engine stream.
extern "C" int isegmi_yolact_pack_records(isegmi_engine* h, void* d_dst, int64_t cap, int with_proto, int64_t* bytes) {
    ARG_CHECK(h && d_dst && bytes, "null");
    Engine& e = h->e;
    const int N = e.last_N;
    ARG_CHECK(N > 0, "pack before forward");
    const int K = (int)e.param("max_num_detections", 100);
    hipStream_t rs = (e.multi_stream && e.tail_pending) ? e.tail : e.stream;  // results stream
    const char* names[5] = {"det.count", "det.box", "det.score", "det.class", "det.coeff"};
    const int64_t sizes[5] = {(int64_t)N * 4, (int64_t)N * K * 16, (int64_t)N * K * 4, (int64_t)N * K * 4, (int64_t)N * K * 32 * 4};
    int64_t off = 0;
    for (int i = 0; i < 5; ++i) {
        ARG_CHECK(off + sizes[i] <= cap, "record buffer too small");
        HIP_TRY(hipMemcpyAsync((char*)d_dst + off, e.bufs[names[i]].d, (size_t)sizes[i], hipMemcpyDeviceToDevice, rs));
        off += sizes[i];
    }
    if (with_proto) {
        RawBuf& p = e.bufs["proto"];
        const int64_t pb = (int64_t)N * p.shape[1] * p.shape[2] * p.shape[3] * 4;
        ARG_CHECK(off + pb <= cap, "record buffer too small (proto)");
        HIP_TRY(hipMemcpyAsync((char*)d_dst + off, p.d, (size_t)pb, hipMemcpyDeviceToDevice, rs));
        off += pb;
    }
    if (rs == e.tail && e.tail) HIP_TRY(hipEventRecord(e.tail_done, e.tail));  // the copies read proto / det.*: extend the WAR fence
    *bytes = off;
    return ISEGMI_OK;
}

// Text report "label\tGFLOP\tms\tTFLOP/s" per conv layer accumulated under conv_timing; clears it.
extern "C" int isegmi_engine_conv_report(isegmi_engine* h, char* buf, int cap) {
    ARG_CHECK(h && buf && cap > 0, "args");
    std::string s;
    for (auto& kv : h->e.conv_layers) {
        char line[256];
        snprintf(line, sizeof(line), "%s\t%.3f\t%.4f\t%.2f\n", kv.first.c_str(), kv.second.first / 1e9, kv.second.second,
                 kv.second.second > 0 ? kv.second.first / (kv.second.second * 1e-3) / 1e12 : 0.0);
        s += line;
    }
    h->e.conv_layers.clear();
    ARG_CHECK((int)s.size() + 1 <= cap, "report buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return ISEGMI_OK;
}
