// results.cpp -- device-side COCO output of a step: run-length encoding of the masks the last postprocess /
// paste produced, ONE fixed-size record block per batch (the unit the N = 1 path downloads and the N > 1 path all-gathers over RCCL), and
// its asynchronous download.  Stands where upstream turns predictions into COCO json: maskrcnn-benchmark inference() ->
// prepare_for_coco_segmentation (README.md:344-347), Yolact eval.py Detections.add_bbox / add_mask -> dump (README.md:243-249).
//
// Record block of a batch of N images, K = detection capacity per image (byte offsets are multiples of 8):
//   [status  i32 x 4]            total runs, total characters, overflow bits (1 runs, 2 characters: raise the "rle_cap_*" params), 0
//   [box     N*K*4]              Mask R-CNN: f32 xyxy in ORIGINAL image coordinates (det.box_resized); Yolact: i64 xyxy (det.box_int)
//   [count   i32 x N]
//   [score   f32 x N*K]
//   [label   i32 x N*K]          Mask R-CNN: 1..80; Yolact: class 0..79
//   [mscore  f32 x N*K]          YOLACT++ only (mask re-scoring)
//   [str_off i32 x (N*K + 1)]    characters of slot (n, k): chars[str_off[n*K+k] .. str_off[n*K+k+1])
//   [pad to 8]
//   [chars   u8 x cap_chars]     pycocotools "counts" strings, back to back
#include <string.h>

#include "engine.h"

namespace isegmi {
int rle_encode_launch(const isegmi_rle_args* a, hipStream_t st);

#define TRY(x)               \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

int maskrcnn_det_cap(Engine& e);
static int det_cap(Engine& e) { return e.kind == 1 ? (int)e.param("max_num_detections", 100) : maskrcnn_det_cap(e); }
// Default RLE capacities per image: a blob mask costs ~2 runs per column it spans and ~1.7 characters per run, so K masks as wide as the
// network canvas need ~4 K W runs; the fixed 256 Ki of round 3 was marginal at 640 px and short at 1500 px with K = 100.  An
// overflow is not fatal any more (isegmi.pipeline.run_record_loop raises these parameters and redoes the step), this only makes it rare.
static int64_t rle_cap_default(Engine& e) {
    const int64_t side = e.H > e.W ? e.H : e.W;
    const int64_t c = 4ll * det_cap(e) * side;
    return (c < 262144 ? 262144 : c) * e.max_batch;
}
static int rle_cap_chars(Engine& e) { return (int)e.param("rle_cap_chars", (float)(2 * rle_cap_default(e))); }
static int rle_cap_runs(Engine& e) {
    int c = (int)e.param("rle_cap_runs", (float)rle_cap_default(e));
    return (c + 1023) / 1024 * 1024;
}

struct Section { const char* buf; int64_t bytes; };

static int64_t align8(int64_t v) { return (v + 7) & ~7ll; }

// sections of the record block in order (chars excluded); `with_mscore`: YOLACT++
static int record_layout(Engine& e, int N, std::vector<Section>* out, int64_t* chars_off, int64_t* total) {
    const int K = det_cap(e);
    const bool y = e.kind == 1;
    const bool ms = y && e.convs.count("maskiou_net.2") != 0;
    std::vector<Section> s;
    s.push_back({"rle.status", 16});
    s.push_back({y ? "det.box_int" : "det.box_resized", (int64_t)N * K * (y ? 32 : 16)});
    s.push_back({"det.count", (int64_t)N * 4});
    s.push_back({"det.score", (int64_t)N * K * 4});
    s.push_back({y ? "det.class" : "det.label", (int64_t)N * K * 4});
    if (ms) s.push_back({"det.mask_score", (int64_t)N * K * 4});
    s.push_back({"rle.str_off", ((int64_t)N * K + 1) * 4});
    int64_t off = 0;
    for (auto& x : s) off += x.bytes;
    off = align8(off);
    if (out) *out = s;
    if (chars_off) *chars_off = off;
    if (total) *total = off + rle_cap_chars(e);
    return ISEGMI_OK;
}

// RLE of det.masks (the planes the last yolact_postprocess / maskrcnn_paste wrote) on the results stream.
int eng_rle(Engine& e, const int32_t* h_image_hw) {
    const int N = e.last_N;
    if (N <= 0) { set_error("rle before forward"); return ISEGMI_ERR_STATE; }
    auto mit = e.bufs.find("det.masks");
    if (mit == e.bufs.end() || mit->second.shape.size() != 4 || mit->second.shape[0] != N) { set_error("rle: run postprocess / paste of this batch first"); return ISEGMI_ERR_STATE; }
    const int K = (int)mit->second.shape[1], ph = (int)mit->second.shape[2], pw = (int)mit->second.shape[3];
    hipStream_t rs = eng_results_stream(e);
    isegmi_rle_args a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.K = K; a.plane_h = ph; a.plane_w = pw; a.cap_runs = rle_cap_runs(e); a.cap_chars = rle_cap_chars(e);
    a.d_masks = (const uint8_t*)mit->second.d;
    a.d_count = (const int32_t*)e.bufs["det.count"].d;
    {   // the windows the paste / mask assembly kernel wrote next to the planes: only they are read
        auto wit = e.bufs.find("det.mask_window");
        if (wit != e.bufs.end() && wit->second.d != nullptr && wit->second.shape.size() == 3 && wit->second.shape[0] == N && wit->second.shape[1] == K)
            a.d_windows = (const int32_t*)wit->second.d;
        else if (e.param("sparse_masks", 0.0f) != 0.0f) { set_error("rle: sparse_masks is set but the last postprocess / paste left no windows"); return ISEGMI_ERR_STATE; }
    }
    void* q;
    if (h_image_hw) {
        for (int i = 0; i < N; ++i)
            if (h_image_hw[2 * i] <= 0 || h_image_hw[2 * i] > ph || h_image_hw[2 * i + 1] <= 0 || h_image_hw[2 * i + 1] > pw) { set_error("rle: image size outside the mask plane"); return ISEGMI_ERR_ARG; }
        TRY(eng_buf(e, "rle.image_hw", (int64_t)e.max_batch * 8, &q, 1, {N, 2}));
        TRY(eng_stage_small(e, h_image_hw, (size_t)N * 8, q, rs));
        a.d_image_hw = (const int32_t*)q;
    }
    int64_t tb, cb, nb, lb, eb, sb;
    TRY(isegmi_rle_workspace(N, K, ph, pw, a.cap_runs, &tb, &cb, &nb, &lb, &eb, &sb));
    TRY(eng_buf(e, "rle.ws_trans", tb, &q)); a.d_ws_trans = q;
    TRY(eng_buf(e, "rle.ws_col", cb, &q, 1)); a.d_ws_col = (int32_t*)q;
    TRY(eng_buf(e, "rle.ws_nruns", nb, &q, 1)); a.d_ws_nruns = (int32_t*)q;
    TRY(eng_buf(e, "rle.ws_tile", lb, &q, 1)); a.d_ws_tile = (int32_t*)q;
    TRY(eng_buf(e, "rle.ws_len", eb, &q, 2)); a.d_ws_len = (uint8_t*)q;
    TRY(eng_buf(e, "rle.ws_starts", sb, &q, 1)); a.d_ws_starts = (uint32_t*)q;
    TRY(eng_buf(e, "rle.run_off", ((int64_t)e.max_batch * K + 1) * 4, &q, 1, {(int64_t)N * K + 1})); a.d_out_run_off = (int32_t*)q;
    TRY(eng_buf(e, "rle.counts", (int64_t)a.cap_runs * 4, &q, 1, {a.cap_runs})); a.d_out_counts = (uint32_t*)q;
    TRY(eng_buf(e, "rle.str_off", ((int64_t)e.max_batch * K + 1) * 4, &q, 1, {(int64_t)N * K + 1})); a.d_out_str_off = (int32_t*)q;
    TRY(eng_buf(e, "rle.chars", (int64_t)a.cap_chars, &q, 2, {a.cap_chars})); a.d_out_chars = (uint8_t*)q;
    TRY(eng_buf(e, "rle.status", 16, &q, 1, {4})); a.d_out_status = (int32_t*)q;
    {
        // the chain reads each mask's box WINDOW of its uint8 plane once; the windows are data (on the device): the caller prices this stage from
        // det.mask_window (bench.py), 0 here
        OpScope op(e, rs, "rle (7 prefix-sum launches: pack, scans, emit, lengths, chars)", 0.0);
        TRY(rle_encode_launch(&a, rs));
    }
    if (rs == e.tail) HIP_TRY(hipEventRecord(e.tail_done, e.tail));  // reads det.masks / det.count: extend the WAR fence
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_engine_rle(isegmi_engine* h, const int32_t* h_image_hw) {
    ARG_CHECK(h, "null");
    return eng_rle(h->e, h_image_hw);
}

extern "C" int isegmi_engine_coco_record_bytes(isegmi_engine* h, int N, int64_t* bytes, int64_t* chars_offset) {
    ARG_CHECK(h && bytes && N > 0 && N <= h->e.max_batch, "record_bytes args");
    return record_layout(h->e, N, nullptr, chars_offset, bytes);
}

// One launch writes everything in front of the strings.  The block is laid out for `nb` image slots (nb >= the last forward's n: a short
// last batch of a data set still fills a block of the fixed per-step size); slots n .. nb-1 carry count 0 and empty strings.
struct PackK {
    const void* box; int box_elem;  // bytes per box coordinate: 4 (f32) or 8 (i64)
    const int* count; const float* score; const int* label; const float* mscore; const int* str_off; const int* status;
    int n, nb, K;
    char* dst;
    int64_t off_box, off_count, off_score, off_label, off_mscore, off_stroff;
};
__global__ void pack_coco_kernel(const PackK p) {
    const int64_t nk = (int64_t)p.n * p.K, nbk = (int64_t)p.nb * p.K;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if (tid < 4) ((int*)p.dst)[tid] = p.status[tid];
    for (int64_t i = tid; i < p.nb; i += nth) ((int*)(p.dst + p.off_count))[i] = i < p.n ? p.count[i] : 0;
    const int words = p.box_elem / 4 * 4;  // 32-bit words per box
    for (int64_t i = tid; i < nbk * words; i += nth) ((uint32_t*)(p.dst + p.off_box))[i] = i < nk * words ? ((const uint32_t*)p.box)[i] : 0u;
    for (int64_t i = tid; i < nbk; i += nth) {
        ((float*)(p.dst + p.off_score))[i] = i < nk ? p.score[i] : 0.0f;
        ((int*)(p.dst + p.off_label))[i] = i < nk ? p.label[i] : 0;
        if (p.mscore) ((float*)(p.dst + p.off_mscore))[i] = i < nk ? p.mscore[i] : 0.0f;
    }
    for (int64_t i = tid; i <= nbk; i += nth) ((int*)(p.dst + p.off_stroff))[i] = p.str_off[i <= nk ? i : nk];
}

extern "C" int isegmi_engine_pack_coco_records(isegmi_engine* h, void* d_dst, int64_t cap, int n_block, int64_t* bytes) {
    ARG_CHECK(h && d_dst && bytes, "null");
    Engine& e = h->e;
    const int N = e.last_N;
    ARG_CHECK(N > 0, "pack before forward");
    ARG_CHECK(n_block >= N && n_block <= e.max_batch, "n_block: between the last forward's batch and max_batch");
    std::vector<Section> secs;
    int64_t coff = 0, total = 0;
    TRY(record_layout(e, n_block, &secs, &coff, &total));
    ARG_CHECK(total <= cap, "record buffer too small");
    hipStream_t rs = eng_results_stream(e);
    const int K = det_cap(e);
    const bool y = e.kind == 1;
    PackK p;
    memset(&p, 0, sizeof(p));
    p.n = N; p.nb = n_block; p.K = K; p.dst = (char*)d_dst; p.box_elem = y ? 8 : 4;
    int64_t off = 0;
    for (auto& sct : secs) {
        auto it = e.bufs.find(sct.buf);
        if (it == e.bufs.end() || it->second.d == nullptr) { set_error(std::string("pack_coco_records: buffer not produced yet: ") + sct.buf + " (run postprocess / paste and isegmi_engine_rle first)"); return ISEGMI_ERR_STATE; }
        const std::string nm = sct.buf;
        void* d = it->second.d;
        if (nm == "rle.status") p.status = (const int*)d;
        else if (nm == "det.box_int" || nm == "det.box_resized") { p.box = d; p.off_box = off; }
        else if (nm == "det.count") { p.count = (const int*)d; p.off_count = off; }
        else if (nm == "det.score") { p.score = (const float*)d; p.off_score = off; }
        else if (nm == "det.class" || nm == "det.label") { p.label = (const int*)d; p.off_label = off; }
        else if (nm == "det.mask_score") { p.mscore = (const float*)d; p.off_mscore = off; }
        else if (nm == "rle.str_off") { p.str_off = (const int*)d; p.off_stroff = off; }
        off += sct.bytes;
    }
    hipLaunchKernelGGL(pack_coco_kernel, dim3(64), dim3(256), 0, rs, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync((char*)d_dst + coff, e.bufs["rle.chars"].d, (size_t)(total - coff), hipMemcpyDeviceToDevice, rs));
    if (rs == e.tail && e.tail) HIP_TRY(hipEventRecord(e.tail_done, e.tail));
    *bytes = total;
    return ISEGMI_OK;
}

static int dl_init(Engine& e) {
    if (e.dl_done[0]) return ISEGMI_OK;
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&e.dl_done[i], hipEventDisableTiming));
    return ISEGMI_OK;
}

// Asynchronous D2H of `bytes` at d_src into PINNED host memory, ordered behind everything enqueued so far on the results stream; two slots so that step t+1 may produce while step t's block is still in flight:
//   download_fence(slot): later work on the results stream waits for the slot's previous download (call before overwriting its d_src);
//   download_wait(slot):  host wait, after which h_dst holds the block.
extern "C" int isegmi_engine_download_async(isegmi_engine* h, int slot, void* h_dst_pinned, const void* d_src, int64_t bytes) {
    ARG_CHECK(h && h_dst_pinned && d_src && bytes > 0 && slot >= 0 && slot < 2, "download args");
    Engine& e = h->e;
    TRY(dl_init(e));
    // The copy rides on the RESULTS stream itself, right behind the kernels that produced the block.  A download stream of its own (first
    // version) is one more stream for the runtime to fold onto its four hardware queues, and it landed on the MAIN stream's: the copy of step i --
    // enqueued after forward i, waiting for step i's last kernel -- then sat in that in-order queue in front of forward i+1's backbone, and the
    // whole cross-step overlap was gone (tools/e2e_timeline.py: backbone i+1 started 70 us after copy i; value_e2e 5 % under value at 8.3 ms
    // per step, 19 % at 2.3 ms).
    hipStream_t rs = eng_results_stream(e);
    HIP_TRY(hipMemcpyAsync(h_dst_pinned, d_src, (size_t)bytes, hipMemcpyDeviceToHost, rs));
    HIP_TRY(hipEventRecord(e.dl_done[slot], rs));
    e.dl_used[slot] = true;
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_download_fence(isegmi_engine* h, int slot) {
    ARG_CHECK(h && slot >= 0 && slot < 2, "download slot");
    Engine& e = h->e;
    if (e.dl_used[slot]) HIP_TRY(hipStreamWaitEvent(eng_results_stream(e), e.dl_done[slot], 0));
    return ISEGMI_OK;
}

extern "C" int isegmi_engine_download_wait(isegmi_engine* h, int slot) {
    ARG_CHECK(h && slot >= 0 && slot < 2, "download slot");
    if (h->e.dl_used[slot]) HIP_TRY(hipEventSynchronize(h->e.dl_done[slot]));
    return ISEGMI_OK;
}
