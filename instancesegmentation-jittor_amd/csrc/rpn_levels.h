// rpn_levels.h -- RPN selection batched over (level, image) (round 5): csrc/rcnn_ops.hip (five launches for all levels) and
// csrc/select.hip (the grouped two-level top-k), called from csrc/maskrcnn.cpp and the op-level C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isegmi {

constexpr int RPN_MAX_LEVELS = 5;
int rpn_topk_plan(int nl, int N, const int* n, int k, int* slices, int64_t* cand_off);
int rpn_topk_levels_launch(int nl, int N, const float* keys, const int64_t* key_off, const int* n, int k, const int* slices, const int64_t* cand_off,
                           float* cand_vals, int* cand_idx, float* out_vals, int* out_idx, int* out_cnt, hipStream_t st);
int rpn_levels_workspace(int nl, int N, const int* HWA, int pre_nms, int64_t* prob_elems, int64_t* cand_elems);
int rpn_levels_select_launch(int nl, const float* const* heads, const float* const* anchors, const int* HWA, const int* level_slot, const int* image_hw, int N,
                             int A, int CH, int pre_nms, int post_nms, float thr, float min_size, int ge, int L, int post_cap, float* prob, float* cand_vals,
                             int* cand_idx, float* tk_vals, int* tk_idx, int* tk_cnt, void* nms_ws, float* out_boxes, float* out_scores, int* out_cnt,
                             hipStream_t st);

}  // namespace isegmi
