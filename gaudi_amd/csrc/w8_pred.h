// w8_pred.h -- placeholder until the 8-wave predictor lands: declarations only, so that EDM-only 8-wave kernels build.
#pragma once
#include "pred_device.h"
#include "w8_edm.h"

namespace gaudi {
namespace w8 {
template <int HP>
__device__ void guidance_update(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad, float* sTmp, float* sMean,
                                float t_val, float sigma, const float* target_w, float scale, float* pred_out, float readout_div,
                                float* stash, int tid, int phase, const float* dpred_ext);
template <int HP>
__device__ void predictor_entry(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad, float* sTmp, float* sMean,
                                float t_val, const float* dpred, bool want_grad, float* pred_out, float readout_div, float* stash,
                                int tid);
}  // namespace w8
}  // namespace gaudi
