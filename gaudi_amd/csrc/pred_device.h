// pred_device.h -- EGNN_predictor forward + hand-written reverse pass (guidance gradient).
#pragma once
#include "device_common.h"
#include "edm_device.h"

namespace gaudi {

struct PredDev {
  const float* w;
  int F, K, L, attention, use_tanh;
  float coords_range_layer;  // coords_range / n_layers (egnn_predictor/models.py:515)
};

template <int HP>
__device__ void guidance_update(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad, float* sTmp,
                                float* sMean, float t_val, float sigma, const float* target_w, float scale,
                                float* pred_out, float readout_div, float* stash, int tid);

template <int HP>
__device__ void predictor_entry(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad, float* sTmp,
                                float* sMean, float t_val, const float* dpred, bool want_grad, float* pred_out,
                                float readout_div, float* stash, int tid);

}  // namespace gaudi
