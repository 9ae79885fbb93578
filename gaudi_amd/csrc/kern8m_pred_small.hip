// kern8m_pred_small.hip -- 8-wave kernels whose predictor runs SEVERAL rounds of eight edge tiles (graphs of more than 128 live-edge slots:
// fully connected molecules of 12+ nodes; w8_pred.h, template flag MR) [the test widths, predictor only, all modes]; own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8m_pred_small.  mode: 0 = fp32 matrix instructions,
// 1 / 2 = split operands with the full / half weight ring.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8m_pred_small(int hpe, int hpp, int mode) {
  if (hpe == 0 && hpp == 32 && mode == 0) return gaudi::sampler_kernel8m<0, 0, 32>;
  if (hpe == 0 && hpp == 48 && mode == 0) return gaudi::sampler_kernel8m<0, 0, 48>;
  if (hpe == 0 && hpp == 64 && mode == 0) return gaudi::sampler_kernel8m<0, 0, 64>;
  if (hpe == 0 && hpp == 32 && mode == 1) return gaudi::sampler_kernel8m<1, 0, 32>;
  if (hpe == 0 && hpp == 48 && mode == 1) return gaudi::sampler_kernel8m<1, 0, 48>;
  if (hpe == 0 && hpp == 64 && mode == 1) return gaudi::sampler_kernel8m<1, 0, 64>;
  if (hpe == 0 && hpp == 32 && mode == 2) return gaudi::sampler_kernel8m<2, 0, 32>;
  if (hpe == 0 && hpp == 48 && mode == 2) return gaudi::sampler_kernel8m<2, 0, 48>;
  if (hpe == 0 && hpp == 64 && mode == 2) return gaudi::sampler_kernel8m<2, 0, 64>;
  return nullptr;
}
