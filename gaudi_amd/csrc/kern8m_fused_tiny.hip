// kern8m_fused_tiny.hip -- 8-wave kernels whose predictor runs SEVERAL rounds of eight edge tiles (graphs of more than 128 live-edge slots:
// fully connected molecules of 12+ nodes; w8_pred.h, template flag MR) [the test widths, all modes]; own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8m_fused_tiny.  mode: 0 = fp32 matrix instructions,
// 1 / 2 = split operands with the full / half weight ring.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8m_fused_tiny(int hpe, int hpp, int mode) {
  if (hpe == 32 && hpp == 48 && mode == 0) return gaudi::sampler_kernel8m<0, 32, 48>;
  if (hpe == 32 && hpp == 32 && mode == 0) return gaudi::sampler_kernel8m<0, 32, 32>;
  if (hpe == 48 && hpp == 48 && mode == 0) return gaudi::sampler_kernel8m<0, 48, 48>;
  if (hpe == 64 && hpp == 64 && mode == 0) return gaudi::sampler_kernel8m<0, 64, 64>;
  if (hpe == 32 && hpp == 48 && mode == 1) return gaudi::sampler_kernel8m<1, 32, 48>;
  if (hpe == 32 && hpp == 32 && mode == 1) return gaudi::sampler_kernel8m<1, 32, 32>;
  if (hpe == 48 && hpp == 48 && mode == 1) return gaudi::sampler_kernel8m<1, 48, 48>;
  if (hpe == 64 && hpp == 64 && mode == 1) return gaudi::sampler_kernel8m<1, 64, 64>;
  if (hpe == 32 && hpp == 48 && mode == 2) return gaudi::sampler_kernel8m<2, 32, 48>;
  if (hpe == 32 && hpp == 32 && mode == 2) return gaudi::sampler_kernel8m<2, 32, 32>;
  if (hpe == 48 && hpp == 48 && mode == 2) return gaudi::sampler_kernel8m<2, 48, 48>;
  if (hpe == 64 && hpp == 64 && mode == 2) return gaudi::sampler_kernel8m<2, 64, 64>;
  return nullptr;
}
