// kern8m_fused_192_208_h.hip -- 8-wave kernels whose predictor runs SEVERAL rounds of eight edge tiles (graphs of more than 128 live-edge slots:
// fully connected molecules of 12+ nodes; w8_pred.h, template flag MR) [(192, 208), mode 2]; own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8m_fused_192_208_h.  mode: 0 = fp32 matrix instructions,
// 1 / 2 = split operands with the full / half weight ring.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8m_fused_192_208_h(int hpe, int hpp, int mode) {
  if (hpe == 192 && hpp == 208 && mode == 2) return gaudi::sampler_kernel8m<2, 192, 208>;
  return nullptr;
}
