// kern8s_pred_small.hip -- sampler_kernel8s (8 waves, edge and node GEMMs on fp16-pair operands: w8_split.h, w8_nodes_f16.h) instantiations [(0, 32), (0, 48), (0, 64), (0, 128)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8s_pred_small).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8s_pred_small(int hpe, int hpp) {
  if (hpe == 0 && hpp == 32) return gaudi::sampler_kernel8s<0, 32>;
  if (hpe == 0 && hpp == 48) return gaudi::sampler_kernel8s<0, 48>;
  if (hpe == 0 && hpp == 64) return gaudi::sampler_kernel8s<0, 64>;
  if (hpe == 0 && hpp == 128) return gaudi::sampler_kernel8s<0, 128>;
  return nullptr;
}
