// kern8_pred_256.hip -- sampler_kernel8 (8 waves, two per SIMD) instantiations [(0, 256)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8_pred_256).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8_pred_256(int hpe, int hpp) {
  if (hpe == 0 && hpp == 256) return gaudi::sampler_kernel8<0, 256>;
  return nullptr;
}
