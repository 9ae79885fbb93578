// kern8_edm_208.hip -- sampler_kernel8 (8 waves, two per SIMD) instantiations [(208, 0)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8_edm_208).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8_edm_208(int hpe, int hpp) {
  if (hpe == 208 && hpp == 0) return gaudi::sampler_kernel8<208, 0>;
  return nullptr;
}
