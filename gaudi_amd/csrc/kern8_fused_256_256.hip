// kern8_fused_256_256.hip -- sampler_kernel8 (8 waves, two per SIMD) instantiations [(256, 256)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8_fused_256_256).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8_fused_256_256(int hpe, int hpp) {
  if (hpe == 256 && hpp == 256) return gaudi::sampler_kernel8<256, 256>;
  return nullptr;
}
