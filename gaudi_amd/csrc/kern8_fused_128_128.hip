// kern8_fused_128_128.hip -- sampler_kernel8 (8 waves, two per SIMD) instantiations [(128, 128)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8_fused_128_128).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8_fused_128_128(int hpe, int hpp) {
  if (hpe == 128 && hpp == 128) return gaudi::sampler_kernel8<128, 128>;
  return nullptr;
}
