// kern8_fused_192_192.hip -- sampler_kernel8 (8 waves, two per SIMD) instantiations [(192, 192)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8_fused_192_192).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8_fused_192_192(int hpe, int hpp) {
  if (hpe == 192 && hpp == 192) return gaudi::sampler_kernel8<192, 192>;
  return nullptr;
}
