// kern8s2_fused_tiny.hip -- sampler_kernel8s2 (see kern8s2_fused_192_208.hip) for the test-sized networks [(32, 48), (64, 64)]; looked up
// by gaudi_hip.hip through gaudi_kern8s2_fused_tiny.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8s2_fused_tiny(int hpe, int hpp) {
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel8s2<32, 48>;
  if (hpe == 64 && hpp == 64) return gaudi::sampler_kernel8s2<64, 64>;
  return nullptr;
}
