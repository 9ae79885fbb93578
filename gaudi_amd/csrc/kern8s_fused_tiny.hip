// kern8s_fused_tiny.hip -- sampler_kernel8s (8 waves, edge and node GEMMs on fp16-pair operands: w8_split.h, w8_nodes_f16.h) instantiations [(32, 48), (32, 32), (48, 48), (64, 64)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8s_fused_tiny).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8s_fused_tiny(int hpe, int hpp) {
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel8s<32, 48>;
  if (hpe == 32 && hpp == 32) return gaudi::sampler_kernel8s<32, 32>;
  if (hpe == 48 && hpp == 48) return gaudi::sampler_kernel8s<48, 48>;
  if (hpe == 64 && hpp == 64) return gaudi::sampler_kernel8s<64, 64>;
  return nullptr;
}
