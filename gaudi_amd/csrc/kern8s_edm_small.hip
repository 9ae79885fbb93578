// kern8s_edm_small.hip -- sampler_kernel8s (8 waves, edge and node GEMMs on fp16-pair operands: w8_split.h, w8_nodes_f16.h) instantiations [(32, 0), (48, 0), (64, 0), (128, 0)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8s_edm_small).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8s_edm_small(int hpe, int hpp) {
  if (hpe == 32 && hpp == 0) return gaudi::sampler_kernel8s<32, 0>;
  if (hpe == 48 && hpp == 0) return gaudi::sampler_kernel8s<48, 0>;
  if (hpe == 64 && hpp == 0) return gaudi::sampler_kernel8s<64, 0>;
  if (hpe == 128 && hpp == 0) return gaudi::sampler_kernel8s<128, 0>;
  return nullptr;
}
