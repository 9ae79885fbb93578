// kern8h_fused_192_208.hip -- sampler_kernel8h (8 waves, edge and node GEMMs on fp16-pair operands with the half-size weight ring: w8_split.h, SplitGeo MODE 2) instantiations [(192, 208)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8h_fused_192_208).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8h_fused_192_208(int hpe, int hpp) {
  if (hpe == 192 && hpp == 208) return gaudi::sampler_kernel8h<192, 208>;
  return nullptr;
}
