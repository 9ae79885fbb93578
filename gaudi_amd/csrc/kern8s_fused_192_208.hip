// kern8s_fused_192_208.hip -- sampler_kernel8s (8 waves, edge and node GEMMs on fp16-pair operands: w8_split.h, w8_nodes_f16.h) instantiations [(192, 208)] (own translation unit so the
// instantiations compile in parallel; looked up by gaudi_hip.hip through gaudi_kern8s_fused_192_208).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8s_fused_192_208(int hpe, int hpp) {
  if (hpe == 192 && hpp == 208) return gaudi::sampler_kernel8s<192, 208>;
  return nullptr;
}
