// kern8g_fused_tiny.hip -- 8-wave sampler kernels with the node buffers in global memory (sampler_kernel.h: V8T<1, true, 1> = V8G, round 4):
// molecules whose node buffers do not fit 160 KiB of LDS beside the weight ring; split edge GEMMs with the full ring, several
// rounds of edge tiles in the predictor [the test widths].  Own translation unit (the instantiations compile in parallel); looked up by
// gaudi_hip.hip through gaudi_kern8g_fused_tiny.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8g_fused_tiny(int hpe, int hpp) {
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel8g<32, 48>;
  if (hpe == 32 && hpp == 32) return gaudi::sampler_kernel8g<32, 32>;
  if (hpe == 48 && hpp == 48) return gaudi::sampler_kernel8g<48, 48>;
  if (hpe == 64 && hpp == 64) return gaudi::sampler_kernel8g<64, 64>;
  return nullptr;
}
