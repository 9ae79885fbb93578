// kern8g_pred_small.hip -- 8-wave sampler kernels with the node buffers in global memory (sampler_kernel.h: V8T<1, true, 1> = V8G, round 4):
// molecules whose node buffers do not fit 160 KiB of LDS beside the weight ring; split edge GEMMs with the full ring, several
// rounds of edge tiles in the predictor [predictor only, the test widths].  Own translation unit (the instantiations compile in parallel); looked up by
// gaudi_hip.hip through gaudi_kern8g_pred_small.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8g_pred_small(int hpe, int hpp) {
  if (hpe == 0 && hpp == 32) return gaudi::sampler_kernel8g<0, 32>;
  if (hpe == 0 && hpp == 48) return gaudi::sampler_kernel8g<0, 48>;
  if (hpe == 0 && hpp == 64) return gaudi::sampler_kernel8g<0, 64>;
  return nullptr;
}
