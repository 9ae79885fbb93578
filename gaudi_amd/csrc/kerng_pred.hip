// kerng_pred.hip -- V4G kernels (node buffers in global memory), predictor only (the unit-test entry points).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kerng_pred(int hpe, int hpp) {
  if (hpe == 0 && hpp == 48) return gaudi::sampler_kernel_g<0, 48>;
  if (hpe == 0 && hpp == 208) return gaudi::sampler_kernel_g<0, 208>;
  return nullptr;
}
