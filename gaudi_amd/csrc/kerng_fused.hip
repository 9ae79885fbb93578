// kerng_fused.hip -- V4G kernels (node buffers in global memory), guided sampling at the default widths and the test widths.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kerng_fused(int hpe, int hpp) {
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel_g<32, 48>;
  if (hpe == 192 && hpp == 208) return gaudi::sampler_kernel_g<192, 208>;
  return nullptr;
}
