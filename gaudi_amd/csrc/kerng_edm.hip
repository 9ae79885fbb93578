// kerng_edm.hip -- 4-wave sampler kernels with the node buffers in global memory (sampler_kernel.h: V4G), EDM only:
// molecules whose working set exceeds 160 KiB of LDS (own translation unit; looked up through gaudi_kerng_edm).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kerng_edm(int hpe, int hpp) {
  if (hpe == 32 && hpp == 0) return gaudi::sampler_kernel_g<32, 0>;
  if (hpe == 192 && hpp == 0) return gaudi::sampler_kernel_g<192, 0>;
  return nullptr;
}
