// kern8gp_tiny.hip -- 8-wave sampler kernels with three of the five node buffers in global memory and P / Q in LDS
// (sampler_kernel.h: V8T<1, true, 2>; w8_edm.h: gn_lds_buffers -- round 6) [the test widths: fused, EDM only, predictor only].
// Own translation unit; looked up through gaudi_kern8gp_tiny.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8gp_tiny(int hpe, int hpp) {
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel8gp<32, 48>;
  if (hpe == 32 && hpp == 0) return gaudi::sampler_kernel8gp<32, 0>;
  if (hpe == 0 && hpp == 48) return gaudi::sampler_kernel8gp<0, 48>;
  return nullptr;
}
