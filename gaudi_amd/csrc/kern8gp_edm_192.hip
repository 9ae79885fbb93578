// kern8gp_edm_192.hip -- 8-wave sampler kernels with three of the five node buffers in global memory and P / Q in LDS
// (sampler_kernel.h: V8T<1, true, 2>; w8_edm.h: gn_lds_buffers -- round 6): what a molecule beyond the resident kernels' LDS limit
// runs on where that plan fits (gaudi_hip.hip: stage_graph8), kern8g_* otherwise.  Own translation unit; looked up through
// gaudi_kern8gp_edm_192.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8gp_edm_192(int hpe, int hpp) {
  if (hpe == 192 && hpp == 0) return gaudi::sampler_kernel8gp<192, 0>;
  return nullptr;
}
