// kernse_edm.hip -- 4-wave sampler kernels for sin_embedding denoisers (sampler_kernel.h: V4S / V4GS; edm_device.h: EF = 24), EDM
// only: resident node buffers at the tiny and the default width (the other widths: kernse_edm_more.hip), node buffers in global
// memory for molecules beyond the LDS limit (own translation unit; looked up through gaudi_kernse_edm).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kernse_edm(int hpe, int hpp, int gn) {
  if (!gn) {
    if (hpe == 32 && hpp == 0) return gaudi::sampler_kernel_se<32, 0>;
    if (hpe == 192 && hpp == 0) return gaudi::sampler_kernel_se<192, 0>;
  } else {
    if (hpe == 32 && hpp == 0) return gaudi::sampler_kernel_gse<32, 0>;
    if (hpe == 192 && hpp == 0) return gaudi::sampler_kernel_gse<192, 0>;
  }
  return nullptr;
}
