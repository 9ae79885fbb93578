// kernse_edm_more.hip -- 4-wave sampler kernels for sin_embedding denoisers (sampler_kernel.h: V4S; edm_device.h: EF = 24), EDM only,
// the remaining hidden sizes of the 4-wave family: an unguided chain runs on these alone, a guided one as two launches per step with
// the ordinary predictor-only kernel (gaudi_hip.hip: run_two).  Own translation unit; looked up through gaudi_kernse_edm_more.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kernse_edm_more(int hpe, int hpp, int gn) {
  if (gn || hpp) return nullptr;
  if (hpe == 48) return gaudi::sampler_kernel_se<48, 0>;
  if (hpe == 64) return gaudi::sampler_kernel_se<64, 0>;
  if (hpe == 128) return gaudi::sampler_kernel_se<128, 0>;
  if (hpe == 208) return gaudi::sampler_kernel_se<208, 0>;
  if (hpe == 256) return gaudi::sampler_kernel_se<256, 0>;
  return nullptr;
}
