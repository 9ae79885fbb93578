// kernse_fused.hip -- 4-wave sampler kernels for sin_embedding denoisers (sampler_kernel.h: V4S; edm_device.h: EF = 24) with the
// guidance predictor fused: the tiny and the default width pairs (own translation unit; looked up through gaudi_kernse_fused).
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kernse_fused(int hpe, int hpp, int gn) {
  if (gn) return nullptr;
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel_se<32, 48>;
  if (hpe == 192 && hpp == 208) return gaudi::sampler_kernel_se<192, 208>;
  return nullptr;
}
