// kern8mp_fused.hip -- 8-wave kernels for WIDE groups on the FULL weight ring (round 6): several rounds of eight edge tiles, split
// operands, the predictor's fifth node buffer in the workgroup's global scratch (sampler_kernel.h: V8T<1, true, 0, false, true>;
// w8_pred.h: PredSmem, PG) -- two cata-11 molecules per workgroup do not fit five resident buffers beside the full ring at the
// default widths [(192, 208) and the test widths (32, 48)].  Own translation unit; looked up through gaudi_kern8mp_fused.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8mp_fused(int hpe, int hpp) {
  if (hpe == 192 && hpp == 208) return gaudi::sampler_kernel8mp<192, 208>;
  if (hpe == 32 && hpp == 48) return gaudi::sampler_kernel8mp<32, 48>;
  return nullptr;
}
