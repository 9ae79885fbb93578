// kern8s2_fused_192_208.hip -- sampler_kernel8s2: the resident full-ring split-operand kernel (kern8s_fused_192_208.hip) with FR set -- node-GEMM
// split passes and epilogues recompute their lane addresses per call (w8_nodes_f16.h: FL); the host runs it when a workgroup has
// more than 16 node slots (two column tiles per node GEMM: C4, packed workgroups).  Instantiations [(192, 208)]; looked up by
// gaudi_hip.hip through gaudi_kern8s2_fused_192_208.
#include "sampler_kernel.h"

typedef void (*kernel_fn)(const gaudi::KParams);

kernel_fn gaudi_kern8s2_fused_192_208(int hpe, int hpp) {
  if (hpe == 192 && hpp == 208) return gaudi::sampler_kernel8s2<192, 208>;
  return nullptr;
}
