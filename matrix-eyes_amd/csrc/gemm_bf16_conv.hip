// bf16 instantiations of the GEMM core, implicit-GEMM convolution A operand.
#include "gemm_core.h"
namespace me {
ME_GEMM_DISPATCH_BODY(bf16, A_CONV, EPI_STORE)
template <>
void gemm_dispatch<bf16, A_CONV, EPI_HEAD_FINAL>(const GemmParams& p, int, hipStream_t stream) {
    gemm_launch_cfg<bf16, 256, 32, 4, 1, A_CONV, EPI_HEAD_FINAL>(p, stream);
}
// the composed head (weights.hip compose_head) on the 128-channel halo tile
void head_composed_launch_bf16(const GemmParams& p, hipStream_t stream) { conv_halo_launch<bf16, EPI_HEAD_COMPOSED, 16, 128>(p, stream); }
}  // namespace me
