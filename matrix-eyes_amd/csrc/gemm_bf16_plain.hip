// bf16 instantiations of the GEMM core, row-major A.
#include "gemm_core.h"
namespace me {
ME_GEMM_DISPATCH_BODY(bf16, A_PLAIN, EPI_STORE)
ME_GEMM_DISPATCH_BODY(bf16, A_PLAIN, EPI_RESID_SCALE)
ME_GEMM_DISPATCH_BODY(bf16, A_PLAIN, EPI_PATCH_EMBED)
ME_GEMM_DISPATCH_BODY(bf16, A_PLAIN, EPI_CONVT)
}  // namespace me
