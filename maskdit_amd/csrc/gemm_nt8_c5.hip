// gemm_nt8, class 5: the epilogue-overlap EXPERIMENT kernel (see E_TRK in gemm_nt8_impl.h); 256 x 192 tiles only
#define NT8_CLASS 5
#include "gemm_nt8_impl.h"
template __global__ void gemm_nt8_kernel<3, 2, 5>(NTParams);
int launch_gemm_nt8_class5(const NTParams& p, hipStream_t stream) {
  const int ntiles = (p.M / 256) * (p.N / 192);
  const int slots = nt8_num_cus();
  const int grid = ntiles < slots ? ntiles : slots;
  hipLaunchKernelGGL((gemm_nt8_kernel<3, 2, 5>), dim3(grid), dim3(512), 0, stream, p);
  return mdt_check_launch("gemm_nt8_trk");
}
