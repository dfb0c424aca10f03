// gemm_nt8 scheduling EXPERIMENTS (tools/nt8_bench.py --sched): the plain-bf16 class with the phase-placement variants
// of gemm_nt8_impl.h's PAIR_BODY, selected by mdt_set_tuning("nt8_sched", v).  Not on the product path.
#define NT8_CLASS 0
#include "gemm_nt8_impl.h"
#define XI(S) template __global__ void gemm_nt8_kernel<3, 2, 0, S>(NTParams); template __global__ void gemm_nt8_kernel<4, 2, 0, S>(NTParams);
XI(8) XI(9) XI(10) XI(1029) XI(517) XI(1541) XI(37) XI(69) XI(133) XI(101) XI(229)
#define XL(S) case S: if (nf == 4) hipLaunchKernelGGL((gemm_nt8_kernel<4, 2, 0, S>), dim3(grid), dim3(512), 0, stream, p); \
                      else hipLaunchKernelGGL((gemm_nt8_kernel<3, 2, 0, S>), dim3(grid), dim3(512), 0, stream, p); break;
int launch_gemm_nt8_sched(const NTParams& p, int nf, int sched, hipStream_t stream) {
  const int ntiles = (p.M / 256) * (p.N / (64 * nf));
  const int slots = nt8_num_cus();
  const int grid = ntiles < slots ? ntiles : slots;
  switch (sched) {  // 1029 = the round-2 tile hand-over (next tile's K-tiles as one burst); 517 / 1541 = + timing stamps
    XL(8) XL(9) XL(10) XL(1029) XL(517) XL(1541) XL(37) XL(69) XL(133) XL(101) XL(229)
    default: mdt_set_error("gemm_nt8: unknown nt8_sched"); return MDT_ERR_ARG;
  }
  return mdt_check_launch("gemm_nt8_sched");
}

// tools/nt8_stamps.py: copy the stamp table of the experiment kernels to the host (64 tiles x 4 events)
extern "C" int nt8x_read_stamps(unsigned long long* host64x4) {
  return hipMemcpyFromSymbol(host64x4, HIP_SYMBOL(nt8_stamps), sizeof(unsigned long long) * 2 * 64 * 4) == hipSuccess ? MDT_OK : MDT_ERR_LAUNCH;
}
