// gemm_nt8 dispatcher: epilogue -> kernel class (the kernels live in gemm_nt8_impl.h, one translation unit
// per class: gemm_nt8_c0..c4.hip).
#include "common.h"
#include "../../include/maskdit_hip.h"
#include "gemm_common.h"

int launch_gemm_nt8_class0(const NTParams& p, int nf, int wr, hipStream_t stream);  // plain bf16
int launch_gemm_nt8_class1(const NTParams& p, int nf, int wr, hipStream_t stream);  // fp32 (+ optional bf16)
int launch_gemm_nt8_class2(const NTParams& p, int nf, int wr, hipStream_t stream);  // GELU / SiLU dual output
int launch_gemm_nt8_class3(const NTParams& p, int nf, int wr, hipStream_t stream);  // gate * y + residual
int launch_gemm_nt8_class4(const NTParams& p, int nf, int wr, hipStream_t stream);  // d-activation
#ifdef MDT_EXPERIMENTS  // gemm_nt8_c5.hip / gemm_nt8_x.hip: part of libmaskdit_hip_exp.so only (`make experiments`)
int launch_gemm_nt8_class5(const NTParams& p, hipStream_t stream);                  // overlap experiment (E_TRK)
int launch_gemm_nt8_sched(const NTParams& p, int nf, int sched, hipStream_t stream);  // phase-placement experiments
#endif

int nt8_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  // "nt8_max_cus" caps the persistent grid: CUs left free for a concurrent RCCL kernel (data-parallel overlap), and
  // the half-chip A/B of tools/nt8_bench.py
  const int cap = mdt_get_tuning_int(MDT_TUNE_NT8_MAX_CUS);
  return (cap > 0 && cap < n) ? cap : n;
}

// widest column tile (in 64-column units) the class of epilogue `epi` is instantiated for
int nt8_max_nf(int epi) { return (epi & 0xff) == MDT_EPI_GATE_RES ? 3 : 4; }

int launch_gemm_nt8(const NTParams& p, int nf, int wr, hipStream_t stream) {
#ifdef MDT_EXPERIMENTS
  if ((p.epi & 0xff) == MDT_EPI_GATE_RES && nf == 3 && wr == 2 && mdt_get_tuning_int(MDT_TUNE_NT8_TRICKLE))
    return launch_gemm_nt8_class5(p, stream);  // timing experiment: outputs are garbage
  const int sched = mdt_get_tuning_int(MDT_TUNE_NT8_SCHED);
  if (sched && (p.epi & 0xff) == MDT_EPI_BF16 && wr == 2 && nf >= 3) return launch_gemm_nt8_sched(p, nf, sched, stream);
#endif
  switch (p.epi & 0xff) {
    case MDT_EPI_BF16: return launch_gemm_nt8_class0(p, nf, wr, stream);
    case MDT_EPI_F32: return launch_gemm_nt8_class1(p, nf, wr, stream);
    case MDT_EPI_GELU:
    case MDT_EPI_SILU: return launch_gemm_nt8_class2(p, nf, wr, stream);
    case MDT_EPI_GATE_RES: return launch_gemm_nt8_class3(p, nf, wr, stream);
    default: return launch_gemm_nt8_class4(p, nf, wr, stream);
  }
}
