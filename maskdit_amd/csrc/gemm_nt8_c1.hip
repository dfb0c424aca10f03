// gemm_nt8, epilogue class 1 (fp32): see gemm_nt8_impl.h
#define NT8_CLASS 1
#include "gemm_nt8_impl.h"
NT8_INSTANTIATE_CLASS(1, NT8_INST(4, 2))
