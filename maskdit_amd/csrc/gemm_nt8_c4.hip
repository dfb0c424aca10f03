// gemm_nt8, epilogue class 4 (d-activation (x saved pre-activation)): see gemm_nt8_impl.h
#define NT8_CLASS 4
#include "gemm_nt8_impl.h"
NT8_INSTANTIATE_CLASS(1, NT8_INST(4, 2))
