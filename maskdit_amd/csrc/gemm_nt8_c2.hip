// gemm_nt8, epilogue class 2 (activation (GELU / SiLU, dual output)): see gemm_nt8_impl.h
#define NT8_CLASS 2
#include "gemm_nt8_impl.h"
NT8_INSTANTIATE_CLASS(1, NT8_INST(4, 2))
