// gemm_nt8, epilogue class 0 (plain bf16): see gemm_nt8_impl.h
#define NT8_CLASS 0
#include "gemm_nt8_impl.h"
NT8_INSTANTIATE_CLASS(1, NT8_INST(4, 2))
