// gemm_nt8, epilogue class 3 (gate * y + residual): see gemm_nt8_impl.h
#define NT8_CLASS 3
#include "gemm_nt8_impl.h"
NT8_INSTANTIATE_CLASS(0, )
