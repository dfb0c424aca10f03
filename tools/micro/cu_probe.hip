// Where does a CU-masked stream run its workgroups?  (run on the GPU box; used by tools/cu_partition_bench.py)
//   cu_probe(out, n_wg, stream): every workgroup records HW_ID / XCC_ID of its wave 0 -> out[wg] = xcc << 16 | se << 8 | sh << 7 ... (see below)
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libcu_probe.so tools/micro/cu_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cu_probe_kernel(uint32_t* out, int spin) {
  // s_getreg_b32: simm16 = (size - 1) << 11 | offset << 6 | id;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20 (gfx940+)
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
  // keep the workgroup resident for a while so that the dispatcher has to spread the grid over every allowed CU
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);  // hw: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
}

extern "C" int cu_probe(uint32_t* out, int n_wg, int spin, void* stream) {
  hipLaunchKernelGGL(cu_probe_kernel, dim3(n_wg), dim3(512), 0, (hipStream_t)stream, out, spin);
  return (int)hipGetLastError();
}
