#!/usr/bin/env python3
"""Writes and builds tools/_ab/scrub.hip -> libscrub.so: a kernel that sets every VGPR (v8-v255), every AGPR and all 160 KiB of LDS of every CU
to a pattern (2 048 blocks of 256 threads), for tools/qf_scrub.py.  Diagnostic only."""
import os, subprocess
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab"); os.makedirs(d, exist_ok=True)
L = ['#include <hip/hip_runtime.h>', '#include <stdint.h>',
     'extern "C" __global__ void __launch_bounds__(256) scrub_kernel(uint32_t pat, uint32_t* sink) {',
     '  extern __shared__ uint32_t lds[];', '  for (int i = threadIdx.x; i < 40960; i += 256) lds[i] = pat;', '  __syncthreads();']
L += ['  asm volatile("v_mov_b32 v%d, %%0" :: "v"(pat) : "v%d");' % (i, i) for i in range(8, 256)]
L += ['  asm volatile("v_accvgpr_write_b32 a%d, %%0" :: "v"(pat) : "a%d");' % (i, i) for i in range(256)]
L += ['  if (pat == 0x12345u && lds[threadIdx.x] == 7u) sink[0] = lds[5];', '}',
      'extern "C" int scrub(uint32_t pat, void* sink, void* stream) {', '  static bool init = false;',
      '  if (!init) { hipFuncSetAttribute((const void*)scrub_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); init = true; }',
      '  hipLaunchKernelGGL(scrub_kernel, dim3(2048), dim3(256), 163840, (hipStream_t)stream, pat, (uint32_t*)sink);',
      '  return (int)hipGetLastError();', '}']
open(os.path.join(d, "scrub.hip"), "w").write("\n".join(L) + "\n")
subprocess.check_call(["hipcc", "-O1", "-shared", "-fPIC", "--offload-arch=gfx950", os.path.join(d, "scrub.hip"), "-o", os.path.join(d, "libscrub.so")])
print(os.path.join(d, "libscrub.so"))
