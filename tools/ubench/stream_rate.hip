// microbenchmark: the streamed-layer inner loop (LDS B reads + MFMAs) in isolation, 8 waves/WG
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "pmbrl_fast.h"
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* w, float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ld = 232;
  for (int i = tid; i < 16 * ld; i += 512) smem[i] = 0.001f * i;
  __syncthreads();
  FragS<7> fa, fb;
  for (int c = 0; c < 7; ++c) { fa.a[c] = ldg4(w + c * 256 + lane * 4); fb.a[c] = ldg4(w + (7 + c) * 256 + lane * 4); }
  f32x4 acc[2][1];
  acc[0][0] = acc[1][0] = f32x4{0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1) {   // + weight loads from L2 as in the kernel
      for (int c = 0; c < 7; ++c) fb.a[c] = ldg4(w + ((i * 14 + 7 + c + wid * 28) % 2548) * 256 + lane * 4);
    }
    frag_compute<1, 7>(fa, 0, smem, ld, lane, acc);
    if (MODE == 1) {
      for (int c = 0; c < 7; ++c) fa.a[c] = ldg4(w + ((i * 14 + c + wid * 28) % 2548) * 256 + lane * 4);
    }
    frag_compute<1, 7>(fb, 1, smem, ld, lane, acc);
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0][0] + acc[1][0];
  out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *w, *out; long long* cyc;
  hipMalloc(&w, 2600 * 1024); hipMemset(w, 0, 2600 * 1024);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  for (int blocks : {1, 157}) for (int mode = 0; mode < 2; ++mode) {
    int iters = 400;
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 16 * 232 * 4, 0, w, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 16 * 232 * 4, 0, w, out, cyc, iters);
    }
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("blocks=%d mode=%d: %.1f cycles per MFMA per wave (56 MFMAs per iter)\n", blocks, mode, (double)h / (iters * 56.0));
  }
  return 0;
}
