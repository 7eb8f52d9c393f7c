// microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 with 1/2/4 accumulator chains
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
void run(int waves, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * waves * 64 * 4); hipMalloc(&cyc, 8);
  int iters = 1000;
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("chains=%d waves/WG=%d blocks=%d: %.1f cycles per MFMA\n", CH, waves, blocks, (double)h / (iters * 8.0 * CH));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1, 157, 256}) {
    run<1>(1, blocks); run<2>(1, blocks); run<4>(1, blocks);
    run<1>(4, blocks); run<2>(4, blocks); run<4>(4, blocks);
    run<2>(8, blocks);
  }
  return 0;
}
