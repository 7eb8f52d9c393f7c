#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void k(long long* out, double* sink, int M, double seed) {
  __shared__ float rows[64 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 512; i += 512) rows[i] = 0.01f * i + (float)seed;
  __syncthreads();
  long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double res = 0;
  for (int rep = 0; rep < 10; ++rep) {
    __syncthreads();
    if (wid == 0) {
      const int g = lane >> 4, c = lane & 15;
      double x = seed + lane, y = seed;
      // A: 4 dependent MFMAs, operands ready
      f64x4 G = {0, 0, 0, 0};
      long long t0 = __builtin_readcyclecounter();
#pragma unroll
      for (int i = 0; i < 4; ++i) G = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G, 0, 0, 0);
      res += G[0]; asm volatile("" : "+v"(res));
      long long t1 = __builtin_readcyclecounter();
      // B: 8 dependent MFMAs
      G = f64x4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 8; ++i) G = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G, 0, 0, 0);
      res += G[0]; asm volatile("" : "+v"(res));
      long long t2 = __builtin_readcyclecounter();
      // C: 4 LDS loads, then operands formed (cvt, sub, fma, select), 4 MFMAs in two chains
      f64x4 G0 = {0, 0, 0, 0}, G1 = {0, 0, 0, 0};
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = rows[((4 * u + g) & 63) * 8 + (c & 7)];
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double xx = __builtin_fma((double)v[u] - y, x, y);
        xx = 4 * u + g < M ? xx : 0.0;
        asm volatile("" : "+v"(xx));
        if (u & 1) G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx, xx, G1, 0, 0, 0);
        else G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx, xx, G0, 0, 0, 0);
      }
      G = G0 + G1;
      res += G[1]; asm volatile("" : "+v"(res));
      long long t3 = __builtin_readcyclecounter();
      // D: the same with a uniform branch around every MFMA
      G0 = f64x4{0, 0, 0, 0}; G1 = f64x4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = rows[((4 * u + g + 1) & 63) * 8 + (c & 7)];
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double xx = __builtin_fma((double)v[u] - y, x, y);
        xx = 4 * u + g < M ? xx : 0.0;
        asm volatile("" : "+v"(xx));
        if (4 * u < M) {
          if (u & 1) G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx, xx, G1, 0, 0, 0);
          else G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx, xx, G0, 0, 0, 0);
        }
      }
      G = G0 + G1;
      res += G[1]; asm volatile("" : "+v"(res));
      long long t4 = __builtin_readcyclecounter();
      // E: one LDS load -> use
      float w = rows[(lane * 3) & 511];
      res += w; asm volatile("" : "+v"(res));
      long long t5 = __builtin_readcyclecounter();
      // F: empty
      long long t6 = __builtin_readcyclecounter();
      acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5;
    }
  }
  if (tid == 0) for (int i = 0; i < 6; ++i) out[i] = acc[i] / 10;
  sink[tid] = res;
}
int main() {
  long long* out; double* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 512 * 8);
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, sink, 13, 1.25); hipDeviceSynchronize(); }
  long long h[8]; hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
  printf("4 dep mfma %lld | 8 dep mfma %lld | lds + operands + 4 mfma (2 chains) %lld | same with uniform branches %lld | lds load->use %lld | empty %lld\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  return 0;
}
