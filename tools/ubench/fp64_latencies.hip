// micro-measurements for the one-wave fp64 chains of the moment matching (cycles per dependent operation, one wave on a CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define N 256
__global__ void k(long long* out, double* sink, double seed) {
  __shared__ double lds[512];
  const int lane = threadIdx.x;
  lds[lane] = seed + lane;
  lds[lane + 64] = seed;
  __syncthreads();
  long long t0, t1;
  double x = seed, y = seed * 0.5;
  // 1. dependent fp64 FMA chain
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i) { x = __builtin_fma(x, y, y); asm volatile("" : "+v"(x)); }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  // 2. independent fp64 FMAs (4 chains)
  double a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    a0 = __builtin_fma(a0, y, y); a1 = __builtin_fma(a1, y, y); a2 = __builtin_fma(a2, y, y); a3 = __builtin_fma(a3, y, y);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[1] = t1 - t0;
  x = a0 + a1 + a2 + a3;
  // 3. dependent MFMA f64 chain
  f64x4 G = {0, 0, 0, 0};
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 32; ++i) G = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, G, 0, 0, 0);
  asm volatile("" : "+v"(G));
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[2] = t1 - t0;
  // 4. two independent MFMA f64 chains
  f64x4 G2 = {0, 0, 0, 0};
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) { G = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, G, 0, 0, 0); G2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, G2, 0, 0, 0); }
  asm volatile("" : "+v"(G), "+v"(G2));
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[3] = t1 - t0;
  x += G[0] + G2[1];
  // 5. dependent LDS broadcast read chain (address depends on the previous value)
  int idx = lane & 1;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 32; ++i) { double v = lds[64 + idx]; idx = (int)(v - seed); asm volatile("" : "+v"(idx)); }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[4] = t1 - t0;
  // 6. readlane -> use chain (double = 2 readlanes)
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int lo = __builtin_amdgcn_readlane((int)__double2loint(x), i & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 63);
    x = __builtin_fma(__hiloint2double(hi, lo), y, x);
    asm volatile("" : "+v"(x));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[5] = t1 - t0;
  // 7. rsqrt (v_rsq_f64 + 2 Newton steps) dependent chain
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double r = __builtin_amdgcn_rsq(x + 2.0);
    double e = __builtin_fma(-(x + 2.0) * r, r, 1.0);
    r = __builtin_fma(r * 0.5, e, r);
    e = __builtin_fma(-(x + 2.0) * r, r, 1.0);
    r = __builtin_fma(r * 0.5, e, r);
    x = r;
    asm volatile("" : "+v"(x));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[6] = t1 - t0;
  // 8. fp32 -> fp64 convert + sub + fma (the Gram operand)
  float f = (float)x;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 64; ++i) { x = __builtin_fma((double)f - x, y, y); f = (float)x; asm volatile("" : "+v"(f)); }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[7] = t1 - t0;
  // 9. LDS write -> wave sync -> read (a round trip through LDS between lanes)
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lds[128 + lane] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    x += lds[128 + ((lane + 1) & 63)];
    asm volatile("" : "+v"(x));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[8] = t1 - t0;
  // 10. __syncthreads with 8 waves all arriving together
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) __syncthreads();
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[9] = t1 - t0;
  sink[threadIdx.x] = x + G[1] + G2[2] + f + idx;
}
int main() {
  long long* out; double* sink;
  hipMalloc(&out, 16 * 8); hipMalloc(&sink, 512 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, sink, 1.000001);
    hipDeviceSynchronize();
  }
  long long h[16]; hipMemcpy(h, out, 16 * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"dep fp64 fma", "4 indep fp64 fma chains (per fma)", "dep mfma f64 16x16x4", "2 indep mfma f64 chains (per mfma)", "dep LDS read",
                      "readlane x2 + fma", "rsq + 2 newton", "cvt+sub+fma+cvt", "LDS write/sync/read", "syncthreads (8 waves)"};
  const int cnt[] = {256, 256, 32, 32, 32, 64, 16, 64, 16, 16};
  for (int i = 0; i < 10; ++i) printf("%-40s %8lld cycles total, %7.1f each (s_memtime ticks: 100 MHz? see below)\n", nm[i], h[i], (double)h[i] / cnt[i]);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("clock rate %d kHz, wall clock rate %d kHz\n", clk, wall);
  return 0;
}
