// microbenchmark: the streamed-layer chunk loop of pmbrl_fast.h in isolation (inline-asm weight
// loads from L2, explicit waits, LDS B-operand pipeline), with 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../prob_mbrl_amd/csrc -I../../include stream_rate.hip -o stream_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "pmbrl_fast.h"
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void k(const float* w, float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ld = 232;
  for (int i = tid; i < 16 * ld; i += NW * 64) smem[i] = 0.001f * i;
  __syncthreads();
  FragS<7> fa, fb;
  const unsigned vo0 = lane * 16u, vo1 = vo0 + 4096u;
  const float* wp = w + (size_t)wid * 14 * 256;
  frag_load_p<7>(fa, wp, vo0, vo1);
  f32x4 acc[2][1];
  acc[0][0] = acc[1][0] = f32x4{0, 0, 0, 0};
  BPair<1> b0;
  bpair_load<1, 7>(b0, smem + (lane & 15) * ld + 4 * (lane >> 4), ld, 0);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    const float* w1 = w + (size_t)((i * 2 + 1 + wid * 3) % 180) * 7 * 256;
    const float* w2 = w + (size_t)((i * 2 + 2 + wid * 3) % 180) * 7 * 256;
    if (MODE == 1) frag_load_p<7>(fb, w1, vo0, vo1);
    if (MODE == 1) frag_wait<7, 7>(fa);
    frag_compute<1, 7>(fa, 0, 7, smem, ld, lane, acc, b0);
    if (MODE == 1) frag_load_p<7>(fa, w2, vo0, vo1);
    if (MODE == 1) frag_wait<7, 7>(fb);
    frag_compute<1, 7>(fb, 7, 0, smem, ld, lane, acc, b0);
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)");
  f32x4 s = acc[0][0] + acc[1][0] + fa.a[0] + fb.a[0];
  out[blockIdx.x * NW * 64 + tid] = s[0] + s[1] + s[2] + s[3];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NW, int MODE>
void run(const float* w, float* out, long long* cyc, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NW, MODE>), dim3(blocks), dim3(NW * 64), 16 * 232 * 4, 0, w, out, cyc, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NW, MODE>), dim3(blocks), dim3(NW * 64), 16 * 232 * 4, 0, w, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("   wall %.1f us -> %.2f ns per MFMA per SIMD; counter rate %.2f GHz\n", ms * 1e3,
         ms * 1e6 / (iters * 56.0 * (NW / 4)), 0.0);
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("   counter ticks %lld in %.1f us -> %.2f GHz\n", h, ms * 1e3, h / (ms * 1e6));
  printf("waves/WG=%d blocks=%d loads=%d: %.1f cycles per MFMA per wave -> %.1f cycles per MFMA per SIMD (ideal 32)\n", NW, blocks,
         MODE, (double)h / (iters * 56.0), (double)h / (iters * 56.0) / (NW / 4));
}
int main() {
  float *w, *out;
  long long* cyc;
  hipMalloc(&w, 2600 * 1024);
  hipMemset(w, 0, 2600 * 1024);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 8);
  for (int blocks : {1, 157}) {
    run<4, 0>(w, out, cyc, blocks);
    run<4, 1>(w, out, cyc, blocks);
    run<8, 0>(w, out, cyc, blocks);
    run<8, 1>(w, out, cyc, blocks);
  }
  return 0;
}
