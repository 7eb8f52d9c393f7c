// the one-wave pieces of the split-group moment matching, timed alone (wave 0 of a 512-thread workgroup, the others at a barrier)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../prob_mbrl_amd/csrc/pmbrl_mm_w.h"
template <int DD>
__global__ __launch_bounds__(512, 2) void k(long long* out, float* sink, int M, int nvalid, int reps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xb = smem;                  // [64][DD]
  float* z = smem + 64 * DD;         // [64][DD]
  float* g = z + 64 * DD;
  double* scr = reinterpret_cast<double*>(g + 64 * DD);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 64 * DD; i += 512) {
    xb[i] = 0.01f * ((i * 37) % 101) + 0.5f * (i % DD) + ((i / DD) % 7) * 0.013f * ((i % DD) + 1);
    z[i] = 0.02f * ((i * 53) % 97) - 1.0f + ((i / DD) % 5) * 0.11f * ((i % DD) + 2);
    g[i] = 0.001f * ((i * 11) % 89);
  }
  __syncthreads();
  long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float res = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    if (wid == 0) {
      long long t0 = __builtin_readcyclecounter();
      const double refl = (double)xb[(lane & 15) < DD ? (lane & 15) : 0];
      pm_f64x4 G = pm_mm_gram_lds<DD, 1, 8>(xb, xb, nvalid, lane, refl);
      asm volatile("" : "+v"(G));
      long long t1 = __builtin_readcyclecounter();
      MMW<DD> q;
      const pm_f64x4 Gz = pm_mm_gram_lds<DD, 2, 8>(z, z, M, lane, 0.0);
      pm_mmw_zstats<DD>(Gz, M, q);
      asm volatile("" : "+v"(q.zi[0]), "+v"(q.zm[DD - 1]));
      long long t2 = __builtin_readcyclecounter();
      // pretend the other part saw the same rows: double the sums
      G[0] *= 2.0; G[1] *= 2.0;
      const bool ok = pm_mmw_factor<DD, false>(G, 2 * nvalid, q);
      asm volatile("" : "+v"(q.L[DD - 1][DD - 1]));
      long long t3 = __builtin_readcyclecounter();
      if (lane < nvalid) {
        double zh[DD];
#pragma unroll
        for (int c = 0; c < DD; ++c) zh[c] = ((double)z[lane * DD + c] - q.zm[c]) * q.zi[c];
#pragma unroll
        for (int j = 0; j < DD; ++j) {
          double a = q.mean[j];
#pragma unroll
          for (int c = 0; c <= j; ++c) a += zh[c] * q.L[j][c];
          xb[lane * DD + j] = (float)a * 0.5f + 0.25f * xb[lane * DD + j];
        }
      }
      long long t4 = __builtin_readcyclecounter();
      // adjoint pieces: raw H tile, LDS tail
      pm_f64x4 H = pm_mm_gram_h_lds<DD, 8>(g, z, nvalid, lane, 0.0, 1.0);
      asm volatile("" : "+v"(H));
      long long t5 = __builtin_readcyclecounter();
      const MMScratch qs = pm_mm_carve(scr, DD);
      if (lane == 0) {
        for (int j = 0; j < DD; ++j) {
          qs.mean[j] = q.mean[j]; qs.zmean[j] = q.zm[j]; qs.zistd[j] = q.zi[j]; qs.invd[j] = q.invd[j]; qs.mbar[j] = 0.1 * j;
          for (int c = 0; c < DD; ++c) { qs.Lm[j * DD + c] = c <= j ? q.L[j][c] : 0.0; qs.P[j * DD + c] = c <= j ? 0.01 * (j + c + 1) : 0.0; }
        }
      }
      pm_wave_sync();
      long long t6 = __builtin_readcyclecounter();
      pm_mm_bwd_l_tail<DD>(qs, lane, M);
      long long t7 = __builtin_readcyclecounter();
      res += (float)qs.P[lane % (DD * DD)] + (ok ? 1.f : 0.f) + (float)H[0];
      acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t7 - t6;
    }
  }
  if (tid == 0) for (int i = 0; i < 6; ++i) out[i] = acc[i] / reps;
  sink[tid] = res + xb[tid % (64 * DD)];
}
template <int DD>
void run(int M, int nvalid) {
  long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 512 * 4);
  const size_t sm = 3 * 64 * DD * 4 + pm_mm_scratch_doubles(DD) * 8 + 64;
  hipLaunchKernelGGL(k<DD>, dim3(1), dim3(512), sm, 0, out, sink, M, nvalid, 20);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
  printf("D=%d M=%d own=%d: own-rows Gram %lld | z Gram + z stats %lld | factor %lld | apply %lld | H tile %lld | adjoint d x d tail %lld cycles\n", DD, M, nvalid, h[0], h[1], h[2], h[3], h[4], h[5]);
}
int main() { run<4>(25, 13); run<6>(50, 25); run<4>(25, 13); return 0; }
