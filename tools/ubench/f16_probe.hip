// probe: does v_mfma_f32_16x16x32_f16 keep subnormal fp16 inputs, and how are the products accumulated?
//   hipcc --offload-arch=gfx950 -O3 f16_probe.hip -o f16_probe && ./f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  // A[i][k]: row i = lane&15, k = 8*(lane>>4)+e ; B[k][j]: col j = lane&15
  f16x8 a, b;
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; ab[e] = (__bf16)0.f; bb[e] = (__bf16)0.f; }
  if ((lane >> 4) == 0) {
    a[0] = (_Float16)3.0e-6f;   // subnormal in fp16 (min normal 6.1e-5)
    b[0] = (_Float16)1024.f;
    a[1] = (_Float16)1.0f;
    b[1] = (_Float16)1.0f;
    ab[0] = (__bf16)1.0e-39f;   // subnormal in bf16 / fp32
    bb[0] = (__bf16)1.0e10f;
  }
  f32x4 c = {0, 0, 0, 0};
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c, 0, 0, 0);
  if (lane == 0) { out[0] = d[0]; out[1] = (float)a[0] * 1024.f + 1.f; out[2] = d2[0]; out[3] = (float)ab[0] * 1e10f; }
}
int main() {
  float* o; hipMalloc(&o, 64); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float h[4]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("f16 mfma: got %.9g expect %.9g (1.0 exactly would mean the subnormal input was flushed)\n", h[0], h[1]);
  printf("bf16 mfma: got %.9g expect %.9g\n", h[2], h[3]);
  return 0;
}
