// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds the 16-bit value i at element i; every lane passes the byte address
// addr[lane] and gets back four 16-bit values.  Prints, per lane, the four element indices it received.
//   hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(const unsigned* addr, unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(unsigned long long)(const void*)lds + addr[threadIdx.x];
  unsigned v0, v1;
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  v0 = (unsigned)r; v1 = (unsigned)(r >> 32);
  out[threadIdx.x * 4 + 0] = v0 & 0xffff; out[threadIdx.x * 4 + 1] = v0 >> 16;
  out[threadIdx.x * 4 + 2] = v1 & 0xffff; out[threadIdx.x * 4 + 3] = v1 >> 16;
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  unsigned h[64]; unsigned short ho[256];
  for (int l = 0; l < 64; ++l) {
    const int t = l & 15, g = l >> 4;
    if (mode == 0) h[l] = l * 8;                                   // lane-linear: lane l at elements 4l .. 4l+3
    else if (mode == 1) h[l] = ((t >> 2) * 64 + (t & 3) * 4 + g * 16) * 2;   // group g: a [4 rows][16 cols] block of a [.][64]-element-pitch image, cols 16g..
    else h[l] = ((t >> 2) * 512 + (t & 3) * 4 + g * 16) * 2;       // the same with row pitch 512 elements
  }
  unsigned* d; unsigned short* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d addr-elem %4u -> %4u %4u %4u %4u\n", l, h[l] / 2, ho[4 * l], ho[4 * l + 1], ho[4 * l + 2], ho[4 * l + 3]);
  return 0;
}
