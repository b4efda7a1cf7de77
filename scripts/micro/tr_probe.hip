// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive?  LDS holds u16 value = element index;
// lane l passes the byte address base + 8 l (pattern 0) or row-strided addresses (pattern 1: lane l -> row (l % 16), 8-byte
// chunk (l / 16) of 64-byte rows).  Prints, per lane, the four element indices it got.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(int pattern, int* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr_elems;
  if (pattern == 0) addr_elems = 4 * l;                       // contiguous: lane l -> elements 4l .. 4l+3
  else if (pattern == 1) addr_elems = (l % 16) * 32 + (l / 16) * 4;   // 64-byte rows (32 elements): row l%16, chunk l/16
  else addr_elems = (l % 4) * 32 + (l / 4) * 4;               // row l%4, chunk l/4
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)(lds + addr_elems));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  for (int pat = 0; pat < 3; ++pat) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pat, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
