#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned a = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  unsigned b = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
  out[blockIdx.x * 2] = a;
  out[blockIdx.x * 2 + 1] = b;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 8);
  hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, d);
  unsigned h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // count distinct (xcc, se, sh, cu, simd)
  unsigned seen[1 << 16] = {0}; int distinct = 0; unsigned ormask = 0, orx = 0;
  for (int i = 0; i < 4096; ++i) {
    unsigned a = h[2 * i], x = h[2 * i + 1] & 0xf;
    ormask |= a; orx |= x;
    unsigned key = (x << 12) | (((a >> 13) & 7) << 9) | (((a >> 12) & 1) << 8) | (((a >> 8) & 15) << 2) | ((a >> 4) & 3);
    if (!seen[key]) { seen[key] = 1; distinct++; }
  }
  printf("distinct simd keys %d ; OR of HW_ID %08x ; OR of XCC_ID %x ; sample %08x %08x\n", distinct, ormask, orx, h[0], h[1]);
  return 0;
}
