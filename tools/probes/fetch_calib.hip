// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS library's access pattern:
// every lane loads / stores one double, a wave covers 512 contiguous bytes (dwordx2 per lane),
// streamed once over a buffer much larger than the 256 MiB Infinity Cache.
// (MI355X_MICROARCH.md calibrates only the 16-B-per-lane pattern.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rd(const double* __restrict__ a, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += stride) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void wr(double* __restrict__ a, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) a[i] = (double)i;
}
int main() {
  const size_t n = (size_t)1 << 29;  // 4 GiB of doubles
  double *a, *o;
  hipMalloc(&a, n * 8); hipMalloc(&o, 8);
  hipMemset(a, 0, n * 8);
  hipDeviceSynchronize();
  rd<<<2048, 64>>>(a, o, n);
  hipDeviceSynchronize();
  wr<<<2048, 64>>>(a, n);
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", n * 8);
  return 0;
}
