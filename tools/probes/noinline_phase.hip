#include <hip/hip_runtime.h>
#include <cstdio>
struct Params { const double* a; double* out; int n; int m; double scale; };
typedef __attribute__((address_space(3))) double lds_d;

struct Carry { int cur; double mu; unsigned long long h; };

template <int K>
struct Solver {
  const Params& P;
  double* lds;
  int T;
  int cur; double mu; unsigned long long h;
  __device__ __forceinline__ Solver(const Params& p, double* l) : P(p), lds(l), T(threadIdx.x) {}
  __device__ __forceinline__ Carry pack() const { return Carry{cur, mu, h}; }
  __device__ __forceinline__ void unpack(const Carry& c) { cur = __builtin_amdgcn_readfirstlane(c.cur); mu = c.mu; h = c.h; }
  __device__ __forceinline__ void phase() {
    double acc = 0.0;
    for (int i = 0; i < P.n; ++i) { acc += P.a[i * 64 + T] * lds[(i + cur) & 63]; }
    lds[T] = acc * mu * P.scale;
    h += (unsigned long long)T * K;
    cur += 1;
  }
};

template <int K>
__device__ __attribute__((noinline)) Carry phase_fn(unsigned long long kp, Carry c) {
  extern __shared__ double lds[];
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)kp), hi = __builtin_amdgcn_readfirstlane((unsigned)(kp >> 32));
  typedef const __attribute__((address_space(4))) Params* CP;
  const Params& P = *(const Params*)(CP)(((unsigned long long)hi << 32) | lo);
  Solver<K> s(P, lds);
  s.unpack(c);
  s.phase();
  return s.pack();
}

template <int K>
__global__ void kern(Params P, int reps) {
  extern __shared__ double lds[];
  Solver<K> s(P, lds);
  s.cur = 0; s.mu = 1.0; s.h = 0;
  lds[threadIdx.x] = 1.0 + threadIdx.x;
  __syncthreads();
  for (int r = 0; r < reps; ++r) {
    Carry c = phase_fn<K>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr(), s.pack());
    s.unpack(c);
    __syncthreads();
  }
  P.out[blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x] + (double)s.h + s.cur;
}

int main() {
  double *a, *out; int n = 8;
  hipMalloc(&a, n * 64 * 8); hipMalloc(&out, 2 * 64 * 8);
  double ha[8 * 64]; for (int i = 0; i < n * 64; ++i) ha[i] = 0.001 * (i % 17);
  hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
  Params P{a, out, n, 3, 0.5};
  hipLaunchKernelGGL(kern<3>, dim3(2), dim3(64), 64 * 8, 0, P, 3);
  double ho[128]; hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  // host reference
  double l[64]; for (int t = 0; t < 64; ++t) l[t] = 1.0 + t;
  int cur = 0; 
  for (int r = 0; r < 3; ++r) { double nl[64]; for (int t = 0; t < 64; ++t) { double acc = 0; for (int i = 0; i < n; ++i) acc += ha[i * 64 + t] * l[(i + cur) & 63]; nl[t] = acc * 0.5; } for (int t = 0; t < 64; ++t) l[t] = nl[t]; cur++; }
  double err = 0; for (int t = 0; t < 64; ++t) { double e = fabs(ho[t] - (l[t] + 3.0 * t * 3 + 3)); if (e > err) err = e; }
  printf("max err %g (%g)\n", err, ho[5]);
  return err < 1e-12 ? 0 : 1;
}
