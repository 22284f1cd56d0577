// Micro-probe (not product code): does the FP64 matrix pipe of gfx950 run beside the FP64 vector
// pipe, or do v_mfma_f64_* and v_fma_f64 share the same multipliers?  And does a DP VALU
// instruction with only one 16-lane row enabled in EXEC issue faster than a full one?
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
//
// Every wave runs ITERS x 64 instructions of its kind on independent accumulators.  Modes:
//   V      : every wave v_fmac_f64 (vector)                        1 or 2 waves per SIMD
//   M4     : every wave v_mfma_f64_4x4x4_4b_f64                    1 or 2 waves per SIMD
//   M16    : every wave v_mfma_f64_16x16x4_f64                     1 or 2 waves per SIMD
//   V|M4   : 512-thread blocks, waves 0-3 vector, waves 4-7 matrix (one of each per SIMD)
//   V|M16  : same with the 16x16x4 form
//   V+M4   : one wave alternates 4 vector FMAs and 1 MFMA in its own stream
//   Vrow   : vector stream with EXEC = 0xFFFF (one DPP row of the wave enabled)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef double v4d __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP2(x) x x

__device__ __forceinline__ void vec64(double (&w)[8], double a, double b) {
  // 64 independent-enough FMAs: 8 accumulators x 8
  REP8(asm volatile("v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %9\n\tv_fmac_f64_e32 %2, %8, %9\n\tv_fmac_f64_e32 %3, %8, %9\n\t"
                    "v_fmac_f64_e32 %4, %8, %9\n\tv_fmac_f64_e32 %5, %8, %9\n\tv_fmac_f64_e32 %6, %8, %9\n\tv_fmac_f64_e32 %7, %8, %9"
                    : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(a), "v"(b));)
}
__device__ __forceinline__ void vec64row(double (&w)[8], double a, double b) {
  // the same stream with only lanes 0..15 enabled; EXEC is narrowed and restored inside each asm
  // statement so that no compiler-generated instruction ever runs under the narrow mask
  REP8(asm volatile("s_mov_b64 exec, 0xffff\n\t"
                    "v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %9\n\tv_fmac_f64_e32 %2, %8, %9\n\tv_fmac_f64_e32 %3, %8, %9\n\t"
                    "v_fmac_f64_e32 %4, %8, %9\n\tv_fmac_f64_e32 %5, %8, %9\n\tv_fmac_f64_e32 %6, %8, %9\n\tv_fmac_f64_e32 %7, %8, %9\n\t"
                    "s_mov_b64 exec, -1"
                    : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(a), "v"(b));)
}
__device__ __forceinline__ void mfma4_64(double (&w)[8], double a, double b) {
  REP8(asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n\tv_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n\tv_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n\t"
                    "v_mfma_f64_4x4x4_4b_f64 %4, %8, %9, %4\n\tv_mfma_f64_4x4x4_4b_f64 %5, %8, %9, %5\n\tv_mfma_f64_4x4x4_4b_f64 %6, %8, %9, %6\n\tv_mfma_f64_4x4x4_4b_f64 %7, %8, %9, %7"
                    : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(a), "v"(b));)
}
__device__ __forceinline__ void mfma16_64(v4d (&w)[4], double a, double b) {
  // 64 MFMAs on 4 accumulators
  REP8(REP2(w[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w[0], 0, 0, 0); w[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w[1], 0, 0, 0);
            w[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w[2], 0, 0, 0); w[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w[3], 0, 0, 0);
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));))
}
__device__ __forceinline__ void mix80(double (&w)[8], double (&q)[8], double a, double b) {
  // 16 x (4 vector FMAs + 1 MFMA 4x4x4): 64 vector + 16 matrix instructions
  REP8(asm volatile("v_fmac_f64_e32 %0, %16, %17\n\tv_fmac_f64_e32 %1, %16, %17\n\tv_fmac_f64_e32 %2, %16, %17\n\tv_fmac_f64_e32 %3, %16, %17\n\t"
                    "v_mfma_f64_4x4x4_4b_f64 %8, %16, %17, %8\n\t"
                    "v_fmac_f64_e32 %4, %16, %17\n\tv_fmac_f64_e32 %5, %16, %17\n\tv_fmac_f64_e32 %6, %16, %17\n\tv_fmac_f64_e32 %7, %16, %17\n\t"
                    "v_mfma_f64_4x4x4_4b_f64 %9, %16, %17, %9"
                    : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                      "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(a), "v"(b));)
}

// kind per wave: 0 vector, 1 mfma 4x4x4, 2 mfma 16x16x4, 3 mixed stream, 4 vector with one row enabled
template <int KLO, int KHI>
__global__ void __launch_bounds__(512) probe(const double* __restrict__ in, double* __restrict__ out, int iters) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  const int wave = threadIdx.x >> 6;
  const int kind = (wave < 4) ? KLO : KHI;
  double a = in[t & 4095] * 1e-3, b = in[(t + 7) & 4095] * 1e-3;
  double w[8], q[8];
  v4d m[4];
  for (int i = 0; i < 8; ++i) { w[i] = 0; q[i] = 0; }
  for (int i = 0; i < 4; ++i) m[i] = v4d{0, 0, 0, 0};
  if (kind == 0) {
    for (int it = 0; it < iters; ++it) vec64(w, a, b);
  } else if (kind == 1) {
    for (int it = 0; it < iters; ++it) mfma4_64(w, a, b);
  } else if (kind == 2) {
    for (int it = 0; it < iters; ++it) mfma16_64(m, a, b);
  } else if (kind == 3) {
    for (int it = 0; it < iters; ++it) mix80(w, q, a, b);
  } else {
    for (int it = 0; it < iters; ++it) vec64row(w, a, b);
  }
  double r = 0;
  for (int i = 0; i < 8; ++i) r += w[i] + q[i];
  for (int i = 0; i < 4; ++i) r += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  out[t] = r;
}

template <int KLO, int KHI>
float run(const char* name, int threads, int blocks, const double* a, double* o, int iters, double flop_lo, double flop_hi) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<KLO, KHI>), dim3(blocks), dim3(threads), 0, 0, a, o, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const int wlo = threads >= 256 ? 4 : threads / 64, whi = threads / 64 - wlo;
  const double fl = (double)blocks * iters * (wlo * flop_lo + whi * flop_hi);
  printf("%-34s threads/block %3d blocks %5d : %8.3f ms  %7.2f TFLOP/s\n", name, threads, blocks, best, fl / best * 1e-9);
  return best;
}

int main() {
  const int iters = 20000;
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *a, *o;
  CHECK(hipMalloc(&a, 4096 * 8));
  CHECK(hipMalloc(&o, (size_t)8192 * 512 * 8));
  std::vector<double> ha(4096);
  for (int i = 0; i < 4096; i++) ha[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
  CHECK(hipMemcpy(a, ha.data(), 4096 * 8, hipMemcpyHostToDevice));
  // flops per wave per loop iteration
  const double FV = 64.0 * 64 * 2, FM4 = 64.0 * 4 * 64 * 2, FM16 = 64.0 * 16 * 16 * 4 * 2, FMIX = FV + 16.0 * 4 * 64 * 2, FROW = 64.0 * 16 * 2;
  const int nb = 256;  // one block per CU
  run<0, 0>("V      1 wave/SIMD", 256, nb, a, o, iters, FV, FV);
  run<0, 0>("V      2 waves/SIMD", 512, nb, a, o, iters, FV, FV);
  run<1, 1>("M4     1 wave/SIMD", 256, nb, a, o, iters, FM4, FM4);
  run<1, 1>("M4     2 waves/SIMD", 512, nb, a, o, iters, FM4, FM4);
  run<2, 2>("M16    1 wave/SIMD", 256, nb, a, o, iters, FM16, FM16);
  run<2, 2>("M16    2 waves/SIMD", 512, nb, a, o, iters, FM16, FM16);
  run<0, 1>("V|M4   one of each per SIMD", 512, nb, a, o, iters, FV, FM4);
  run<0, 2>("V|M16  one of each per SIMD", 512, nb, a, o, iters, FV, FM16);
  run<3, 3>("V+M4   one stream, 1 wave/SIMD", 256, nb, a, o, iters, FMIX, FMIX);
  run<3, 3>("V+M4   one stream, 2 waves/SIMD", 512, nb, a, o, iters, FMIX, FMIX);
  run<4, 4>("Vrow   EXEC=0xFFFF 1 wave/SIMD", 256, nb, a, o, iters, FROW, FROW);
  run<4, 4>("Vrow   EXEC=0xFFFF 2 waves/SIMD", 512, nb, a, o, iters, FROW, FROW);
  run<0, 4>("V|Vrow one of each per SIMD", 512, nb, a, o, iters, FV, FROW);
  return 0;
}
