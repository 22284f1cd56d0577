// Latency of dependent FP64 operations on one wave (gfx950): s_memtime ticks per operation of a serial chain.
// Measured on MI355X: v_add/fma/mul_f64 8.3; v_fmac_f64_dpp row_newbcast (after s_nop 1) 16.3; a butterfly stage (two
// v_mov_b32_dpp + v_add_f64) 24.3; four INDEPENDENT v_fmac_f64_dpp 28.3 (7.1 each).  The closed-loop rollout's recurrence
// (dx, product, four butterfly stages, two adds, select, four chained DPP FMAs, two adds) is ~230 ticks per knot.
//   hipcc --offload-arch=gfx950 -O3 -o dep_latency tools/probes/dep_latency.hip && ./dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define LOOPS 256
template <int KIND>
__global__ void chain(double* out, long long* cyc, double a, double b) {
  double x = a + threadIdx.x, y = b;
  double y1 = b + 1, y2 = b + 2, y3 = b + 3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int l = 0; l < LOOPS; ++l) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if constexpr (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
      if constexpr (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
      if constexpr (KIND == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
      if constexpr (KIND == 3) {  // 32-bit DPP moves of both halves, then an add: one butterfly stage
        const int l2 = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, false);
        const int h2 = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, false);
        double t = __hiloint2double(h2, l2);
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(t));
      }
      if constexpr (KIND == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
      if constexpr (KIND == 7) {  // four independent plain FMA chains (issue rate without DPP)
        asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5"
                     : "+v"(y), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x), "v"(a));
      }
      if constexpr (KIND == 8) {  // eight DPP FMAs over four accumulators, no s_nop between groups (inside one block)
        asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf"
                     : "+v"(y), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x), "v"(a));
      }
      if constexpr (KIND == 9) {  // DPP FMA and plain FMA alternating (does a plain FMA fill the DPP FMA's issue gap?)
        asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %1, %4, %5\n\t"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %3, %4, %5"
                     : "+v"(y), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x), "v"(a));
      }
      if constexpr (KIND == 10) {  // 32-bit VALU between DPP FMAs (v_cndmask / v_mov: does 32-bit work fill the gap?)
        int t0_, t1_;
        asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32 %6, %7\n\t"
                     "v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32 %7, %6\n\t"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32 %6, %7\n\t"
                     "v_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_mov_b32 %7, %6"
                     : "+v"(y), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x), "v"(a), "v"(t0_), "v"(t1_));
      }
      if constexpr (KIND == 11) {  // v_rcp_f64 dependent chain
        asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
      }
      if constexpr (KIND == 12) {  // permlane32_swap + permlane16_swap dependent pair on one register pair
        int lo = __double2loint(x), hi = __double2hiint(x);
        asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        x = __hiloint2double(hi, lo);
      }
      if constexpr (KIND == 13) {  // v_mov_b64_dpp row_newbcast dependent chain
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x));
      }
      if constexpr (KIND == 14) {  // ds_bpermute round trip (dependent)
        int lo = __builtin_amdgcn_ds_bpermute(threadIdx.x << 2, __double2loint(x));
        x = __hiloint2double(__double2hiint(x), lo);
      }
      if constexpr (KIND == 6) {  // four independent DPP FMA chains (issue rate)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                     : "+v"(y), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x), "v"(a));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x + y + y1 + y2 + y3;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
void run(const char* name, int per) {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, sizeof(long long));
  chain<KIND><<<1, 64>>>(out, cyc, 1.0, 1e-9);
  chain<KIND><<<1, 64>>>(out, cyc, 1.0, 1e-9);
  long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-46s %.1f s_memtime ticks per step (%d dependent instruction(s) per step)\n", name, (double)c / (REP * LOOPS), per);
}
int main() {
  run<0>("v_add_f64 chain", 1);
  run<1>("v_fma_f64 chain", 1);
  run<5>("v_mul_f64 chain", 1);
  run<2>("v_fmac_f64_dpp row_newbcast chain (+ s_nop 1)", 1);
  run<3>("2 x v_mov_b32_dpp + v_add_f64 (butterfly stage)", 2);
  run<6>("4 independent v_fmac_f64_dpp (per group of 4)", 4);
  run<7>("4 independent plain v_fmac_f64 (per group of 4)", 4);
  run<8>("8 v_fmac_f64_dpp over 4 accumulators (per group of 8)", 8);
  run<9>("2 DPP + 2 plain FMAs alternating (per group of 4)", 4);
  run<10>("4 DPP FMAs + 4 v_mov_b32 alternating (per group)", 8);
  run<11>("v_rcp_f64 chain", 1);
  run<12>("permlane32_swap + permlane16_swap chain (per pair)", 2);
  run<13>("v_mov_b64_dpp row_newbcast chain (+ s_nop 1)", 1);
  run<14>("ds_bpermute_b32 chain", 1);
  return 0;
}
