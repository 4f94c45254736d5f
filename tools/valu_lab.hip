// valu_lab.hip -- development harness: issue cost of VALU instruction classes on gfx950
// (cycles per wave-instruction per SIMD at 8 waves/SIMD, 8 independent chains per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP> __global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float f[8]; double d[8]; unsigned u[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { f[i] = a + i + threadIdx.x; d[i] = a + i + threadIdx.x; u[i] = threadIdx.x * 7 + i; p[i] = v2f{a + i, b + i + threadIdx.x}; }
  v2f pa = {a, b}, pb = {b, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) f[i] = __builtin_fmaf(f[i], a, b);
      if (OP == 1) d[i] = __builtin_fma(d[i], (double)a, (double)b);
      if (OP == 2) u[i] = (u[i] << 3) ^ u[i];            // 2 int ops (lshl + xor) or 1 (v_lshl_xor?)
      if (OP == 3) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
      if (OP == 4) f[i] = __builtin_ceilf(f[i] * a);      // mul + ceil
      if (OP == 5) d[i] = d[i] + (double)a;               // f64 add
      if (OP == 6) u[i] = __builtin_amdgcn_alignbit(u[i], u[(i + 1) & 7], 5);
      if (OP == 7) f[i] = __builtin_fmaxf(__builtin_fmaxf(f[i], a), b);  // max3
      if (OP == 8) u[i] = (unsigned)(f[i] = f[i] + a);    // add + cvt_u32
      if (OP == 9) d[i] = __builtin_floor(d[i] * (double)a);  // f64 mul + floor
      if (OP == 10) asm volatile("v_min_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) & 7]));
      if (OP == 11) { unsigned long long x = __double_as_longlong(d[i]), y = __double_as_longlong(d[(i + 1) & 7]); d[i] = __longlong_as_double(x < y ? x : y); }
      if (OP == 12) d[i] = d[i] < d[(i + 1) & 7] ? d[i] : d[(i + 1) & 7];
      if (OP == 13) asm volatile("v_max_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) & 7]));
      if (OP == 14) asm volatile("v_cvt_u32_f32_e32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
      if (OP == 15) asm volatile("v_ceil_f32_e32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
      if (OP == 16) asm volatile("v_bfm_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
      if (OP == 17) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
      if (OP == 18) asm volatile("v_sub_u32 %0, %1, %2 clamp" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
      if (OP == 19) asm volatile("v_rcp_f32_e32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
      if (OP == 20) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      if (OP == 21) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      if (OP == 22) asm volatile("v_rndne_f64_e32 %0, %1" : "=v"(d[i]) : "v"(d[i]));
      if (OP == 23) asm volatile("v_cvt_i32_f64_e32 %0, %1" : "=v"(u[i]) : "v"(d[i]));
      if (OP == 24) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
      if (OP == 25) { unsigned long long x = __double_as_longlong(d[i]); asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(x) : "v"(u[i]), "v"(x)); d[i] = __longlong_as_double(x); }
      if (OP == 26) asm volatile("v_bfe_u32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
      if (OP == 27) asm volatile("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
      if (OP == 28) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc");
      if (OP == 29) asm volatile("v_max_u32_e32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
      if (OP == 30) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
      if (OP == 31) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(p[(i + 1) & 7]));
      if (OP == 32) asm volatile("v_floor_f64_e32 %0, %1" : "=v"(d[i]) : "v"(d[i]));
      if (OP == 33) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) & 7]));
      if (OP == 34) asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
      if (OP == 35) asm volatile("v_ffbl_b32_e32 %0, %1" : "=v"(u[i]) : "v"(u[i]));
      if (OP == 36) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
      if (OP == 37) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
      if (OP == 38) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i] + (float)u[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; CK(hipMalloc(&out, 256 * 8192 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096, blocks = 256 * 8;  // 8 blocks/CU = 8 waves/SIMD
  auto run = [&](auto kern, const char* name, int ops_per_iter) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // waves per SIMD = blocks*4/1024 ; wave-instrs per SIMD = that * iters * ops
    double winstr = (double)blocks * 4 / 1024 * iters * ops_per_iter;
    printf("%-26s %8.3f ms  %6.2f ns per wave-instr per SIMD (x2.4 GHz = %5.2f cyc)\n", name, ms, ms * 1e6 / winstr, ms * 1e6 / winstr * 2.4);
  };
  run(k<0>, "v_fma_f32", 8);
  run(k<1>, "v_fma_f64", 8);
  run(k<2>, "lshl+xor (int)", 16);
  run(k<3>, "v_pk_fma_f32", 8);
  run(k<4>, "mul_f32 + ceil_f32", 16);
  run(k<5>, "v_add_f64", 8);
  run(k<6>, "v_alignbit_b32", 8);
  run(k<7>, "v_max3_f32", 8);
  run(k<8>, "add_f32 + cvt_u32_f32", 16);
  run(k<9>, "mul_f64 + floor_f64", 16);
  run(k<10>, "v_min_f64", 8);
  run(k<11>, "u64 cmp + 2 cndmask", 8);
  run(k<12>, "f64 cmp + 2 cndmask", 8);
  run(k<13>, "v_max_f64", 8);
  run(k<14>, "v_cvt_u32_f32", 8);
  run(k<15>, "v_ceil_f32", 8);
  run(k<16>, "v_bfm_b32", 8);
  run(k<17>, "v_and_or_b32", 8);
  run(k<18>, "v_sub_u32 clamp", 8);
  run(k<19>, "v_rcp_f32", 8);
  run(k<20>, "v_cvt_f32_f64", 8);
  run(k<21>, "v_cvt_f64_f32", 8);
  run(k<22>, "v_rndne_f64", 8);
  run(k<23>, "v_cvt_i32_f64", 8);
  run(k<24>, "v_min3_f32", 8);
  run(k<25>, "v_lshlrev_b64", 8);
  run(k<26>, "v_bfe_u32", 8);
  run(k<27>, "v_lshl_or_b32", 8);
  run(k<28>, "v_cmp_f32 + v_cndmask", 16);
  run(k<29>, "v_max_u32", 8);
  run(k<30>, "v_add_f32", 8);
  run(k<31>, "v_pk_add_f32", 8);
  run(k<32>, "v_floor_f64", 8);
  run(k<33>, "v_mul_f64", 8);
  run(k<34>, "v_cvt_f32_u32", 8);
  run(k<35>, "v_ffbl_b32", 8);
  run(k<36>, "v_mov_b32", 8);
  run(k<37>, "v_mul_lo_u32", 8);
  run(k<38>, "v_mad_u32_u24", 8);
  return 0;
}
