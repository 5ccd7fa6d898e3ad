// ubench.hip -- gfx950 issue-rate microbenchmarks behind DESIGN.md's roofline discussion (development aid, not product).
// Each test: W workgroups of 256 threads per CU (= W waves per SIMD), a loop whose body is 64 copies of one
// instruction pattern; reports SIMD cycles per wave-instruction assuming the measured shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench.bin && tools/ubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

typedef unsigned long long u64;

// 0: dependent fma_f64 chain            1: 8 independent fma_f64 (VGPR,VGPR,VGPR)
// 2: 8 independent fma_f64 with one SGPR operand
// 3: independent mul_f64, normal        4: mul_f64 subnormal input x normal -> subnormal
// 5: fma_f64 entirely subnormal domain  6: v_perm_b32 independent
// 7: 7 fma_f64 + 1 ds_add_u64 (conflict-free)   8: 7 fma_f64 + 1 ds_add_u32   9: 7 fma + 1 ds_read_b64
// 10: 8 fma only (same loop shape as 7-9)       11: v_fma_f32 independent      12: v_pk_fma_f32 independent
// 13: v_cndmask_b32                              14: v_rcp_f64                  15: v_cvt_f64_f32
template <int MODE>
__global__ __launch_bounds__(256) void k_ub(int iters, double seed, double sub, u64* out) {
  __shared__ u64 lds[256 * 16];
  const int tid = threadIdx.x;
  for (int k = tid; k < 256 * 16; k += 256) lds[k] = 0;
  __syncthreads();
  double a0 = seed + tid * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double m = 0.999999, c = 1e-7;
  float f0 = float(a0), f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fm = 0.99999f, fc = 1e-6f;
  unsigned p0 = tid, p1 = tid * 3, p2 = tid * 5, p3 = tid * 7, sel = 0x07060100u;
  unsigned laddr = (unsigned)(size_t)(&lds[0]) + (tid & 15) * 8 + (tid >> 4) * 128;  // lane-private cells
  double s0 = sub * (1.0 + tid * 1e-3), s1 = s0 * 1.1, s2 = s0 * 1.2, s3 = s0 * 1.3;  // subnormal values when sub is
  double one_eps = 0.9999;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
    } else if (MODE == 1) {
      asm volatile(REP4(REP4("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
    } else if (MODE == 2) {
      asm volatile(REP4(REP4("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m), "v"(c));
    } else if (MODE == 3) {
      asm volatile(REP4(REP4("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(one_eps));
    } else if (MODE == 4) {
      asm volatile(REP4(REP4("v_mul_f64 %0, %4, %8\n v_mul_f64 %1, %5, %8\n v_mul_f64 %2, %6, %8\n v_mul_f64 %3, %7, %8\n"))
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(one_eps));
    } else if (MODE == 5) {
      asm volatile(REP4(REP4("v_fma_f64 %0, %4, %8, %5\n v_fma_f64 %1, %5, %8, %6\n v_fma_f64 %2, %6, %8, %7\n v_fma_f64 %3, %7, %8, %4\n"))
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(one_eps));
    } else if (MODE == 6) {
      asm volatile(REP4(REP4("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sel), "v"(sel));
    } else if (MODE == 7 || MODE == 8 || MODE == 9 || MODE == 10) {
#define SEVEN "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n"
      if (MODE == 7) asm volatile(REP4(REP4(SEVEN "ds_add_u64 %6, %7\n")) "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c), "v"(laddr), "v"(a4) : "memory");
      if (MODE == 8) asm volatile(REP4(REP4(SEVEN "ds_add_u32 %6, %7\n")) "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c), "v"(laddr), "v"(p0) : "memory");
      if (MODE == 9) asm volatile(REP4(REP4(SEVEN "ds_read_b64 %8, %6\n")) "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c), "v"(laddr), "v"(a4), "v"(a5) : "memory");
      if (MODE == 10) asm volatile(REP4(REP4(SEVEN "v_fma_f64 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
    } else if (MODE == 11) {
      asm volatile(REP4(REP4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"))
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fm), "v"(fc));
    } else if (MODE == 12) {
      asm volatile(REP4(REP4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
    } else if (MODE == 13) {
      asm volatile(REP4(REP4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sel) : "vcc");
    } else if (MODE == 14) {
      asm volatile(REP4(REP4("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (MODE == 15) {
      asm volatile(REP4(REP4("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"))
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
    }
  }
  u64 r = __double_as_longlong(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) ^ p0 ^ p1 ^ p2 ^ p3 ^ __float_as_uint(f0 + f1 + f2 + f3) ^ lds[tid];
  if (r == 0x123456789abcdefull) out[0] = r;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, double ghz, u64* d_out, double sub = 1.0) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_ub<MODE>, dim3(cus * waves_per_simd), dim3(256), 0, 0, 10, 1.5, sub, d_out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_ub<MODE>, dim3(cus * waves_per_simd), dim3(256), 0, 0, iters, 1.5, sub, d_out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts = double(iters) * 64.0 * waves_per_simd;  // wave-instructions per SIMD
  std::printf("%-44s waves/SIMD=%d  %8.3f ms  %6.2f clk per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e-3 * ghz * 1e9 / insts);
}

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  u64* d_out = nullptr;
  (void)hipMalloc(&d_out, 64);
  const double sub = 4.9e-324 * 1e11;  // ~2^-1037: a subnormal of the magnitude the fixed-point weights have
  for (int w : {1, 2, 4}) {
    run<0>("fma_f64 dependent chain", w, ghz, d_out);
    run<1>("fma_f64 4 independent (VVV)", w, ghz, d_out);
    run<2>("fma_f64 4 independent (one SGPR operand)", w, ghz, d_out);
    run<3>("mul_f64 normal", w, ghz, d_out);
    run<4>("mul_f64 subnormal x normal -> subnormal", w, ghz, d_out, sub);
    run<5>("fma_f64 subnormal domain", w, ghz, d_out, sub);
    run<6>("v_perm_b32", w, ghz, d_out);
    run<10>("8 fma_f64", w, ghz, d_out);
    run<7>("7 fma_f64 + 1 ds_add_u64", w, ghz, d_out);
    run<8>("7 fma_f64 + 1 ds_add_u32", w, ghz, d_out);
    run<9>("7 fma_f64 + 1 ds_read_b64", w, ghz, d_out);
    run<11>("fma_f32", w, ghz, d_out);
    run<12>("pk_fma_f32", w, ghz, d_out);
    run<13>("v_cndmask_b32", w, ghz, d_out);
    run<14>("v_rcp_f64", w, ghz, d_out);
    run<15>("v_cvt_f64_f32", w, ghz, d_out);
  }
  return 0;
}
