// ubench2.hip -- per-instruction issue cost of everything that appears in the spline kernels' inner loops (gfx950).
// 4 waves per SIMD, 64 independent-ish copies per loop iteration; prints clk per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define KERNEL(NAME, ASM4)                                                                                                  \
  __global__ __launch_bounds__(256) void NAME(int iters, double seed, u64* out) {                                           \
    __shared__ u64 lds[256 * 16];                                                                                           \
    const int tid = threadIdx.x;                                                                                            \
    for (int k = tid; k < 256 * 16; k += 256) lds[k] = 0;                                                                   \
    __syncthreads();                                                                                                        \
    double a0 = seed + tid * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, m = 0.999999, c = 1e-7;                          \
    float f0 = float(a0), f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                                                           \
    unsigned p0 = tid, p1 = tid * 3, p2 = tid * 5, p3 = tid * 7, q = 0x07060100u;                                           \
    unsigned laddr = (unsigned)(size_t)(&lds[0]) + (tid & 15) * 8 + (tid >> 4) * 128;                                       \
    u64 sm = 0x5555555555555555ull;                                                                                         \
    for (int it = 0; it < iters; it++) {                                                                                    \
      asm volatile(REP16(ASM4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)  \
                   : "v"(m), "v"(c), "v"(q), "v"(laddr), "s"(sm) : "vcc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                    \
    }                                                                                                                       \
    u64 r = __double_as_longlong(a0 + a1 + a2 + a3) ^ p0 ^ p1 ^ p2 ^ p3 ^ __float_as_uint(f0 + f1 + f2 + f3) ^ lds[tid];    \
    if (r == 0x123456789abcdefull) out[0] = r;                                                                              \
  }
// operands: %0-3 doubles, %4-7 floats, %8-11 uints, %12 m, %13 c, %14 q, %15 laddr, %16 sgpr-pair mask
KERNEL(k_fma64, "v_fma_f64 %0, %0, %12, %13\n v_fma_f64 %1, %1, %12, %13\n v_fma_f64 %2, %2, %12, %13\n v_fma_f64 %3, %3, %12, %13\n")
KERNEL(k_add64, "v_add_f64 %0, %0, %13\n v_add_f64 %1, %1, %13\n v_add_f64 %2, %2, %13\n v_add_f64 %3, %3, %13\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %8, %8, %14, vcc\n v_cndmask_b32 %9, %9, %14, vcc\n v_cndmask_b32 %10, %10, %14, vcc\n v_cndmask_b32 %11, %11, %14, vcc\n")
KERNEL(k_cnd_sgpr, "v_cndmask_b32_e64 %8, %8, %14, %16\n v_cndmask_b32_e64 %9, %9, %14, %16\n v_cndmask_b32_e64 %10, %10, %14, %16\n v_cndmask_b32_e64 %11, %11, %14, %16\n")
KERNEL(k_cmp64_vcc, "v_cmp_lt_f64 vcc, %0, %12\n v_cmp_lt_f64 vcc, %1, %12\n v_cmp_lt_f64 vcc, %2, %12\n v_cmp_lt_f64 vcc, %3, %12\n")
KERNEL(k_cmp64_sgpr, "v_cmp_lt_f64 s[20:21], %0, %12\n v_cmp_lt_f64 s[22:23], %1, %12\n v_cmp_lt_f64 s[24:25], %2, %12\n v_cmp_lt_f64 s[26:27], %3, %12\n")
KERNEL(k_cmp_then_cnd, "v_cmp_lt_f64 vcc, %0, %12\n v_cndmask_b32 %8, %8, %14, vcc\n v_cmp_lt_f64 vcc, %1, %12\n v_cndmask_b32 %9, %9, %14, vcc\n")
KERNEL(k_mul_lo, "v_mul_lo_u32 %8, %8, %14\n v_mul_lo_u32 %9, %9, %14\n v_mul_lo_u32 %10, %10, %14\n v_mul_lo_u32 %11, %11, %14\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %8, %8, %14\n v_mul_u32_u24 %9, %9, %14\n v_mul_u32_u24 %10, %10, %14\n v_mul_u32_u24 %11, %11, %14\n")
KERNEL(k_lshl_add64, "v_lshl_add_u64 %0, %0, 4, %1\n v_lshl_add_u64 %1, %1, 4, %2\n v_lshl_add_u64 %2, %2, 4, %3\n v_lshl_add_u64 %3, %3, 4, %0\n")
KERNEL(k_lshl_add32, "v_lshl_add_u32 %8, %8, 2, %14\n v_lshl_add_u32 %9, %9, 2, %14\n v_lshl_add_u32 %10, %10, 2, %14\n v_lshl_add_u32 %11, %11, 2, %14\n")
KERNEL(k_alignbyte, "v_alignbyte_b32 %8, %8, %14, %9\n v_alignbyte_b32 %9, %9, %14, %10\n v_alignbyte_b32 %10, %10, %14, %11\n v_alignbyte_b32 %11, %11, %14, %8\n")
KERNEL(k_cvt_i32_f64, "v_cvt_i32_f64 %8, %0\n v_cvt_i32_f64 %9, %1\n v_cvt_i32_f64 %10, %2\n v_cvt_i32_f64 %11, %3\n")
KERNEL(k_fract64, "v_fract_f64 %0, %0\n v_fract_f64 %1, %1\n v_fract_f64 %2, %2\n v_fract_f64 %3, %3\n")
KERNEL(k_floor64, "v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3\n")
KERNEL(k_sdwa, "v_lshlrev_b32_sdwa %8, %14, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %9, %14, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %10, %14, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %11, %14, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n")
KERNEL(k_mov64, "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n")
KERNEL(k_ds_add64, "ds_add_u64 %15, %0\n ds_add_u64 %15, %1\n ds_add_u64 %15, %2\n ds_add_u64 %15, %3\n")
KERNEL(k_ds_read64, "ds_read_b64 %0, %15\n ds_read_b64 %1, %15\n ds_read_b64 %2, %15\n ds_read_b64 %3, %15\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_fma64_ds1, "v_fma_f64 %0, %0, %12, %13\n v_fma_f64 %1, %1, %12, %13\n v_fma_f64 %2, %2, %12, %13\n ds_add_u64 %15, %3\n")
KERNEL(k_perm_mul_ds, "v_perm_b32 %8, %9, %14, %14\n v_mul_f64 %0, %1, %12\n ds_add_u64 %8, %0\n v_fma_f64 %2, %2, %12, %13\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 5\n v_readlane_b32 s22, %10, 7\n v_readlane_b32 s23, %11, 9\n")
KERNEL(k_fma32, "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %4\n v_fma_f32 %7, %7, %4, %5\n")
KERNEL(k_mul32, "v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %4\n")
KERNEL(k_add_u32, "v_add_u32 %8, %8, %14\n v_add_u32 %9, %9, %14\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %14\n")
KERNEL(k_and_b32, "v_and_b32 %8, %8, %14\n v_and_b32 %9, %9, %14\n v_and_b32 %10, %10, %14\n v_and_b32 %11, %11, %14\n")

template <typename K>
static void run(const char* name, K kern, u64* d_out) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 3000, w = 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), 0, 0, 10, 1.5, d_out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), 0, 0, iters, 1.5, d_out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-22s %8.3f ms  %6.2f clk/wave-instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (double(iters) * 64.0 * w));
}
int main() {
  u64* d_out = nullptr;
  (void)hipMalloc(&d_out, 64);
#define RUN(k) run(#k, k, d_out)
  RUN(k_fma64); RUN(k_add64); RUN(k_cnd_vcc); RUN(k_cnd_sgpr); RUN(k_cmp64_vcc); RUN(k_cmp64_sgpr); RUN(k_cmp_then_cnd); RUN(k_mul_lo); RUN(k_mul_u24);
  RUN(k_lshl_add64); RUN(k_lshl_add32); RUN(k_alignbyte); RUN(k_cvt_i32_f64); RUN(k_fract64); RUN(k_floor64); RUN(k_sdwa); RUN(k_mov64); RUN(k_ds_add64);
  RUN(k_ds_read64); RUN(k_fma64_ds1); RUN(k_perm_mul_ds); RUN(k_readlane); RUN(k_fma32); RUN(k_mul32); RUN(k_add_u32); RUN(k_and_b32);
  return 0;
}
