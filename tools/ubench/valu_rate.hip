// Micro-benchmark: issue rate of the vector instructions the extractor kernels are made of, on gfx950.
// Each kernel runs ITER x 64 independent instructions of one kind per wave (8 accumulators), W waves per SIMD; the result is
// shader cycles (s_memtime) per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 256
#define REP8(x) x x x x x x x x

template <int OP>
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned long long* clk, unsigned seed) {
  unsigned a0 = threadIdx.x * 3 + seed, a1 = a0 * 5 + 1, a2 = a0 * 7 + 2, a3 = a0 * 11 + 3, a4 = a0 * 13 + 4, a5 = a0 * 17 + 5, a6 = a0 * 19 + 6, a7 = a0 * 23 + 7;
  unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5bd1e995u;
  __shared__ unsigned char lds[4096];
  if (OP == 100) { for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (unsigned char)(i * 7 + seed); __syncthreads(); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#define ONE(ACC)                                                                                                        \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                             \
    if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                             \
    if (OP == 2) asm volatile("v_min_u32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                             \
    if (OP == 3) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(ACC) : "v"(b));                                          \
    if (OP == 4) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(ACC) : "v"(b));                                          \
    if (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                                \
    if (OP == 6) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(ACC) : "v"(b));                                    \
    if (OP == 7) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(ACC) : "v"(b));                                      \
    if (OP == 8) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(ACC) : "v"(b));                                    \
    if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                          \
    if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                                \
    if (OP == 11) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(ACC));                                                     \
    if (OP == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                            \
    if (OP == 13) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(ACC));                                                   \
    if (OP == 14) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                            \
    if (OP == 15) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                                 \
    if (OP == 16) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(ACC), "v"(b) : "vcc");                                \
    if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ACC) : "v"(b) : "vcc");                           \
    if (OP == 18) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                             \
    if (OP == 19) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(ACC) : "v"(b));                                      \
    if (OP == 20) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                               \
    if (OP == 21) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ACC) : "v"(b), "v"(c));                             \
    if (OP == 22) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(ACC));                                                  \
    if (OP == 23) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                            \
    if (OP == 24) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ACC##ACC) : "v"(bb));                                   \
    if (OP == 25) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                       \
    if (OP == 26) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ACC) : "v"(b));                                            \
    if (OP == 27) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(ACC) : "s20");                                       \
    if (OP == 28) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(b));        \
    if (OP == 29) asm volatile("v_max_i16 %0, %0, %1" : "+v"(ACC) : "v"(b));                                            \
    if (OP == 30) asm volatile("v_sub_u16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(ACC) : "v"(b)); \
    if (OP == 100) asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(ACC) : "v"((ACC & 4095u)));
    unsigned long long bb = 0; (void)bb; unsigned long long a0a0 = 0, a1a1 = 0, a2a2 = 0, a3a3 = 0, a4a4 = 0, a5a5 = 0, a6a6 = 0, a7a7 = 0;
    (void)a0a0; (void)a1a1; (void)a2a2; (void)a3a3; (void)a4a4; (void)a5a5; (void)a6a6; (void)a7a7;
    REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int wavesPerSimd) {
  const int threads = 64 * 4 * wavesPerSimd > 512 ? 512 : 64 * 4 * wavesPerSimd;       // waves of one block spread over the 4 SIMDs
  const int blocksPerCU = (64 * 4 * wavesPerSimd) / threads;
  const int blocks = 256 * blocksPerCU;
  unsigned* out; unsigned long long* clk;
  hipMalloc(&out, sizeof(unsigned) * blocks * threads);
  hipMalloc(&clk, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, clk, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, clk, 2u);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
  // s_memtime ticks at 100 MHz on gfx9 (constant clock): convert with the wall time of the launch instead
  const double instrPerSimd = (double)ITER * 64 * wavesPerSimd;                        // wave-instructions one SIMD issued
  printf("%-22s waves/SIMD %d: %8.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz), memtime ticks/blk %.0f\n", name, wavesPerSimd, ms,
         ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4, avg);
  hipFree(out); hipFree(clk);
}

int main() {
#define R(OP, NAME) run<OP>(NAME, 1); run<OP>(NAME, 2); run<OP>(NAME, 8);
  R(0, "v_add_u32") R(1, "v_and_b32") R(2, "v_min_u32") R(3, "v_pk_min_u16") R(4, "v_pk_sub_i16") R(5, "v_perm_b32") R(6, "v_alignbyte_b32")
  R(7, "v_lshl_or_b32") R(8, "v_mbcnt_lo") R(9, "v_mul_lo_u32") R(10, "v_fma_f32") R(11, "v_cvt_f32_u32") R(12, "v_mad_u32_u24") R(13, "v_bfe_u32")
  R(14, "v_dot4_u32_u8") R(15, "v_sad_u8") R(16, "v_cmp_gt_u32") R(17, "v_cndmask_b32") R(18, "v_pk_mad_i16") R(19, "v_pk_mul_lo_u16")
  R(20, "v_max3_u32") R(21, "v_and_or_b32") R(22, "v_lshlrev_b32") R(23, "v_add_f32") R(25, "v_bcnt_u32_b32") R(26, "v_xor_b32")
  R(27, "v_readlane_b32") R(28, "v_mov_b32_dpp") R(29, "v_max_i16") R(30, "v_sub_u16_sdwa") R(100, "ds_read_u8+wait")
  return 0;
}
