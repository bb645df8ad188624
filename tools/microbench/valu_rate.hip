// Micro-benchmark: VALU issue rate per SIMD on gfx950 as a function of waves per SIMD and instruction kind,
// to price "VALU-bound" kernels (k_score, k_bfs_wave) against a measured instruction roofline.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// One workgroup per CU, W waves per SIMD (threads = 256 * W); every wave runs N independent instructions of one kind
// on 8 private registers; cycles per instruction per SIMD = elapsed / (N * W).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void k(float* out, int iters, long long* cyc) {
  float f[8];
  double d[8];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) {
    f[i] = threadIdx.x * 0.001f + i;
    d[i] = threadIdx.x * 0.001 + i;
    u[i] = threadIdx.x * 2654435761u + i;
  }
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 1) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 2) {
#define X(i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 3) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 4) {
#define X(i) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 5) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 6) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 7) {
#define X(i) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 8) {
#define X(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 9) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 10) {
#define X(i) asm volatile("v_min_f64 %0, %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 11) {
#define X(i) asm volatile("v_cmp_gt_f64 vcc, %0, %0" : : "v"(d[i]) : "vcc");
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 12) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[i]) : : "vcc");
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 13) {  // scalar ALU: how many per clock does a CU's scalar unit retire with W waves per SIMD asking?
      uint32_t s0 = it, s1 = it + 1, s2 = it + 2, s3 = it + 3;
#define X(i) asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
      REP8(X)
#undef X
      u[0] += s0 + s1 + s2 + s3;
    } else if (KIND == 14) {  // exec-mask bookkeeping as the compiler emits it around a divergent if
#define X(i) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]" : : : "s20", "s21", "s22", "s23", "scc");
      REP8(X)
#undef X
    } else if (KIND == 15) {  // VALU and SALU interleaved 1:1 in one wave: do they overlap across waves?
      uint32_t s0 = it, s1 = it + 1;
#define X(i) asm volatile("v_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %0\n s_add_u32 %2, %2, %1\n v_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %0\n s_add_u32 %2, %2, %1" : "+v"(u[i]), "+s"(s0), "+s"(s1) : : "scc");
      REP8(X)
#undef X
      u[0] += s0 + s1;
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  printf("%-14s", name);
  for (int w : {1, 2, 4, 8}) {
    // 1024 threads per workgroup at most: 8 waves per SIMD = two 1024-thread workgroups per CU
    hipLaunchKernelGGL(k<KIND>, dim3(w > 4 ? 512 : 256), dim3(256 * (w > 4 ? 4 : w)), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)h[i];
    mean /= 256;
    // clock64 = s_memtime ticks (100 MHz constant clock on some parts, shader clock on others): report both raw per instr
    printf("  w=%d: %7.3f ticks/instr/SIMD", w, mean / ((double)iters * 32 * w));
  }
  printf("\n");
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 2048 * sizeof(float));
  hipMalloc(&cyc, 512 * sizeof(long long));
  setvbuf(stdout, nullptr, _IOLBF, 0);
  // wall-clock calibration of the tick: time a long fma run with events
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<0>, dim3(512), dim3(1024), 0, 0, out, 20000, cyc);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<0>, dim3(512), dim3(1024), 0, 0, out, 20000, cyc);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  long long h0;
  hipMemcpy(&h0, cyc, 8, hipMemcpyDeviceToHost);
  printf("calibration: 20000 x 32 v_fma_f32 x 8 waves/SIMD: %.3f ms, %lld ticks -> %.1f MHz tick; %.2f ns per wave-instruction per SIMD\n", ms, h0,
         h0 / (ms * 1e3), ms * 1e6 / (20000.0 * 32 * 8));
  run<0>("v_fma_f32", out, cyc);
  run<1>("v_fma_f64", out, cyc);
  run<3>("v_mul_f64", out, cyc);
  run<4>("v_add_f64", out, cyc);
  run<2>("v_add_u32", out, cyc);
  run<5>("v_mul_lo_u32", out, cyc);
  run<6>("v_cvt_f32_f64", out, cyc);
  run<7>("v_floor_f64", out, cyc);
  run<8>("v_cvt_i32_f64", out, cyc);
  run<9>("v_cvt_f64_f32", out, cyc);
  run<10>("v_min_f64", out, cyc);
  run<11>("v_cmp_gt_f64", out, cyc);
  run<12>("v_cndmask_b32", out, cyc);
  run<13>("s_add_u32", out, cyc);
  run<14>("saveexec pair", out, cyc);
  run<15>("v_add+s_add 1:1 (per pair: x2)", out, cyc);
  return 0;
}
