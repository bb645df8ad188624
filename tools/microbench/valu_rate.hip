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
  return 0;
}
