// Micro-benchmark: cost of a workgroup barrier and of dependent LDS operations on gfx950, to size
// level-synchronous kernels (k_bfs).  hipcc --offload-arch=gfx950 -O3 lds_latency.hip -o lds_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(uint32_t* out, int iters, long long* cyc) {
  __shared__ uint32_t a[8192];
  __shared__ uint32_t cnt[4];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 8192; i += blockDim.x) a[i] = i * 2654435761u;
  if (tid < 4) cnt[tid] = 0;
  __syncthreads();
  uint32_t x = tid * 7 + 1;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // barrier only
    } else if (MODE == 1) {  // 1 dependent LDS read
      x = a[x & 8191];
    } else if (MODE == 2) {  // 4 dependent LDS reads
      x = a[x & 8191]; x = a[x & 8191]; x = a[x & 8191]; x = a[x & 8191];
    } else if (MODE == 3) {  // 1 returning atomic
      x = atomicOr(&a[x & 8191], 1u);
    } else if (MODE == 4) {  // 4 dependent returning atomics
      x = atomicOr(&a[x & 8191], 1u); x = atomicOr(&a[x & 8191], 2u); x = atomicOr(&a[x & 8191], 4u); x = atomicOr(&a[x & 8191], 8u);
    } else if (MODE == 5) {  // 5 independent atomics + 5 independent reads, then 5 dependent atomics
      uint32_t o[5], f[5];
      for (int j = 0; j < 5; ++j) o[j] = atomicOr(&a[(x + j * 977) & 8191], 1u);
      for (int j = 0; j < 5; ++j) f[j] = a[(x + j * 31 + 5) & 8191];
      uint32_t y = 0;
      for (int j = 0; j < 5; ++j) y += atomicOr(&a[(o[j] ^ f[j]) & 8191], 2u);
      x = y;
    } else if (MODE == 6) {  // one global store per lane (not waited)
      out[(x & 1023) + 1024 * (it & 63)] = x;
      x = x * 3 + 1;
    } else if (MODE == 7) {  // wave scan with shfl_up (6 steps) + shfl
      uint32_t incl = x & 1;
      for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)(tid & 63) >= off) incl += v; }
      x = __shfl(incl, 63) + x;
    } else if (MODE == 8) {  // counter atomic by one lane per wave + broadcast
      uint32_t b = 0;
      if ((tid & 63) == 63) b = atomicAdd(&cnt[it & 3], 1u);
      x += __shfl(b, 63);
    }
    __syncthreads();
  }
  long long t1 = clock64();
  if (tid == 0) *cyc = t1 - t0;
  out[tid] = x;
}

int main() {
  uint32_t* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  const char* names[] = {"barrier only", "1 dep LDS read", "4 dep LDS reads", "1 rtn atomic", "4 dep rtn atomics",
                         "5+5 indep then 5 dep atomics", "1 global store (unwaited)", "wave scan shfl_up x6", "wave atomicAdd+bcast"};
  for (int threads : {1024, 256, 64}) {
    for (int mode = 0; mode < 9; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        switch (mode) {
          case 0: k<0><<<1, threads>>>(out, iters, cyc); break; case 1: k<1><<<1, threads>>>(out, iters, cyc); break;
          case 2: k<2><<<1, threads>>>(out, iters, cyc); break; case 3: k<3><<<1, threads>>>(out, iters, cyc); break;
          case 4: k<4><<<1, threads>>>(out, iters, cyc); break; case 5: k<5><<<1, threads>>>(out, iters, cyc); break;
          case 6: k<6><<<1, threads>>>(out, iters, cyc); break; case 7: k<7><<<1, threads>>>(out, iters, cyc); break;
          case 8: k<8><<<1, threads>>>(out, iters, cyc); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("threads %4d  %-32s  %7.1f ns/iter  %7.1f clk/iter\n", threads, names[mode], ms * 1e6 / iters, (double)c / iters);
    }
  }
  return 0;
}
