// fp32 atomic-add throughput in the access pattern a single-body (5-GEMM) backward would produce for dQ:
// workgroup = (b, h, 256-key block) walks all 32-row query steps; per step each of its 4 waves adds a 32 x 64 fp32 tile
// (or, REDUCED: one tile per workgroup) into the (b, h) slice of a (B*H, S, 64) fp32 accumulator.  All key blocks of one (b, h)
// sit on one XCD (bid % 8), like the real kernels.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/mb_atomic.hip -o tools/bin/mb_atomic && tools/bin/mb_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>  // 0: every wave adds its tile (4 x traffic); 1: one tile per workgroup; 2: plain stores (no atomics) for reference
__global__ __launch_bounds__(256) void k(float* acc, int S, int nblk, int nbh) {
  const int bid = blockIdx.x;
  // (b,h) -> XCD: units of one XCD are bid % 8; consecutive blocks of a unit are bid / 8 ...
  const int xcd = bid & 7, idx = bid >> 3;
  const int per_x = nbh / 8;
  const int u = xcd * per_x + idx / nblk, blk = idx % nblk;
  (void)blk;
  float* base = acc + (size_t)u * S * 64;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int steps = S / 32;
  float v = 1.0f + l * 1e-3f;
  for (int st = 0; st < steps; ++st) {
    float* tile = base + (size_t)st * 32 * 64;
    if (MODE == 0) {
#pragma unroll 8
      for (int r = 0; r < 32; ++r) __hip_atomic_fetch_add(tile + r * 64 + l, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 1) {
#pragma unroll 8
      for (int r = 0; r < 8; ++r) __hip_atomic_fetch_add(tile + (8 * w + r) * 64 + l, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll 8
      for (int r = 0; r < 8; ++r) tile[(8 * w + r) * 64 + l] = v;
    }
    v += 1e-6f;
  }
}

int main() {
  const int B = 4, H = 12, S = 8192, nbh = B * H, nblk = S / 256;
  float* acc;
  const size_t bytes = (size_t)nbh * S * 64 * 4;
  hipMalloc(&acc, bytes);
  hipMemset(acc, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, double traffic) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nbh * nblk), dim3(256), 0, 0, acc, S, nblk, nbh);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nbh * nblk), dim3(256), 0, 0, acc, S, nblk, nbh);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-44s %8.3f ms  %7.1f GB/s of added operands\n", name, ms, traffic / ms / 1e6);
  };
  const double per_wg_tile = (double)nbh * nblk * (S / 32) * 32 * 64 * 4;
  run(k<0>, "atomic add, one tile per wave (12.9 GB)", 4 * per_wg_tile);
  run(k<1>, "atomic add, one tile per workgroup (3.2 GB)", per_wg_tile);
  run(k<2>, "plain stores, one tile per workgroup", per_wg_tile);
  return 0;
}
