// Hardware-layout probe for gfx950 (run on the GPU box):
//   1. checks the MFMA 32x32x16 bf16 A/B/C lane mappings assumed in flasht5_amd/csrc/attn_common.h
//   2. dumps the ds_read_b64_tr_b16 gather pattern
//   3. checks v_permlane32_swap semantics
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_layout.hip -o gpurun_out/probe_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void mfma_probe(const float* A, const float* B, float* C) {  // A: 32x16, B: 16x32 row-major fp32
  const int l = threadIdx.x, lq = l & 31, hi = l >> 5;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[lq * 16 + 8 * hi + j];       // A[row = lq][k = 8hi + j]
    b[j] = (__bf16)B[(8 * hi + j) * 32 + lq];     // B[k = 8hi + j][col = lq]
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    C[row * 32 + lq] = c[r];
  }
}

__global__ void tr_probe(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int off;  // element offset supplied by this lane
  if (mode == 0) off = l * 4;                                       // natural: lane i -> elements 4i..4i+3
  else off = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 256;    // lane = 4*row + chunk inside a 16-lane group, row stride 64
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

__global__ void swap_probe(int* out) {
  const int l = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(l, 1000 + l, false, false);
  out[2 * l] = r[0];
  out[2 * l + 1] = r[1];
}

int main() {
  // ---- 1. MFMA ----
  std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
  srand(1);
  for (auto& x : A) x = (float)((rand() % 17) - 8);
  for (auto& x : B) x = (float)((rand() % 13) - 6);
  for (int i = 0; i < 32; ++i)
    for (int n = 0; n < 32; ++n)
      for (int k = 0; k < 16; ++k) R[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(C[i] - R[i]));
  printf("[probe] mfma_32x32x16_bf16 assumed A/B/C layout: max err %.3f -> %s\n", err, err == 0 ? "OK" : "MISMATCH");

  // ---- 2. tr16_b64 ----
  short* dO; hipMalloc(&dO, 64 * 4 * 2);
  std::vector<short> O(256);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dO, mode);
    hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost);
    printf("[probe] ds_read_b64_tr_b16 mode %d (lane: e0 e1 e2 e3)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  %2d: %4d %4d %4d %4d%s", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3], (l % 4 == 3) ? "\n" : " |");
    }
    // guide formula check for mode 0: lane l elem j = lds[(l&15) + 16j + (l>>4)*64]
    if (mode == 0) {
      int bad = 0;
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += O[l * 4 + j] != (l & 15) + 16 * j + (l >> 4) * 64;
      printf("[probe] tr16 natural-address formula: %s\n", bad ? "MISMATCH" : "OK");
    } else {
      // expectation: lane l elem j = element at row j (address of lane 4j + (l&15)/4 in its 16-group) + (l&15)%4
      int bad = 0;
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const int src = 4 * j + ((l & 15) >> 2);  // lane within group supplying the address
        const int off = ((src & 15) >> 2) * 64 + (src & 3) * 4 + (l >> 4) * 256;
        bad += O[l * 4 + j] != off + ((l & 15) & 3);
      }
      printf("[probe] tr16 row-strided gather model: %s\n", bad ? "MISMATCH" : "OK");
    }
  }
  // ---- 3. permlane32_swap ----
  int* dS; hipMalloc(&dS, 128 * 4);
  std::vector<int> S(128);
  hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, dS);
  hipMemcpy(S.data(), dS, 512, hipMemcpyDeviceToHost);
  printf("[probe] permlane32_swap(vdst=l, src=1000+l): lane0 -> (%d,%d) lane31 -> (%d,%d) lane32 -> (%d,%d) lane63 -> (%d,%d)\n",
         S[0], S[1], S[62], S[63], S[64], S[65], S[126], S[127]);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("[probe] device %s CUs %d clock %d kHz mem clock %d kHz L2 %d smem/block %zu\n", prop.name, prop.multiProcessorCount,
         prop.clockRate, prop.memoryClockRate, prop.l2CacheSize, prop.sharedMemPerBlock);
  return 0;
}
