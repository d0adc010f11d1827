// Does the f32-input MFMA share issue/execute cycles with ordinary vector ALU work on gfx950?
// Each wave runs ITER x (4 independent v_mfma_f32_32x32x2_f32 + NV independent v_fma_f32); 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o gpurun_out/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i % 16]) : "v"(b), "v"(a));
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV>
void run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NV>, dim3(256), dim3(512), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NV>, dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * iters * 4 * 4096.0;
  printf("NV=%2d per 4 MFMA: %.3f ms  %.1f TFLOP/s (MFMA only)\n", NV, ms, flops / ms / 1e9);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int iters = 20000;
  run<0>(d, iters); run<4>(d, iters); run<8>(d, iters); run<16>(d, iters); run<32>(d, iters); run<64>(d, iters);
  return 0;
}
