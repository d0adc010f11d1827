// Probe: fp32 product on the matrix cores as six bf16 MFMAs of a three-way split (hi + mid + lo, 8 mantissa bits each,
// exact by truncation) against the native fp32 MFMA: accuracy against a double-precision host product and rate.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_split_probe.hip -o ab/mfma_split_probe && ab/mfma_split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Split3 { bf16x8 hi, mid, lo; };
// eight consecutive floats -> three bf16x8 planes
__device__ __forceinline__ Split3 split8(const float* x) {
  u32x4 h, m, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = x[2 * p], a1 = x[2 * p + 1];
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    h[p] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m[p] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  Split3 s;
  s.hi = __builtin_bit_cast(bf16x8, h); s.mid = __builtin_bit_cast(bf16x8, m); s.lo = __builtin_bit_cast(bf16x8, l);
  return s;
}
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// C[32x32] = A[32xK] B[KxN=32]^T-style: A row-major [32][K], Bt row-major [32][K] (both reduction-contiguous)
__global__ void k_split(const float* A, const float* Bt, float* C, int K, int terms) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const Split3 a = split8(A + r * K + k0 + 8 * h), b = split8(Bt + r * K + k0 + 8 * h);
    if (terms >= 6) { acc = MF(a.lo, b.hi, acc); acc = MF(a.hi, b.lo, acc); acc = MF(a.mid, b.mid, acc); }
    if (terms >= 3) { acc = MF(a.mid, b.hi, acc); acc = MF(a.hi, b.mid, acc); }
    acc = MF(a.hi, b.hi, acc);
  }
  for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + r] = acc[v];
}
__global__ void k_f32(const float* A, const float* Bt, float* C, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], Bt[r * K + k + h], acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + r] = acc[v];
}
// rate: every wave splits one A operand per step and multiplies it with NT column tiles whose B planes sit in registers
template <int NT, int MODE>      // MODE 0: native fp32 MFMA, 1: 6 bf16 products with the split done in the loop
__global__ __launch_bounds__(256) void k_rate(const float* A, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  float x[8];
  for (int q = 0; q < 8; ++q) x[q] = A[lane * 8 + q];
  Split3 b = split8(x);
  for (int it = 0; it < iters; ++it) {
    for (int q = 0; q < 8; ++q) x[q] += 1.0f;            // fresh operand words every step
    if (MODE == 0) {
      for (int q = 0; q < 8; ++q) for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[q], x[(q + j) & 7], acc[j], 0, 0, 0);
    } else {
      const Split3 a = split8(x);
      for (int j = 0; j < NT; ++j) {
        acc[j] = MF(a.lo, b.hi, acc[j]); acc[j] = MF(a.hi, b.lo, acc[j]); acc[j] = MF(a.mid, b.mid, acc[j]);
        acc[j] = MF(a.mid, b.hi, acc[j]); acc[j] = MF(a.hi, b.mid, acc[j]); acc[j] = MF(a.hi, b.hi, acc[j]);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < NT; ++j) for (int v = 0; v < 16; ++v) s += acc[j][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int K = 512;
  std::vector<float> A(32 * K), B(32 * K);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  std::vector<double> ref(1024, 0.0);
  double scale = 0.0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k];
    ref[i * 32 + j] = s; scale = fmax(scale, fabs(s));
  }
  float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> C(1024);
  auto err = [&](const char* name) {
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - ref[i]));
    printf("%-28s max abs err %.3e  relative to max|C| %.3e\n", name, e, e / scale);
  };
  hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); err("native fp32 MFMA");
  hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 6); err("bf16 x 6 products");
  hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 3); err("bf16 x 3 products");
  hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1); err("bf16 x 1 product");
  float* dO; hipMalloc(&dO, 1024 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 1024;
#define RATE(NT, MODE, name)                                                                                     \
  {                                                                                                               \
    hipLaunchKernelGGL((k_rate<NT, MODE>), dim3(blocks), dim3(256), 0, 0, dA, dO, 10);                            \
    hipEventRecord(e0); hipLaunchKernelGGL((k_rate<NT, MODE>), dim3(blocks), dim3(256), 0, 0, dA, dO, iters); hipEventRecord(e1); \
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);                                          \
    const double fl = 2.0 * 32 * 32 * 16 * NT * (double)iters * blocks * 4;                                       \
    printf("%-44s %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)\n", name, ms, fl / ms / 1e9);                         \
  }
  RATE(2, 0, "fp32 MFMA, 2 column tiles per operand");
  RATE(2, 1, "6 x bf16 MFMA + split, 2 column tiles");
  RATE(4, 0, "fp32 MFMA, 4 column tiles");
  RATE(4, 1, "6 x bf16 MFMA + split, 4 column tiles");
  RATE(1, 1, "6 x bf16 MFMA + split, 1 column tile");
  return 0;
}
