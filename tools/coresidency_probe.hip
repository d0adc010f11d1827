// Does the node-path LayerNorm backward (rowops.hip, 256 features: one wave per row) return other bits when waves of
// ANOTHER kernel share its SIMDs?  The partner kernels below are synthetic and touch no memory the victim uses: a
// matrix-core spin loop (bf16 or f32 MFMA), a vector-ALU spin loop, an LDS read loop.  They are sized (few registers,
// little LDS) so that victim waves fit next to them on the same SIMD.
// Build:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on [-DVARIANT...] tools/coresidency_probe.hip -o gpurun_out/coresidency_probe
#include "../mb-vae-dti_amd/csrc/rowops.hip"
#include <cstring>
#include <vector>

typedef float p_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 p_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void spin_mfma_bf16(float* out, int iters) {
  p_f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
  p_bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void spin_mfma_f32(float* out, int iters) {
  p_f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void spin_valu(float* out, int iters) {
  float x[8];
  const float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void spin_lds(float* out, int iters) {
  __shared__ float4 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 512) buf[i] = make_float4(i, 1.f, 2.f, 3.f);
  __syncthreads();
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
    const float4 v = buf[(threadIdx.x + it * 17) & 2047];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int64_t rows = 16384, D = 256;
  const size_t nb = rows * D * sizeof(float);
  std::vector<float> h(rows * D);
  float *x, *r, *dy, *gamma, *beta, *mean, *rstd, *y, *ws, *spin_out;
  CK(hipMalloc(&x, nb)); CK(hipMalloc(&r, nb)); CK(hipMalloc(&dy, nb)); CK(hipMalloc(&y, nb));
  CK(hipMalloc(&gamma, D * 4)); CK(hipMalloc(&beta, D * 4)); CK(hipMalloc(&mean, rows * 4)); CK(hipMalloc(&rstd, rows * 4));
  const int64_t ws_bytes = 8 << 20;
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMalloc(&spin_out, 1024 * 512 * 4));
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
  for (float** p : {&x, &r, &dy}) { for (auto& v : h) v = rnd(); CK(hipMemcpy(*p, h.data(), nb, hipMemcpyHostToDevice)); }
  for (int i = 0; i < D; ++i) h[i] = 1.f + 0.5f * rnd();
  CK(hipMemcpy(gamma, h.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, h.data(), D * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  if (mbvae_ln_residual_fwd(x, r, gamma, beta, y, mean, rstd, rows, D, 1e-5f, 0.1f, 1234, sb) != 0) return 2;
  const int REPS = 40;
  const size_t outb = 2 * nb + 2 * D * 4;
  std::vector<char*> outs(REPS + 1);
  for (auto& o : outs) CK(hipMalloc(&o, outb));
  auto victim = [&](char* o) {
    float* dx = (float*)o; float* dr = dx + rows * D; float* dg = dr + rows * D; float* db = dg + D;
    return mbvae_ln_residual_bwd(x, r, gamma, mean, rstd, dy, dx, dr, dg, db, rows, D, 0.1f, 1234, ws, ws_bytes, sb);
  };
  if (victim(outs[REPS]) != 0) return 3;
  CK(hipDeviceSynchronize());
  std::vector<char> ref(outb), got(outb);
  CK(hipMemcpy(ref.data(), outs[REPS], outb, hipMemcpyDeviceToHost));
  const char* names[] = {"(idle)", "bf16 MFMA spin", "f32 MFMA spin", "vector-ALU spin", "LDS read spin"};
  for (int p = 0; p < 5; ++p) {
    // the partner runs for a few milliseconds on every CU; the victims are issued right behind its launch
    const int iters = 400000;
    if (p == 1) hipLaunchKernelGGL(spin_mfma_bf16, dim3(512), dim3(512), 0, sa, spin_out, iters / 8);
    if (p == 2) hipLaunchKernelGGL(spin_mfma_f32, dim3(512), dim3(512), 0, sa, spin_out, iters / 16);
    if (p == 3) hipLaunchKernelGGL(spin_valu, dim3(512), dim3(512), 0, sa, spin_out, iters / 4);
    if (p == 4) hipLaunchKernelGGL(spin_lds, dim3(512), dim3(512), 0, sa, spin_out, iters);
    for (int i = 0; i < REPS; ++i) if (victim(outs[i]) != 0) return 4;
    hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, sb));
    CK(hipEventSynchronize(ev));
    const bool partner_still_running = p > 0 && hipStreamQuery(sa) == hipErrorNotReady;
    CK(hipDeviceSynchronize());
    int bad = 0, bad_rows = 0;
    for (int i = 0; i < REPS; ++i) {
      CK(hipMemcpy(got.data(), outs[i], outb, hipMemcpyDeviceToHost));
      if (memcmp(got.data(), ref.data(), outb) != 0) {
        ++bad;
        if (bad == 1) {
          const float* a = (const float*)ref.data(); const float* b = (const float*)got.data();
          for (int64_t row = 0; row < rows; ++row) bad_rows += memcmp(a + row * D, b + row * D, D * 4) != 0;
          const float* ga = a + 2 * rows * D; const float* gb = b + 2 * rows * D;
          printf("    first bad result: %d rows of dx differ; dgamma columns that differ:", bad_rows);
          for (int c = 0; c < D; ++c) if (ga[c] != gb[c]) printf(" %d", c);
          printf("\n");
        }
      }
    }
    printf("LayerNorm backward [16384 x 256] next to %-18s: %2d of %d results differ from the idle-GPU result%s\n", names[p], bad, REPS,
           p > 0 ? (partner_still_running ? "  (partner ran throughout)" : "  (partner ended early)") : "");
    fflush(stdout);
  }
  return 0;
}
