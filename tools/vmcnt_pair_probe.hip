// Micro-test of the instruction pattern behind the node-path LayerNorm backward's rare wrong rows (see
// tools/concurrency_probe.py): two per-row scalars sit in ONE register pair that two separate global_load_dword fill,
// wider loads are issued behind them, a counted s_waitcnt leaves the last two of those in flight, and a packed-fp32
// instruction then reads the pair with op_sel (both halves from the HIGH register).  Variant PLAIN reads the high
// register with an ordinary v_mov instead.  A streaming kernel runs on a second stream meanwhile.
// Build: hipcc -O3 --offload-arch=gfx950 tools/vmcnt_pair_probe.hip -o tools/_bin/vmcnt_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// rows: 16384 of 256 floats; block = 4 waves, wave w of block b takes rows b*4+w + k*4096 (+2048 for the second row of a trip)
template <bool PLAIN>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ x,
                                               const float* __restrict__ dy, const float* __restrict__ r, unsigned* __restrict__ flags,
                                               float* __restrict__ sink, int rows) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  float acc = 0.f;
  for (int row0 = blockIdx.x * 4 + grp; row0 < rows; row0 += 4096) {
    const int row1 = row0 + 2048;
    const float* pm0 = mean + row0; const float* pr0 = rstd + row0;
    const float* pm1 = mean + row1; const float* pr1 = rstd + row1;
    const float* px0 = x + (size_t)row0 * 256 + lane * 4; const float* pd0 = dy + (size_t)row0 * 256 + lane * 4; const float* pq0 = r + (size_t)row0 * 256 + lane * 4;
    const float* px1 = x + (size_t)row1 * 256 + lane * 4; const float* pd1 = dy + (size_t)row1 * 256 + lane * 4; const float* pq1 = r + (size_t)row1 * 256 + lane * 4;
    float lo, hi, direct, s;
    asm volatile(
        "v_mov_b32 v200, 0x7fc00001\n\t"
        "v_mov_b32 v201, 0x7fc00001\n\t"
        "v_mov_b32 v218, 0\n\t"
        "v_mov_b32 v219, 0\n\t"
        "global_load_dword v200, %[pm0], off\n\t"
        "global_load_dword v210, %[pr0], off\n\t"
        "global_load_dwordx4 v[202:205], %[px0], off\n\t"
        "global_load_dwordx4 v[206:209], %[pd0], off\n\t"
        "global_load_dwordx4 v[212:215], %[pq0], off\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_add_f32 v220, v202, v206\n\t"
        "v_add_f32 v220, v220, v212\n\t"
        "v_add_f32 v220, v220, v210\n\t"
        "global_load_dword v201, %[pm1], off\n\t"
        "global_load_dword v211, %[pr1], off\n\t"
        "global_load_dwordx4 v[202:205], %[px1], off\n\t"
        "global_load_dwordx4 v[206:209], %[pd1], off\n\t"
        "global_load_dwordx4 v[212:215], %[pq1], off\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        ".if %[plain]\n\t"
        "v_mov_b32 v216, v201\n\t"
        "v_mov_b32 v217, v201\n\t"
        ".else\n\t"
        "v_pk_add_f32 v[216:217], v[218:219], v[200:201] op_sel:[0,1]\n\t"
        ".endif\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[lo], v216\n\t"
        "v_mov_b32 %[hi], v217\n\t"
        "v_mov_b32 %[direct], v201\n\t"
        "v_add_f32 %[s], v220, v205\n\t"
        "v_add_f32 %[s], %[s], v209\n\t"
        "v_add_f32 %[s], %[s], v215\n\t"
        "v_add_f32 %[s], %[s], v211\n\t"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [direct] "=&v"(direct), [s] "=&v"(s)
        : [pm0] "v"(pm0), [pr0] "v"(pr0), [px0] "v"(px0), [pd0] "v"(pd0), [pq0] "v"(pq0), [pm1] "v"(pm1), [pr1] "v"(pr1), [px1] "v"(px1),
          [pd1] "v"(pd1), [pq1] "v"(pq1), [plain] "n"(PLAIN ? 1 : 0)
        : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215",
          "v216", "v217", "v218", "v219", "v220");
    acc += s;
    const unsigned bad = (__float_as_uint(lo) != __float_as_uint(direct) ? 1u : 0u) | (__float_as_uint(hi) != __float_as_uint(direct) ? 2u : 0u);
    if (bad) flags[(size_t)row1 * 64 + lane] = bad;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

// partner: every wave streams through a large buffer with many 16-byte non-temporal loads in flight and writes a little
__global__ __launch_bounds__(512) void stream_reader(const vf4* __restrict__ src, float* __restrict__ out, size_t n4, int passes) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = (size_t)gridDim.x * 512;
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i + 3 * stride < n4; i += 4 * stride) {
      const vf4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
      const vf4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
      s.x += a.x + b.x + c.x + d.x; s.y += a.y + b.y + c.y + d.y; s.z += a.z + b.z + c.z + d.z; s.w += a.w + b.w + c.w + d.w;
    }
  out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

__global__ __launch_bounds__(512) void stream_writer(vf4* __restrict__ dst, size_t n4, int passes) {
  const size_t stride = (size_t)gridDim.x * 512;
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += stride)
      __builtin_nontemporal_store(vf4{(float)p, 1.f, 2.f, 3.f}, dst + i);
}

int main() {
  const int rows = 16384;
  const size_t nb = (size_t)rows * 256 * 4;
  float *mean, *rstd, *x, *dy, *r, *sink, *big, *pout;
  unsigned* flags;
  CK(hipMalloc(&mean, rows * 4)); CK(hipMalloc(&rstd, rows * 4)); CK(hipMalloc(&x, nb)); CK(hipMalloc(&dy, nb)); CK(hipMalloc(&r, nb));
  CK(hipMalloc(&sink, 4096 * 256 * 4)); CK(hipMalloc(&flags, (size_t)rows * 64 * 4)); CK(hipMalloc(&pout, 2048 * 512 * 4));
  const size_t big_bytes = (size_t)1 << 30;
  CK(hipMalloc(&big, big_bytes));
  std::vector<float> h(rows * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
  CK(hipMemcpy(x, h.data(), nb, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, h.data(), nb, hipMemcpyHostToDevice)); CK(hipMemcpy(r, h.data(), nb, hipMemcpyHostToDevice));
  for (int i = 0; i < rows; ++i) h[i] = 1.f + i;          // mean[row] = row + 1: a stale value names the row it came from
  CK(hipMemcpy(mean, h.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rstd, h.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMemset(big, 0, big_bytes));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  std::vector<unsigned> hf((size_t)rows * 64);
  const char* pn[] = {"(idle)", "streaming reader", "streaming writer"};
  for (int variant = 0; variant < 2; ++variant)
    for (int p = 0; p < 3; ++p) {
      CK(hipMemset(flags, 0, (size_t)rows * 64 * 4));
      CK(hipDeviceSynchronize());
      if (p == 1) hipLaunchKernelGGL(stream_reader, dim3(1024), dim3(512), 0, sa, (const vf4*)big, pout, big_bytes / 16, 12);
      if (p == 2) hipLaunchKernelGGL(stream_writer, dim3(1024), dim3(512), 0, sa, (vf4*)big, big_bytes / 16, 12);
      for (int rep = 0; rep < 200; ++rep) {
        if (variant == 0) hipLaunchKernelGGL(victim<false>, dim3(512), dim3(256), 0, sb, mean, rstd, x, dy, r, flags, sink, rows);
        else hipLaunchKernelGGL(victim<true>, dim3(512), dim3(256), 0, sb, mean, rstd, x, dy, r, flags, sink, rows);
      }
      CK(hipStreamSynchronize(sb));
      const bool running = p > 0 && hipStreamQuery(sa) == hipErrorNotReady;
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hf.data(), flags, hf.size() * 4, hipMemcpyDeviceToHost));
      size_t bad_lo = 0, bad_hi = 0; int lane_hist[4] = {0, 0, 0, 0};
      for (size_t i = 0; i < hf.size(); ++i) if (hf[i]) { bad_lo += hf[i] & 1; bad_hi += (hf[i] >> 1) & 1; lane_hist[(i & 63) >> 4]++; }
      printf("%-34s next to %-17s: low-half results stale in %zu (row, lane) slots, high-half in %zu; by lane quarter %d %d %d %d%s\n",
             variant == 0 ? "v_pk_add_f32 ... op_sel:[0,1]" : "v_mov_b32 from the high register", pn[p], bad_lo, bad_hi, lane_hist[0], lane_hist[1],
             lane_hist[2], lane_hist[3], p > 0 ? (running ? "  (partner ran throughout)" : "  (partner ended early)") : "");
      fflush(stdout);
    }
  return 0;
}
