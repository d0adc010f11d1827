// Loss kernels: masked node/edge cross-entropy (+ per-class BCE trackers) and the reparameterised KL head.
//
// Masked CE: rows whose one-hot target is all-zero are skipped, the target class is the arg-max of the
// target row, loss = sum_rows CE(logits, class); the caller divides by the kept-row count.  Logit rows are
// tiny (C <= 16 classes) so one thread owns a row; everything the train step needs from the logits
// (CE sum, kept-row count, per-class sum of BCE-with-logits) comes out of ONE pass over them.
// Backward: dlogits = keep * (softmax - onehot) * coef, coef read from device memory (no host sync).
#include "common.h"
#include <algorithm>

#define CE_MAX_C 16

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                     float* __restrict__ part, int64_t rows, int C) {
  __shared__ float red[4][2 + CE_MAX_C];
  float loss = 0.f, cnt = 0.f, bce[CE_MAX_C];
#pragma unroll
  for (int c = 0; c < CE_MAX_C; ++c) bce[c] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    const float* l = logits + row * C; const float* t = target + row * C;
    float lv[CE_MAX_C], tv[CE_MAX_C];
    bool keep = false; int cls = 0; float tmax = -INFINITY, lmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) {
        lv[c] = l[c]; tv[c] = t[c];
        keep = keep || (tv[c] != 0.f);
        if (tv[c] > tmax) { tmax = tv[c]; cls = c; }
        lmax = fmaxf(lmax, lv[c]);
      }
    if (!keep) continue;
    float s = 0.f, lt = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) {
        s += expf(lv[c] - lmax);
        if (c == cls) lt = lv[c];
        bce[c] += fmaxf(lv[c], 0.f) - lv[c] * tv[c] + log1pf(expf(-fabsf(lv[c])));
      }
    loss += (lmax + logf(s)) - lt;
    cnt += 1.f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  loss = wave_sum(loss); cnt = wave_sum(cnt);
#pragma unroll
  for (int c = 0; c < CE_MAX_C; ++c) bce[c] = wave_sum(bce[c]);
  if (lane == 0) {
    red[wave][0] = loss; red[wave][1] = cnt;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) red[wave][2 + c] = bce[c];
  }
  __syncthreads();
  if (threadIdx.x < 2 + CE_MAX_C)
    part[(int64_t)blockIdx.x * (2 + CE_MAX_C) + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Sums the per-block partials column by column: 16 waves, wave w owns columns w and w + 16; every lane adds a strided
// subset in double, then a fixed xor tree joins the lanes (same order every run, insensitive to nparts).
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < 2 + CE_MAX_C; c += 16) {
    double s = 0.0;
    for (int p = lane; p < nparts; p += 64) s += (double)part[(int64_t)p * (2 + CE_MAX_C) + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[c] = (float)s;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                     const float* __restrict__ coef, float* __restrict__ dlogits,
                                                     int64_t rows, int C) {
  const float k = coef[0];
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    const float* l = logits + row * C; const float* t = target + row * C;
    float lv[CE_MAX_C];
    bool keep = false; int cls = 0; float tmax = -INFINITY, lmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) {
        lv[c] = l[c]; const float tv = t[c];
        keep = keep || (tv != 0.f);
        if (tv > tmax) { tmax = tv; cls = c; }
        lmax = fmaxf(lmax, lv[c]);
      }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) { lv[c] = expf(lv[c] - lmax); s += lv[c]; }
    const float inv = keep ? k / s : 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) dlogits[row * C + c] = keep ? (lv[c] * inv - (c == cls ? k : 0.f)) : 0.f;
  }
}

// z = mu + eps * exp(0.5*logvar);  part = per-block sum of (1 + logvar - mu^2 - exp(logvar))
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, float* __restrict__ z,
                                                     float* __restrict__ part, int64_t numel) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (int64_t)gridDim.x * blockDim.x) {
    const float m = mu[e], lv = logvar[e];
    z[e] = m + eps[e] * expf(0.5f * lv);
    s += 1.f + lv - m * m - expf(lv);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void kl_finalize_kernel(const float* __restrict__ part, float* __restrict__ kl, int nparts, float scale) {
  const int lane = threadIdx.x;
  double s = 0.0;
  for (int p = lane; p < nparts; p += 64) s += (double)part[p];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) kl[0] = (float)(s * scale);
}

// dmu = dz + dkl*mu/numel ; dlogvar = dz*eps*0.5*exp(0.5 lv) + dkl*0.5*(exp(lv)-1)/numel
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, const float* __restrict__ dz,
                                                     const float* __restrict__ dkl, float* __restrict__ dmu,
                                                     float* __restrict__ dlogvar, int64_t numel) {
  const float g = dkl ? dkl[0] / (float)numel : 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (int64_t)gridDim.x * blockDim.x) {
    const float m = mu[e], lv = logvar[e], d = dz ? dz[e] : 0.f;
    dmu[e] = d + g * m;
    dlogvar[e] = d * eps[e] * 0.5f * expf(0.5f * lv) + g * 0.5f * (expf(lv) - 1.f);
  }
}

// ---- scratch of the allocation-free entry point ------------------------------------------------------------
// One partial-sum row per workgroup of ce_fwd_kernel (at most 1024 of them): 72 KB that every GPU the library runs on
// holds until the process ends.  The array's NAME is only a host-side handle: kernels get the device address that
// hipGetSymbolAddress returns for the current device (each GPU has its own copy at its own address).
#define CE_STATIC_MAX_BLOCKS 1024
#define CE_STATIC_MAX_ROWS ((int64_t)1 << 27)
static __device__ __align__(256) float ce_static_scratch[CE_STATIC_MAX_BLOCKS * (2 + CE_MAX_C)];

// Device address of ce_static_scratch on the current device, looked up once per device; nullptr on failure.
static float* ce_static_scratch_ptr() {
  static std::atomic<float*> table[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  float* p = table[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ce_static_scratch));
    if (e != hipSuccess || sym == nullptr) {
      fprintf(stderr, "libmbvae_hip: hipGetSymbolAddress(ce_static_scratch) failed on device %d: %s\n", dev, hipGetErrorString(e));
      return nullptr;
    }
    p = (float*)sym;
    table[dev].store(p, std::memory_order_release);     // (racing threads look up the same address)
  }
  return p;
}

extern "C" {

int64_t mbvae_masked_ce_workspace() { return (int64_t)sizeof(float) * 1024 * (2 + CE_MAX_C); }

// out[0] = sum of CE over kept rows, out[1] = #kept rows, out[2+c] = sum over kept rows of BCEWithLogits(logit_c, target_c)
int mbvae_masked_ce_fwd(const float* logits, const float* target, float* out, int64_t rows, int64_t C,
                        void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!logits || !target || !out || !workspace || rows <= 0) return MBVAE_EINVAL;
  if (C <= 0 || C > CE_MAX_C) return MBVAE_EUNSUPPORTED;
  if (ws_bytes < mbvae_masked_ce_workspace()) return MBVAE_EINVAL;
  const int blocks = (int)std::min<int64_t>((rows + 255) / 256, 1024);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(blocks), dim3(256), 0, stream, logits, target, (float*)workspace, rows, (int)C);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, (const float*)workspace, out, blocks);
  return mbvae_launch_status();
}

// mbvae_masked_ce_fwd without a workspace argument and without any allocation: the per-workgroup partial sums go to a
// file-scope device array (72 KB per GPU).  Same kernels, same launch shape, hence the same bits as mbvae_masked_ce_fwd.
// Nothing in the array has to start at zero: the first kernel overwrites every row the second one reads.  All calls
// on one GPU share the array, so calls that overlap on one GPU (different streams or threads) race; serialise them.
int mbvae_masked_ce_fwd_static(const float* logits, const float* target, float* out, int64_t rows, int64_t C,
                               hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!logits || !target || !out || rows <= 0) return MBVAE_EINVAL;
  if (C <= 0 || C > CE_MAX_C) return MBVAE_EUNSUPPORTED;
  if (rows > CE_STATIC_MAX_ROWS) {
    fprintf(stderr, "libmbvae_hip: mbvae_masked_ce_fwd_static takes at most %lld rows (got %lld); use mbvae_masked_ce_fwd\n",
            (long long)CE_STATIC_MAX_ROWS, (long long)rows);
    return MBVAE_EUNSUPPORTED;
  }
  float* scratch = ce_static_scratch_ptr();
  if (!scratch) return MBVAE_ELAUNCH;
  const int blocks = (int)std::min<int64_t>((rows + 255) / 256, CE_STATIC_MAX_BLOCKS);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(blocks), dim3(256), 0, stream, logits, target, scratch, rows, (int)C);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, (const float*)scratch, out, blocks);
  return mbvae_launch_status();
}

// dlogits = keep * (softmax(logits) - onehot(class)) * coef[0]   (coef on device: lambda * upstream / count)
int mbvae_masked_ce_bwd(const float* logits, const float* target, const float* coef, float* dlogits, int64_t rows,
                        int64_t C, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!logits || !target || !coef || !dlogits || rows <= 0) return MBVAE_EINVAL;
  if (C <= 0 || C > CE_MAX_C) return MBVAE_EUNSUPPORTED;
  const int blocks = (int)std::min<int64_t>((rows + 255) / 256, 4096);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(blocks), dim3(256), 0, stream, logits, target, coef, dlogits, rows, (int)C);
  return mbvae_launch_status();
}

int64_t mbvae_kl_workspace() { return (int64_t)sizeof(float) * 1024; }

// z = mu + eps*exp(logvar/2); kl[0] = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) / numel
int mbvae_kl_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, float* kl, int64_t numel,
                         void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!mu || !logvar || !eps || !z || !kl || !workspace || numel <= 0) return MBVAE_EINVAL;
  if (ws_bytes < mbvae_kl_workspace()) return MBVAE_EINVAL;
  const int blocks = (int)std::min<int64_t>((numel + 255) / 256, 1024);
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(blocks), dim3(256), 0, stream, mu, logvar, eps, z, (float*)workspace, numel);
  hipLaunchKernelGGL(kl_finalize_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, kl, blocks,
                     -0.5f / (float)numel);
  return mbvae_launch_status();
}

// dz (may be null) is the gradient wrt z, dkl (device scalar, may be null) wrt the KL value.
int mbvae_kl_reparam_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, const float* dkl,
                         float* dmu, float* dlogvar, int64_t numel, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!mu || !logvar || !eps || !dmu || !dlogvar || numel <= 0) return MBVAE_EINVAL;
  const int blocks = (int)std::min<int64_t>((numel + 255) / 256, 1024);
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(blocks), dim3(256), 0, stream, mu, logvar, eps, dz, dkl, dmu, dlogvar, numel);
  return mbvae_launch_status();
}

}  // extern "C"
