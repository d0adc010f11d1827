// k-cycle counts of a batch of small graphs (n <= 64) in one launch: per node the numbers of 3-, 4- and 5-cycles it
// lies on, per graph the numbers of 3-, 4-, 5- and 6-cycles, from powers of the adjacency matrix
// (mb_vae_dti/training/diffusion/augmentation.py:177-250, KNodeCycles).  The reference forms A^2 .. A^6 with five
// batched matrix products and about thirty element-wise / reduction launches; only A^2 and A^3 are needed in full,
// because the higher powers enter through their diagonals and traces alone:
//     diag(A^4)_i = sum_j A2[i,j] A2[j,i],   diag(A^5)_i = sum_j A2[i,j] A3[j,i],   tr(A^6) = sum_ij A3[i,j] A3[j,i].
// One 256-thread workgroup per graph, A / A^2 / A^3 in LDS (50 KB), thread (row i, quarter q) owns 16 columns of row i.
// Counts of molecule-sized graphs are small integers, exact in fp32 whatever the summation order.
#include "common.h"

__global__ __launch_bounds__(256) void cycle_counts_kernel(const float* __restrict__ A, float* __restrict__ xc,
                                                           float* __restrict__ yc, int n) {
  constexpr int LD = 65;
  __shared__ float sA[64 * LD], s2[64 * LD], s3[64 * LD];
  __shared__ float sd[64], stri[64], srow[64][10];
  const int tid = threadIdx.x, i = tid >> 2, q = tid & 3;
  const int64_t b = blockIdx.x;
  const float* Ab = A + b * n * n;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    sA[r * LD + c] = (r < n && c < n) ? Ab[r * n + c] : 0.f;
  }
  __syncthreads();
  // A2 = A A, degree
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  float di = 0.f;
  for (int k = 0; k < 64; ++k) {
    const float a = sA[i * LD + k];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] += a * sA[k * LD + q * 16 + c];
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) { s2[i * LD + q * 16 + c] = acc[c]; di += sA[i * LD + q * 16 + c]; }
  di += __shfl_xor(di, 1); di += __shfl_xor(di, 2);
  if (q == 0) sd[i] = di;
  __syncthreads();
  // A3 = A2 A
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (int k = 0; k < 64; ++k) {
    const float a = s2[i * LD + k];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] += a * sA[k * LD + q * 16 + c];
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) s3[i * LD + q * 16 + c] = acc[c];
  __syncthreads();
  if (q == 0) stri[i] = s3[i * LD + i] * 0.5f;
  __syncthreads();
  // row quantities: partial sums over this thread's 16 columns, joined across the row's four threads
  float Ad = 0.f, dA4 = 0.f, t1 = 0.f, Atri = 0.f, dA5 = 0.f, tr6 = 0.f, aa22 = 0.f, s3sum = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int j = q * 16 + c;
    const float a = sA[i * LD + j], a2 = s2[i * LD + j], a3 = s3[i * LD + j];
    Ad += a * sd[j];
    dA4 += a2 * s2[j * LD + i];
    t1 += a2 * a * sd[j];
    Atri += a * stri[j];
    dA5 += a2 * s3[j * LD + i];
    tr6 += a3 * s3[j * LD + i];
    aa22 += a * a2 * a2;
    s3sum += a3;
  }
#define JOIN4(v) v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
  JOIN4(Ad) JOIN4(dA4) JOIN4(t1) JOIN4(Atri) JOIN4(dA5) JOIN4(tr6) JOIN4(aa22) JOIN4(s3sum)
#undef JOIN4
  if (q == 0) {
    const float d = sd[i], c3 = s3[i * LD + i], d2 = s2[i * LD + i], tri = c3 * 0.5f;
    const float c4 = dA4 - d * (d - 1.f) - Ad;
    const float c5 = dA5 - 2.f * t1 - 4.f * d * tri - 2.f * Atri + 10.f * tri;
    if (i < n) {
      float* o = xc + (b * n + i) * 3;
      o[0] = c3 * 0.5f; o[1] = c4 * 0.5f; o[2] = c5 * 0.5f;
    }
    float* rw = srow[i];
    rw[0] = c3; rw[1] = c4; rw[2] = c5; rw[3] = tr6; rw[4] = c3 * c3; rw[5] = aa22; rw[6] = d2 * dA4; rw[7] = dA4;
    rw[8] = d2; rw[9] = s3sum;
  }
  __syncthreads();
  if (tid < 64) {                                   // one wave: sums over the 64 rows (rows >= n hold zeros)
    float v[10], d23, d22;
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = srow[tid][k];
    d23 = v[8] * v[8] * v[8]; d22 = v[8] * v[8];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = wave_sum(v[k]);
    d23 = wave_sum(d23); d22 = wave_sum(d22);
    if (tid == 0) {
      const float c6 = v[3] - 3.f * v[4] + 9.f * v[5] - 6.f * v[6] + 6.f * v[7] - 4.f * v[0] + 4.f * d23 + 3.f * v[9]
                       - 12.f * d22 + 4.f * v[8];
      float* o = yc + b * 4;
      o[0] = v[0] / 6.f; o[1] = v[1] / 8.f; o[2] = v[2] / 10.f; o[3] = c6 / 12.f;
    }
  }
}

extern "C" {

// A [bs,n,n] fp32 adjacency (any real weights; the reference passes the summed bond one-hots) -> xc [bs,n,3] =
// (3-, 4-, 5-cycles through each node), yc [bs,4] = (3-, 4-, 5-, 6-cycles of the graph), unclamped.  n <= 64.
int mbvae_cycle_counts(const float* A, float* xc, float* yc, int64_t bs, int64_t n, hipStream_t stream) {
  (void)hipGetLastError();
  if (!A || !xc || !yc || bs <= 0 || n <= 0) return MBVAE_EINVAL;
  if (n > 64 || bs > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  hipLaunchKernelGGL(cycle_counts_kernel, dim3((unsigned)bs), dim3(256), 0, stream, A, xc, yc, (int)n);
  return mbvae_launch_status();
}

}  // extern "C"
