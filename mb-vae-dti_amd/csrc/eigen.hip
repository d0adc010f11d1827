// Batched symmetric eigendecomposition of small (n <= 64) matrices: one workgroup per matrix, everything in LDS.
//
// Replaces the torch.linalg.eigh call on the [bs, n, n] graph Laplacians of the extra-feature path
// (rocSOLVER's generic batched path spends ~100 ms on 256 matrices of 64x64; this kernel needs ~1 ms).
// Method: two-sided cyclic Jacobi with the round-robin ("circle") parallel ordering -- n/2 disjoint rotations
// per round, n-1 rounds per sweep -- with the rotations of a round applied to rows, then to columns of A and
// to the eigenvector matrix V by all 1024 threads (the kernel is latency bound: one matrix per CU, so more waves per
// matrix shorten every round).  Rotations between decoupled index sets (a_pq == 0) are
// identities, so a block-diagonal Laplacian (disconnected graph) yields eigenvectors supported on single
// components.  Eigenvalues come out ascending, eigenvectors in the columns of V, as LAPACK orders them.
#include "common.h"

#define EIG_MAX_N 64
#define EIG_LD 65

#define EIG_NT 1024
__global__ __launch_bounds__(EIG_NT) void sym_eig_jacobi_kernel(const float* __restrict__ A_in, float* __restrict__ W,
                                                             float* __restrict__ V_out, int n, int max_sweeps) {
  __shared__ float A[EIG_MAX_N * EIG_LD];
  __shared__ float V[EIG_MAX_N * EIG_LD];
  __shared__ float rot_c[EIG_MAX_N / 2], rot_s[EIG_MAX_N / 2];
  __shared__ int rot_p[EIG_MAX_N / 2], rot_q[EIG_MAX_N / 2];
  __shared__ float red[EIG_NT / 64];
  __shared__ float wsort[EIG_MAX_N];
  __shared__ int rank_of[EIG_MAX_N];
  const int tid = threadIdx.x;
  const float* src = A_in + (int64_t)blockIdx.x * n * n;
  for (int e = tid; e < EIG_MAX_N * EIG_MAX_N; e += EIG_NT) {
    const int i = e >> 6, j = e & 63;
    A[i * EIG_LD + j] = (i < n && j < n) ? src[i * n + j] : 0.f;
    V[i * EIG_LD + j] = (i == j) ? 1.f : 0.f;
  }
  __syncthreads();
  // scale for the convergence test: squared Frobenius norm
  float nrm = 0.f;
  for (int e = tid; e < n * n; e += EIG_NT) { const float v = A[(e / n) * EIG_LD + (e % n)]; nrm += v * v; }
  nrm = wave_sum(nrm);
  if ((tid & 63) == 0) red[tid >> 6] = nrm;
  __syncthreads();
  float total2 = 0.f;
  for (int w = 0; w < EIG_NT / 64; ++w) total2 += red[w];
  __syncthreads();

  // Trailing all-zero rows (padding nodes of a ragged batch, isolated last atoms) are decoupled from everything:
  // their eigenpairs (a_ii, e_i) are already in place, so the tournament only runs over the leading ne indices.
  __shared__ int ne_s;
  if (tid == 0) ne_s = 0;
  __syncthreads();
  if (tid < n) {
    bool nz = false;
    for (int j = 0; j < n; ++j) nz |= (j != tid) && A[tid * EIG_LD + j] != 0.f;
    if (nz) atomicMax(&ne_s, tid + 1);
  }
  __syncthreads();
  const int ne = ne_s;
  const int m = (ne + 1) & ~1;           // even number of players (index ne is a bye when ne is odd)
  const int half = m / 2;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    // off-diagonal mass
    float off = 0.f;
    for (int e = tid; e < ne * ne; e += EIG_NT) {
      const int i = e / ne, j = e % ne;
      if (i != j) { const float v = A[i * EIG_LD + j]; off += v * v; }
    }
    off = wave_sum(off);
    if ((tid & 63) == 0) red[tid >> 6] = off;
    __syncthreads();
    float off2 = 0.f;
    for (int w = 0; w < EIG_NT / 64; ++w) off2 += red[w];
    __syncthreads();
    if (off2 <= 1e-14f * total2 || off2 == 0.f) break;

    for (int r = 0; r < m - 1; ++r) {
      if (tid < half) {
        int p, q;
        if (tid == 0) { p = m - 1; q = r; }
        else { p = (r + tid) % (m - 1); q = (r - tid + (m - 1)) % (m - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        float c = 1.f, s = 0.f;
        if (q < ne) {
          const float apq = A[p * EIG_LD + q];
          if (fabsf(apq) > 1e-30f) {
            const float tau = (A[q * EIG_LD + q] - A[p * EIG_LD + p]) / (2.f * apq);
            const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
            c = 1.f / sqrtf(1.f + t * t);
            s = t * c;
          }
        }
        rot_p[tid] = p; rot_q[tid] = q; rot_c[tid] = c; rot_s[tid] = s;
      }
      __syncthreads();
      // rows: A <- J^T A
      for (int e = tid; e < half * EIG_MAX_N; e += EIG_NT) {
        const int pr = e >> 6, k = e & 63;
        const int p = rot_p[pr], q = rot_q[pr];
        const float c = rot_c[pr], s = rot_s[pr];
        if (q < ne && s != 0.f && k < ne) {
          const float ap = A[p * EIG_LD + k], aq = A[q * EIG_LD + k];
          A[p * EIG_LD + k] = c * ap - s * aq;
          A[q * EIG_LD + k] = s * ap + c * aq;
        }
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int e = tid; e < half * EIG_MAX_N; e += EIG_NT) {
        const int pr = e >> 6, k = e & 63;
        const int p = rot_p[pr], q = rot_q[pr];
        const float c = rot_c[pr], s = rot_s[pr];
        if (q < ne && s != 0.f && k < ne) {
          const float ap = A[k * EIG_LD + p], aq = A[k * EIG_LD + q];
          A[k * EIG_LD + p] = c * ap - s * aq;
          A[k * EIG_LD + q] = s * ap + c * aq;
          const float vp = V[k * EIG_LD + p], vq = V[k * EIG_LD + q];
          V[k * EIG_LD + p] = c * vp - s * vq;
          V[k * EIG_LD + q] = s * vp + c * vq;
        }
      }
      __syncthreads();
    }
  }
  // ascending order (ties by index), scatter eigenvalues and eigenvector columns
  if (tid < n) wsort[tid] = A[tid * EIG_LD + tid];
  __syncthreads();
  if (tid < n) {
    const float w = wsort[tid];
    int rk = 0;
    for (int j = 0; j < n; ++j) { const float u = wsort[j]; rk += (u < w || (u == w && j < tid)) ? 1 : 0; }
    rank_of[tid] = rk;
    W[(int64_t)blockIdx.x * n + rk] = w;
  }
  __syncthreads();
  float* dst = V_out + (int64_t)blockIdx.x * n * n;
  for (int e = tid; e < n * n; e += EIG_NT) {
    const int k = e / n, j = e % n;                      // V[k][j] -> column rank_of[j]
    dst[k * n + rank_of[j]] = V[k * EIG_LD + j];
  }
}

extern "C" {

// A [bs,n,n] symmetric -> W [bs,n] ascending eigenvalues, V [bs,n,n] eigenvectors in columns.  n <= 64.
int mbvae_sym_eig(const float* A, float* W, float* V, int64_t bs, int64_t n, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!A || !W || !V || bs <= 0 || n <= 0) return MBVAE_EINVAL;
  if (n > EIG_MAX_N || bs > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  hipLaunchKernelGGL(sym_eig_jacobi_kernel, dim3((unsigned)bs), dim3(EIG_NT), 0, stream, A, W, V, (int)n, 16);
  return mbvae_launch_status();
}

}  // extern "C"
