// Batched symmetric eigendecomposition of small (n <= 64) matrices: one workgroup per matrix, everything in LDS.
//
// Replaces the torch.linalg.eigh call on the [bs, n, n] graph Laplacians of the extra-feature path
// (rocSOLVER's generic batched path spends ~100 ms on 256 matrices of 64x64; this kernel needs ~0.4 ms).
// Method: two-sided cyclic Jacobi with the round-robin ("circle") parallel ordering -- n/2 disjoint rotations
// per round, n-1 rounds per sweep -- with the rotations of a round applied to A from both sides and to the
// eigenvector matrix V by all 1024 threads (the kernel is latency bound: one matrix per CU, so more waves per
// matrix and fewer barriers shorten every round).  Rotations between decoupled index sets (a_pq == 0) are
// identities, so a block-diagonal Laplacian (disconnected graph) yields eigenvectors supported on single
// components.  Eigenvalues come out ascending, eigenvectors in the columns of V, as LAPACK orders them.
#include "common.h"

#define EIG_MAX_N 64
#define EIG_LD 65

#define EIG_NT 1024

// Rotation that annihilates a_pq (identity for a pair with the bye / a decoupled index, or a_pq == 0).
__device__ __forceinline__ void eig_rotation(const float* __restrict__ A, int p, int q, int ne, float& c, float& s) {
  c = 1.f; s = 0.f;
  if (q < ne) {
    const float apq = A[p * EIG_LD + q];
    if (fabsf(apq) > 1e-30f) {
      const float tau = (A[q * EIG_LD + q] - A[p * EIG_LD + p]) / (2.f * apq);
      const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
      c = 1.f / sqrtf(1.f + t * t);
      s = t * c;
    }
  }
}

// Pair `k` of round r of the round-robin tournament over m players (p < q).
__device__ __forceinline__ void eig_pair(int k, int r, int m, int& p, int& q) {
  if (k == 0) { p = m - 1; q = r; }
  else {                                  // (r + k) mod (m - 1), (r - k) mod (m - 1): r < m - 1 and k < m / 2
    p = r + k; if (p >= m - 1) p -= m - 1;
    q = r - k; if (q < 0) q += m - 1;
  }
  if (p > q) { const int t = p; p = q; q = t; }
}

// Two barriers per round.  Phase 1: lane k < m/2 derives rotation k of the round and publishes (p, q, c, s).  Phase 2:
// thread (a, b) owns the 2x2 block of A at rows (p_a, q_a) and columns (p_b, q_b): it applies J_a^T from the left and J_b
// from the right in registers - the same arithmetic, element for element, as a pass over the rows followed by a pass over
// the columns, without the barrier between them - and writes the block back (blocks are disjoint); it also rotates rows
// 2b, 2b+1 of the eigenvector matrix by J_a (no other thread touches those entries).
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
  const int ba = half > 0 ? tid / half : 0, bb = half > 0 ? tid % half : 0;
  const bool owner = half > 0 && ba < half;
  float prev_off2 = INFINITY;
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
    // Converged, or at the rounding floor: in fp32 the off-diagonal mass stalls at 1e-14 .. 2e-14 of the total (the square of
    // the unit round-off), so a fixed bound there is met by some matrices and never by others; a sweep that no longer
    // halves a mass already below 1e-12 of the total (ten times the floor's upper end) ends the iteration.
    if (off2 <= 1e-14f * total2 || off2 == 0.f || (off2 <= 1e-12f * total2 && off2 >= 0.5f * prev_off2)) break;
    prev_off2 = off2;

    for (int r = 0; r < m - 1; ++r) {
      if (tid < half) {
        int p, q;
        float c, sn;
        eig_pair(tid, r, m, p, q);
        eig_rotation(A, p, q, ne, c, sn);
        rot_p[tid] = p; rot_q[tid] = q; rot_c[tid] = c; rot_s[tid] = sn;
      }
      __syncthreads();
      if (owner) {
        const int pa = rot_p[ba], qa = rot_q[ba], pb = rot_p[bb], qb = rot_q[bb];
        const float ca = rot_c[ba], sa = rot_s[ba], cb = rot_c[bb], sb = rot_s[bb];
        if (sa != 0.f || sb != 0.f) {
          float b00 = A[pa * EIG_LD + pb], b01 = A[pa * EIG_LD + qb], b10 = A[qa * EIG_LD + pb], b11 = A[qa * EIG_LD + qb];
          // rows: J_a^T from the left (only inside the coupled part, as the identity elsewhere)
          if (sa != 0.f) {
            if (pb < ne) { const float u = b00, v = b10; b00 = ca * u - sa * v; b10 = sa * u + ca * v; }
            if (qb < ne) { const float u = b01, v = b11; b01 = ca * u - sa * v; b11 = sa * u + ca * v; }
          }
          // columns: J_b from the right
          if (sb != 0.f) {
            if (pa < ne) { const float u = b00, v = b01; b00 = cb * u - sb * v; b01 = sb * u + cb * v; }
            if (qa < ne) { const float u = b10, v = b11; b10 = cb * u - sb * v; b11 = sb * u + cb * v; }
          }
          A[pa * EIG_LD + pb] = b00; A[pa * EIG_LD + qb] = b01; A[qa * EIG_LD + pb] = b10; A[qa * EIG_LD + qb] = b11;
        }
        if (sa != 0.f) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * bb + kk;
            if (k < ne) {
              const float vp = V[k * EIG_LD + pa], vq = V[k * EIG_LD + qa];
              V[k * EIG_LD + pa] = ca * vp - sa * vq;
              V[k * EIG_LD + qa] = sa * vp + ca * vq;
            }
          }
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
