// Discrete-diffusion sampling kernels.
//
// Both samplers draw a category as argmax_k prob[k] / e[k] with e ~ Exp(1) supplied by the caller
// (that is how ATen's multinomial(1) consumes its generator), so given the same Exp(1) tensors the
// class ids are reproducible against the PyTorch reference.  All n*n pairs own a slot in the Exp(1)
// tensor (keeps the RNG stream aligned with the reference) but only the strict upper triangle is
// sampled; the result is mirrored, the diagonal gets class 0, masked nodes/pairs get all-zero rows.
//
// Forward noising: prob = onehot(x0) @ Qbar_t with Qbar_t = abar*I + (1-abar)*1 m^T, i.e. row x0 of
// Qbar_t.  The arithmetic is kept un-contracted (no fma) so the probabilities are bit-identical to
// the reference's elementwise construction of Qbar_t followed by a one-hot matmul.
// Reverse step: p(z_s | z_t) = sum_x0 softmax(logits)[x0] * q(z_s | z_t, x0), closed form for the
// rank-one-plus-diagonal transition matrices (no [bs,N,c,c] intermediate).
#include "common.h"
#include <algorithm>

#define DF_MAX_C 16

// class of a one-hot row (first arg-max), or -1 if the row is all zero
__device__ __forceinline__ int onehot_class(const float* row, int C) {
  int cls = -1; float best = 0.f;
  for (int c = 0; c < C; ++c) { const float v = row[c]; if (v != 0.f && (cls < 0 || v > best)) { cls = c; best = v; } }
  return cls;
}

__device__ __forceinline__ int sample_argmax(const float* prob, const float* e, int C) {
  int best = 0; float bv = __fdiv_rn(prob[0], e[0]);
  for (int c = 1; c < C; ++c) { const float v = __fdiv_rn(prob[c], e[c]); if (v > bv) { bv = v; best = c; } }
  return best;
}

__device__ __forceinline__ void forward_probs(int cls, float abar, const float* marg, int C, bool valid, float* prob) {
  const float om = __fsub_rn(1.f, abar);
  for (int c = 0; c < C; ++c) {
    if (!valid) prob[c] = 1.0f / C;
    else if (cls < 0) prob[c] = 0.f;
    else prob[c] = __fadd_rn(c == cls ? abar : 0.f, __fmul_rn(om, marg[c]));
  }
}

// one thread per node / per pair.  x0: [bs,n,C] one-hot (all-zero rows allowed).
__global__ __launch_bounds__(256) void noise_nodes_kernel(const float* __restrict__ x0, const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ abar, const float* __restrict__ marg,
                                                          const float* __restrict__ expo, int64_t* __restrict__ cls_out,
                                                          float* __restrict__ onehot_out, int64_t nodes, int n, int C) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nodes) return;
  const int64_t b = r / n;
  const bool valid = mask[r] != 0;
  float prob[DF_MAX_C];
  forward_probs(onehot_class(x0 + r * C, C), abar[b], marg, C, valid, prob);
  const int k = sample_argmax(prob, expo + r * C, C);
  cls_out[r] = k;
  for (int c = 0; c < C; ++c) onehot_out[r * C + c] = (valid && c == k) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void noise_pairs_kernel(const float* __restrict__ e0, const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ abar, const float* __restrict__ marg,
                                                          const float* __restrict__ expo, int64_t* __restrict__ cls_out,
                                                          float* __restrict__ onehot_out, int64_t pairs, int n, int C) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pairs) return;
  const int64_t bi = p / n; const int j = (int)(p - bi * n);
  const int64_t b = bi / n; const int i = (int)(bi - b * n);
  const bool valid = mask[b * n + i] && mask[b * n + j];
  int k = 0;
  if (i != j) {
    const int lo = min(i, j), hi = max(i, j);
    const int64_t src = (b * n + lo) * n + hi;          // upper-triangle slot decides both (i,j) and (j,i)
    float prob[DF_MAX_C];
    forward_probs(onehot_class(e0 + src * C, C), abar[b], marg, C, valid, prob);
    k = sample_argmax(prob, expo + src * C, C);
  }
  cls_out[p] = k;
  for (int c = 0; c < C; ++c) onehot_out[p * C + c] = (valid && c == k) ? 1.f : 0.f;
}

// p(z_s | z_t) for one element, reference order of operations (training/modules/full.py:1120-1153 of the reference)
__device__ __forceinline__ void reverse_probs(const float* logits, int zt, float beta, float abar_s, float abar_t,
                                              const float* marg, int C, float* prob) {
  float p0[DF_MAX_C];
  float lmax = -INFINITY;
  for (int c = 0; c < C; ++c) lmax = fmaxf(lmax, logits[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) { p0[c] = expf(logits[c] - lmax); s += p0[c]; }
  for (int c = 0; c < C; ++c) p0[c] = p0[c] / s;
  float tot = 0.f;
  for (int k = 0; k < C; ++k) {
    // left[k] = Qt[k][zt]  (zero row if z_t is all-zero)
    const float left = zt < 0 ? 0.f : beta * marg[zt] + (k == zt ? 1.f - beta : 0.f);
    float u = 0.f;
    for (int x0 = 0; x0 < C; ++x0) {
      const float qsb = (1.f - abar_s) * marg[k] + (x0 == k ? abar_s : 0.f);
      float den = zt < 0 ? 0.f : (1.f - abar_t) * marg[zt] + (x0 == zt ? abar_t : 0.f);
      if (den == 0.f) den = 1e-6f;
      u += p0[x0] * (left * qsb / den);
    }
    prob[k] = u; tot += u;
  }
  if (tot == 0.f) { for (int k = 0; k < C; ++k) prob[k] = 1e-5f; tot = 1e-5f * C; }
  for (int k = 0; k < C; ++k) prob[k] = prob[k] / tot;
}

__global__ __launch_bounds__(256) void reverse_nodes_kernel(const float* __restrict__ logits, const float* __restrict__ zt,
                                                            const uint8_t* __restrict__ mask, const float* __restrict__ beta,
                                                            const float* __restrict__ abar_s, const float* __restrict__ abar_t,
                                                            const float* __restrict__ marg, const float* __restrict__ expo,
                                                            int64_t* __restrict__ cls_out, float* __restrict__ onehot_out,
                                                            float* __restrict__ prob_out, int64_t nodes, int n, int C) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nodes) return;
  const int64_t b = r / n;
  const bool valid = mask[r] != 0;
  float prob[DF_MAX_C];
  reverse_probs(logits + r * C, onehot_class(zt + r * C, C), beta[b], abar_s[b], abar_t[b], marg, C, prob);
  if (prob_out) for (int c = 0; c < C; ++c) prob_out[r * C + c] = prob[c];
  if (!valid) for (int c = 0; c < C; ++c) prob[c] = 1.0f / C;
  const int k = sample_argmax(prob, expo + r * C, C);
  cls_out[r] = k;
  for (int c = 0; c < C; ++c) onehot_out[r * C + c] = (valid && c == k) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void reverse_pairs_kernel(const float* __restrict__ logits, const float* __restrict__ zt,
                                                            const uint8_t* __restrict__ mask, const float* __restrict__ beta,
                                                            const float* __restrict__ abar_s, const float* __restrict__ abar_t,
                                                            const float* __restrict__ marg, const float* __restrict__ expo,
                                                            int64_t* __restrict__ cls_out, float* __restrict__ onehot_out,
                                                            float* __restrict__ prob_out, int64_t pairs, int n, int C) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pairs) return;
  const int64_t bi = p / n; const int j = (int)(p - bi * n);
  const int64_t b = bi / n; const int i = (int)(bi - b * n);
  const bool valid = mask[b * n + i] && mask[b * n + j];
  if (prob_out) {
    float prob[DF_MAX_C];
    reverse_probs(logits + p * C, onehot_class(zt + p * C, C), beta[b], abar_s[b], abar_t[b], marg, C, prob);
    for (int c = 0; c < C; ++c) prob_out[p * C + c] = prob[c];
  }
  int k = 0;
  if (i != j) {
    const int lo = min(i, j), hi = max(i, j);
    const int64_t src = (b * n + lo) * n + hi;
    float prob[DF_MAX_C];
    if (valid) reverse_probs(logits + src * C, onehot_class(zt + src * C, C), beta[b], abar_s[b], abar_t[b], marg, C, prob);
    else for (int c = 0; c < C; ++c) prob[c] = 1.0f / C;
    k = sample_argmax(prob, expo + src * C, C);
  }
  cls_out[p] = k;
  for (int c = 0; c < C; ++c) onehot_out[p * C + c] = (valid && c == k) ? 1.f : 0.f;
}


// ---- class-id wire format -> dense one-hot batch --------------------------------------------------------
// atom ids [bs,n] (int8, < 0 = padding) and bond ids [bs,n,n] (int8: 0 none, 1.. bond types) expand to the dense
// convention the losses expect: X rows of padded nodes all-zero; E diagonal all-zero; EVERY off-diagonal pair
// without a bond - pairs that touch padding included - is class 0; nothing is masked.
__global__ void dense_nodes_kernel(const int8_t* __restrict__ atoms, float* __restrict__ X, uint8_t* __restrict__ mask,
                                   int64_t nodes, int cx) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nodes) return;
  const int a = atoms[r];
  mask[r] = a >= 0 ? 1 : 0;
  for (int c = 0; c < cx; ++c) X[r * cx + c] = (c == a) ? 1.f : 0.f;
}
__global__ void dense_pairs_kernel(const int8_t* __restrict__ bonds, float* __restrict__ E, int64_t pairs, int n, int ce) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= pairs) return;
  const unsigned ij = (unsigned)(r % ((int64_t)n * n));
  const bool diag = (ij / (unsigned)n) == (ij % (unsigned)n);
  int b = bonds[r];
  if (b < 0 || b >= ce) b = 0;
  for (int c = 0; c < ce; ++c) E[r * ce + c] = (!diag && c == b) ? 1.f : 0.f;
}

extern "C" {

// Dense clean graphs from class ids: X [bs,n,cx], E [bs,n,n,ce] one-hot floats, mask [bs,n] (u8).
int mbvae_dense_from_ids(const int8_t* atoms, const int8_t* bonds, float* X, float* E, uint8_t* mask, int64_t bs, int64_t n,
                         int64_t cx, int64_t ce, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!atoms || !bonds || !X || !E || !mask) return MBVAE_EINVAL;
  if (bs <= 0 || n <= 0 || cx <= 0 || ce <= 0 || cx > 127 || ce > 127 || n > 46340) return MBVAE_EUNSUPPORTED;
  const int64_t nodes = bs * n, pairs = bs * n * n;
  hipLaunchKernelGGL(dense_nodes_kernel, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, stream, atoms, X, mask, nodes, (int)cx);
  hipLaunchKernelGGL(dense_pairs_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, bonds, E, pairs, (int)n, (int)ce);
  return mbvae_launch_status();
}

// Forward noising of one batch: X0 [bs,n,cx], E0 [bs,n,n,ce] one-hot floats, abar_t [bs], marginals, Exp(1) draws.
// Outputs int64 class ids (E symmetric, zero diagonal) and masked one-hot floats.
int mbvae_apply_noise(const float* X0, const float* E0, const uint8_t* mask, const float* abar_t, const float* mx,
                      const float* me, const float* expX, const float* expE, int64_t* Xc, int64_t* Ec, float* Xt,
                      float* Et, int64_t bs, int64_t n, int64_t cx, int64_t ce, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!X0 || !E0 || !mask || !abar_t || !mx || !me || !expX || !expE || !Xc || !Ec || !Xt || !Et) return MBVAE_EINVAL;
  if (bs <= 0 || n <= 0 || cx <= 0 || ce <= 0 || cx > DF_MAX_C || ce > DF_MAX_C || n > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int64_t nodes = bs * n, pairs = bs * n * n;
  hipLaunchKernelGGL(noise_nodes_kernel, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, stream, X0, mask, abar_t, mx,
                     expX, Xc, Xt, nodes, (int)n, (int)cx);
  hipLaunchKernelGGL(noise_pairs_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, E0, mask, abar_t, me,
                     expE, Ec, Et, pairs, (int)n, (int)ce);
  return mbvae_launch_status();
}

// One reverse-diffusion sampling step from network logits.  probX/probE (optional) receive p(z_s|z_t).
int mbvae_reverse_sample(const float* logitsX, const float* logitsE, const float* Xt, const float* Et, const uint8_t* mask,
                         const float* beta_t, const float* abar_s, const float* abar_t, const float* mx, const float* me,
                         const float* expX, const float* expE, int64_t* Xc, int64_t* Ec, float* Xs, float* Es,
                         float* probX, float* probE, int64_t bs, int64_t n, int64_t cx, int64_t ce, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!logitsX || !logitsE || !Xt || !Et || !mask || !beta_t || !abar_s || !abar_t || !mx || !me || !expX || !expE ||
      !Xc || !Ec || !Xs || !Es)
    return MBVAE_EINVAL;
  if (bs <= 0 || n <= 0 || cx <= 0 || ce <= 0 || cx > DF_MAX_C || ce > DF_MAX_C || n > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int64_t nodes = bs * n, pairs = bs * n * n;
  hipLaunchKernelGGL(reverse_nodes_kernel, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, stream, logitsX, Xt, mask,
                     beta_t, abar_s, abar_t, mx, expX, Xc, Xs, probX, nodes, (int)n, (int)cx);
  hipLaunchKernelGGL(reverse_pairs_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, logitsE, Et, mask,
                     beta_t, abar_s, abar_t, me, expE, Ec, Es, probE, pairs, (int)n, (int)ce);
  return mbvae_launch_status();
}

}  // extern "C"
