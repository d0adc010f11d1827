// Shared device/host helpers for the mbvae HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MBVAE_OK 0
#define MBVAE_EINVAL (-1)
#define MBVAE_EUNSUPPORTED (-2)
#define MBVAE_ELAUNCH (-3)

#define MBVAE_MASK_NONE 0
#define MBVAE_MASK_NODE 1   // row r is a node (b,i): keep iff mask[r]
#define MBVAE_MASK_EDGE 2   // row r is a pair (b,i,j): keep iff mask[b*n+i] && mask[b*n+j]

static inline int mbvae_launch_status() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return MBVAE_OK;
  fprintf(stderr, "libmbvae_hip: HIP error %d: %s\n", (int)e, hipGetErrorString(e));
  return MBVAE_ELAUNCH;
}

// Counter-based uniform in [0,1): splitmix64 of (seed, element index).  Used for dropout so that
// the backward pass regenerates the keep-mask instead of storing it.
__device__ __forceinline__ float mbvae_u01(uint64_t seed, uint64_t idx) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// keep-scale of one element under dropout p: 0 if dropped, 1/(1-p) if kept.
__device__ __forceinline__ float mbvae_drop_scale(float p, uint64_t seed, uint64_t idx) {
  if (p <= 0.f) return 1.f;
  return mbvae_u01(seed, idx) < p ? 0.f : 1.f / (1.f - p);
}

// 1.0f if row r survives the mask, else 0.0f.  Row counts are < 2^31 (checked by the callers): 32-bit divides.
__device__ __forceinline__ float mbvae_row_keep(const uint8_t* mask, int mode, int64_t r, int n) {
  if (mode == MBVAE_MASK_NONE) return 1.f;
  if (mode == MBVAE_MASK_NODE) return mask[r] ? 1.f : 0.f;
  const unsigned un = (unsigned)n, ur = (unsigned)r;
  const unsigned bi = ur / un;          // b*n + i
  const unsigned j = ur - bi * un;
  const unsigned b = bi / un;
  return (mask[bi] && mask[b * un + j]) ? 1.f : 0.f;
}

// Raw mask values deciding whether row r survives (both must be non-zero).  Lets a kernel issue the loads early and
// combine them later.  The outputs are full 32-bit registers and the index selection is branch-free on purpose: byte
// variables get packed by the compiler, which consumes (and therefore waits for) each load right where it is issued
// and serialises every other load in flight behind it.
__device__ __forceinline__ void mbvae_row_mask_bytes(const uint8_t* mask, int mode, int64_t r, int n, const void* safe,
                                                     unsigned& a, unsigned& b) {
  // `safe` is any readable address: with no mask the two loads still execute (from there) and are overridden, which
  // keeps this function free of control flow.
  const bool none = mode == MBVAE_MASK_NONE;
  const uint8_t* mp = none ? reinterpret_cast<const uint8_t*>(safe) : mask;
  const unsigned un = none ? 1u : (unsigned)n, ur = (unsigned)r;
  const unsigned bi = ur / un, j = ur - bi * un, bb = bi / un;
  const unsigned ia = none ? 0u : (mode == MBVAE_MASK_EDGE ? bi : ur);
  const unsigned ib = none ? 0u : (mode == MBVAE_MASK_EDGE ? bb * un + j : ur);
  a = mp[ia] | (unsigned)none; b = mp[ib] | (unsigned)none;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
