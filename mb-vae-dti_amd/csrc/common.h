// Shared device/host helpers for the mbvae HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

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

// Counter-based uniform in [0,1): a 32-bit integer hash of (seed, element index).  Used for dropout so that the
// backward pass regenerates the keep-mask instead of storing it.  32-bit on purpose: the epilogues that call this
// once per output element sit next to the MFMA stream, where 64-bit integer multiplies would make the vector ALU the
// bottleneck.  The seed and the upper index half (wave-uniform in practice) fold into one word on the scalar unit.
// Process-wide tuning switches.  Each starts from its environment variable (MBVAE_<NAME>, read ONCE when the first
// entry point runs; the per-call paths never call getenv) and can be changed afterwards with mbvae_set_option():
// they are plain atomics read with relaxed loads, so changing one while kernels are being issued from another
// thread merely takes effect at some later call.  They select between equivalent HIP routes (kernel A/B tests,
// parity tests of the fallback kernels), never a CPU path.
enum MbvaeOption {
  MBVAE_OPT_NO_TSGEMM, MBVAE_OPT_NO_TSGEMM2, MBVAE_OPT_TSGEMM_N64, MBVAE_OPT_NO_TSWGRAD, MBVAE_OPT_NO_TSWGRAD2,
  MBVAE_OPT_NO_EDGE_FUSED, MBVAE_OPT_NO_EDGE_FFN, MBVAE_OPT_NO_SMALLM, MBVAE_OPT_NO_EDGE_TRAIN_FUSED,
  MBVAE_OPT_EA_KEYS,       // keys per wave of the edge-attention backward: 8, anything else = 16
  MBVAE_OPT_NO_EA_PREFETCH,   // edge-attention backward: do NOT request the next query row's operands a row ahead
  MBVAE_OPT_GEMM_TILE,     // -1: automatic; 0 = 64x64, 1 = 128x64, 2 = 128x128
  MBVAE_OPT_NO_SPLIT_MFMA, // edge-path products on the native fp32 MFMA instead of six bf16 MFMAs of a three-way split
  MBVAE_OPT_NO_E12_RECOMPUTE,   // training: store the e_mul|e_add projection output in the forward instead of recomputing it in the backward
  MBVAE_OPT_EFB_SYNC,      // recomputing edge backward: waves meet every (value) query rows, a power of two; 0 = default (8)
  MBVAE_OPT_EF_STAGGER,    // fused edge kernels, training forms: 0 = default (forward 1, backward in phase); 1 = waves 0-3 / 4-7 meet half a tile apart; 2 = even / odd waves; 3 = all in phase
  MBVAE_OPT_NO_COLGROUPS,  // node-path Linear shapes on the generic LDS-tiled kernel instead of column groups of the tall-skinny one
  MBVAE_OPT_NO_EF_COOP,    // fused edge forward: every wave fetches and splits the whole operand tile itself (no shared planes in LDS)
  MBVAE_OPT_NO_EFB_COOP,   // the same choice for the recomputing backward
  MBVAE_OPT_NO_FFN_HANDOVER,   // feed-forward blocks: the ReLU gate stays inside the first Linear's gradient kernels (read by the host side)
  MBVAE_OPT_NO_FILM,       // node update: the torch expression instead of the FiLM kernel pair (read by the host side)
  MBVAE_OPT_NO_BWD_BOTH,   // e_mul|e_add projection: separate input- and weight-gradient kernels instead of the single pass over dY
  MBVAE_OPT_NO_LN_EPILOGUE, // 64-wide edge Linears feeding a residual LayerNorm: separate Linear and LayerNorm kernels instead of the fused epilogue
  MBVAE_OPT_NO_THIN,        // Linear layers with <= 8 input or output features over the pair rows: the tiled kernels instead of the streaming ones
  MBVAE_OPT_NO_EDGE_FFN_TRAIN,   // training: the edge feed-forward block as two Linear kernels instead of the one-pass kernel (read by the host side too)
  MBVAE_OPT_NO_GEMM_SPLIT,  // generic LDS-tiled kernel (node / global-feature / encoder Linears): native fp32 matrix instructions instead of the bf16 pipeline
  MBVAE_OPT_COUNT
};
inline const char* const* mbvae_option_names() {
  static const char* const names[MBVAE_OPT_COUNT] = {"NO_TSGEMM", "NO_TSGEMM2", "TSGEMM_N64", "NO_TSWGRAD", "NO_TSWGRAD2",
                                                     "NO_EDGE_FUSED", "NO_EDGE_FFN", "NO_SMALLM", "NO_EDGE_TRAIN_FUSED",
                                                     "EA_KEYS", "NO_EA_PREFETCH", "GEMM_TILE", "NO_SPLIT_MFMA",
                                                     "NO_E12_RECOMPUTE", "EFB_SYNC", "EF_STAGGER", "NO_COLGROUPS", "NO_EF_COOP", "NO_EFB_COOP", "NO_FFN_HANDOVER", "NO_FILM", "NO_BWD_BOTH",
                                                     "NO_LN_EPILOGUE", "NO_THIN", "NO_EDGE_FFN_TRAIN", "NO_GEMM_SPLIT"};
  return names;
}
struct MbvaeOptions {
  std::atomic<int> v[MBVAE_OPT_COUNT];
  MbvaeOptions() {
    for (int i = 0; i < MBVAE_OPT_COUNT; ++i) {
      char name[64];
      snprintf(name, sizeof(name), "MBVAE_%s", mbvae_option_names()[i]);
      const char* e = getenv(name);
      int val = i == MBVAE_OPT_GEMM_TILE ? -1 : 0;
      // "MBVAE_X=<number>" sets the number (so "=0" switches a flag OFF); set but empty or non-numeric means 1
      if (e) val = (*e == '-' || (*e >= '0' && *e <= '9')) ? atoi(e) : 1;
      v[i].store(val, std::memory_order_relaxed);
    }
  }
};
inline MbvaeOptions& mbvae_options() {
  static MbvaeOptions o;
  return o;
}
inline int mbvae_opt(int id) { return mbvae_options().v[id].load(std::memory_order_relaxed); }

// Compute units of the current device (queried once per device, then a table lookup).
inline int mbvae_num_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::atomic<int>& slot = cus[dev & 63];
  int c = slot.load(std::memory_order_relaxed);
  if (c == 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    slot.store(c, std::memory_order_relaxed);
  }
  return c;
}

// Raises `kernel`'s dynamic-LDS limit once per device (the attribute is a cap, not an allocation) to 152 KB: the
// 160 KB a workgroup can have on gfx950 minus room for the kernels' small static __shared__ variables, which count
// against the same total.  One static flag word per call site; evaluates to false if the runtime call fails.
#define MBVAE_LDS_LIMIT 155648
#define MBVAE_ALLOW_BIG_LDS(kernel)                                                                                   \
  ([&]() -> bool {                                                                                                    \
    static std::atomic<unsigned long long> done_{0ull};                                                               \
    int dev_ = 0;                                                                                                     \
    if (hipGetDevice(&dev_) != hipSuccess) return false;                                                              \
    const unsigned long long bit_ = 1ull << (dev_ & 63);                                                              \
    if (done_.load(std::memory_order_acquire) & bit_) return true;                                                    \
    if (hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MBVAE_LDS_LIMIT) !=    \
        hipSuccess)                                                                                                   \
      return false;                                                                                                   \
    done_.fetch_or(bit_, std::memory_order_release);                                                                  \
    return true;                                                                                                      \
  }())

__device__ __forceinline__ float mbvae_u01(uint64_t seed, uint64_t idx) {
  const uint32_t s = (uint32_t)seed * 0x9E3779B9u ^ (uint32_t)(seed >> 32) * 0x85EBCA77u ^ (uint32_t)(idx >> 32) * 0xC2B2AE3Du;
  uint32_t x = (uint32_t)idx ^ s;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16; x += s;
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 13;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
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
  // `safe` is any readable address: with no mask the two loads still execute (from there) and are ignored by
  // mbvae_mask_keep(), which keeps this function free of control flow.
  const bool none = mode == MBVAE_MASK_NONE;
  const uint8_t* mp = none ? reinterpret_cast<const uint8_t*>(safe) : mask;
  const unsigned un = none ? 1u : (unsigned)n, ur = (unsigned)r;
  const unsigned bi = ur / un, j = ur - bi * un, bb = bi / un;
  const unsigned ia = none ? 0u : (mode == MBVAE_MASK_EDGE ? bi : ur);
  const unsigned ib = none ? 0u : (mode == MBVAE_MASK_EDGE ? bb * un + j : ur);
  a = mp[ia]; b = mp[ib];      // raw: combine with mbvae_mask_keep() where the flag is needed, not here
}
__device__ __forceinline__ bool mbvae_mask_keep(int mode, unsigned a, unsigned b) {
  return mode == MBVAE_MASK_NONE || (a != 0u && b != 0u);
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

// ------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipeline.  gfx950 runs v_mfma_f32_32x32x2_f32 at 64 flop/clk per SIMD and
// v_mfma_f32_32x32x16_bf16 at 1024: an fp32 value cut by TRUNCATION into three bf16 words (hi + mid + lo, 8 mantissa
// bits each, so the three sum to the value exactly) lets a product of two fp32 operands be formed from six bf16
// MFMAs - hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the three dropped terms are below 2^-24 of the product - with
// fp32 accumulation, at up to 1.8x the native fp32 rate (measured: 269 vs 148 TFLOP/s, error against a double
// product 8.5e-7 vs 6.1e-7 of the result scale at K = 512).  Non-finite operands are not preserved (inf splits into
// inf + nan); every caller selects masked rows to zero BEFORE splitting.
typedef float mbvae_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mbvae_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned mbvae_u32x4 __attribute__((ext_vector_type(4)));
struct MbvaeSplit3 { mbvae_bf16x8 hi, mid, lo; };
// eight values -> the three planes in the bf16 MFMA's operand order (element i = reduction index i of the lane's octet)
__device__ __forceinline__ MbvaeSplit3 mbvae_split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
  const float x[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
  mbvae_u32x4 h, m, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = x[2 * p], a1 = x[2 * p + 1];
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    h[p] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);          // upper halves of (a1, a0) side by side
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);   // exact
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m[p] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);   // exact
    l[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  MbvaeSplit3 s;
  s.hi = __builtin_bit_cast(mbvae_bf16x8, h);
  s.mid = __builtin_bit_cast(mbvae_bf16x8, m);
  s.lo = __builtin_bit_cast(mbvae_bf16x8, l);
  return s;
}
// fp32 -> bf16 bits, round to nearest even (NaN stays a quiet NaN)
__device__ __forceinline__ unsigned short mbvae_f2bf(float v) {
  const unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
#define MBVAE_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
