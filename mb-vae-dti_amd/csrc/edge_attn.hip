// Edge-feature attention of the node/edge block: the n x n x dx streaming core.
//
//   Y[b,i,j,c]    = Q[b,i,c] * K[b,j,c] / sqrt(df) * (E1[b,i,j,c] + 1) + E2[b,i,j,c]
//   P[b,i,j,c]    = ye_add[b,c] + (ye_mul[b,c] + 1) * Y[b,i,j,c]                (edge update, pre e_out)
//   A[b,i,:,c]    = softmax over valid keys j of Y[b,i,:,c]     (one softmax PER CHANNEL c)
//   wV[b,i,c]     = sum_j A[b,i,j,c] * V[b,j,c]
//
// Every channel has its own softmax, so this is an HBM-bound streaming kernel, not a GEMM:
// forward reads E1,E2 and writes P (3 x bs*n*n*dx floats); backward reads E1,E2,dP and writes
// dE1,dE2 (5 x).  Layout: channels contiguous; a lane owns 4 consecutive channels (16-byte
// accesses), a wave covers 256 channels = one 1 KiB row segment per instruction.
//
// Forward: one wave per query row (b,i), online (running max / sum) softmax over j; also stores
// lse = max + log(sum) per (b,i,c) so that backward can rebuild A without a second reduction.
// Backward: every (i,j,c) element is independent once lse and wV are known.  One 512-thread
// workgroup per (graph, 16 keys): a wave owns two keys (dK/dV accumulate in registers over the
// i loop), the per-query sums dQ[i] are combined across the 8 waves with LDS float adds and across
// the key groups with global float adds.
#include "common.h"


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Streaming accesses (every edge element is touched exactly once per kernel): the non-temporal hint keeps them from
// displacing the node rows (Q, K, V, lse, ...) that all waves of a graph re-read out of L1/L2.
typedef float ea_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float* p) {
#ifdef MBVAE_EA_NO_NT
  return ld4(p);
#else
  const ea_v4f v = __builtin_nontemporal_load(reinterpret_cast<const ea_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st4s(float* p, float4 v) {
#ifdef MBVAE_EA_NO_NT
  st4(p, v);
#else
  ea_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<ea_v4f*>(p));
#endif
}

#define F4_OP(out, expr)                                   \
  {                                                        \
    { const int _q = 0; (out).x = (expr); }                \
    { const int _q = 1; (out).y = (expr); }                \
    { const int _q = 2; (out).z = (expr); }                \
    { const int _q = 3; (out).w = (expr); }                \
  }
__device__ __forceinline__ float f4get(const float4& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

// grid: (ceil(n/ROWS), ceil(dx/256), bs), block 64*ROWS: wave = query row, lane = 4 channels.
// STAGE: the graph's K and V rows (n <= 64) are copied to LDS once per block; every query row re-reads all of them,
// and from global memory they keep getting evicted by the edge streams and fetched again.
template <int ROWS, bool STAGE>
__global__ __launch_bounds__(64 * ROWS) void edge_attn_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ E1, const float* __restrict__ E2,
    const float* __restrict__ ye_add, const float* __restrict__ ye_mul, const uint8_t* __restrict__ mask,
    float* __restrict__ P, float* __restrict__ wV, float* __restrict__ lse, int n, int dx, float scale,
    int ldq, int lde) {
  extern __shared__ __align__(16) float kv_rows[];          // STAGE: [2][n][256]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row addressing on the scalar unit
  const int b = blockIdx.z, i = blockIdx.x * ROWS + wave;
  const int c = blockIdx.y * 256 + lane * 4;
  const int64_t node = (int64_t)b * n;
  if (STAGE) {
    if (c < dx)
      for (int j = wave; j < n; j += ROWS) {
        st4(&kv_rows[j * 256 + lane * 4], ld4(K + (node + j) * ldq + c));
        st4(&kv_rows[(n + j) * 256 + lane * 4], ld4(V + (node + j) * ldq + c));
      }
    __syncthreads();
  }
  if (i >= n || c >= dx) return;
  const float4 q = ld4(Q + (node + i) * ldq + c);
  const float4 ya = ld4(ye_add + (int64_t)b * dx + c);
  float4 ym = ld4(ye_mul + (int64_t)b * dx + c);
  ym.x += 1.f; ym.y += 1.f; ym.z += 1.f; ym.w += 1.f;
  float4 qs; F4_OP(qs, f4get(q, _q) * scale);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  float4 l = make_float4(0.f, 0.f, 0.f, 0.f), acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t rowbase = ((node + i) * n) * (int64_t)dx + c;
  const int64_t erow = ((node + i) * n) * (int64_t)lde + c;
#pragma unroll 4
  for (int j = 0; j < n; ++j) {
    const float4 e1 = ld4s(E1 + erow + (int64_t)j * lde);
    const float4 e2 = ld4s(E2 + erow + (int64_t)j * lde);
    const float4 k = STAGE ? ld4(&kv_rows[j * 256 + lane * 4]) : ld4(K + (node + j) * ldq + c);
    const float4 v = STAGE ? ld4(&kv_rows[(n + j) * 256 + lane * 4]) : ld4(V + (node + j) * ldq + c);
    float4 y; F4_OP(y, f4get(qs, _q) * f4get(k, _q) * (f4get(e1, _q) + 1.f) + f4get(e2, _q));
    float4 p; F4_OP(p, f4get(ya, _q) + f4get(ym, _q) * f4get(y, _q));
    st4s(P + rowbase + (int64_t)j * dx, p);
    if (mask[node + j]) {
      float4 mn; F4_OP(mn, fmaxf(f4get(m, _q), f4get(y, _q)));
      float4 corr; F4_OP(corr, __expf(f4get(m, _q) - f4get(mn, _q)));
      float4 w; F4_OP(w, __expf(f4get(y, _q) - f4get(mn, _q)));
      F4_OP(l, f4get(l, _q) * f4get(corr, _q) + f4get(w, _q));
      F4_OP(acc, f4get(acc, _q) * f4get(corr, _q) + f4get(w, _q) * f4get(v, _q));
      m = mn;
    }
  }
  float4 o, s;
  F4_OP(o, f4get(l, _q) > 0.f ? f4get(acc, _q) / f4get(l, _q) : 0.f);
  F4_OP(s, f4get(l, _q) > 0.f ? f4get(m, _q) + __logf(f4get(l, _q)) : INFINITY);
  st4(wV + (node + i) * dx + c, o);
  st4(lse + (node + i) * dx + c, s);
}

// grid: (ceil(dx/256), ceil(n/KPB), bs), block 64*NW: wave w owns keys j = KPB*blockIdx.y + w + NW*jj (jj < 2, KPB =
// 2*NW keys per block), lane = 4 channels.  dQ/dye_* are accumulated with global float adds across the key groups
// (zeroed by the host).  Block shape (measured at bs=256, n=64, dx=256): 8 waves x 32 query rows per LDS pass keeps
// three independent workgroups on a CU and runs 14 % faster than one 16-wave workgroup with 64 rows; requesting the
// next query row's streams ahead of the arithmetic (two rows in flight per wave) did not help at either shape.
#define EA_JPW 2
template <int NW, int ROWS>
__global__ __launch_bounds__(64 * NW) void edge_attn_bwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ E1, const float* __restrict__ E2,
    const float* __restrict__ ye_mul, const uint8_t* __restrict__ mask,
    const float* __restrict__ wV, const float* __restrict__ lse,
    const float* __restrict__ dP, const float* __restrict__ dwV,
    float* __restrict__ dE1, float* __restrict__ dE2, float* __restrict__ dQ, float* __restrict__ dK,
    float* __restrict__ dV, float* __restrict__ dye_add, float* __restrict__ dye_mul, int n, int dx, float scale,
    int ldq, int lde) {
  constexpr int KPB = NW * EA_JPW, NT = 64 * NW;
  __shared__ float dq_acc[ROWS * 256];
  __shared__ float dya_acc[256], dym_acc[256];
  // V rows of this block's keys: re-read for every query row; from global memory they get evicted from L1/L2 by
  // the edge streams in between and come back from HBM (measured: +1 GB of fetches per launch)
  __shared__ __align__(16) float v_rows[KPB * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row addressing on the scalar unit
  const int b = blockIdx.z;
  const int c = blockIdx.x * 256 + lane * 4;
  const int cbase = blockIdx.x * 256;
  const bool active = c < dx;
  const int cc = active ? c : 0;                   // inactive lanes load from channel 0 and store nothing
  const int64_t node = (int64_t)b * n;
  if (threadIdx.x < 256) { dya_acc[threadIdx.x] = 0.f; dym_acc[threadIdx.x] = 0.f; }

  float4 ks[EA_JPW], dk[EA_JPW], dv[EA_JPW];
  bool ok[EA_JPW], inr[EA_JPW];
  int jc[EA_JPW];
  float4 ym = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 dya = make_float4(0.f, 0.f, 0.f, 0.f), dym = dya;
#pragma unroll
  for (int jj = 0; jj < EA_JPW; ++jj) {
    const int j = blockIdx.y * KPB + wave + NW * jj;
    inr[jj] = j < n;
    jc[jj] = min(j, n - 1);
    dk[jj] = make_float4(0.f, 0.f, 0.f, 0.f); dv[jj] = dk[jj];
    const float4 k = ld4(K + (node + jc[jj]) * ldq + cc);
    st4(&v_rows[(wave + NW * jj) * 256 + lane * 4], ld4(V + (node + jc[jj]) * ldq + cc));   // read back by this lane only
    F4_OP(ks[jj], f4get(k, _q) * scale);
    ok[jj] = inr[jj] && mask[node + jc[jj]] != 0;
  }
  {
    const float4 t = ld4(ye_mul + (int64_t)b * dx + cc);
    ym.x += t.x; ym.y += t.y; ym.z += t.z; ym.w += t.w;
  }

  float4 e1[1][EA_JPW], e2[1][EA_JPW], dp[1][EA_JPW], q4[1], ls4[1], wv4[1], dwv4[1];
#define EA_LOAD(i_, S)                                                                     \
  {                                                                                        \
    const int64_t nrow = node + min((i_), n - 1);                                          \
    q4[S] = ld4(Q + nrow * ldq + cc);                                                      \
    ls4[S] = ld4(lse + nrow * dx + cc);                                                    \
    wv4[S] = ld4(wV + nrow * dx + cc);                                                     \
    dwv4[S] = ld4(dwV + nrow * dx + cc);                                                   \
    _Pragma("unroll") for (int jj = 0; jj < EA_JPW; ++jj) {                                \
      e1[S][jj] = ld4s(E1 + (nrow * n + jc[jj]) * (int64_t)lde + cc);                      \
      e2[S][jj] = ld4s(E2 + (nrow * n + jc[jj]) * (int64_t)lde + cc);                      \
      dp[S][jj] = ld4s(dP + (nrow * n + jc[jj]) * (int64_t)dx + cc);                       \
    }                                                                                      \
    asm volatile("" : : : "memory");         /* issue the loads here */                    \
  }
#define EA_ROW(i_, ii_, S)                                                                 \
  {                                                                                        \
    const float4 q = q4[S], ls = ls4[S], wv = wv4[S], dwv = dwv4[S];                       \
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);                                           \
    const int64_t nrow = node + (i_);                                                      \
    _Pragma("unroll") for (int jj = 0; jj < EA_JPW; ++jj) {                                \
      const float4 x1 = e1[S][jj], x2 = e2[S][jj], dpp = dp[S][jj];                        \
      const float4 vj = ld4(&v_rows[(wave + NW * jj) * 256 + lane * 4]);                   \
      float4 qk; F4_OP(qk, f4get(q, _q) * f4get(ks[jj], _q));                              \
      float4 y; F4_OP(y, f4get(qk, _q) * (f4get(x1, _q) + 1.f) + f4get(x2, _q));           \
      float4 a; F4_OP(a, ok[jj] ? __expf(f4get(y, _q) - f4get(ls, _q)) : 0.f);             \
      float4 dy; F4_OP(dy, f4get(dpp, _q) * f4get(ym, _q) +                                \
                             f4get(a, _q) * f4get(dwv, _q) * (f4get(vj, _q) - f4get(wv, _q))); \
      if (inr[jj] && active) {                                                             \
        float4 t; F4_OP(t, f4get(dy, _q) * f4get(qk, _q));                                 \
        st4s(dE1 + (nrow * n + jc[jj]) * (int64_t)lde + c, t);                             \
        st4s(dE2 + (nrow * n + jc[jj]) * (int64_t)lde + c, dy);                            \
        float4 g; F4_OP(g, f4get(dy, _q) * (f4get(x1, _q) + 1.f));                         \
        F4_OP(dq, f4get(dq, _q) + f4get(g, _q) * f4get(ks[jj], _q));                       \
        F4_OP(dk[jj], f4get(dk[jj], _q) + f4get(g, _q) * f4get(q, _q) * scale);            \
        F4_OP(dv[jj], f4get(dv[jj], _q) + f4get(a, _q) * f4get(dwv, _q));                  \
        F4_OP(dya, f4get(dya, _q) + f4get(dpp, _q));                                       \
        F4_OP(dym, f4get(dym, _q) + f4get(dpp, _q) * f4get(y, _q));                        \
      }                                                                                    \
    }                                                                                      \
    float* slot = &dq_acc[(ii_) * 256 + lane * 4];                                         \
    atomicAdd(slot + 0, dq.x); atomicAdd(slot + 1, dq.y); atomicAdd(slot + 2, dq.z); atomicAdd(slot + 3, dq.w); \
  }

  for (int i0 = 0; i0 < n; i0 += ROWS) {
    const int rows = min(ROWS, n - i0);
    for (int t = threadIdx.x; t < rows * 256; t += NT) dq_acc[t] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int ii = 0; ii < rows; ++ii) {
      EA_LOAD(i0 + ii, 0)
      EA_ROW(i0 + ii, ii, 0)
    }
    __syncthreads();
    for (int t = threadIdx.x; t < rows * 256; t += NT) {
      const int ii = t >> 8, ch = t & 255;
      if (cbase + ch < dx) atomicAdd(&dQ[(node + i0 + ii) * ldq + cbase + ch], dq_acc[t]);
    }
    __syncthreads();
  }
#undef EA_LOAD
#undef EA_ROW
  if (active) {
#pragma unroll
    for (int jj = 0; jj < EA_JPW; ++jj) {
      const int j = blockIdx.y * KPB + wave + NW * jj;
      if (j < n) { st4(dK + (node + j) * ldq + c, dk[jj]); st4(dV + (node + j) * ldq + c, dv[jj]); }
    }
    atomicAdd(&dya_acc[lane * 4 + 0], dya.x); atomicAdd(&dya_acc[lane * 4 + 1], dya.y);
    atomicAdd(&dya_acc[lane * 4 + 2], dya.z); atomicAdd(&dya_acc[lane * 4 + 3], dya.w);
    atomicAdd(&dym_acc[lane * 4 + 0], dym.x); atomicAdd(&dym_acc[lane * 4 + 1], dym.y);
    atomicAdd(&dym_acc[lane * 4 + 2], dym.z); atomicAdd(&dym_acc[lane * 4 + 3], dym.w);
  }
  __syncthreads();
  if (threadIdx.x < 256 && cbase + threadIdx.x < dx) {
    atomicAdd(&dye_add[(int64_t)b * dx + cbase + threadIdx.x], dya_acc[threadIdx.x]);
    atomicAdd(&dye_mul[(int64_t)b * dx + cbase + threadIdx.x], dym_acc[threadIdx.x]);
  }
}

static bool ea_aligned(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" {

// Q,K,V [bs,n,dx] rows ldq apart (masked), E1,E2 [bs,n,n,dx] rows lde apart (masked), ye_add,ye_mul [bs,dx],
// mask [bs,n] (u8) -> P [bs,n,n,dx], wV [bs,n,dx], lse [bs,n,dx] (dense).   scale = 1/sqrt(df).
// The row strides let Q|K|V and E1|E2 be column blocks of one projection output ([bs,n,3dx], [bs,n,n,2dx]).
int mbvae_edge_attn_fwd(const float* Q, const float* K, const float* V, int64_t ldq, const float* E1, const float* E2,
                        int64_t lde, const float* ye_add, const float* ye_mul, const uint8_t* mask,
                        float* P, float* wV, float* lse, int64_t bs, int64_t n, int64_t dx, float scale,
                        hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (bs <= 0 || n <= 0 || dx <= 0 || (dx & 3)) return MBVAE_EUNSUPPORTED;
  if (ldq < dx || lde < dx || (ldq & 3) || (lde & 3) || ldq > 0x7fffffff || lde > 0x7fffffff) return MBVAE_EINVAL;
  if (bs > 65535 || n > 0x7fffffff / 16) return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {Q, K, V, E1, E2, ye_add, ye_mul, P, wV, lse};
  for (const void* p : ptrs) if (!p || !ea_aligned(p)) return MBVAE_EINVAL;
  if (!mask) return MBVAE_EINVAL;
  // (Staging the graph's K,V rows in LDS - template flag STAGE, 16 query rows per block - brings the measured reads
  // down to the algorithmic bytes but costs occupancy: 128 KB of LDS leave one block per CU, and the kernel ran 5 %
  // slower.  The direct version below re-reads K,V through L1/L2 and fetches ~1.18x of its algorithmic bytes.)
  dim3 grid((unsigned)((n + 3) / 4), (unsigned)((dx + 255) / 256), (unsigned)bs);
  hipLaunchKernelGGL((edge_attn_fwd_kernel<4, false>), grid, dim3(256), 0, stream, Q, K, V, E1, E2, ye_add, ye_mul, mask,
                     P, wV, lse, (int)n, (int)dx, scale, (int)ldq, (int)lde);
  return mbvae_launch_status();
}

// Gradients of the above given dP [bs,n,n,dx] and dwV [bs,n,dx] (dense).
// Writes dE1,dE2 (rows lde apart, like E1,E2), dQ,dK,dV (rows ldq apart, like Q,K,V), dye_add,dye_mul [bs,dx].
int mbvae_edge_attn_bwd(const float* Q, const float* K, const float* V, int64_t ldq, const float* E1, const float* E2,
                        int64_t lde, const float* ye_mul, const uint8_t* mask, const float* wV, const float* lse,
                        const float* dP, const float* dwV, float* dE1, float* dE2, float* dQ, float* dK,
                        float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t dx,
                        float scale, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (bs <= 0 || n <= 0 || dx <= 0 || (dx & 3)) return MBVAE_EUNSUPPORTED;
  if (ldq < dx || lde < dx || (ldq & 3) || (lde & 3) || ldq > 0x7fffffff || lde > 0x7fffffff) return MBVAE_EINVAL;
  if (bs > 65535 || n > 16 * 65535) return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {Q, K, V, E1, E2, ye_mul, wV, lse, dP, dwV, dE1, dE2, dQ, dK, dV, dye_add, dye_mul};
  for (const void* p : ptrs) if (!p || !ea_aligned(p)) return MBVAE_EINVAL;
  if (!mask) return MBVAE_EINVAL;
  // dQ is accumulated with float adds; with ldq > dx the gaps between its rows (dK/dV blocks) are zeroed too and
  // then overwritten by the kernel's plain stores
  if (hipMemsetAsync(dQ, 0, sizeof(float) * ((bs * n - 1) * ldq + dx), stream) != hipSuccess) return MBVAE_ELAUNCH;
  if (hipMemsetAsync(dye_add, 0, sizeof(float) * bs * dx, stream) != hipSuccess) return MBVAE_ELAUNCH;
  if (hipMemsetAsync(dye_mul, 0, sizeof(float) * bs * dx, stream) != hipSuccess) return MBVAE_ELAUNCH;
  constexpr int NW = 8, ROWS = 32;
  dim3 grid((unsigned)((dx + 255) / 256), (unsigned)((n + 2 * NW - 1) / (2 * NW)), (unsigned)bs);
  hipLaunchKernelGGL((edge_attn_bwd_kernel<NW, ROWS>), grid, dim3(64 * NW), 0, stream, Q, K, V, E1, E2, ye_mul, mask,
                     wV, lse, dP, dwV, dE1, dE2, dQ, dK, dV, dye_add, dye_mul, (int)n, (int)dx, scale, (int)ldq,
                     (int)lde);
  return mbvae_launch_status();
}

}  // extern "C"
