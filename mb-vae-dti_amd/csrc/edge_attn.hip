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
// Backward: every (i,j,c) element is independent once lse and wV are known; see edge_attn_bwd_kernel.
#include "common.h"
#include <algorithm>


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

// Backward.  One wave owns (graph b, a group of KPW consecutive keys, 64 consecutive channels): lane = ONE channel.
// The wave walks over the query rows i; for every row it requests the 3 * KPW stream words (E1, E2, dP of its keys:
// each a 256-byte row segment per instruction, touched once, non-temporal) plus the row's q / lse / wV / dwV, then
// works through its keys.  Everything that is summed lives in the lane's own registers:
//   dK[j], dV[j]  (sum over i)            -> stored once at the end, each key has exactly one owner per channel
//   dQ[i]         (sum over the KPW keys) -> one partial per key group, plain store
//   dye_add/mul   (sum over i and keys)   -> one partial per key group
// and a second tiny kernel adds the key groups' partials in index order.  No LDS, no barrier, no atomic: the result
// is bitwise reproducible from run to run.  The four waves of a workgroup are the four channel quarters of one
// (graph, key group), so together they read whole 1 KiB rows.
// BF16: the edge streams (E1, E2, dP in; dE1, dE2 out) are bf16 in memory - the opt-in reduced-precision storage
// mode; all arithmetic and every sum stay fp32, the outputs are rounded to nearest-even when stored.
template <int KPW, bool PREFETCH, bool BF16 = false>
__global__ __launch_bounds__(256) void edge_attn_bwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const void* __restrict__ E1v, const void* __restrict__ E2v,
    const float* __restrict__ ye_mul, const uint8_t* __restrict__ mask,
    const float* __restrict__ wV, const float* __restrict__ lse,
    const void* __restrict__ dPv, const float* __restrict__ dwV,
    void* __restrict__ dE1v, void* __restrict__ dE2v, float* __restrict__ dK, float* __restrict__ dV,
    float* __restrict__ dq_out, int64_t dq_ld, int64_t dq_group_stride,      // dQ itself (one key group) or the partials
    float* __restrict__ dye_out, int64_t dye_group_stride,                   // [group][b][2][dx] (or dye_add|dye_mul ...)
    float* __restrict__ dye_add, float* __restrict__ dye_mul,                // ... written directly when there is one group
    int b0, int nb, int n, int dx, float scale, int ldq, int lde, int n_kg, int n_cq) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  const int cq = (int)(w % n_cq);
  const int64_t kgb = w / n_cq;
  const int kg = (int)(kgb % n_kg);
  const int bl = (int)(kgb / n_kg);                 // graph within this launch's chunk
  if (bl >= nb) return;
  const int b = b0 + bl;
  const int c = cq * 64 + lane;
  const bool active = c < dx;
  const int cc = active ? c : 0;                   // inactive lanes read channel 0 and store nothing
  const int64_t node = (int64_t)b * n;
  const int j0 = kg * KPW;

  float ks[KPW], vj[KPW], dk[KPW], dv[KPW];
  unsigned okbits = 0u;                            // bit t: key j0 + t exists and is not masked (wave-uniform)
#pragma unroll
  for (int t = 0; t < KPW; ++t) {
    const int j = min(j0 + t, n - 1);
    ks[t] = K[(node + j) * ldq + cc] * scale;
    vj[t] = V[(node + j) * ldq + cc];
    if ((j0 + t < n) && mask[node + j] != 0) okbits |= 1u << t;
    dk[t] = 0.f; dv[t] = 0.f;
  }
  okbits = __builtin_amdgcn_readfirstlane(okbits);
  const float ym = ye_mul[(int64_t)b * dx + cc] + 1.f;
  float dya = 0.f, dym = 0.f;
  const int nk = min(KPW, n - j0);                 // keys of this group that exist (wave-uniform)

  // Stream addressing through buffer descriptors: the row base sits in scalar registers, the key offset is a scalar
  // immediate-like operand and only the lane's channel offset is a vector register.  (Flat 64-bit addresses for the
  // 48 loads and 32 stores of a row would by themselves fill the register file.)
  constexpr int ES = BF16 ? 2 : 4;                 // bytes per stream element
  const char* E1 = (const char*)E1v; const char* E2 = (const char*)E2v; const char* dP = (const char*)dPv;
  char* dE1 = (char*)dE1v; char* dE2 = (char*)dE2v;
  const int voff = cc * ES;
  const unsigned e_key = (unsigned)lde * ES, p_key = (unsigned)dx * ES;        // bytes from one key's row to the next
  auto ld = [&](const __amdgpu_buffer_rsrc_t& rs, int soff) -> float {
    if (BF16) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 2) << 16);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 2));
  };
  auto st = [&](float v, const __amdgpu_buffer_rsrc_t& rs, int soff) {
    if (BF16) __builtin_amdgcn_raw_buffer_store_b16(mbvae_f2bf(v), rs, voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 2);
  };
  struct RowRegs { float x1[KPW], x2[KPW], dpv[KPW], q, ls, wv, dwv; };
  // requests one query row's operands (nothing is waited for here)
  auto load_row = [&](int i, RowRegs& r) {
    const int64_t nrow = node + min(i, n - 1);
    const int64_t eoff = (nrow * n + j0) * (int64_t)lde, poff = (nrow * n + j0) * (int64_t)dx;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(E1 + eoff * ES), 0, nk * (int)e_key, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(E2 + eoff * ES), 0, nk * (int)e_key, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dP + poff * ES), 0, nk * (int)p_key, 0x00020000);
#pragma unroll
    for (int t = 0; t < KPW; ++t) {
      const int tt = min(t, nk - 1);               // clamped: the loads stay unconditional
      r.x1[t] = ld(r1, tt * (int)e_key);
      r.x2[t] = ld(r2, tt * (int)e_key);
      r.dpv[t] = ld(rp, tt * (int)p_key);
    }
    r.q = Q[nrow * ldq + cc]; r.ls = lse[nrow * dx + cc]; r.wv = wV[nrow * dx + cc]; r.dwv = dwV[nrow * dx + cc];
    asm volatile("" : : : "memory");               // the loads are issued HERE
  };
  auto compute_row = [&](int i, const RowRegs& r) {
    const int64_t nrow = node + i;
    const int64_t eoff = (nrow * n + j0) * (int64_t)lde;
    const __amdgpu_buffer_rsrc_t w1 = __builtin_amdgcn_make_buffer_rsrc(dE1 + eoff * ES, 0, nk * (int)e_key, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2 = __builtin_amdgcn_make_buffer_rsrc(dE2 + eoff * ES, 0, nk * (int)e_key, 0x00020000);
    float dq = 0.f;
#pragma unroll
    for (int t = 0; t < KPW; ++t) {
      if (t < nk) {
        const float qk = r.q * ks[t];
        const float y = qk * (r.x1[t] + 1.f) + r.x2[t];
        const float a = ((okbits >> t) & 1u) ? __expf(y - r.ls) : 0.f;
        const float dy = r.dpv[t] * ym + a * r.dwv * (vj[t] - r.wv);
        if (active) {
          st(dy * qk, w1, t * (int)e_key);
          st(dy, w2, t * (int)e_key);
        }
        const float g = dy * (r.x1[t] + 1.f);
        dq += g * ks[t];
        dk[t] += g * r.q * scale;
        dv[t] += a * r.dwv;
        dya += r.dpv[t];
        dym += r.dpv[t] * y;
      }
    }
    if (active) dq_out[(int64_t)kg * dq_group_stride + ((int64_t)bl * n + i) * dq_ld + c] = dq;
#ifdef EA_ROW_BARRIER
    asm volatile("s_barrier" : : : "memory");
#endif
  };
  if (PREFETCH) {
    // the next row's operands are in flight underneath this row's arithmetic (two register sets, loop unrolled by 2)
    RowRegs ra, rb;
    load_row(0, ra);
    for (int i = 0; i < n; i += 2) {
      load_row(i + 1, rb);
      compute_row(i, ra);
      load_row(i + 2, ra);
      if (i + 1 < n) compute_row(i + 1, rb);
    }
  } else {
    RowRegs ra;
    for (int i = 0; i < n; ++i) {
      load_row(i, ra);
      compute_row(i, ra);
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < KPW; ++t)
      if (t < nk) { dK[(node + j0 + t) * ldq + c] = dk[t]; dV[(node + j0 + t) * ldq + c] = dv[t]; }
    if (n_kg == 1) {
      dye_add[(int64_t)b * dx + c] = dya; dye_mul[(int64_t)b * dx + c] = dym;
    } else {
      float* o = dye_out + (int64_t)kg * dye_group_stride + (int64_t)bl * 2 * dx;
      o[c] = dya; o[dx + c] = dym;
    }
  }
}

// Adds the key groups' partial dQ / dye_add / dye_mul in index order (fixed order: reproducible).
__global__ __launch_bounds__(256) void edge_attn_bwd_reduce_kernel(const float* __restrict__ dq_part, int64_t dq_group_stride,
                                                                   const float* __restrict__ dye_part, int64_t dye_group_stride,
                                                                   float* __restrict__ dQ, float* __restrict__ dye_add,
                                                                   float* __restrict__ dye_mul, int b0, int nb, int n, int dx,
                                                                   int ldq, int n_kg) {
  const int64_t rows = (int64_t)nb * n, total = rows * dx, extra = (int64_t)nb * 2 * dx;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total + extra; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < total) {
      float s = 0.f;
      for (int g = 0; g < n_kg; ++g) s += dq_part[(int64_t)g * dq_group_stride + e];
      const int64_t row = e / dx; const int c = (int)(e - row * dx);
      dQ[((int64_t)b0 * n + row) * ldq + c] = s;
    } else {
      const int64_t r = e - total;
      float s = 0.f;
      for (int g = 0; g < n_kg; ++g) s += dye_part[(int64_t)g * dye_group_stride + r];
      const int64_t bl = r / (2 * dx); const int rem = (int)(r - bl * 2 * dx);
      if (rem < dx) dye_add[((int64_t)b0 + bl) * dx + rem] = s;
      else dye_mul[((int64_t)b0 + bl) * dx + rem - dx] = s;
    }
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
// workspace: partial sums of the key groups (16 keys each); mbvae_edge_attn_bwd_workspace(bs, n, dx) bytes let the whole
// batch run in one launch, less makes the call walk over the batch in chunks of graphs (at least one graph's worth is
// needed; none at all when n <= 16).  Bitwise reproducible.
int64_t mbvae_edge_attn_bwd_workspace(int64_t bs, int64_t n, int64_t dx) {
  const int64_t n_kg = (n + 7) / 8;                 // (sized for the smaller of the two key-group widths)
  return n <= 16 ? 0 : (int64_t)sizeof(float) * n_kg * bs * (n * dx + 2 * dx);
}

static int edge_attn_bwd_impl(const float* Q, const float* K, const float* V, int64_t ldq, const void* E1, const void* E2,
                              int64_t lde, const float* ye_mul, const uint8_t* mask, const float* wV, const float* lse,
                              const void* dP, const float* dwV, void* dE1, void* dE2, float* dQ, float* dK,
                              float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t dx,
                              float scale, void* workspace, int64_t ws_bytes, hipStream_t stream, bool bf16) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (bs <= 0 || n <= 0 || dx <= 0) return MBVAE_EUNSUPPORTED;
  if (ldq < dx || lde < dx || ldq > 0x7fffffff || lde > 0x7fffffff) return MBVAE_EINVAL;
  if (bs > 0x7fffffff / 64 || n > 0x7fffffff / 64 || dx > 0x7fffffff / 4) return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {Q, K, V, E1, E2, ye_mul, wV, lse, dP, dwV, dE1, dE2, dQ, dK, dV, dye_add, dye_mul};
  for (const void* p : ptrs) if (!p) return MBVAE_EINVAL;
  if (!mask) return MBVAE_EINVAL;
  const int KPW = (!bf16 && mbvae_opt(MBVAE_OPT_EA_KEYS) == 8) ? 8 : 16;
  const int n_kg = (int)((n + KPW - 1) / KPW), n_cq = (int)((dx + 63) / 64);
  const int64_t per_graph = (int64_t)sizeof(float) * n_kg * (n * dx + 2 * dx);
  int64_t chunk = bs;
  if (n_kg > 1) {
    if (!workspace || ws_bytes < per_graph) return MBVAE_EINVAL;
    chunk = std::min<int64_t>(bs, ws_bytes / per_graph);
  }
  for (int64_t b0 = 0; b0 < bs; b0 += chunk) {
    const int nb = (int)std::min<int64_t>(chunk, bs - b0);
    float* dq_part = n_kg > 1 ? (float*)workspace : dQ + b0 * n * ldq;
    const int64_t dq_stride = n_kg > 1 ? (int64_t)nb * n * dx : 0;
    float* dye_part = n_kg > 1 ? (float*)workspace + (int64_t)n_kg * dq_stride : nullptr;
    const int64_t dye_stride = (int64_t)nb * 2 * dx;
    const int64_t waves = (int64_t)nb * n_kg * n_cq;
    const bool pf = mbvae_opt(MBVAE_OPT_NO_EA_PREFETCH) == 0;   // measured at bs=256, n=64, dx=256: 16 keys + prefetch is fastest
    auto kern = bf16 ? edge_attn_bwd_kernel<16, true, true>
                : KPW == 8 ? (pf ? edge_attn_bwd_kernel<8, true> : edge_attn_bwd_kernel<8, false>)
                           : (pf ? edge_attn_bwd_kernel<16, true> : edge_attn_bwd_kernel<16, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, Q, K, V, E1, E2,
                       ye_mul, mask, wV, lse, dP, dwV, dE1, dE2, dK, dV, dq_part, n_kg > 1 ? dx : ldq, dq_stride, dye_part,
                       dye_stride, dye_add, dye_mul, (int)b0, nb, (int)n, (int)dx, scale, (int)ldq, (int)lde, n_kg, n_cq);
    if (n_kg > 1) {
      const int64_t total = (int64_t)nb * n * dx + (int64_t)nb * 2 * dx;
      const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
      hipLaunchKernelGGL(edge_attn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, stream, dq_part, dq_stride, dye_part,
                         dye_stride, dQ, dye_add, dye_mul, (int)b0, nb, (int)n, (int)dx, (int)ldq, n_kg);
    }
  }
  return mbvae_launch_status();
}

int mbvae_edge_attn_bwd(const float* Q, const float* K, const float* V, int64_t ldq, const float* E1, const float* E2,
                        int64_t lde, const float* ye_mul, const uint8_t* mask, const float* wV, const float* lse,
                        const float* dP, const float* dwV, float* dE1, float* dE2, float* dQ, float* dK,
                        float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t dx,
                        float scale, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  return edge_attn_bwd_impl(Q, K, V, ldq, E1, E2, lde, ye_mul, mask, wV, lse, dP, dwV, dE1, dE2, dQ, dK, dV, dye_add, dye_mul,
                            bs, n, dx, scale, workspace, ws_bytes, stream, false);
}

// The same with the edge streams stored as bf16: E1, E2, dP are read and dE1, dE2 written as bf16 (lde counts bf16
// elements); Q, K, V, wV, lse, dwV and every other output stay fp32.
int mbvae_edge_attn_bwd_bf16(const float* Q, const float* K, const float* V, int64_t ldq, const uint16_t* E1,
                             const uint16_t* E2, int64_t lde, const float* ye_mul, const uint8_t* mask, const float* wV,
                             const float* lse, const uint16_t* dP, const float* dwV, uint16_t* dE1, uint16_t* dE2,
                             float* dQ, float* dK, float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n,
                             int64_t dx, float scale, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  return edge_attn_bwd_impl(Q, K, V, ldq, E1, E2, lde, ye_mul, mask, wV, lse, dP, dwV, dE1, dE2, dQ, dK, dV, dye_add, dye_mul,
                            bs, n, dx, scale, workspace, ws_bytes, stream, true);
}

}  // extern "C"
