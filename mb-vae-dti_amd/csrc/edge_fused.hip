// Fused forward of the edge path's attention core: the e_mul|e_add projection of E runs on the matrix cores INSIDE
// the attention kernel, so the two [bs,n,n,dx] projection outputs never make a round trip through HBM.
//
//   E1|E2[b,i,j,:] = E[b,i,j,:] @ W12^T + b12            (zero where the pair (i,j) is masked)
//   Y[b,i,j,c]     = Q[b,i,c] * K[b,j,c] / sqrt(df) * (E1[b,i,j,c] + 1) + E2[b,i,j,c]
//   P[b,i,j,c]     = ye_add[b,c] + (ye_mul[b,c] + 1) * Y[b,i,j,c]
//   wV[b,i,c]      = sum_j softmax_j(Y[b,i,:,c] over valid keys) * V[b,j,c],   lse = log-sum-exp of that softmax
//
// Fixed widths de = 64, dx = 256 and n <= 64 (the molecule-graph configuration); other shapes use the separate
// projection and attention kernels.  One 512-thread workgroup per graph (or per block of query rows of a graph).
// Wave w owns channels [32w, 32w+32) of E1 and of E2: its slice of W12 (64 x 64 floats) lives in 64 registers per
// lane for the whole kernel, laid out as the MFMA B operand, so the products need no LDS at all; LDS holds the
// graph's K and V rows.  A tile = 32 keys x 32 channels of one query row: the lane that owns (channel c, half h)
// gets 16 keys of E1 and E2 in its accumulators, i.e. the softmax over keys runs down a lane's own registers and
// meets the other half's 16 (or 48) keys in one cross-half exchange per query row.  The E rows (A operand) go from
// global memory straight into MFMA operand registers, one tile ahead of the products.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned ef_u32x2 __attribute__((ext_vector_type(2)));

struct EdgeFusedArgs {
  const float* E; const float* W; const float* bias;
  const float* Q; const float* K; const float* V; int ldq;
  const float* ye_add; const float* ye_mul; const uint8_t* mask;
  float* P; float* wV; float* lse; float* E12;
  int bs, n; float scale; int ichunks;
  int* queue;      // optional work counter (zeroed before the launch): workgroups take tasks in order as they finish
  const int* node_off;   // optional [bs+1]: Q, K, V, wV, lse hold only the valid nodes, graph b's at rows node_off[b]..
  const int* graph_order;   // optional [bs]: the k-th group of tasks works on graph graph_order[k] (largest graphs first)
  int stagger;              // training forms: 0 all waves in phase; 1 waves 0-3 vs 4-7 half a tile apart; 2 even vs odd waves
};

__device__ __forceinline__ float4 ef_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// four consecutive fp32 values -> their three bf16 planes (8 bytes each), 4096 bytes apart, at dst
__device__ __forceinline__ void ef_stage4(unsigned char* dst, float4 v4) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 ph, pm, pl;
  const float xv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2) {
    const float a0 = xv[2 * p2], a1 = xv[2 * p2 + 1];
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    ph[p2] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    pm[p2] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    pl[p2] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
  }
  *reinterpret_cast<u32x2*>(dst) = ph; *reinterpret_cast<u32x2*>(dst + 4096) = pm; *reinterpret_cast<u32x2*>(dst + 8192) = pl;
}

// SPLIT: the projection runs on the bf16 matrix pipeline (six bf16 MFMAs per fp32 product, see mbvae_split8): the
// lane's W12 slice is split once into three bf16 planes (96 registers instead of 64), the E rows are split in
// registers right before their products, one split per 16 reduction indices shared by the E1 and the E2 tile.
// BF16OUT: P and E1|E2 are stored as bf16 (round to nearest even) - the opt-in reduced-precision storage mode; the
// softmax statistics and wV are computed from the unrounded fp32 values.
// COOP (with SPLIT): the eight waves of the workgroup multiply the SAME E rows by different weight slices, so splitting
// those rows in every wave does the work eight times.  Instead each wave fetches and splits one eighth of the next tile
// (one 16-byte load and 22 vector instructions per lane instead of eight loads and 176) and parks the planes in the
// 24 KB of LDS next to the K / V rows (two tile buffers, 16-byte groups XOR-swizzled by row: the A-operand fetch of
// the products - lane (key, half) reads eight consecutive reduction indices - is then free of bank conflicts); one
// barrier per tile hands the buffer over.  The operand double buffer in registers (32) shrinks to one float4.
template <bool WRITE_E12, bool SPLIT, bool BF16OUT = false, bool COOP = false>
__global__ __launch_bounds__(512) void edge_fused_fwd_kernel(EdgeFusedArgs a) {
  static_assert(!COOP || SPLIT, "the shared operand planes are bf16 planes");
  constexpr int OS = BF16OUT ? 2 : 4;              // bytes per stored P / E1|E2 element
  extern __shared__ __align__(16) float ef_kv[];   // K rows [64][256], then V rows [64][256]
  float* Ks = ef_kv;
  float* Vs = ef_kv + 64 * 256;
  constexpr int EF_LO_LDS = 3;                     // SPLIT: lo weight planes kept in LDS behind the K / V rows
  mbvae_bf16x8* wlo = reinterpret_cast<mbvae_bf16x8*>(ef_kv + 2 * 64 * 256);
  unsigned char* Ap = reinterpret_cast<unsigned char*>(ef_kv + 2 * 64 * 256);   // COOP: [2 buffers][3 planes][32 keys][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int c = wave * 32 + r;
  const int n = a.n, ldq = a.ldq;
  const int RT = (n + 31) >> 5;                    // key tiles per query row (1 or 2)

  // B operands: step (p, e) of a tile multiplies reduction index 32h + 4p + e (the A registers hold 32 consecutive
  // reduction indices of the lane's row), so the lane keeps W12[c][32h .. 32h+31] and W12[256 + c][32h .. 32h+31]
  float w1[32], w2[32];
  MbvaeSplit3 s1[4], s2[4];                          // SPLIT: octet o = reduction indices 32h + 8o .. + 7
  {
    const float* p1 = a.W + (int64_t)c * 64 + 32 * h;
    const float* p2 = a.W + (int64_t)(256 + c) * 64 + 32 * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 t = ef_ld4(p1 + 4 * q), u = ef_ld4(p2 + 4 * q);
      w1[4 * q] = t.x; w1[4 * q + 1] = t.y; w1[4 * q + 2] = t.z; w1[4 * q + 3] = t.w;
      w2[4 * q] = u.x; w2[4 * q + 1] = u.y; w2[4 * q + 2] = u.z; w2[4 * q + 3] = u.w;
    }
    if (SPLIT) {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        s1[o] = mbvae_split8(w1[8 * o], w1[8 * o + 1], w1[8 * o + 2], w1[8 * o + 3], w1[8 * o + 4], w1[8 * o + 5], w1[8 * o + 6], w1[8 * o + 7]);
        s2[o] = mbvae_split8(w2[8 * o], w2[8 * o + 1], w2[8 * o + 2], w2[8 * o + 3], w2[8 * o + 4], w2[8 * o + 5], w2[8 * o + 6], w2[8 * o + 7]);
        // the LDS left over next to the K / V rows (24 KB) takes three of the eight lo planes (thread-private 16-byte
        // slots, no barrier): 12 registers less, which is what keeps scratch reloads out of the tile loop
        if (!COOP && 2 * o < EF_LO_LDS) wlo[(2 * o) * 512 + tid] = s1[o].lo;     // (COOP: that LDS holds the operand planes,
        if (!COOP && 2 * o + 1 < EF_LO_LDS) wlo[(2 * o + 1) * 512 + tid] = s2[o].lo;   //  and the registers it frees hold these)
      }
    }
  }
  const float b1 = a.bias ? a.bias[c] : 0.f, b2 = a.bias ? a.bias[256 + c] : 0.f;
  const int ntasks = a.bs * a.ichunks;
  const int ipc = (n + a.ichunks - 1) / a.ichunks;   // query rows per task
  const bool st_e12 = a.E12 != nullptr;            // training form without E1|E2 output: the backward recomputes it
  // Two waves share a SIMD.  In phase they would both multiply (queueing on the matrix pipe), then both run their
  // element-wise epilogues (queueing on the vector ALU).  The rendezvous therefore sits at DIFFERENT points of the tile
  // for the two groups - after the products for group A, after the epilogue for group B - which keeps the groups half
  // a tile apart: while one multiplies the other runs its epilogue.  Same number of barriers per tile for every wave.
  const bool grpB = a.stagger == 1 ? wave >= 4 : (a.stagger == 2 ? (wave & 1) != 0 : false);
  const bool stag = WRITE_E12 && !COOP && a.stagger != 0;
  // COOP: this lane's share of a tile is 4 reduction indices of one key
  const int prow_ = 4 * wave + (lane >> 4), pk4 = (lane & 15) * 4;
  const int pst = prow_ * 128 + (((pk4 >> 3) ^ (prow_ & 7)) << 4) + ((pk4 >> 2) & 1) * 8;   // byte offset inside a plane
  float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 ra[2][8];
  float qv[2];
  float m_run = -INFINITY, l_run = 0.f, av_run = 0.f;

  // Task order: static round robin, or (with a queue) first come first served -- graphs of a ragged batch cost
  // very different amounts, and a workgroup that drew small ones simply takes more of them.  Every workgroup leaves
  // through the same test (task >= ntasks), so the grid always drains.
  __shared__ int next_task;
  for (int task = blockIdx.x, round = 0;; ++round) {
    if (a.queue) {
      __syncthreads();                             // everyone has read the previous value
      if (tid == 0) next_task = atomicAdd(a.queue, 1);
      __syncthreads();
      task = next_task;
    } else if (round > 0) {
      task += gridDim.x;
    }
    if (task >= ntasks) break;
    const int bt = task / a.ichunks, ic = task - bt * a.ichunks;
    const int b = a.graph_order ? a.graph_order[bt] : bt;
    const int i0 = ic * ipc, i1 = min(n, i0 + ipc);
    const int64_t node = (int64_t)b * n;
    // packed node rows (inference, masks that keep a prefix of each graph's nodes): node (b, i) is row pnode + i for
    // i < nb; the rows of masked keys are never used, any readable row stands in for them
    const int64_t pnode = a.node_off ? a.node_off[b] : node;
    const int nb = a.node_off ? a.node_off[b + 1] - a.node_off[b] : n;
    __syncthreads();                               // every wave is done with the previous graph's K, V rows
    for (int e = tid; e < n * 64; e += 512) {
      const int j = e >> 6, c4 = (e & 63) * 4;
      const int64_t src = pnode + (j < nb ? j : 0);
      *reinterpret_cast<float4*>(Ks + j * 256 + c4) = ef_ld4(a.K + src * ldq + c4);
      *reinterpret_cast<float4*>(Vs + j * 256 + c4) = ef_ld4(a.V + src * ldq + c4);
    }
    const unsigned long long kmask = __ballot(lane < n && a.mask[node + min(lane, n - 1)] != 0);   // bit j: key j valid
    const float ya = a.ye_add[(int64_t)b * 256 + c], ym1 = a.ye_mul[(int64_t)b * 256 + c] + 1.f;
    __syncthreads();
    // The training form writes P and E1|E2 for every pair.  The inference form only walks the tiles up to the last
    // valid query row and the last valid key: nothing reads P of a masked pair, and a masked query row's wV / lse
    // are the constants 0 / +inf.
    int RTq = RT, iend = i1;
    if (!WRITE_E12) {
      const int top = kmask ? 64 - __builtin_clzll(kmask) : 0;   // 1 + index of the last valid node
      RTq = top > 32 ? 2 : 1;
      iend = min(i1, min(top, nb));
      if (!a.node_off)
        for (int i = max(i0, top); i < i1; ++i)
          if (h == 0) { a.wV[(node + i) * 256 + c] = 0.f; a.lse[(node + i) * 256 + c] = INFINITY; }
    }
    const int T = (iend - i0) * RTq;               // tiles of this task
    if (T <= 0) continue;

    // tile t -> (query row, key tile); past the end the last tile is fetched again and never used
#define EF_FETCH(t, S)                                                                                \
    {                                                                                                 \
      const int tt = min((t), T - 1);                                                                  \
      const int ii = RTq == 2 ? tt >> 1 : tt, mt = RTq == 2 ? tt & 1 : 0;                              \
      const int64_t qrow = node + i0 + ii;                                                             \
      const int j = min(32 * mt + r, n - 1);                                                           \
      const float* src = a.E + ((qrow * n + j) * 64 + 32 * h);                                         \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) ra[SPLIT ? 0 : S][p] = ef_ld4(src + 4 * p);        \
      qv[S] = a.Q[(pnode + i0 + ii) * ldq + c];                                                        \
      asm volatile("" : : : "memory");       /* issue the loads here, do not sink them to their use */ \
    }
    /* COOP: this lane's 16 bytes of tile t; its three planes into buffer `buf` */                   \
#define EF_PIECE(t)                                                                                   \
    ([&]() -> float4 {                                                                                \
      const int tt = min((t), T - 1);                                                                  \
      const int ii = RTq == 2 ? tt >> 1 : tt, mt = RTq == 2 ? tt & 1 : 0;                              \
      return ef_ld4(a.E + (((node + i0 + ii) * n + min(32 * mt + prow_, n - 1)) * 64 + pk4));          \
    }())
#define EF_STAGE(buf, v4)                                                                             \
    {                                                                                                 \
      ef_u32x2 ph, pm, pl;                                                                             \
      const float xv[4] = {(v4).x, (v4).y, (v4).z, (v4).w};                                            \
      _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                               \
        const float a0 = xv[2 * p2], a1 = xv[2 * p2 + 1];                                              \
        const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);                             \
        ph[p2] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                                           \
        const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u); \
        const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);                             \
        pm[p2] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);                                           \
        const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u); \
        pl[p2] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);         \
      }                                                                                                \
      unsigned char* dst = Ap + (buf) * 12288 + pst;                                                   \
      *reinterpret_cast<ef_u32x2*>(dst) = ph; *reinterpret_cast<ef_u32x2*>(dst + 4096) = pm;           \
      *reinterpret_cast<ef_u32x2*>(dst + 8192) = pl;                                                   \
    }
#define EF_ST(ptr_rsrc, val, voff, soff)                                                             \
    do {                                                                                              \
      if (BF16OUT) __builtin_amdgcn_raw_buffer_store_b16(mbvae_f2bf(val), ptr_rsrc, voff, (soff) / 2, 2); \
      else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ptr_rsrc, voff, soff, 2);       \
    } while (0)
#define EF_TILE(t, S)                                                                                 \
    if ((t) < T) {                                                                                    \
      const int ii = RTq == 2 ? (t) >> 1 : (t), mt = RTq == 2 ? (t) & 1 : 0;                           \
      const int i = i0 + ii;                                                                           \
      const bool qi_ok = (kmask >> i) & 1ull;                                                          \
      const unsigned km32 = (unsigned)(kmask >> (32 * mt));                                            \
      if (mt == 0) { m_run = -INFINITY; l_run = 0.f; av_run = 0.f; }                                   \
      if (SPLIT && !COOP && !(qi_ok && km32 != 0u)) EF_FETCH((t) + 1, 1 - S)   /* no products here: request the next tile whole */ \
      if (COOP) {                                  /* hand tile t+1 to buffer 1-S, request this lane's share of tile t+2 */ \
        const int tn = min((t) + 1, T - 1);                                                            \
        qv[1 - S] = a.Q[(pnode + i0 + (RTq == 2 ? tn >> 1 : tn)) * ldq + c];                           \
        EF_STAGE(1 - S, nx)                                                                            \
        nx = EF_PIECE((t) + 2);                                                                        \
        asm volatile("" : : : "memory");                                                               \
      }                                                                                                \
      if (WRITE_E12 || (qi_ok && km32 != 0u)) {    /* inference: a tile without a valid pair is skipped */ \
      f32x16 acc1, acc2;                                                                               \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) { acc1[v] = b1; acc2[v] = b2; }                   \
      if (COOP && qi_ok && km32 != 0u) {                                                               \
        const unsigned char* Ab = Ap + S * 12288 + r * 128;                                            \
        _Pragma("unroll") for (int o = 0; o < 4; ++o) {                                                \
          const int sw = ((4 * h + o) ^ (r & 7)) << 4;                                                 \
          const mbvae_bf16x8 ahi = *reinterpret_cast<const mbvae_bf16x8*>(Ab + sw);                    \
          const mbvae_bf16x8 ami = *reinterpret_cast<const mbvae_bf16x8*>(Ab + 4096 + sw);             \
          const mbvae_bf16x8 alo = *reinterpret_cast<const mbvae_bf16x8*>(Ab + 8192 + sw);             \
          acc1 = MBVAE_MFMA_BF16(alo, s1[o].hi, acc1);  acc2 = MBVAE_MFMA_BF16(alo, s2[o].hi, acc2);   \
          acc1 = MBVAE_MFMA_BF16(ahi, s1[o].lo, acc1);  acc2 = MBVAE_MFMA_BF16(ahi, s2[o].lo, acc2);   \
          acc1 = MBVAE_MFMA_BF16(ami, s1[o].mid, acc1); acc2 = MBVAE_MFMA_BF16(ami, s2[o].mid, acc2);  \
          acc1 = MBVAE_MFMA_BF16(ami, s1[o].hi, acc1);  acc2 = MBVAE_MFMA_BF16(ami, s2[o].hi, acc2);   \
          acc1 = MBVAE_MFMA_BF16(ahi, s1[o].mid, acc1); acc2 = MBVAE_MFMA_BF16(ahi, s2[o].mid, acc2);  \
          acc1 = MBVAE_MFMA_BF16(ahi, s1[o].hi, acc1);  acc2 = MBVAE_MFMA_BF16(ahi, s2[o].hi, acc2);   \
        }                                                                                              \
      } else if (SPLIT && qi_ok && km32 != 0u) {                                                       \
        /* ONE operand buffer: an octet is dead once split, so the next tile's octet is requested into the same     \
           registers right there and has the rest of this tile (products, softmax, stores) to arrive */              \
        const int tn = min((t) + 1, T - 1);                                                            \
        const int iin = RTq == 2 ? tn >> 1 : tn, mtn = RTq == 2 ? tn & 1 : 0;                          \
        const float* nsrc = a.E + (((node + i0 + iin) * n + min(32 * mtn + r, n - 1)) * 64 + 32 * h);  \
        qv[1 - S] = a.Q[(pnode + i0 + iin) * ldq + c];                                                 \
        _Pragma("unroll") for (int o = 0; o < 4; ++o) {                                                \
          float4 t4 = ra[0][2 * o], u4 = ra[0][2 * o + 1];                                             \
          asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w), "+v"(u4.x), "+v"(u4.y), "+v"(u4.z), "+v"(u4.w) : : "memory"); \
          const MbvaeSplit3 as = mbvae_split8(t4.x, t4.y, t4.z, t4.w, u4.x, u4.y, u4.z, u4.w);         \
          ra[0][2 * o] = ef_ld4(nsrc + 8 * o); ra[0][2 * o + 1] = ef_ld4(nsrc + 8 * o + 4);            \
          asm volatile("" : : : "memory");                                                             \
          acc1 = MBVAE_MFMA_BF16(as.lo, s1[o].hi, acc1);   acc2 = MBVAE_MFMA_BF16(as.lo, s2[o].hi, acc2);   \
          acc1 = MBVAE_MFMA_BF16(as.hi, 2 * o < EF_LO_LDS ? wlo[(2 * o) * 512 + tid] : s1[o].lo, acc1);    \
          acc2 = MBVAE_MFMA_BF16(as.hi, 2 * o + 1 < EF_LO_LDS ? wlo[(2 * o + 1) * 512 + tid] : s2[o].lo, acc2); \
          acc1 = MBVAE_MFMA_BF16(as.mid, s1[o].mid, acc1); acc2 = MBVAE_MFMA_BF16(as.mid, s2[o].mid, acc2); \
          acc1 = MBVAE_MFMA_BF16(as.mid, s1[o].hi, acc1);  acc2 = MBVAE_MFMA_BF16(as.mid, s2[o].hi, acc2);  \
          acc1 = MBVAE_MFMA_BF16(as.hi, s1[o].mid, acc1);  acc2 = MBVAE_MFMA_BF16(as.hi, s2[o].mid, acc2);  \
          acc1 = MBVAE_MFMA_BF16(as.hi, s1[o].hi, acc1);   acc2 = MBVAE_MFMA_BF16(as.hi, s2[o].hi, acc2);   \
        }                                                                                              \
      } else if (qi_ok && km32 != 0u) {            /* a tile without a valid (query, key) pair needs no products */ \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                \
          float4 t4 = ra[S][p];                                                                        \
          asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w) : : "memory");              \
          const float avs[4] = {t4.x, t4.y, t4.z, t4.w};                                               \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                              \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(avs[e], w1[4 * p + e], acc1, 0, 0, 0);         \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(avs[e], w2[4 * p + e], acc2, 0, 0, 0);         \
          }                                                                                            \
        }                                                                                              \
      }                                                                                                \
      if (stag && !grpB) asm volatile("s_barrier" : : : "memory");                                     \
      const float qs = qv[S] * a.scale;                                                                \
      const float* Kp = Ks + (32 * mt + 4 * h) * 256 + c;                                              \
      const float* Vp = Vs + (32 * mt + 4 * h) * 256 + c;                                              \
      const int64_t prow = (node + i) * n;                                                             \
      const __amdgpu_buffer_rsrc_t rsP =                                                               \
          __builtin_amdgcn_make_buffer_rsrc((char*)a.P + prow * 256 * OS, 0, n * 256 * OS, 0x00020000); \
      const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(                            \
          WRITE_E12 ? (char*)a.E12 + prow * 512 * OS : (char*)a.P, 0, WRITE_E12 ? n * 512 * OS : 0, 0x00020000); \
      const int voffP = (4 * h * 256 + c) * OS, voffE = (4 * h * 512 + c) * OS;                        \
      float y[16];                                                                                     \
      if (qi_ok && km32 == 0xffffffffu) {          /* full tile, nothing masked */                     \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          const float e1 = acc1[v], e2 = acc2[v];                                                      \
          if (WRITE_E12 && st_e12) {                                                                   \
            EF_ST(rsE, e1, voffE, (32 * mt + rr) * 2048);                                              \
            EF_ST(rsE, e2, voffE, (32 * mt + rr) * 2048 + 1024);                                       \
          }                                                                                            \
          const float qk = qs * Kp[rr * 256];                                                          \
          y[v] = qk * (e1 + 1.f) + e2;                                                                 \
          EF_ST(rsP, ya + ym1 * y[v], voffP, (32 * mt + rr) * 1024);                                   \
        }                                                                                              \
        float tmax = y[0];                                                                             \
        _Pragma("unroll") for (int v = 1; v < 16; ++v) tmax = fmaxf(tmax, y[v]);                       \
        const float m_new = fmaxf(m_run, tmax);                                                        \
        const float corr = __expf(m_run - m_new);                                                      \
        l_run *= corr; av_run *= corr;                                                                 \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          const float w = __expf(y[v] - m_new);                                                        \
          l_run += w; av_run += w * Vp[rr * 256];                                                      \
        }                                                                                              \
        m_run = m_new;                                                                                 \
      } else {                                                                                         \
        const int rows_here = min(32, n - 32 * mt);                                                    \
        const unsigned kml = km32 >> (4 * h);      /* bit rr: this lane's row rr is a valid key */     \
        const unsigned inl = (rows_here >= 32 ? 0xffffffffu : ((1u << rows_here) - 1u)) >> (4 * h);    \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          /* inference: a group of 8 keys (4 per half-wave) without a valid one needs nothing (wave-uniform test) */ \
          if (!WRITE_E12 && ((km32 >> (8 * (v >> 2))) & 0xffu) == 0u) { y[v] = -INFINITY; continue; }  \
          const bool key_ok = (kml >> rr) & 1u, in = (inl >> rr) & 1u;                                 \
          const bool pair_ok = qi_ok && key_ok;                                                        \
          const float e1 = pair_ok ? acc1[v] : 0.f, e2 = pair_ok ? acc2[v] : 0.f;                      \
          const float kv = in ? Kp[rr * 256] : 0.f;                                                    \
          const float yy = qs * kv * (e1 + 1.f) + e2;                                                  \
          if (in && (WRITE_E12 || pair_ok)) {      /* inference: P of a masked pair is never read */  \
            if (WRITE_E12 && st_e12) {                                                                 \
              EF_ST(rsE, e1, voffE, (32 * mt + rr) * 2048);                                            \
              EF_ST(rsE, e2, voffE, (32 * mt + rr) * 2048 + 1024);                                     \
            }                                                                                          \
            EF_ST(rsP, ya + ym1 * yy, voffP, (32 * mt + rr) * 1024);                                   \
          }                                                                                            \
          y[v] = key_ok ? yy : -INFINITY;                                                              \
        }                                                                                              \
        float tmax = y[0];                                                                             \
        _Pragma("unroll") for (int v = 1; v < 16; ++v) tmax = fmaxf(tmax, y[v]);                       \
        const float m_new = fmaxf(m_run, tmax);                                                        \
        if (m_new > -INFINITY) {                                                                       \
          const float corr = __expf(m_run - m_new);                                                    \
          l_run *= corr; av_run *= corr;                                                               \
          _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                             \
            const int rr = (v & 3) + 8 * (v >> 2);                                                     \
            if (!WRITE_E12 && ((km32 >> (8 * (v >> 2))) & 0xffu) == 0u) continue;                      \
            const bool key_ok = (kml >> rr) & 1u;                                                      \
            const float w = key_ok ? __expf(y[v] - m_new) : 0.f;                                       \
            const float vv = key_ok ? Vp[rr * 256] : 0.f;                                              \
            l_run += w; av_run += w * vv;                                                              \
          }                                                                                            \
          m_run = m_new;                                                                               \
        }                                                                                              \
      }                                                                                                \
      }                                                                                                \
      if (mt == RTq - 1) {                         /* query row finished: join the two halves' keys */ \
        const float mo = __shfl_xor(m_run, 32), lo = __shfl_xor(l_run, 32), ao = __shfl_xor(av_run, 32); \
        const float M = fmaxf(m_run, mo);                                                              \
        float o = 0.f, s = INFINITY;                                                                   \
        if (M > -INFINITY) {                                                                           \
          const float ea = __expf(m_run - M), eb = __expf(mo - M);                                     \
          const float lt = l_run * ea + lo * eb;                                                       \
          o = (av_run * ea + ao * eb) / lt;                                                            \
          s = M + __logf(lt);                                                                          \
        }                                                                                              \
        if (h == 0) { a.wV[(pnode + i) * 256 + c] = o; a.lse[(pnode + i) * 256 + c] = s; }             \
      }                                                                                                \
      if (stag && grpB) asm volatile("s_barrier" : : : "memory");                                      \
      if (COOP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");   /* buffer 1-S is complete, buffer S is free */ \
    }

    if (COOP) {                                      // tile 0 into buffer 0, this lane's share of tile 1 on its way
      qv[0] = a.Q[(pnode + i0) * ldq + c];
      nx = EF_PIECE(0);
      EF_STAGE(0, nx)
      nx = EF_PIECE(1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
    } else {
      EF_FETCH(0, 0)
    }
    for (int t = 0; t < T; t += 2) {
      if (!SPLIT) EF_FETCH(t + 1, 1)
      EF_TILE(t, 0)
      if (!SPLIT) EF_FETCH(t + 2, 0)
      EF_TILE(t + 1, 1)
      // keep the eight waves within two tiles of each other: they all read the same E rows, and a wave that falls
      // behind finds them evicted from L1/L2 and fetches them from HBM again (measured: E read 2.7 times).
      // Pure rendezvous, no data is exchanged, so no memory counters are waited for.  Only in the HBM-bound form
      // that also writes E1|E2; the inference form is bound by the products and ran 2 % slower in step.
      if (WRITE_E12 && !stag) asm volatile("s_barrier" : : : "memory");
    }
#undef EF_FETCH
#undef EF_TILE
#undef EF_ST
#undef EF_PIECE
#undef EF_STAGE
  }
}

// ------------------------------------------------------------------------------------------------------------
// Backward of the fused block with the projection RECOMPUTED: E1|E2 = E W12^T + b12 is formed again on the matrix
// cores from E (64 floats per pair) instead of being read back (512 floats per pair), so the forward does not store
// it either - 4 GB less HBM traffic per layer at bs = 256, n = 64.  Same tiling as the forward: one workgroup per
// graph, wave w owns channels [32w, 32w+32) of E1 and E2, a tile is 32 keys x 32 channels of one query row with a
// lane holding 16 keys of its channel.  The sums follow from that ownership without LDS, atomics or a second kernel:
//   dK[j], dV[j]   (sum over query rows)  lane-private accumulators of the current KEY tile: the key tiles are the
//                                         OUTER loop, so 2 x 16 registers suffice and every (key, channel) has one owner
//   dQ[i]          (sum over keys)        in-tile sum + the other half-wave; the second key tile adds to what the same
//                                         lane stored for the first one
//   dye_add/mul    (sum over everything)  lane-private, the two half-waves meet once per graph
// in a fixed order: bitwise reproducible.  dE1|dE2 [bs,n,n,2dx] is written for the projection's gradient kernels.
struct EdgeFusedBwdArgs {
  const float* E; const float* W; const float* bias;
  const float* Q; const float* K; const float* V; int ldq;
  const float* ye_mul; const uint8_t* mask; const float* wV; const float* lse;
  const float* dP; const float* dwV;
  float* dE12; float* dQ; float* dK; float* dV; float* dye_add; float* dye_mul;
  int bs, n; float scale;
  int sync_mask;            // the waves of a workgroup meet after every query row i with (i & sync_mask) == sync_mask
  int stagger;              // 0: all waves meet at the end of the row; 1: waves 0-3 meet after their products, waves 4-7 after
                            // their epilogue (half a tile apart: one group multiplies while the other runs its epilogue); 2: even / odd
};

// BF16: dP is read and dE1|dE2 written as bf16 (the opt-in bf16 storage mode); everything else as in fp32.
// COOP: the operand tile (the E rows of 32 keys) is fetched and split once per workgroup - each wave converts an eighth -
// and shared as bf16 planes through 24 KB of LDS (two buffers, one barrier per query row), as in the forward; the
// weight planes then get 16 LDS slots (hi, mid) and all eight lo planes sit in registers freed by the operand buffer.
template <bool BF16, bool COOP>
__global__ __launch_bounds__(512) void edge_fused_bwd_kernel(EdgeFusedBwdArgs a) {
  constexpr int SS = BF16 ? 2 : 4;                  // bytes per dP / dE1|dE2 element
  // Register budget (256 per lane at 512 threads): the lane's W12 slice as three bf16 planes would take 96.  Most of
  // them therefore live in LDS - 16 bytes per (plane, octet, E1/E2, thread), written and read by their owner only,
  // so no barrier guards them.  K and V need no LDS here: with the key
  // tile as the outer loop a lane's 16 keys are fixed for a whole pass over the query rows and sit in registers.
  extern __shared__ __align__(16) float ef_kv[];
  mbvae_bf16x8* wl = reinterpret_cast<mbvae_bf16x8*>(ef_kv);      // [(plane * 4 + octet) * 2 + which][512]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int c = wave * 32 + r;
  const int n = a.n, ldq = a.ldq;
  const int RT = (n + 31) >> 5;

  // 20 of the 24 (plane, octet, E1/E2) slots fill the CU's whole 160 KB of LDS (this kernel has no static __shared__
  // data): all hi and mid planes and four lo planes; the other four lo planes are registers.
  constexpr int LO_LDS = COOP ? 0 : 4;
  unsigned char* Ap = reinterpret_cast<unsigned char*>(ef_kv) + 16 * 512 * 16;   // COOP: [2 buffers][3 planes][32 keys][128 B]
  mbvae_bf16x8 lo[8];                                // lo planes, slot = 2 * octet + (E2 ? 1 : 0); slots < LO_LDS live in LDS
  {
    const float* p1 = a.W + (int64_t)c * 64 + 32 * h;
    const float* p2 = a.W + (int64_t)(256 + c) * 64 + 32 * h;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float4 t = ef_ld4(p1 + 8 * o), u = ef_ld4(p1 + 8 * o + 4), v = ef_ld4(p2 + 8 * o), w = ef_ld4(p2 + 8 * o + 4);
      const MbvaeSplit3 s1 = mbvae_split8(t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w);
      const MbvaeSplit3 s2 = mbvae_split8(v.x, v.y, v.z, v.w, w.x, w.y, w.z, w.w);
      wl[((0 * 4 + o) * 2 + 0) * 512 + tid] = s1.hi; wl[((0 * 4 + o) * 2 + 1) * 512 + tid] = s2.hi;
      wl[((1 * 4 + o) * 2 + 0) * 512 + tid] = s1.mid; wl[((1 * 4 + o) * 2 + 1) * 512 + tid] = s2.mid;
      if (2 * o < LO_LDS) wl[(16 + 2 * o) * 512 + tid] = s1.lo; else lo[2 * o] = s1.lo;
      if (2 * o + 1 < LO_LDS) wl[(16 + 2 * o + 1) * 512 + tid] = s2.lo; else lo[2 * o + 1] = s2.lo;
    }
  }
  const float b1 = a.bias ? a.bias[c] : 0.f, b2 = a.bias ? a.bias[256 + c] : 0.f;
  const bool grpB = a.stagger == 1 ? wave >= 4 : (a.stagger == 2 ? (wave & 1) != 0 : false);
  float4 ra[8];
  // COOP: this lane's share of a tile is 4 reduction indices of one key
  const int prow_ = 4 * wave + (lane >> 4), pk4 = (lane & 15) * 4;
  const int pst = prow_ * 128 + (((pk4 >> 3) ^ (prow_ & 7)) << 4) + ((pk4 >> 2) & 1) * 8;
  float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int b = blockIdx.x; b < a.bs; b += gridDim.x) {
    const int64_t node = (int64_t)b * n;
    const unsigned long long kmask = __ballot(lane < n && a.mask[node + min(lane, n - 1)] != 0);
    const float ym1 = a.ye_mul[(int64_t)b * 256 + c] + 1.f;
    float dya = 0.f, dym = 0.f;
    for (int mt = 0; mt < RT; ++mt) {
      float dk[16], dv[16], kv[16], vv[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int j = 32 * mt + 4 * h + (v & 3) + 8 * (v >> 2);
        dk[v] = 0.f; dv[v] = 0.f;
        kv[v] = j < n ? a.K[(node + j) * ldq + c] : 0.f;
        vv[v] = j < n ? a.V[(node + j) * ldq + c] : 0.f;
      }
      const unsigned km32 = (unsigned)(kmask >> (32 * mt));
      const unsigned kml = km32 >> (4 * h);          // bit rr: this lane's key rr of the tile is valid
      const int jrow = min(32 * mt + r, n - 1);      // the key whose E row this lane feeds to the products
      const int64_t pkey = min(32 * mt + prow_, n - 1);     // COOP: the key of this lane's share
      if (COOP) {                                    // query row 0 into buffer 0, this lane's share of row 1 on its way
        ef_stage4(Ap + pst, ef_ld4(a.E + ((node * n + pkey) * 64 + pk4)));
        nx = ef_ld4(a.E + (((node + min(1, n - 1)) * n + pkey) * 64 + pk4));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
      } else {
        const float* src = a.E + ((node * n + jrow) * 64 + 32 * h);
#pragma unroll
        for (int p = 0; p < 8; ++p) ra[p] = ef_ld4(src + 4 * p);
      }
      for (int i = 0; i < n; ++i) {
        const bool qi_ok = (kmask >> i) & 1ull;
        const int64_t qn = node + i;
        const float qv = a.Q[qn * ldq + c], ls = a.lse[qn * 256 + c], wv = a.wV[qn * 256 + c], dwv = a.dwV[qn * 256 + c];
        // the first key tile's dQ partial (stored by this very lane) is requested here, a whole tile before it is added
        float* const dqp = a.dQ + qn * ldq + c;
        const float dq_prev = (mt > 0 && h == 0) ? *dqp : 0.f;
        // this tile's dP words are requested before the products and arrive underneath them
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(const_cast<float*>(a.dP)) + qn * n * 256 * SS, 0, n * 256 * SS, 0x00020000);
        const int voffP = (4 * h * 256 + c) * SS, voffE = (4 * h * 512 + c) * SS;
        float dp[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {                // (rows past n are out of the descriptor's range: they read 0)
          const int so = (32 * mt + (v & 3) + 8 * (v >> 2)) * 256 * SS;
          if (BF16) dp[v] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsP, voffP, so, 2) << 16);
          else dp[v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsP, voffP, so, 2));
        }
        asm volatile("" : : : "memory");
        f32x16 acc1, acc2;
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc1[v] = b1; acc2[v] = b2; }
        // ONE operand buffer: an octet is dead once split, the next query row's octet is requested into the same registers
        const float* nsrc = a.E + (((node + min(i + 1, n - 1)) * n + jrow) * 64 + 32 * h);
        if (COOP) {                                  // hand row i+1 to the other buffer, request this lane's share of row i+2
          ef_stage4(Ap + ((i + 1) & 1) * 12288 + pst, nx);
          nx = ef_ld4(a.E + (((node + min(i + 2, n - 1)) * n + pkey) * 64 + pk4));
          asm volatile("" : : : "memory");
        }
        const unsigned char* Ab = Ap + (i & 1) * 12288 + r * 128;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          MbvaeSplit3 as;
          if (COOP) {
            const int sw = ((4 * h + o) ^ (r & 7)) << 4;
            as.hi = *reinterpret_cast<const mbvae_bf16x8*>(Ab + sw);
            as.mid = *reinterpret_cast<const mbvae_bf16x8*>(Ab + 4096 + sw);
            as.lo = *reinterpret_cast<const mbvae_bf16x8*>(Ab + 8192 + sw);
            asm volatile("" : : : "memory");
          } else {
            float4 t4 = ra[2 * o], u4 = ra[2 * o + 1];
            asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w), "+v"(u4.x), "+v"(u4.y), "+v"(u4.z), "+v"(u4.w) : : "memory");
            as = mbvae_split8(t4.x, t4.y, t4.z, t4.w, u4.x, u4.y, u4.z, u4.w);
            ra[2 * o] = ef_ld4(nsrc + 8 * o); ra[2 * o + 1] = ef_ld4(nsrc + 8 * o + 4);
            asm volatile("" : : : "memory");
          }
          if (qi_ok && km32 != 0u) {                  // a tile without a valid pair needs no products (wave-uniform)
            // weight planes are fetched from LDS one pair at a time, in the order lo, mid, hi, each right before its
            // products (8 registers in use at a time instead of 24); smaller terms still enter the sums first
            {
              const mbvae_bf16x8 w1l = 2 * o < LO_LDS ? wl[(16 + 2 * o) * 512 + tid] : lo[2 * o];
              const mbvae_bf16x8 w2l = 2 * o + 1 < LO_LDS ? wl[(16 + 2 * o + 1) * 512 + tid] : lo[2 * o + 1];
              acc1 = MBVAE_MFMA_BF16(as.hi, w1l, acc1);   acc2 = MBVAE_MFMA_BF16(as.hi, w2l, acc2);
            }
            {
              const mbvae_bf16x8 m1 = wl[((1 * 4 + o) * 2 + 0) * 512 + tid], m2 = wl[((1 * 4 + o) * 2 + 1) * 512 + tid];
              acc1 = MBVAE_MFMA_BF16(as.mid, m1, acc1);   acc2 = MBVAE_MFMA_BF16(as.mid, m2, acc2);
              acc1 = MBVAE_MFMA_BF16(as.hi, m1, acc1);    acc2 = MBVAE_MFMA_BF16(as.hi, m2, acc2);
            }
            {
              const mbvae_bf16x8 h1 = wl[((0 * 4 + o) * 2 + 0) * 512 + tid], h2 = wl[((0 * 4 + o) * 2 + 1) * 512 + tid];
              acc1 = MBVAE_MFMA_BF16(as.lo, h1, acc1);    acc2 = MBVAE_MFMA_BF16(as.lo, h2, acc2);
              acc1 = MBVAE_MFMA_BF16(as.mid, h1, acc1);   acc2 = MBVAE_MFMA_BF16(as.mid, h2, acc2);
              acc1 = MBVAE_MFMA_BF16(as.hi, h1, acc1);    acc2 = MBVAE_MFMA_BF16(as.hi, h2, acc2);
            }
          }
        }
        const bool meet = !COOP && (i & a.sync_mask) == a.sync_mask;
        if (meet && a.stagger != 0 && !grpB) asm volatile("s_barrier" : : : "memory");
        const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(a.dE12) + qn * n * 512 * SS, 0, n * 512 * SS, 0x00020000);
        const float qs = qv * a.scale;
        float dq = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int rr = (v & 3) + 8 * (v >> 2);
          const bool key_ok = (kml >> rr) & 1u;
          const bool pair_ok = qi_ok && key_ok;
          const float e1 = pair_ok ? acc1[v] : 0.f, e2 = pair_ok ? acc2[v] : 0.f;    // as the forward used them
          const float qk = qs * kv[v];
          const float y = qk * (e1 + 1.f) + e2;
          const float aa = key_ok ? __expf(y - ls) : 0.f;
          const float dy = dp[v] * ym1 + aa * dwv * (vv[v] - wv);
          // (rows past n are out of the descriptor's range: the stores are dropped)
          if (BF16) {
            __builtin_amdgcn_raw_buffer_store_b16(mbvae_f2bf(dy * qk), rsE, voffE, (32 * mt + rr) * 1024, 2);
            __builtin_amdgcn_raw_buffer_store_b16(mbvae_f2bf(dy), rsE, voffE, (32 * mt + rr) * 1024 + 512, 2);
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dy * qk), rsE, voffE, (32 * mt + rr) * 2048, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dy), rsE, voffE, (32 * mt + rr) * 2048 + 1024, 2);
          }
          const float g = dy * (e1 + 1.f);
          dq += g * (kv[v] * a.scale);
          dk[v] += g * qs;
          dv[v] += aa * dwv;
          dya += dp[v];
          dym += dp[v] * y;
        }
        dq += __shfl_xor(dq, 32);
        if (h == 0) *dqp = dq_prev + dq;
        if (meet && (a.stagger == 0 || grpB)) asm volatile("s_barrier" : : : "memory");   // keep the waves together: they share the E rows in L1
        if (COOP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");       // the other buffer is complete, this one is free
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int j = 32 * mt + 4 * h + (v & 3) + 8 * (v >> 2);
        if (j < n) { a.dK[(node + j) * ldq + c] = dk[v]; a.dV[(node + j) * ldq + c] = dv[v]; }
      }
    }
    dya += __shfl_xor(dya, 32);
    dym += __shfl_xor(dym, 32);
    if (h == 0) { a.dye_add[(int64_t)b * 256 + c] = dya; a.dye_mul[(int64_t)b * 256 + c] = dym; }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Inference-only edge feed-forward block in one pass over the rows:
//     out = LayerNorm(x + W2 relu(W1 x + b1) + b2) * gamma + beta          x, out [M, 64], hidden width 128
// (training keeps the separate kernels: its backward needs the hidden activations and the pre-norm sum in memory).
// Both products run TRANSPOSED on the matrix cores so that they chain through registers: H^T = W1 X^T puts the row
// index m on the lanes and the 128 hidden units in the accumulator registers, which is exactly the B-operand layout
// of Y^T = W2 H^T (step (t, v) multiplies hidden unit 32t + (v&3) + 8(v>>2) of the lower half-wave with that unit
// + 4 of the upper one; the A operand W2 is read from LDS at the matching indices).  After the second product a
// lane holds 32 of its row's 64 outputs and lane + 32 the others: residual add and LayerNorm are register
// arithmetic plus one cross-half exchange per statistic.  Per row 256 B are read and 256 B written; the separate
// kernels move 2.3 KB.
// helpers of the bf16-pipeline forms below
// one fp32 weight -> its three bf16 planes, `plane` elements apart
__device__ __forceinline__ void ef_store_planes(unsigned short* dst, int plane, float w) {
  const unsigned u = __float_as_uint(w);
  const float r1 = w - __uint_as_float(u & 0xffff0000u);
  const unsigned v = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(v & 0xffff0000u);
  dst[0] = (unsigned short)(u >> 16); dst[plane] = (unsigned short)(v >> 16); dst[2 * plane] = (unsigned short)(__float_as_uint(r2) >> 16);
}
// eight consecutive bf16 (16-byte aligned) / two runs of four, eight elements apart (8-byte aligned each)
__device__ __forceinline__ mbvae_bf16x8 ef_ld8(const unsigned short* p) { return *reinterpret_cast<const mbvae_bf16x8*>(p); }
__device__ __forceinline__ mbvae_bf16x8 ef_ld44(const unsigned short* p) {
  const ef_u32x2 a = *reinterpret_cast<const ef_u32x2*>(p), b = *reinterpret_cast<const ef_u32x2*>(p + 8);
  mbvae_u32x4 v; v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  return __builtin_bit_cast(mbvae_bf16x8, v);
}
// acc += W (A operand: planes wh, wm, wl) x activations (B operand: split xs), smaller terms first
#define EF_SPLIT_MFMA(acc, wh, wm, wl, xs)                                                              \
  {                                                                                                     \
    const mbvae_bf16x8 wh_ = (wh), wm_ = (wm), wl_ = (wl);                                              \
    acc = MBVAE_MFMA_BF16(wl_, xs.hi, acc); acc = MBVAE_MFMA_BF16(wh_, xs.lo, acc); acc = MBVAE_MFMA_BF16(wm_, xs.mid, acc); \
    acc = MBVAE_MFMA_BF16(wm_, xs.hi, acc); acc = MBVAE_MFMA_BF16(wh_, xs.mid, acc); acc = MBVAE_MFMA_BF16(wh_, xs.hi, acc); \
  }

// 16-byte store.  Plain on purpose: each lane writes 32-byte pieces of its own row, which the L2 assembles into whole
// lines; with the non-temporal hint the same stores ran 2.6 times slower (measured, 1 M rows).
#define EF_ST4(ptr, v4) (*reinterpret_cast<float4*>(ptr) = (v4))

struct EdgeFfnArgs {
  const float* x; const float* W1; const float* b1; const float* W2; const float* b2;
  const float* gamma; const float* beta; float* out; int64_t M; float eps; const uint8_t* tile_ok;
  const int32_t* row_idx; const int32_t* nvalid;   // packed row order (see ef_row_order), or null
  // training form: the hidden activation after ReLU / dropout [M,128] and the pre-LayerNorm sum [M,64] are stored for
  // the backward; dropout keep-scales are hashed from (seed, row * width + column) as in the separate Linear kernels
  float* hidden; float* z; float p1, p2; uint64_t seed1, seed2;
};

// Packed row order for ragged batches (optional, both kernels below): row_idx lists the nvalid[0] valid rows, padded
// to a multiple of 32 by repeating the last one, then the M - nvalid[0] padding rows, padded likewise.  Tile t of the
// kernels is then rows row_idx[32t .. 32t+31] instead of rows 32t .. 32t+31: valid rows fill whole tiles (no products
// are spent on padding rows that merely share a tile with valid ones) and the padding tiles follow.  Repeated
// entries are evaluated and stored twice with the same values.  nvalid has two entries: the count and a flag.
struct EfRowOrder {
  const int32_t* idx; int64_t nvt, ntiles;
  __device__ __forceinline__ EfRowOrder(const int32_t* row_idx, const int32_t* nvalid, int64_t M) : idx(row_idx) {
    if (row_idx) {
      const int64_t nv = nvalid[0];
      nvt = (nv + 31) / 32;
      // nvalid[1] != 0: only the FIRST padding tile is evaluated and stored (it carries the padding rows' common value
      // from kernel to kernel); the caller reads no other padding row (mbvae_pool_stats_pairs_fwd does not).
      ntiles = nvt + (nvalid[1] != 0 ? (int64_t)(nv < M) : (M - nv + 31) / 32);
    } else {
      nvt = 0;
      ntiles = (M + 31) / 32;       // the last tile may be partial: loads clamp, stores are guarded
    }
  }
  // row of lane r in tile t, and whether it may be stored (always, in the packed order)
  __device__ __forceinline__ int row(int64_t t, int r, int64_t M) const {
    return idx ? idx[t * 32 + r] : (int)min(t * 32 + r, M - 1);
  }
  __device__ __forceinline__ bool inb(int64_t t, int r, int64_t M) const { return idx ? true : t * 32 + r < M; }
};

// Stores the 64-float row `cst` (LDS) to the 32 rows of a tile: 16 lanes per row, four whole rows per instruction
// (the products' own layout, one row per lane, would write 32-byte pieces).  row / inb: lane r (< 32) holds the tile's
// r-th row index and whether it may be written.
__device__ __forceinline__ void ef_store_row_tile(float* out, const float* cst, int row, bool inb, int lane) {
  const int sub = lane & 15, rr = lane >> 4;
  const float4 cv = *reinterpret_cast<const float4*>(cst + sub * 4);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int src = it * 4 + rr;
    const int rw = __shfl(row, src, 64);
    const int ok = __shfl((int)inb, src, 64);
    if (ok) *reinterpret_cast<float4*>(out + (int64_t)rw * 64 + sub * 4) = cv;
  }
}

// Validity of the 32-row tiles a wave owns (tile k of the wave = global tile wid + k*nw), 64 at a time: one byte
// gather + ballot per window; k only ever grows.  A null map marks every tile valid.
struct EfTileWindow {
  unsigned long long bits = 0ull;
  int64_t base = -64;
  __device__ __forceinline__ bool ok(const uint8_t* map, int64_t k, int64_t wid, int64_t nw, int64_t mine, int lane,
                                     const EfRowOrder& ord) {
    if (ord.idx) return wid + k * nw < ord.nvt;
    if (!map) return true;
    if (k >= base + 64) {
      base = k;
      const int64_t kk = k + lane;
      const uint8_t v = kk < mine ? map[wid + kk * nw] : (uint8_t)0;
      bits = __ballot(v != 0);
    }
    return (bits >> (k - base)) & 1ull;
  }
};

// SPLIT: both products on the bf16 matrix pipeline (six bf16 MFMAs per fp32 product, see mbvae_split8).  The weights
// are split once, at staging time, into three bf16 planes per matrix, rows padded by 16 bytes so that the A-operand
// fetches are conflict-free; the activations (x rows, then the ReLU'd hidden accumulators - eight consecutive
// accumulator registers ARE one B-operand octet) are split in registers, once per 16 reduction indices.
// TRAIN: dropout (p1) on the hidden activation and (p2) on the second Linear's output; the hidden activation and the
// pre-LayerNorm sum are written out (once each, never read back here) for the backward kernels.
template <bool SPLIT, bool TRAIN = false>
__global__ __launch_bounds__(512) void edge_ffn_fwd_kernel(EdgeFfnArgs a) {
  constexpr int L1 = 65, L2 = 129;                  // LDS row strides of W1 [128][64] and W2 [64][128]
  constexpr int K1 = 72, K2 = 136;                  // SPLIT: bf16 elements per row of the W1 / W2 planes
  extern __shared__ __align__(16) float ff_lds[];
  float* W1s = ff_lds;                              // [128][L1]
  float* W2s = W1s + 128 * L1;                      // [64][L2]
  unsigned short* W1p = reinterpret_cast<unsigned short*>(ff_lds);     // SPLIT: [3][128][K1]
  unsigned short* W2p = W1p + 3 * 128 * K1;                            // SPLIT: [3][64][K2]
  float* b1s = SPLIT ? reinterpret_cast<float*>(W2p + 3 * 64 * K2) : W2s + 64 * L2;   // [128]
  float* b2s = b1s + 128;                           // [64]
  float* gs = b2s + 64;                             // [64]
  float* bs_ = gs + 64;                             // [64]
  float* cst = bs_ + 64;                            // [8][64]: per wave, the output row shared by all padding rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  if (SPLIT) {
    for (int e = tid; e < 128 * 64; e += 512) {
      ef_store_planes(W1p + (e >> 6) * K1 + (e & 63), 128 * K1, a.W1[e]);
      ef_store_planes(W2p + (e >> 7) * K2 + (e & 127), 64 * K2, a.W2[e]);
    }
  } else {
    for (int e = tid; e < 128 * 64; e += 512) { W1s[(e >> 6) * L1 + (e & 63)] = a.W1[e]; W2s[(e >> 7) * L2 + (e & 127)] = a.W2[e]; }
  }
  if (tid < 128) b1s[tid] = a.b1 ? a.b1[tid] : 0.f;
  if (tid < 64) { b2s[tid] = a.b2 ? a.b2[tid] : 0.f; gs[tid] = a.gamma[tid]; bs_[tid] = a.beta[tid]; }
  __syncthreads();

  const EfRowOrder ord(a.row_idx, a.nvalid, a.M);
  const int64_t ntiles = ord.ntiles;
  const int64_t wid = (int64_t)blockIdx.x * 8 + wave, nw = (int64_t)gridDim.x * 8;
  const int64_t mine = wid < ntiles ? (ntiles - wid + nw - 1) / nw : 0;
  if (mine <= 0) return;
  float4 ra[2][8];
  int srow[2];                                      // this lane's row in the staged tile
  bool sinb[2];
  int nxt_row = ord.row(wid, r, a.M);               // row of the NEXT tile to be fetched (its index load runs a tile ahead)
  // Tiles marked 0 in tile_ok hold padding rows, all with the same values (the caller's guarantee): the first one a
  // wave meets is evaluated like any other tile and its output row kept in LDS; the later ones only store that row.
  // mode: 0 = evaluate, 1 = evaluate and keep the row, 2 = store the kept row (its operand fetch is pointed at
  // tile 0, which stays in cache, so that the fetch itself remains unconditional).
  EfTileWindow win;
  int mode[2] = {0, 0};
  bool have_cst = false;
  float* my_cst = cst + wave * 64;
#define FF_FETCH(q, S)                                                                                \
  {                                                                                                   \
    const int64_t k = min((int64_t)(q), mine - 1);                                                     \
    int md = 0;                                                                                        \
    if (!win.ok(a.tile_ok, k, wid, nw, mine, lane, ord)) { md = have_cst ? 2 : 1; have_cst = true; }   \
    mode[S] = md;                                                                                      \
    srow[S] = nxt_row;                                                                                 \
    sinb[S] = ord.inb(wid + k * nw, r, a.M);                                                           \
    /* the next tile's row index is requested BEFORE this tile's operands: the memory counter is in order, so the  \
       wait for the index (next fetch) then leaves the operand loads issued after it in flight */      \
    nxt_row = ord.row(wid + min((int64_t)(q) + 1, mine - 1) * nw, r, a.M);                             \
    asm volatile("" : : : "memory");                                                                   \
    const float* src = a.x + (int64_t)(md == 2 ? min(r, (int)a.M - 1) : srow[S]) * 64 + 32 * h;        \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) ra[S][p] = ef_ld4(src + 4 * p);                      \
    asm volatile("" : : : "memory");                                                                   \
  }
#define FF_TILE(q, S)                                                                                 \
  if ((q) < mine && mode[S] == 2) {                                                                   \
    ef_store_row_tile(a.out, my_cst, srow[S], sinb[S], lane);                                          \
  } else if ((q) < mine) {                                                                            \
    /* a zero the optimiser cannot see through, added to the LDS indices of the small per-column vectors: their   \
       160 values are loop invariant and would otherwise be hoisted into registers, spilling the accumulators */ \
    int oz;                                                                                            \
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));                                                        \
    const int h4 = 4 * h + oz;                                                                         \
    /* the residual operand (this lane's 32 output columns of its row; cache hits, the row was just read) is      \
       requested before the products: asked for in the epilogue it costs a full memory latency per tile AND, the  \
       memory counter being in order, drags the wait for the next tile's prefetch along */            \
    float4 xres[8];                                                                                    \
    {                                                                                                  \
      const float* xrow = a.x + (int64_t)srow[S] * 64 + 4 * h;                                         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                    \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) xres[4 * u + gq] = ef_ld4(xrow + 32 * u + 8 * gq); \
      asm volatile("" : : : "memory");                                                                 \
    }                                                                                                  \
    /* hidden layer, transposed: rows = hidden units (4 tiles of 32), columns = the tile's 32 rows */  \
    f32x16 hh[4];                                                                                      \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                      \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) hh[t][v] = b1s[32 * t + h4 + (v & 3) + 8 * (v >> 2)]; \
    if (SPLIT) {                                                                                       \
      _Pragma("unroll") for (int o = 0; o < 4; ++o) {                                                  \
        float4 t4 = ra[S][2 * o], u4 = ra[S][2 * o + 1];                                               \
        asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w), "+v"(u4.x), "+v"(u4.y), "+v"(u4.z), "+v"(u4.w) : : "memory"); \
        const MbvaeSplit3 xs = mbvae_split8(t4.x, t4.y, t4.z, t4.w, u4.x, u4.y, u4.z, u4.w);           \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
          const unsigned short* wp = W1p + (32 * t + r) * K1 + 32 * h + 8 * o + oz;   /* W1[unit r + 32t][k octet] */ \
          EF_SPLIT_MFMA(hh[t], ef_ld8(wp), ef_ld8(wp + 128 * K1), ef_ld8(wp + 2 * 128 * K1), xs)       \
        }                                                                                              \
      }                                                                                                \
    } else {                                                                                           \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                    \
      float4 t4 = ra[S][p];                                                                            \
      asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w) : : "memory");                  \
      const float xs[4] = {t4.x, t4.y, t4.z, t4.w};                                                    \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
        const float* wcol = W1s + r * L1 + 32 * h + 4 * p + e + oz;      /* W1[unit r + 32t][k], k as in xs */ \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                  \
          hh[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[32 * t * L1], xs[e], hh[t], 0, 0, 0);      \
      }                                                                                                \
    }                                                                                                  \
    }                                                                                                  \
    if (TRAIN) {                 /* ReLU + dropout in place, then the hidden row goes out: 16 x 16 bytes per lane */ \
      const unsigned hbase = (unsigned)srow[S] * 128u + 4u * h;                                        \
      float* hrow = a.hidden + (int64_t)srow[S] * 128 + 4 * h;                                         \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                             \
          float4 o;                                                                                    \
          o.x = fmaxf(hh[t][4 * gq + 0], 0.f); o.y = fmaxf(hh[t][4 * gq + 1], 0.f);                    \
          o.z = fmaxf(hh[t][4 * gq + 2], 0.f); o.w = fmaxf(hh[t][4 * gq + 3], 0.f);                    \
          if (a.p1 > 0.f) {                                                                            \
            const unsigned i0 = hbase + 32 * t + 8 * gq;                                               \
            o.x *= mbvae_drop_scale(a.p1, a.seed1, i0); o.y *= mbvae_drop_scale(a.p1, a.seed1, i0 + 1); \
            o.z *= mbvae_drop_scale(a.p1, a.seed1, i0 + 2); o.w *= mbvae_drop_scale(a.p1, a.seed1, i0 + 3); \
          }                                                                                            \
          hh[t][4 * gq + 0] = o.x; hh[t][4 * gq + 1] = o.y; hh[t][4 * gq + 2] = o.z; hh[t][4 * gq + 3] = o.w; \
          if (sinb[S]) EF_ST4(hrow + 32 * t + 8 * gq, o);                                              \
        }                                                                                              \
    }                                                                                                  \
    /* output layer: the (ReLU'd) hidden registers are the B operand */                                \
    f32x16 yy[2];                                                                                      \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) yy[u][v] = b2s[32 * u + h4 + (v & 3) + 8 * (v >> 2)]; \
    if (SPLIT) {                                                                                       \
      /* accumulator registers 8g .. 8g+7 of hidden tile t hold units 32t + 16g + 4h + {0..3, 8..11}: one octet */ \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
        _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {                                             \
          const MbvaeSplit3 hs = mbvae_split8(fmaxf(hh[t][8 * gp], 0.f), fmaxf(hh[t][8 * gp + 1], 0.f), fmaxf(hh[t][8 * gp + 2], 0.f), \
                                              fmaxf(hh[t][8 * gp + 3], 0.f), fmaxf(hh[t][8 * gp + 4], 0.f), fmaxf(hh[t][8 * gp + 5], 0.f), \
                                              fmaxf(hh[t][8 * gp + 6], 0.f), fmaxf(hh[t][8 * gp + 7], 0.f)); \
          _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                              \
            const unsigned short* wp = W2p + (32 * u + r) * K2 + 32 * t + 16 * gp + h4;                \
            EF_SPLIT_MFMA(yy[u], ef_ld44(wp), ef_ld44(wp + 64 * K2), ef_ld44(wp + 2 * 64 * K2), hs)    \
          }                                                                                            \
        }                                                                                              \
    } else {                                                                                           \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                                 \
        const float hv = fmaxf(hh[t][v], 0.f);                                                         \
        const float* wcol = W2s + r * L2 + 32 * t + h4 + (v & 3) + 8 * (v >> 2);                    \
        yy[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[0], hv, yy[0], 0, 0, 0);                     \
        yy[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[32 * L2], hv, yy[1], 0, 0, 0);               \
      }                                                                                                \
    }                                                                                                  \
    }                                                                                                  \
    /* residual + LayerNorm: this lane has columns 32u + 8g + 4h + (0..3) of row `tile*32 + r` */      \
    const bool inb = sinb[S];                                                                          \
    float z[2][16];                                                                                    \
    float sum = 0.f;                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
      _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                               \
        const float4 xr = xres[4 * u + gq];                                                            \
        if (TRAIN && a.p2 > 0.f) {                                                                     \
          const unsigned i0 = (unsigned)srow[S] * 64u + 4u * h + 32 * u + 8 * gq;                      \
          yy[u][4 * gq + 0] *= mbvae_drop_scale(a.p2, a.seed2, i0); yy[u][4 * gq + 1] *= mbvae_drop_scale(a.p2, a.seed2, i0 + 1); \
          yy[u][4 * gq + 2] *= mbvae_drop_scale(a.p2, a.seed2, i0 + 2); yy[u][4 * gq + 3] *= mbvae_drop_scale(a.p2, a.seed2, i0 + 3); \
        }                                                                                              \
        z[u][4 * gq + 0] = yy[u][4 * gq + 0] + xr.x; z[u][4 * gq + 1] = yy[u][4 * gq + 1] + xr.y;      \
        z[u][4 * gq + 2] = yy[u][4 * gq + 2] + xr.z; z[u][4 * gq + 3] = yy[u][4 * gq + 3] + xr.w;      \
        if (TRAIN && sinb[S]) {                                                                        \
          float4 zo; zo.x = z[u][4 * gq]; zo.y = z[u][4 * gq + 1]; zo.z = z[u][4 * gq + 2]; zo.w = z[u][4 * gq + 3]; \
          EF_ST4(a.z + (int64_t)srow[S] * 64 + 4 * h + 32 * u + 8 * gq, zo);                           \
        }                                                                                              \
        sum += (z[u][4 * gq] + z[u][4 * gq + 1]) + (z[u][4 * gq + 2] + z[u][4 * gq + 3]);              \
      }                                                                                                \
    sum += __shfl_xor(sum, 32);                                                                        \
    const float mean = sum * (1.f / 64.f);                                                             \
    float sq = 0.f;                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) { z[u][v] -= mean; sq += z[u][v] * z[u][v]; }     \
    sq += __shfl_xor(sq, 32);                                                                          \
    const float rstd = rsqrtf(sq * (1.f / 64.f) + a.eps);                                              \
    float* orow = a.out + (int64_t)srow[S] * 64 + 4 * h;                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                      \
      _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                               \
        const float4 gm = *reinterpret_cast<const float4*>(gs + 32 * u + 8 * gq + h4);              \
        const float4 bt = *reinterpret_cast<const float4*>(bs_ + 32 * u + 8 * gq + h4);             \
        float4 o;                                                                                      \
        o.x = z[u][4 * gq + 0] * rstd * gm.x + bt.x; o.y = z[u][4 * gq + 1] * rstd * gm.y + bt.y;      \
        o.z = z[u][4 * gq + 2] * rstd * gm.z + bt.z; o.w = z[u][4 * gq + 3] * rstd * gm.w + bt.w;      \
        if (inb) *reinterpret_cast<float4*>(orow + 32 * u + 8 * gq) = o;                               \
        if (mode[S] == 1 && r == 0) *reinterpret_cast<float4*>(my_cst + 32 * u + 8 * gq + 4 * h) = o;  \
      }                                                                                                \
    if (mode[S] == 1) asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");   /* the row is in LDS */  \
  }
  FF_FETCH(0, 0)
  for (int64_t q = 0; q < mine; q += 2) {
    FF_FETCH(q + 1, 1)
    FF_TILE(q, 0)
    FF_FETCH(q + 2, 0)
    FF_TILE(q + 1, 1)
  }
#undef FF_FETCH
#undef FF_TILE
}

// ------------------------------------------------------------------------------------------------------------
// Backward data path of the same block in one pass over the rows (training):
//   dh [M,128] = (h > 0 ? dr W2 : 0) * gate_scale        (h = the stored hidden activation: ReLU and dropout gate in one)
//   dx [M,64]  = dh W1 + dres                              (dres = gradient of the residual branch)
// with dr, dres [M,64] (the same tensor when the output dropout is off).  Same transposed products as the forward (rows
// on the lanes): first the hidden units of one 32-unit tile at a time - its accumulators are gated, stored and fed
// straight into the second product as the B operand - so only 16 hidden accumulators are live.  The weight planes in
// LDS are the TRANSPOSES of the forward's (W2^T as [unit][out], W1^T as [in][unit]).  Separate kernels read dr, h, dh and
// dres (2.3 KB per row in all) and write dh, dx; here 1.25 KB are read.  The weight gradients stay with their kernels.
struct EdgeFfnBwdArgs {
  const float* dr; const float* dres; const float* h; const float* W1; const float* W2;
  float* dh; float* dx; int64_t M; float gate_scale;
};

__global__ __launch_bounds__(512) void edge_ffn_bwd_kernel(EdgeFfnBwdArgs a) {
  constexpr int K1 = 72, K2 = 136;                  // bf16 elements per row of the two plane sets
  extern __shared__ __align__(16) float fb_lds[];
  unsigned short* P1 = reinterpret_cast<unsigned short*>(fb_lds);      // [3][128 units][K1]: W2^T
  unsigned short* P2 = P1 + 3 * 128 * K1;                              // [3][64 inputs][K2]: W1^T
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  for (int e = tid; e < 128 * 64; e += 512) {
    ef_store_planes(P1 + (e & 127) * K1 + (e >> 7), 128 * K1, a.W2[e]);      // W2[o = e >> 7][u = e & 127]
    ef_store_planes(P2 + (e & 63) * K2 + (e >> 6), 64 * K2, a.W1[e]);        // W1[u = e >> 6][k = e & 63]
  }
  __syncthreads();
  const int64_t ntiles = a.M / 32;
  const int64_t wid = (int64_t)blockIdx.x * 8 + wave, nw = (int64_t)gridDim.x * 8;
  const int64_t mine = wid < ntiles ? (ntiles - wid + nw - 1) / nw : 0;
  if (mine <= 0) return;
  const float gs = a.gate_scale;
  float4 ra[8];
#define FB_FETCH(q)                                                                                    \
  {                                                                                                   \
    const int64_t row = (wid + min((int64_t)(q), mine - 1) * nw) * 32 + r;                             \
    const float* src = a.dr + row * 64 + 32 * h;                                                       \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) ra[p] = ef_ld4(src + 4 * p);                         \
    asm volatile("" : : : "memory");                                                                   \
  }
  FB_FETCH(0)
  for (int64_t q = 0; q < mine; ++q) {
    const int64_t row = (wid + q * nw) * 32 + r;
    const float* hrow = a.h + row * 128 + 4 * h;
    float* dhrow = a.dh + row * 128 + 4 * h;
    // a zero the optimiser cannot see through (keeps loop-invariant LDS addresses out of registers)
    int oz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
    const int h4 = 4 * h + oz;
    // the second product's accumulators start at the residual branch's gradient (their layout: columns 32u + 8g + 4h + 0..3)
    f32x16 yy[2];
    {
      const float* rrow = a.dres + row * 64 + 4 * h;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 t4 = ef_ld4(rrow + 32 * u + 8 * gq);
          yy[u][4 * gq] = t4.x; yy[u][4 * gq + 1] = t4.y; yy[u][4 * gq + 2] = t4.z; yy[u][4 * gq + 3] = t4.w;
        }
    }
    float4 hg[2][4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) hg[0][gq] = ef_ld4(hrow + 8 * gq);
    asm volatile("" : : : "memory");
    MbvaeSplit3 xs[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float4 t4 = ra[2 * o], u4 = ra[2 * o + 1];
      asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w), "+v"(u4.x), "+v"(u4.y), "+v"(u4.z), "+v"(u4.w) : : "memory");
      xs[o] = mbvae_split8(t4.x, t4.y, t4.z, t4.w, u4.x, u4.y, u4.z, u4.w);
    }
    FB_FETCH(q + 1)                                   // the operand registers are free again: next tile's rows
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < 3) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) hg[(t + 1) & 1][gq] = ef_ld4(hrow + 32 * (t + 1) + 8 * gq);
        asm volatile("" : : : "memory");
      }
      f32x16 hh;
#pragma unroll
      for (int v = 0; v < 16; ++v) hh[v] = 0.f;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const unsigned short* wp = P1 + (32 * t + r) * K1 + 32 * h + 8 * o + oz;     // W2^T[unit r + 32t][out octet]
        EF_SPLIT_MFMA(hh, ef_ld8(wp), ef_ld8(wp + 128 * K1), ef_ld8(wp + 2 * 128 * K1), xs[o])
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 g4 = hg[t & 1][gq];
        float4 o4;
        o4.x = g4.x > 0.f ? hh[4 * gq] * gs : 0.f;     o4.y = g4.y > 0.f ? hh[4 * gq + 1] * gs : 0.f;
        o4.z = g4.z > 0.f ? hh[4 * gq + 2] * gs : 0.f; o4.w = g4.w > 0.f ? hh[4 * gq + 3] * gs : 0.f;
        hh[4 * gq] = o4.x; hh[4 * gq + 1] = o4.y; hh[4 * gq + 2] = o4.z; hh[4 * gq + 3] = o4.w;
        *reinterpret_cast<float4*>(dhrow + 32 * t + 8 * gq) = o4;
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const MbvaeSplit3 hs = mbvae_split8(hh[8 * gp], hh[8 * gp + 1], hh[8 * gp + 2], hh[8 * gp + 3], hh[8 * gp + 4], hh[8 * gp + 5],
                                            hh[8 * gp + 6], hh[8 * gp + 7]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned short* wp = P2 + (32 * u + r) * K2 + 32 * t + 16 * gp + h4;   // W1^T[input r + 32u][unit octet]
          EF_SPLIT_MFMA(yy[u], ef_ld44(wp), ef_ld44(wp + 64 * K2), ef_ld44(wp + 2 * 64 * K2), hs)
        }
      }
    }
    float* orow = a.dx + row * 64 + 4 * h;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(orow + 32 * u + 8 * gq) = make_float4(yy[u][4 * gq], yy[u][4 * gq + 1], yy[u][4 * gq + 2], yy[u][4 * gq + 3]);
  }
#undef FB_FETCH
}

// ------------------------------------------------------------------------------------------------------------
// Inference-only `normE1(E + e_out(P))`: out = LayerNorm(E + mask_ij * (W P + b)) * gamma + beta with P [M,256],
// E, out [M,64].  Same transposed product and in-register LayerNorm as the feed-forward kernel above (rows on the
// lanes, the 64 outputs of a row in the registers of two lanes); the reduction runs over four 64-wide groups of P's
// channels, each group's operands fetched one group ahead.  Saves the round trip of e_out's output and one read of
// E per row.
struct EdgeOutArgs {
  const float* P; const float* W; const float* b; const float* E; const uint8_t* mask; int n;
  const float* gamma; const float* beta; float* out; int64_t M; float eps; const uint8_t* tile_ok;
  const int32_t* row_idx; const int32_t* nvalid;   // packed row order (EfRowOrder), or null
  int uniform;     // the padding rows of E all hold the same values: one is evaluated per wave, the rest only stored
};

template <bool SPLIT>                               // SPLIT: the product on the bf16 matrix pipeline (see edge_ffn_fwd_kernel)
__global__ __launch_bounds__(512) void edge_out_ln_fwd_kernel(EdgeOutArgs a) {
  constexpr int LW = 257;                           // LDS row stride of W [64][256]
  constexpr int KW = 264;                           // SPLIT: bf16 elements per row of the W planes
  extern __shared__ __align__(16) float eo_lds[];
  float* Ws = eo_lds;                               // [64][LW]
  unsigned short* Wp = reinterpret_cast<unsigned short*>(eo_lds);      // SPLIT: [3][64][KW]
  float* bsv = SPLIT ? reinterpret_cast<float*>(Wp + 3 * 64 * KW) : Ws + 64 * LW;   // [64]
  float* gs = bsv + 64;
  float* bt_ = gs + 64;
  float* cst = bt_ + 64;                            // [8][64]: per wave, the output row shared by all padding rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  if (SPLIT) { for (int e = tid; e < 64 * 256; e += 512) ef_store_planes(Wp + (e >> 8) * KW + (e & 255), 64 * KW, a.W[e]); }
  else { for (int e = tid; e < 64 * 256; e += 512) Ws[(e >> 8) * LW + (e & 255)] = a.W[e]; }
  if (tid < 64) { bsv[tid] = a.b ? a.b[tid] : 0.f; gs[tid] = a.gamma[tid]; bt_[tid] = a.beta[tid]; }
  __syncthreads();

  const EfRowOrder ord(a.row_idx, a.nvalid, a.M);
  const int64_t ntiles = ord.ntiles;
  const int64_t wid = (int64_t)blockIdx.x * 8 + wave, nw = (int64_t)gridDim.x * 8;
  const int64_t mine = wid < ntiles ? (ntiles - wid + nw - 1) / nw : 0;
  if (mine <= 0) return;
  const int64_t total = mine * 4;                   // 64-wide reduction groups of this wave
  float4 ra[2][8];
  int srow[2];                                      // this lane's row in the staged tile
  bool sinb[2];
  f32x16 yy[2];
  // A tile marked 0 in tile_ok (or past the valid tiles of the packed order) has no valid pair: its P rows are not
  // read (the operand fetch is pointed at tile 0, which stays in cache, so that the fetch itself remains
  // unconditional) and the products are skipped.  mode: 0 = valid tile; 3 = padding tile, each row normalised on
  // its own; with a.uniform 1 = padding tile evaluated and its output row kept in LDS, 2 = only store the kept row.
  EfTileWindow win;
  int mode[2] = {0, 0};
  int tile_mode = 0;
  bool have_cst = false;
  float* my_cst = cst + wave * 64;
  // rows of the tile being fetched and of the one after it (its index load runs a tile ahead)
  int cur_row = ord.row(wid, r, a.M), nxt_row = ord.row(wid + min((int64_t)1, mine - 1) * nw, r, a.M);
#define EO_FETCH(q, S)                                                                                \
  {                                                                                                   \
    const int64_t qq = min((int64_t)(q), total - 1);                                                   \
    if ((qq & 3) == 0) {                               /* first group of a tile (no loads in here) */  \
      tile_mode = 0;                                                                                   \
      if (!win.ok(a.tile_ok, qq >> 2, wid, nw, mine, lane, ord)) {                                     \
        tile_mode = !a.uniform ? 3 : have_cst ? 2 : 1;                                                 \
        have_cst = have_cst || a.uniform;                                                              \
      }                                                                                                \
    }                                                                                                  \
    mode[S] = tile_mode;                                                                               \
    srow[S] = cur_row;                                                                                 \
    sinb[S] = ord.inb(wid + (qq >> 2) * nw, r, a.M);                                                   \
    const float* src = a.P + (int64_t)(tile_mode == 0 ? cur_row : min(r, (int)a.M - 1)) * 256 + (qq & 3) * 64 + 32 * h; \
    {                                                  /* step to the next group's tile */             \
      const int64_t q1 = min(qq + 1, total - 1);                                                       \
      cur_row = (q1 >> 2) != (qq >> 2) ? nxt_row : cur_row;                                            \
      /* requested BEFORE this group's operands: the memory counter is in order, so the wait for the index (next     \
         fetch) then leaves the operand loads issued after it in flight */                             \
      nxt_row = ord.row(wid + min((q1 >> 2) + 1, mine - 1) * nw, r, a.M);                              \
      asm volatile("" : : : "memory");                                                                 \
    }                                                                                                  \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) ra[S][p] = ef_ld4(src + 4 * p);                      \
    asm volatile("" : : : "memory");                                                                   \
  }
  float4 xres[8];                                   // residual operand of the tile being finished
  unsigned mka = 1u, mkb = 1u;                      // its row's two mask bytes
#define EO_GROUP(q, S, GI)                                                                            \
  if ((q) < total && mode[S] == 2) {                                                                  \
    if (GI == 3) ef_store_row_tile(a.out, my_cst, srow[S], sinb[S], lane);                             \
  } else if ((q) < total) {                                                                           \
    constexpr int gi = GI;                                                                             \
    if (GI == 3) {                                                                                     \
      /* residual row and mask bytes are requested before the last group's products (see the feed-forward kernel) */ \
      const float* xrow = a.E + (int64_t)srow[S] * 64 + 4 * h;                                         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                    \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) xres[4 * u + gq] = ef_ld4(xrow + 32 * u + 8 * gq); \
      mbvae_row_mask_bytes(a.mask, a.mask ? MBVAE_MASK_EDGE : MBVAE_MASK_NONE, srow[S], a.n, a.P, mka, mkb); \
      asm volatile("" : : : "memory");                                                                 \
    }                                                                                                  \
    int oz;          /* opaque zero: keeps the loop-invariant LDS values from being hoisted into registers */ \
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));                                                        \
    const int h4 = 4 * h + oz;                                                                         \
    if (gi == 0) {                                                                                     \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                    \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) yy[u][v] = bsv[32 * u + h4 + (v & 3) + 8 * (v >> 2)]; \
    }                                                                                                  \
    if (SPLIT && mode[S] == 0) {                                                                       \
      _Pragma("unroll") for (int o = 0; o < 4; ++o) {                                                  \
        float4 t4 = ra[S][2 * o], u4 = ra[S][2 * o + 1];                                               \
        asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w), "+v"(u4.x), "+v"(u4.y), "+v"(u4.z), "+v"(u4.w) : : "memory"); \
        const MbvaeSplit3 xs = mbvae_split8(t4.x, t4.y, t4.z, t4.w, u4.x, u4.y, u4.z, u4.w);           \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                \
          const unsigned short* wp = Wp + (32 * u + r) * KW + gi * 64 + 32 * h + 8 * o + oz;           \
          EF_SPLIT_MFMA(yy[u], ef_ld8(wp), ef_ld8(wp + 64 * KW), ef_ld8(wp + 2 * 64 * KW), xs)         \
        }                                                                                              \
      }                                                                                                \
    } else if (mode[S] == 0)                                                                           \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                    \
      float4 t4 = ra[S][p];                                                                            \
      asm volatile("" : "+v"(t4.x), "+v"(t4.y), "+v"(t4.z), "+v"(t4.w) : : "memory");                  \
      const float xs[4] = {t4.x, t4.y, t4.z, t4.w};                                                    \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
        const float* wcol = Ws + r * LW + gi * 64 + 32 * h + 4 * p + e + oz;                           \
        yy[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[0], xs[e], yy[0], 0, 0, 0);                  \
        yy[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcol[32 * LW], xs[e], yy[1], 0, 0, 0);            \
      }                                                                                                \
    }                                                                                                  \
    if (gi == 3) {                                   /* row finished: mask, residual, LayerNorm */     \
      const bool inb = sinb[S];                                                                        \
      const int64_t row = srow[S];                                                                     \
      const bool keep = mode[S] == 0 && mbvae_mask_keep(a.mask ? MBVAE_MASK_EDGE : MBVAE_MASK_NONE, mka, mkb); \
      float z[2][16];                                                                                  \
      float sum = 0.f;                                                                                 \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                    \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                             \
          const float4 xr = xres[4 * u + gq];                                                          \
          /* a select, not a product: P of a masked pair may hold anything (the attention kernel skips it) */ \
          z[u][4 * gq + 0] = (keep ? yy[u][4 * gq + 0] : 0.f) + xr.x; z[u][4 * gq + 1] = (keep ? yy[u][4 * gq + 1] : 0.f) + xr.y; \
          z[u][4 * gq + 2] = (keep ? yy[u][4 * gq + 2] : 0.f) + xr.z; z[u][4 * gq + 3] = (keep ? yy[u][4 * gq + 3] : 0.f) + xr.w; \
          sum += (z[u][4 * gq] + z[u][4 * gq + 1]) + (z[u][4 * gq + 2] + z[u][4 * gq + 3]);            \
        }                                                                                              \
      sum += __shfl_xor(sum, 32);                                                                      \
      const float mean = sum * (1.f / 64.f);                                                           \
      float sq = 0.f;                                                                                  \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                    \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) { z[u][v] -= mean; sq += z[u][v] * z[u][v]; }   \
      sq += __shfl_xor(sq, 32);                                                                        \
      const float rstd = rsqrtf(sq * (1.f / 64.f) + a.eps);                                            \
      float* orow = a.out + row * 64 + 4 * h;                                                          \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                    \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                             \
          const float4 gm = *reinterpret_cast<const float4*>(gs + 32 * u + 8 * gq + h4);               \
          const float4 bt = *reinterpret_cast<const float4*>(bt_ + 32 * u + 8 * gq + h4);              \
          float4 o;                                                                                    \
          o.x = z[u][4 * gq + 0] * rstd * gm.x + bt.x; o.y = z[u][4 * gq + 1] * rstd * gm.y + bt.y;    \
          o.z = z[u][4 * gq + 2] * rstd * gm.z + bt.z; o.w = z[u][4 * gq + 3] * rstd * gm.w + bt.w;    \
          if (inb) *reinterpret_cast<float4*>(orow + 32 * u + 8 * gq) = o;                             \
          if (mode[S] == 1 && r == 0) *reinterpret_cast<float4*>(my_cst + 32 * u + 8 * gq + 4 * h) = o; \
        }                                                                                              \
      if (mode[S] == 1) asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");   /* the row is in LDS */ \
    }                                                                                                  \
  }
  EO_FETCH(0, 0)
  for (int64_t q = 0; q < total; q += 4) {          // total is a multiple of 4: the group index is a constant per line
    EO_FETCH(q + 1, 1)
    EO_GROUP(q, 0, 0)
    EO_FETCH(q + 2, 0)
    EO_GROUP(q + 1, 1, 1)
    EO_FETCH(q + 3, 1)
    EO_GROUP(q + 2, 0, 2)
    EO_FETCH(q + 4, 0)
    EO_GROUP(q + 3, 1, 3)
  }
#undef EO_FETCH
#undef EO_GROUP
}

static int ef_num_cus() { return mbvae_num_cus(); }

extern "C" {

// E [bs,n,n,de] dense, W12 [2dx,de] (rows 0..dx-1: e_mul, dx..2dx-1: e_add), b12 [2dx] or null; Q,K,V [bs,n,dx]
// rows ldq floats apart; ye_add, ye_mul [bs,dx]; mask [bs,n].  Outputs P [bs,n,n,dx], wV, lse [bs,n,dx] dense, and,
// when E12 is not null, the projection output E1|E2 [bs,n,n,2dx] (what the backward kernels read).
// work_counter: NULL, or one int32 of device memory the call may overwrite (it is zeroed on `stream` first): the
// workgroups then draw (graph, query-row block) tasks from it in order instead of a fixed round robin.
// node_off: NULL, or int32 [bs+1] (inference form only, masks that keep a prefix of every graph's nodes): Q, K, V, wV
// and lse then hold the valid nodes only, graph b's nodes 0..nb-1 at rows node_off[b] .. node_off[b+1]-1.
// graph_order: NULL, or a permutation of 0..bs-1 (int32): the order in which the graphs' tasks are handed out (largest
// first, together with work_counter, gives every workgroup a similar amount of work whatever the batch order).
// MBVAE_EUNSUPPORTED unless de == 64, dx == 256, n <= 64.
static int edge_fused_fwd_impl(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                               const float* V, int64_t ldq, const float* ye_add, const float* ye_mul, const uint8_t* mask,
                               float* P, float* wV, float* lse, float* E12, int32_t* work_counter, const int32_t* node_off,
                               const int32_t* graph_order, int64_t bs, int64_t n, int64_t de, int64_t dx, float scale,
                               hipStream_t stream, bool bf16out, bool train_form = false) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (bs <= 0 || n <= 0) return MBVAE_EINVAL;
  if (de != 64 || dx != 256 || n > 64 || bs > 0x7fffffff / 8) return MBVAE_EUNSUPPORTED;
  if (ldq < dx || (ldq & 3) || ldq > 0x7fffffff) return MBVAE_EINVAL;
  const void* ptrs[] = {E, W12, Q, K, V, ye_add, ye_mul, P, wV, lse};
  for (const void* p : ptrs) if (!p || ((uintptr_t)p & 15)) return MBVAE_EINVAL;
  if (!mask || ((uintptr_t)E12 & 15) || ((uintptr_t)b12 & 3)) return MBVAE_EINVAL;
  if (node_off && E12) return MBVAE_EINVAL;        // packed node rows: inference form only
  if (mbvae_opt(MBVAE_OPT_NO_EDGE_FUSED)) return MBVAE_EUNSUPPORTED;
  // tasks = graphs x blocks of query rows; pick the split that fills the last round of workgroups best
  const int cus = ef_num_cus();
  int ichunks = 1;
  double best = 0.0;
  for (int cand = 1; cand <= 8 && cand <= n; cand *= 2) {
    const int64_t tasks = bs * cand;
    const double eff = (double)tasks / (double)(((tasks + cus - 1) / cus) * cus);
    if (eff > best + 0.02) { best = eff; ichunks = cand; }
  }
  if (work_counter) {
    // dynamic order: about four tasks per workgroup are enough to even out a ragged batch
    ichunks = 1;
    while (ichunks < 8 && 2 * ichunks <= n && bs * ichunks < 4 * (int64_t)cus) ichunks *= 2;
    if (((uintptr_t)work_counter & 3) || hipMemsetAsync(work_counter, 0, sizeof(int32_t), stream) != hipSuccess)
      return MBVAE_EINVAL;
  }
  EdgeFusedArgs a{E, W12, b12, Q, K, V, (int)ldq, ye_add, ye_mul, mask, P, wV, lse, E12, (int)bs, (int)n, scale, ichunks,
                  work_counter, node_off, graph_order, 1};
  if (const int sg = mbvae_opt(MBVAE_OPT_EF_STAGGER); sg > 0) a.stagger = sg == 3 ? 0 : sg;
  const size_t smem = sizeof(float) * 2 * 64 * 256 + 3 * 512 * 16;    // K, V rows + three thread-private weight-plane slots
  const bool split = !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA);
  const bool wr = E12 != nullptr || train_form;    // training form: P (and E1|E2 when given) for every pair
  void (*k)(EdgeFusedArgs);
  bool lds_ok;
  const bool coop = !mbvae_opt(MBVAE_OPT_NO_EF_COOP);
#define EF_PICK(...) { k = __VA_ARGS__; lds_ok = MBVAE_ALLOW_BIG_LDS((__VA_ARGS__)); }
  if (bf16out) {
    if (!wr) return MBVAE_EINVAL;                    // bf16 storage is a training-form feature
    if (coop) EF_PICK(edge_fused_fwd_kernel<true, true, true, true>) else EF_PICK(edge_fused_fwd_kernel<true, true, true>)
  } else if (split && coop) {
    if (wr) EF_PICK(edge_fused_fwd_kernel<true, true, false, true>) else EF_PICK(edge_fused_fwd_kernel<false, true, false, true>)
  } else if (split) {
    if (wr) EF_PICK(edge_fused_fwd_kernel<true, true>) else EF_PICK(edge_fused_fwd_kernel<false, true>)
  } else {
    if (wr) EF_PICK(edge_fused_fwd_kernel<true, false>) else EF_PICK(edge_fused_fwd_kernel<false, false>)
  }
#undef EF_PICK
  if (!lds_ok) return MBVAE_ELAUNCH;
  const int grid = (int)std::min<int64_t>(bs * ichunks, cus);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, a);
  return mbvae_launch_status();
}

int mbvae_edge_fused_fwd(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                         const float* V, int64_t ldq, const float* ye_add, const float* ye_mul, const uint8_t* mask,
                         float* P, float* wV, float* lse, float* E12, int32_t* work_counter, const int32_t* node_off,
                         const int32_t* graph_order, int64_t bs, int64_t n, int64_t de, int64_t dx, float scale,
                         hipStream_t stream) {
  return edge_fused_fwd_impl(E, W12, b12, Q, K, V, ldq, ye_add, ye_mul, mask, P, wV, lse, E12, work_counter, node_off,
                             graph_order, bs, n, de, dx, scale, stream, false);
}

// Training form with bf16 storage: P [bs,n,n,dx] and E12 [bs,n,n,2dx] (required) are written as bf16, everything else
// as above (static task order, no packed rows).
int mbvae_edge_fused_fwd_bf16(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                              const float* V, int64_t ldq, const float* ye_add, const float* ye_mul, const uint8_t* mask,
                              uint16_t* P, float* wV, float* lse, uint16_t* E12, int64_t bs, int64_t n, int64_t de,
                              int64_t dx, float scale, hipStream_t stream) {
  return edge_fused_fwd_impl(E, W12, b12, Q, K, V, ldq, ye_add, ye_mul, mask, (float*)P, wV, lse, (float*)E12, nullptr,
                             nullptr, nullptr, bs, n, de, dx, scale, stream, true);
}

// Training-form forward WITHOUT the E1|E2 output (P for every pair, wV, lse): to be paired with
// mbvae_edge_fused_bwd, which recomputes the projection.
int mbvae_edge_fused_fwd_recompute(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                                   const float* V, int64_t ldq, const float* ye_add, const float* ye_mul,
                                   const uint8_t* mask, float* P, float* wV, float* lse, int64_t bs, int64_t n, int64_t de,
                                   int64_t dx, float scale, hipStream_t stream) {
  // the recomputing backward runs on the bf16 pipeline only, one workgroup per graph
  if (mbvae_opt(MBVAE_OPT_NO_E12_RECOMPUTE) || mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA) || 2 * bs < ef_num_cus()) return MBVAE_EUNSUPPORTED;
  return edge_fused_fwd_impl(E, W12, b12, Q, K, V, ldq, ye_add, ye_mul, mask, P, wV, lse, nullptr, nullptr, nullptr,
                             nullptr, bs, n, de, dx, scale, stream, false, true);
}

// Gradients of the fused block from E itself (the projection output is recomputed, not read): given dP [bs,n,n,dx]
// and dwV [bs,n,dx] writes dE12 = dE1|dE2 [bs,n,n,2dx] (the gradient w.r.t. the projection output, for the
// projection's input / weight gradient products), dQ, dK, dV (rows ldq apart) and dye_add, dye_mul [bs,dx].
// One workgroup per graph: meant for batches of at least about one graph per compute unit.
// MBVAE_EUNSUPPORTED unless de == 64, dx == 256, n <= 64.  Bitwise reproducible.
static int edge_fused_bwd_impl(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                               const float* V, int64_t ldq, const float* ye_mul, const uint8_t* mask, const float* wV,
                               const float* lse, const void* dP, const float* dwV, void* dE12, float* dQ, float* dK, float* dV,
                               float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t de, int64_t dx, float scale,
                               hipStream_t stream, bool bf16) {
  (void)hipGetLastError();
  if (bs <= 0 || n <= 0) return MBVAE_EINVAL;
  if (de != 64 || dx != 256 || n > 64 || bs > 0x7fffffff / 8) return MBVAE_EUNSUPPORTED;
  if (ldq < dx || (ldq & 3) || ldq > 0x7fffffff) return MBVAE_EINVAL;
  const void* ptrs[] = {E, W12, Q, K, V, ye_mul, wV, lse, dP, dwV, dE12, dQ, dK, dV, dye_add, dye_mul};
  for (const void* p : ptrs) if (!p || ((uintptr_t)p & 15)) return MBVAE_EINVAL;
  if (!mask || ((uintptr_t)b12 & 3)) return MBVAE_EINVAL;
  if (mbvae_opt(MBVAE_OPT_NO_EDGE_FUSED)) return MBVAE_EUNSUPPORTED;
  const bool coop = !mbvae_opt(MBVAE_OPT_NO_EFB_COOP);
  constexpr int kLds = 20 * 512 * 16;                // 20 thread-private 16-byte slots = the CU's whole 160 KB (per-wave form)
  constexpr int kLdsCoop = 16 * 512 * 16 + 2 * 12288;   // 16 weight slots + two operand tile buffers
  {
    static std::atomic<unsigned long long> done{0ull};   // per-device flag: the attribute is raised once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MBVAE_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      if (hipFuncSetAttribute((const void*)edge_fused_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess ||
          hipFuncSetAttribute((const void*)edge_fused_bwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess ||
          hipFuncSetAttribute((const void*)edge_fused_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess ||
          hipFuncSetAttribute((const void*)edge_fused_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess)
        return MBVAE_ELAUNCH;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  EdgeFusedBwdArgs a{E, W12, b12, Q, K, V, (int)ldq, ye_mul, mask, wV, lse, (const float*)dP, dwV, (float*)dE12, dQ, dK, dV,
                     dye_add, dye_mul, (int)bs, (int)n, scale, 7, 0};   // every 8 rows: measured 2: 7.26, 4: 7.28, 8: 6.95, never: 8.17 ms per step;
                                                                    // meeting half a tile apart (stagger) costs 2 % here, gains 4 % in the forward
  if (const int sy = mbvae_opt(MBVAE_OPT_EFB_SYNC); sy > 0) a.sync_mask = sy - 1;
  if (const int sg = mbvae_opt(MBVAE_OPT_EF_STAGGER); sg > 0) a.stagger = sg == 3 ? 0 : sg;
  const int grid = (int)std::min<int64_t>(bs, ef_num_cus());
  auto k = coop ? (bf16 ? edge_fused_bwd_kernel<true, true> : edge_fused_bwd_kernel<false, true>)
                : (bf16 ? edge_fused_bwd_kernel<true, false> : edge_fused_bwd_kernel<false, false>);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), coop ? kLdsCoop : kLds, stream, a);
  return mbvae_launch_status();
}

int mbvae_edge_fused_bwd(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                         const float* V, int64_t ldq, const float* ye_mul, const uint8_t* mask, const float* wV,
                         const float* lse, const float* dP, const float* dwV, float* dE12, float* dQ, float* dK, float* dV,
                         float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t de, int64_t dx, float scale,
                         hipStream_t stream) {
  return edge_fused_bwd_impl(E, W12, b12, Q, K, V, ldq, ye_mul, mask, wV, lse, dP, dwV, dE12, dQ, dK, dV, dye_add, dye_mul, bs, n,
                             de, dx, scale, stream, false);
}

// The recomputing pair in the bf16 storage mode: P (forward), dP and dE1|dE2 (backward) are bf16 in memory.
int mbvae_edge_fused_fwd_recompute_bf16(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                                        const float* V, int64_t ldq, const float* ye_add, const float* ye_mul,
                                        const uint8_t* mask, uint16_t* P, float* wV, float* lse, int64_t bs, int64_t n,
                                        int64_t de, int64_t dx, float scale, hipStream_t stream) {
  if (mbvae_opt(MBVAE_OPT_NO_E12_RECOMPUTE) || 2 * bs < ef_num_cus()) return MBVAE_EUNSUPPORTED;
  return edge_fused_fwd_impl(E, W12, b12, Q, K, V, ldq, ye_add, ye_mul, mask, (float*)P, wV, lse, nullptr, nullptr, nullptr,
                             nullptr, bs, n, de, dx, scale, stream, true, true);
}
int mbvae_edge_fused_bwd_bf16(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                              const float* V, int64_t ldq, const float* ye_mul, const uint8_t* mask, const float* wV,
                              const float* lse, const uint16_t* dP, const float* dwV, uint16_t* dE12, float* dQ, float* dK,
                              float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t de, int64_t dx,
                              float scale, hipStream_t stream) {
  return edge_fused_bwd_impl(E, W12, b12, Q, K, V, ldq, ye_mul, mask, wV, lse, dP, dwV, dE12, dQ, dK, dV, dye_add, dye_mul, bs, n,
                             de, dx, scale, stream, true);
}

// out [M,64] = LayerNorm(x + W2 relu(W1 x + b1) + b2) * gamma + beta with x [M,64], W1 [128,64], W2 [64,128]
// (torch Linear layouts), b1 / b2 may be NULL.  Forward only (no dropout): the inference form of the edge
// feed-forward block.  MBVAE_EUNSUPPORTED unless d == 64 and dff == 128.
int mbvae_edge_ffn_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* gamma, const float* beta, float* out, const uint8_t* tile_ok,
                       const int32_t* row_idx, const int32_t* nvalid, int64_t M, int64_t d, int64_t dff, float eps,
                       hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M <= 0) return MBVAE_EINVAL;
  if (d != 64 || dff != 128 || M > 0x7fffffff || mbvae_opt(MBVAE_OPT_NO_EDGE_FFN)) return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {x, W1, W2, gamma, beta, out};
  for (const void* p : ptrs) if (!p || ((uintptr_t)p & 15)) return MBVAE_EINVAL;
  if ((row_idx == nullptr) != (nvalid == nullptr) || (row_idx && M < 32)) return MBVAE_EINVAL;
  EdgeFfnArgs a{x, W1, b1, W2, b2, gamma, beta, out, M, eps, tile_ok, row_idx, nvalid, nullptr, nullptr, 0.f, 0.f, 0, 0};
  const bool split = !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA);
  const size_t smem = split ? 2 * 3 * (128 * 72 + 64 * 136) + sizeof(float) * (128 + 3 * 64 + 8 * 64)
                            : sizeof(float) * (128 * 65 + 64 * 129 + 128 + 3 * 64 + 8 * 64);
  if (!(split ? MBVAE_ALLOW_BIG_LDS(edge_ffn_fwd_kernel<true>) : MBVAE_ALLOW_BIG_LDS(edge_ffn_fwd_kernel<false>))) return MBVAE_ELAUNCH;
  const int grid = (int)std::min<int64_t>(((M + 31) / 32 + 1 + 7) / 8, ef_num_cus());   // (+1: the packed order may add a tile)
  hipLaunchKernelGGL(split ? edge_ffn_fwd_kernel<true> : edge_ffn_fwd_kernel<false>, dim3(grid), dim3(512), smem, stream, a);
  return mbvae_launch_status();
}

// Training form of the same block, one pass over the rows:
//   hidden [M,128] = dropout_p1(relu(x W1^T + b1))                      (stored for the backward)
//   z      [M,64]  = x + dropout_p2(hidden W2^T + b2)                   (stored for the backward: the LayerNorm's input)
//   out    [M,64]  = LayerNorm(z) * gamma + beta
// Dropout keep-scales are hashed from (seed, row * width + column) exactly as mbvae_linear_fwd / mbvae_linear_ln_fwd
// hash them, so the backward entry points of those (mbvae_ln_sum_bwd, mbvae_linear_bwd_input_outgate, ...) apply
// unchanged.  Per row 256 B are read and 1 KB written; the two separate kernels read 1 KB and write 1 KB.
// MBVAE_EUNSUPPORTED unless d == 64, dff == 128, M a multiple of 32 with M * 128 < 2^32.
int mbvae_edge_ffn_fwd_train(const float* x, const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* gamma, const float* beta, float* hidden, float* z, float* out, int64_t M,
                             int64_t d, int64_t dff, float eps, float p1, uint64_t seed1, float p2, uint64_t seed2,
                             hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M <= 0 || !(p1 >= 0.f && p1 < 1.f) || !(p2 >= 0.f && p2 < 1.f)) return MBVAE_EINVAL;
  if (d != 64 || dff != 128 || (M & 31) || M * 128 > 0xffffffffll || mbvae_opt(MBVAE_OPT_NO_EDGE_FFN_TRAIN)) return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {x, W1, W2, gamma, beta, hidden, z, out};
  for (const void* p : ptrs) if (!p || ((uintptr_t)p & 15)) return MBVAE_EINVAL;
  EdgeFfnArgs a{x, W1, b1, W2, b2, gamma, beta, out, M, eps, nullptr, nullptr, nullptr, hidden, z, p1, p2, seed1, seed2};
  const bool split = !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA);
  const size_t smem = split ? 2 * 3 * (128 * 72 + 64 * 136) + sizeof(float) * (128 + 3 * 64 + 8 * 64)
                            : sizeof(float) * (128 * 65 + 64 * 129 + 128 + 3 * 64 + 8 * 64);
  if (!(split ? MBVAE_ALLOW_BIG_LDS((edge_ffn_fwd_kernel<true, true>)) : MBVAE_ALLOW_BIG_LDS((edge_ffn_fwd_kernel<false, true>)))) return MBVAE_ELAUNCH;
  const int grid = (int)std::min<int64_t>((M / 32 + 7) / 8, ef_num_cus());
  if (split) hipLaunchKernelGGL((edge_ffn_fwd_kernel<true, true>), dim3(grid), dim3(512), smem, stream, a);
  else hipLaunchKernelGGL((edge_ffn_fwd_kernel<false, true>), dim3(grid), dim3(512), smem, stream, a);
  return mbvae_launch_status();
}

// Backward data path of that block in one pass: dh [M,128] = (h > 0 ? dr W2 : 0) * gate_scale, dx [M,64] = dh W1 + dres.
// dr / dres [M,64]: the gradient entering the second Linear's output and the one of the residual branch (what
// mbvae_ln_sum_bwd returns; the same pointer when the output dropout is off); h [M,128]: the hidden activation stored by
// the forward (its sign pattern is the ReLU-and-dropout gate; gate_scale = 1 / (1 - p1)).  W1 [128,64], W2 [64,128].
// MBVAE_EUNSUPPORTED unless d == 64, dff == 128, M a multiple of 32 (and with the fp32-MFMA switch on): callers then use
// mbvae_linear_bwd_input_outgate and mbvae_linear_bwd_input_add.
int mbvae_edge_ffn_bwd_data(const float* dr, const float* dres, const float* h, const float* W1, const float* W2,
                            float* dh, float* dx, int64_t M, int64_t d, int64_t dff, float gate_scale, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M <= 0) return MBVAE_EINVAL;
  if (d != 64 || dff != 128 || (M & 31) || M * 128 > 0xffffffffll || mbvae_opt(MBVAE_OPT_NO_EDGE_FFN_TRAIN) ||
      mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA))
    return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {dr, dres, h, W1, W2, dh, dx};
  for (const void* p : ptrs) if (!p || ((uintptr_t)p & 15)) return MBVAE_EINVAL;
  EdgeFfnBwdArgs a{dr, dres, h, W1, W2, dh, dx, M, gate_scale};
  const size_t smem = 2 * 3 * (128 * 72 + 64 * 136);
  if (!MBVAE_ALLOW_BIG_LDS(edge_ffn_bwd_kernel)) return MBVAE_ELAUNCH;
  const int grid = (int)std::min<int64_t>((M / 32 + 7) / 8, ef_num_cus());
  hipLaunchKernelGGL(edge_ffn_bwd_kernel, dim3(grid), dim3(512), smem, stream, a);
  return mbvae_launch_status();
}

// out [M,64] = LayerNorm(E + keep(row) * (P W^T + b)) * gamma + beta with P [M,256], W [64,256], E [M,64]; keep(row)
// is the edge mask of row (b,i,j) (mask [bs,n] with M = bs*n*n; NULL: no masking).  Forward only, no dropout: the
// inference form of e_out followed by normE1.  MBVAE_EUNSUPPORTED unless dx == 256, de == 64.
int mbvae_edge_out_ln_fwd(const float* P, const float* W, const float* b, const float* E, const uint8_t* mask,
                          int64_t n, const float* gamma, const float* beta, float* out, const uint8_t* tile_ok,
                          const int32_t* row_idx, const int32_t* nvalid, int uniform_padding, int64_t M, int64_t dx,
                          int64_t de, float eps, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M <= 0 || (mask && n <= 0)) return MBVAE_EINVAL;
  if (dx != 256 || de != 64 || M > 0x7fffffff || mbvae_opt(MBVAE_OPT_NO_EDGE_FFN)) return MBVAE_EUNSUPPORTED;
  const void* ptrs[] = {P, W, E, gamma, beta, out};
  for (const void* p : ptrs) if (!p || ((uintptr_t)p & 15)) return MBVAE_EINVAL;
  if ((row_idx == nullptr) != (nvalid == nullptr) || (row_idx && M < 32)) return MBVAE_EINVAL;
  EdgeOutArgs a{P, W, b, E, mask, (int)n, gamma, beta, out, M, eps, tile_ok, row_idx, nvalid, uniform_padding};
  const bool split = !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA);
  const size_t smem = split ? 2 * 3 * 64 * 264 + sizeof(float) * (3 * 64 + 8 * 64) : sizeof(float) * (64 * 257 + 3 * 64 + 8 * 64);
  if (!(split ? MBVAE_ALLOW_BIG_LDS(edge_out_ln_fwd_kernel<true>) : MBVAE_ALLOW_BIG_LDS(edge_out_ln_fwd_kernel<false>))) return MBVAE_ELAUNCH;
  const int grid = (int)std::min<int64_t>(((M + 31) / 32 + 1 + 7) / 8, ef_num_cus());   // (+1: the packed order may add a tile)
  hipLaunchKernelGGL(split ? edge_out_ln_fwd_kernel<true> : edge_out_ln_fwd_kernel<false>, dim3(grid), dim3(512), smem, stream, a);
  return mbvae_launch_status();
}

}  // extern "C"
