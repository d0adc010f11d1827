// fp32 projection GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
//
// One kernel template serves the three products a Linear layer needs
//   forward        y  = epi(x  @ W^T + b)      "NT": A = x [M,K] (k contiguous),  B = W [N,K] (k contiguous)
//   input grad     dx = pro(dy) @ W            "NN": A = dy [M,K'] (k contiguous), B = W [K',N] (n contiguous)
//   weight grad    dW = pro(dy)^T @ x, db      "TN": A = dy [R,M'] (m contiguous), B = x [R,N'] (n contiguous), R = rows
// with the reference's masking / ReLU / dropout folded into the operand prologue and the epilogue,
// so the [bs*n*n, d] edge tensors are never touched by a stand-alone elementwise pass.
//
// Tiling: 256 threads = 4 waves, BK = 32.  Operand tiles live in LDS k-major (As[k][m], Bs[k][n]) so
// that the MFMA operand fetch (lane l -> A[m = l&31][k = l>>5]) is a stride-1, conflict-free
// ds_read_b32 for each 32-lane half.  Operands whose reduction index is contiguous in memory are
// transposed on the way into LDS with row stride == 1 (mod 32) (conflict-free scatter); operands
// that are already m/n-contiguous are copied with 16-byte LDS stores.  The next k-tile is fetched
// into registers while the current one is multiplied.  Tall reductions (weight grads over bs*n*n
// rows, small-batch y-path products) are split over blockIdx.z into a workspace and summed by a
// second kernel that also applies the epilogue, which keeps results bitwise reproducible.
#include "common.h"
// The kernels of the node / global-feature path raise their waves' issue priority: in training they run on a second stream
// beside the long edge-tensor kernels, whose waves otherwise win every issue slot they ask for; the edge stream has slack,
// the node / global stream is what the step waits for (measured: 50.4-50.6 against 50.6-50.9 ms per step; no effect when
// a kernel has the GPU to itself).
#define MBVAE_RAISE_PRIO() __builtin_amdgcn_s_setprio(3)
#include <algorithm>
#include <cstdlib>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
// Streaming operand load for kernels whose wave-wide access covers whole cache lines that nobody touches again
// (the chunk copies of the weight-gradient kernel): the non-temporal hint keeps them from occupying L2.  NOT for
// accesses where a lane walks along its own row (the line is re-touched by the lane's next load and must stay in L1).
typedef float gemm_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
  const gemm_v4f v = __builtin_nontemporal_load(reinterpret_cast<const gemm_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  // prologue on A: a = (gate > 0 ? a : 0) * gate_scale * keep(row)
  const float* a_gate; float a_gate_scale;
  const uint8_t* mask; int a_mask_mode; int c_mask_mode; int n_nodes;
  // epilogue on C: +bias[n], relu, dropout, row mask, (+= C)
  const float* bias; int act; float drop_p; uint64_t seed; int accumulate;
  // TN only: column sums of pro(A) (bias gradient)
  float* colsum;
  // split-K
  int k_per_split; float* partial; float* partial_colsum;
  // optional addend with C's layout (C = epi(...) + addend): the wave-autonomous tall-skinny kernel and the tiled kernel
  // (epilogue, or the split reduction's) apply it
  const float* addend;
  // optional output gate with C's layout (C = c_gate > 0 ? epi(...) * c_gate_scale : 0): the ReLU / dropout gate of the
  // layer BELOW applied where this layer's input gradient is produced; only the wave-autonomous tall-skinny kernel
  const float* c_gate; float c_gate_scale;
  // reduced-precision storage: the operand / output is bf16 in memory (only the bf16-pipeline edge kernels honour these)
  int a_bf16, b_bf16, c_bf16;
  // wave-autonomous tall-skinny kernel only: columns per workgroup column group (blockIdx.y picks the group); 0 = all N
  int col_group;
  // LayerNorm epilogue of the wave-autonomous kernel (N == 64): z = addend + dropout(rowmask(x W^T + b)) goes to ln_z
  // and LayerNorm(z) * ln_gamma + ln_beta to C
  const float* ln_gamma; const float* ln_beta; float* ln_z; float ln_eps;
};

// Sum over the 32 lanes of a half-wave (every lane receives it): four DPP steps inside each row of 16 lanes
// (two quad permutes, half-row mirror, row mirror) and one swizzle that pairs the two rows of the half.
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));                     // lane ^ 16
  return v;
}

__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, float v, int row, int col, float keep_row) {
  if (g.bias) v += g.bias[col];
  if (g.act == 1) v = fmaxf(v, 0.f);
  if (g.drop_p > 0.f) v *= mbvae_drop_scale(g.drop_p, g.seed, (uint64_t)row * (uint64_t)g.ldc + col);
  v = keep_row != 0.f ? v : 0.f;      // a select: a masked row may hold non-finite values
  if (g.accumulate) v += g.C[(int64_t)row * g.ldc + col];
  if (g.addend) v += g.addend[(int64_t)row * g.ldc + col];
  return v;
}

// FAST: every tile is full and 16-byte aligned (M % BM == 0, N % BN == 0, k range % 32 == 0, ld % 4 == 0), so the
// staging loads are unconditional float4 loads with no bounds logic; GATE: the A prologue has a gate operand.
// SPLIT (with FAST): the products run on the bf16 matrix pipeline (six bf16 MFMAs per fp32 product, see mbvae_split8).
// Every thread stages whole operand OCTETS - eight consecutive reduction indices of one row / column: two 16-byte loads
// where the reduction index is contiguous in memory, eight coalesced 4-byte loads where it is the row index - splits them
// once and parks the three bf16 planes in LDS, row-major with the reduction index contiguous (rows padded to 80 bytes:
// the 16-byte operand fetches of the MFMA loop and the 16-byte stores of the staging are then spread over all banks).
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool FAST, bool GATE, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  MBVAE_RAISE_PRIO();
  constexpr int BK = 32;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int LDAS = A_KC ? BM + 1 : BM + 4;
  constexpr int LDBS = B_KC ? BN + 1 : BN + 4;
  constexpr int RA = BM * BK / 256, RB = BN * BK / 256;   // staged floats per thread
  constexpr int KP = BK + 8;                               // SPLIT: bf16 elements per plane row
  static_assert(WM * WN == 4, "4 waves");
  static_assert(!SPLIT || (FAST && BM % 64 == 0 && BN % 64 == 0), "the bf16-pipeline form takes whole aligned tiles");
  __shared__ __align__(16) float smem[SPLIT ? 3 * (BM + BN) * KP / 2 : BK * LDAS + BK * LDBS];
  __shared__ float csum4[SPLIT ? 4 * BM : 1];     // SPLIT: column-sum partials of the four octet groups
  __shared__ float keepM[BM];        // row-mask flags of the tile's M rows (epilogue mask, or A prologue when A_KC)
  __shared__ float keepK[2][BK];     // row-mask flags of the current k rows (A prologue when !A_KC), double-buffered
  float* As = smem;
  float* Bs = smem + BK * LDAS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps index arithmetic on the scalar unit
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroups are dealt round-robin over the 8 XCDs (each with a private L2).  Remap the linear block id so that
  // the N-tiles of one M-tile (which re-read the same A rows) run on ONE XCD, back to back: speed only.
  const int tiles_n = (g.N + BN - 1) / BN;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, per = nt / 8, rem = nt % 8, xcd = tile % 8, k = tile / 8;
    tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + k;     // bijective for any grid size
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const bool vecA = (((uintptr_t)g.A & 15) == 0) && ((g.lda & 3) == 0) &&
                    (g.a_gate == nullptr || ((uintptr_t)g.a_gate & 15) == 0);
  const bool vecB = (((uintptr_t)g.B & 15) == 0) && ((g.ldb & 3) == 0);
  {
    const int mode_m = A_KC ? (g.a_mask_mode | g.c_mask_mode) : g.c_mask_mode;
    if (tid < BM) keepM[tid] = (m0 + tid < g.M) ? mbvae_row_keep(g.mask, mode_m, m0 + tid, g.n_nodes) : 0.f;
  }
  __syncthreads();

  // Staging registers: raw operand words (and gate words) of the NEXT k-tile.  fetch() only issues the global
  // loads; the prologue arithmetic (gate, row mask) runs in commit(), one k-tile later, so the loads stay in
  // flight underneath the MFMA loop instead of being waited for right after issue.
  float ra[RA], rb[RB], rg[RA];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  float colsum = 0.f;
  constexpr bool has_gate = GATE;

  // element (row, col) of a [rows x cols] row-major operand tile slot: 4 consecutive floats starting at `off`,
  // valid while col + q < col_end; rows beyond row_end give zeros.
  auto load4 = [&](const float* base, bool vec, int64_t off, bool row_ok, int col, int col_end, float* dst) {
    if (FAST) {
      const float4 t = *reinterpret_cast<const float4*>(base + off);
      dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
      return;
    }
    dst[0] = dst[1] = dst[2] = dst[3] = 0.f;
    if (!row_ok) return;
    if (vec && col + 3 < col_end) {
      const float4 t = *reinterpret_cast<const float4*>(base + off);
      dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (col + q < col_end) dst[q] = base[off + q];
    }
  };

  auto fetch = [&](int kt) {
    if (!A_KC && g.a_mask_mode != MBVAE_MASK_NONE && tid < BK)
      keepK[(kt >> 5) & 1][tid] = (kt + tid < kend) ? mbvae_row_keep(g.mask, g.a_mask_mode, kt + tid, g.n_nodes) : 0.f;
    if (SPLIT) {
      // octet (row / column, o): A_KC: row tid >> 2 (+ 64 p), octet tid & 3; else column tid & 63 (+ 64 p), octet tid >> 6
#pragma unroll
      for (int p = 0; p < BM / 64; ++p) {
        if (A_KC) {
          const int row = min(m0 + (tid >> 2) + 64 * p, g.M - 1);          // (a ragged last M tile re-reads row M-1)
          const int64_t off = (int64_t)row * g.lda + kt + 8 * (tid & 3);
          load4(g.A, true, off, true, 0, 4, &ra[p * 8]); load4(g.A, true, off + 4, true, 0, 4, &ra[p * 8 + 4]);
          if (has_gate) { load4(g.a_gate, true, off, true, 0, 4, &rg[p * 8]); load4(g.a_gate, true, off + 4, true, 0, 4, &rg[p * 8 + 4]); }
        } else {
          const int64_t off = (int64_t)(kt + 8 * (tid >> 6)) * g.lda + m0 + (tid & 63) + 64 * p;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ra[p * 8 + i] = g.A[off + (int64_t)i * g.lda];
            if (has_gate) rg[p * 8 + i] = g.a_gate[off + (int64_t)i * g.lda];
          }
        }
      }
#pragma unroll
      for (int p = 0; p < BN / 64; ++p) {
        if (B_KC) {
          const int64_t off = (int64_t)(n0 + (tid >> 2) + 64 * p) * g.ldb + kt + 8 * (tid & 3);
          load4(g.B, true, off, true, 0, 4, &rb[p * 8]); load4(g.B, true, off + 4, true, 0, 4, &rb[p * 8 + 4]);
        } else {
          const int64_t off = (int64_t)(kt + 8 * (tid >> 6)) * g.ldb + n0 + (tid & 63) + 64 * p;
#pragma unroll
          for (int i = 0; i < 8; ++i) rb[p * 8 + i] = g.B[off + (int64_t)i * g.ldb];
        }
      }
      return;
    }
    if (A_KC) {
#pragma unroll
      for (int p = 0; p < BM / 32; ++p) {
        // FAST allows a ragged last M-tile for this layout: its surplus rows re-read row M-1 and are never stored
        const int row = FAST ? min(m0 + (tid >> 3) + 32 * p, g.M - 1) : m0 + (tid >> 3) + 32 * p, kc = kt + (tid & 7) * 4;
        const int64_t off = (int64_t)row * g.lda + kc;
        load4(g.A, vecA, off, row < g.M, kc, kend, &ra[p * 4]);
        if (has_gate) load4(g.a_gate, vecA, off, row < g.M, kc, kend, &rg[p * 4]);
      }
    } else {
      constexpr int ROWS_PER_PASS = 1024 / BM;
#pragma unroll
      for (int p = 0; p < BM / 32; ++p) {
        const int krow = kt + tid / (BM / 4) + ROWS_PER_PASS * p, mc = m0 + (tid % (BM / 4)) * 4;
        const int64_t off = (int64_t)krow * g.lda + mc;
        load4(g.A, vecA, off, krow < kend, mc, g.M, &ra[p * 4]);
        if (has_gate) load4(g.a_gate, vecA, off, krow < kend, mc, g.M, &rg[p * 4]);
      }
    }
    if (B_KC) {
#pragma unroll
      for (int p = 0; p < BN / 32; ++p) {
        const int row = n0 + (tid >> 3) + 32 * p, kc = kt + (tid & 7) * 4;
        load4(g.B, vecB, (int64_t)row * g.ldb + kc, row < g.N, kc, kend, &rb[p * 4]);
      }
    } else {
      constexpr int ROWS_PER_PASS = 1024 / BN;
#pragma unroll
      for (int p = 0; p < BN / 32; ++p) {
        const int krow = kt + tid / (BN / 4) + ROWS_PER_PASS * p, nc = n0 + (tid % (BN / 4)) * 4;
        load4(g.B, vecB, (int64_t)krow * g.ldb + nc, krow < kend, nc, g.N, &rb[p * 4]);
      }
    }
  };

  // kt = first reduction index of the tile held in the staging registers
  float colsum_s[SPLIT ? BM / 64 : 1];             // SPLIT: this thread's share of the column sums (column tid & 63 + 64 p)
#pragma unroll
  for (int p = 0; p < (SPLIT ? BM / 64 : 1); ++p) colsum_s[p] = 0.f;
  unsigned short* Ap = reinterpret_cast<unsigned short*>(smem);          // SPLIT: [3][BM][KP]
  unsigned short* Bp = Ap + 3 * BM * KP;                                   //        [3][BN][KP]
  auto commit = [&](int kt) {
    const bool prologue = has_gate || g.a_mask_mode != MBVAE_MASK_NONE || g.a_gate_scale != 1.f;
    if (SPLIT) {
#pragma unroll
      for (int p = 0; p < BM / 64; ++p) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = ra[p * 8 + i];
          if (has_gate && !(rg[p * 8 + i] > 0.f)) v[i] = 0.f;
          if (prologue) {
            float keep = g.a_gate_scale;
            if (g.a_mask_mode != MBVAE_MASK_NONE)
              keep *= A_KC ? keepM[(tid >> 2) + 64 * p] : keepK[(kt >> 5) & 1][8 * (tid >> 6) + i];
            v[i] = keep != 0.f ? v[i] * keep : 0.f;      // select: masked rows may hold non-finite values
          }
        }
        if (!A_KC) colsum_s[p] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const MbvaeSplit3 sp = mbvae_split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        unsigned short* dst = A_KC ? Ap + ((tid >> 2) + 64 * p) * KP + 8 * (tid & 3) : Ap + ((tid & 63) + 64 * p) * KP + 8 * (tid >> 6);
        *reinterpret_cast<mbvae_bf16x8*>(dst) = sp.hi;
        *reinterpret_cast<mbvae_bf16x8*>(dst + BM * KP) = sp.mid;
        *reinterpret_cast<mbvae_bf16x8*>(dst + 2 * BM * KP) = sp.lo;
      }
#pragma unroll
      for (int p = 0; p < BN / 64; ++p) {
        const MbvaeSplit3 sp = mbvae_split8(rb[p * 8], rb[p * 8 + 1], rb[p * 8 + 2], rb[p * 8 + 3], rb[p * 8 + 4], rb[p * 8 + 5],
                                            rb[p * 8 + 6], rb[p * 8 + 7]);
        unsigned short* dst = B_KC ? Bp + ((tid >> 2) + 64 * p) * KP + 8 * (tid & 3) : Bp + ((tid & 63) + 64 * p) * KP + 8 * (tid >> 6);
        *reinterpret_cast<mbvae_bf16x8*>(dst) = sp.hi;
        *reinterpret_cast<mbvae_bf16x8*>(dst + BN * KP) = sp.mid;
        *reinterpret_cast<mbvae_bf16x8*>(dst + 2 * BN * KP) = sp.lo;
      }
      return;
    }
    if (A_KC) {
#pragma unroll
      for (int p = 0; p < BM / 32; ++p) {
        float keep = 1.f;
        if (prologue) keep = (g.a_mask_mode != MBVAE_MASK_NONE ? keepM[(tid >> 3) + 32 * p] : 1.f) * g.a_gate_scale;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = ra[p * 4 + q];
          if (has_gate && !(rg[p * 4 + q] > 0.f)) v = 0.f;
          if (prologue) v = keep != 0.f ? v * keep : 0.f;      // select: masked rows may hold non-finite values
          As[((tid & 7) * 4 + q) * LDAS + (tid >> 3) + 32 * p] = v;
        }
      }
    } else {
      constexpr int ROWS_PER_PASS = 1024 / BM;
#pragma unroll
      for (int p = 0; p < BM / 32; ++p) {
        float keep = 1.f;
        if (prologue)
          keep = (g.a_mask_mode != MBVAE_MASK_NONE ? keepK[(kt >> 5) & 1][tid / (BM / 4) + ROWS_PER_PASS * p] : 1.f) * g.a_gate_scale;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = ra[p * 4 + q];
          if (has_gate && !(rg[p * 4 + q] > 0.f)) v[q] = 0.f;
          if (prologue) v[q] = keep != 0.f ? v[q] * keep : 0.f;
        }
        *reinterpret_cast<float4*>(&As[(tid / (BM / 4) + ROWS_PER_PASS * p) * LDAS + (tid % (BM / 4)) * 4]) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    if (B_KC) {
#pragma unroll
      for (int p = 0; p < BN / 32; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) Bs[((tid & 7) * 4 + q) * LDBS + (tid >> 3) + 32 * p] = rb[p * 4 + q];
    } else {
      constexpr int ROWS_PER_PASS = 1024 / BN;
#pragma unroll
      for (int p = 0; p < BN / 32; ++p)
        *reinterpret_cast<float4*>(&Bs[(tid / (BN / 4) + ROWS_PER_PASS * p) * LDBS + (tid % (BN / 4)) * 4]) =
            make_float4(rb[p * 4], rb[p * 4 + 1], rb[p * 4 + 2], rb[p * 4 + 3]);
    }
  };

  const bool want_colsum = (g.colsum != nullptr || g.partial_colsum != nullptr) && n0 == 0;
  // The prefetch is unconditional (past the end it re-reads the last tile, which is never committed): loads issued
  // under a branch make the compiler wait for them at the join, i.e. before the MFMA loop they should overlap.
  const int klast = kbeg + (max(kend - kbeg, 1) - 1) / BK * BK;
  if (kbeg < kend) fetch(kbeg);
  for (int kt = kbeg; kt < kend; kt += BK) {
    __syncthreads();                 // previous tile fully consumed
    commit(kt);
    __syncthreads();
    fetch(min(kt + BK, klast));
    if (SPLIT) {
#pragma unroll
      for (int st = 0; st < BK / 16; ++st) {
        mbvae_bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned short* ap = Ap + ((wm * TM + i) * 32 + r) * KP + 16 * st + 8 * h;
          ah[i] = *reinterpret_cast<const mbvae_bf16x8*>(ap);
          am[i] = *reinterpret_cast<const mbvae_bf16x8*>(ap + BM * KP);
          al[i] = *reinterpret_cast<const mbvae_bf16x8*>(ap + 2 * BM * KP);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned short* bp = Bp + ((wn * TN + j) * 32 + r) * KP + 16 * st + 8 * h;
          bh[j] = *reinterpret_cast<const mbvae_bf16x8*>(bp);
          bm[j] = *reinterpret_cast<const mbvae_bf16x8*>(bp + BN * KP);
          bl[j] = *reinterpret_cast<const mbvae_bf16x8*>(bp + 2 * BN * KP);
        }
        // smaller terms first
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MBVAE_MFMA_BF16(al[i], bh[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MBVAE_MFMA_BF16(ah[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MBVAE_MFMA_BF16(am[i], bm[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MBVAE_MFMA_BF16(am[i], bh[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MBVAE_MFMA_BF16(ah[i], bm[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MBVAE_MFMA_BF16(ah[i], bh[j], acc[i][j]);
      }
      continue;
    }
    if (want_colsum && tid < BM) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) colsum += As[k * LDAS + tid];
    }
#pragma unroll 4
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(2 * kk + h) * LDAS + (wm * TM + i) * 32 + r];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * kk + h) * LDBS + (wn * TN + j) * 32 + r];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  if (SPLIT && want_colsum) {                        // the four octet groups' shares of the column sums meet in LDS, in order
#pragma unroll
    for (int p = 0; p < BM / 64; ++p) csum4[(tid >> 6) * BM + (tid & 63) + 64 * p] = colsum_s[p];
    __syncthreads();
    if (tid < BM) colsum = ((csum4[tid] + csum4[BM + tid]) + csum4[2 * BM + tid]) + csum4[3 * BM + tid];
  }
  // ---- write back.  C/D layout of the 32x32 tile: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5).
  // Bias values and row-mask flags are fetched up front: loads issued between the stores would each be waited
  // for (the stores may alias them as far as the compiler knows) and serialise the epilogue on L2 latency.
  const bool split = g.partial != nullptr;
  float bj[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + r;
    bj[j] = (!split && g.bias && (FAST || col < g.N)) ? g.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float keep[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
      keep[v] = (split || g.c_mask_mode == MBVAE_MASK_NONE) ? 1.f : keepM[row - m0];
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + r;
        if (!FAST && col >= g.N) continue;
        if (split) {
          g.partial[((int64_t)blockIdx.z * g.M + row) * g.N + col] = acc[i][j][v];
        } else {
          float o = acc[i][j][v] + bj[j];
          if (g.act == 1) o = fmaxf(o, 0.f);
          if (g.drop_p > 0.f) o *= mbvae_drop_scale(g.drop_p, g.seed, (uint64_t)row * (uint64_t)g.ldc + col);
          o = keep[v] != 0.f ? o : 0.f;
          if (g.accumulate) o += g.C[(int64_t)row * g.ldc + col];
          if (g.addend) o += g.addend[(int64_t)row * g.ldc + col];
          g.C[(int64_t)row * g.ldc + col] = o;
        }
      }
    }
  }
  if (want_colsum && tid < BM && m0 + tid < g.M) {
    if (split) g.partial_colsum[(int64_t)blockIdx.z * g.M + m0 + tid] = colsum;
    else g.colsum[m0 + tid] = colsum;
  }
}

__global__ void gemm_reduce_kernel(GemmArgs g, int splits) {
  MBVAE_RAISE_PRIO();
  const int64_t total = (int64_t)g.M * g.N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total + g.M; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < total) {
      float s = 0.f;
      int z = 0;
      for (; z + 4 <= splits; z += 4) {               // four loads in flight; the additions keep their order
        const float a0 = g.partial[(int64_t)z * total + e], a1 = g.partial[(int64_t)(z + 1) * total + e];
        const float a2 = g.partial[(int64_t)(z + 2) * total + e], a3 = g.partial[(int64_t)(z + 3) * total + e];
        s += a0; s += a1; s += a2; s += a3;
      }
      for (; z < splits; ++z) s += g.partial[(int64_t)z * total + e];
      const int row = (int)(e / g.N), col = (int)(e - (int64_t)row * g.N);
      const float keep = mbvae_row_keep(g.mask, g.c_mask_mode, row, g.n_nodes);
      g.C[(int64_t)row * g.ldc + col] = gemm_epilogue(g, s, row, col, keep);
    } else if (g.colsum) {
      const int64_t m = e - total;
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += g.partial_colsum[(int64_t)z * g.M + m];
      g.colsum[m] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Persistent "tall-skinny" product for the edge path: C[M,N] = epi( pro(A)[M,K] @ Bm[K,N] ) with M ~ 10^6 rows and a
// weight matrix small enough (K*N*4 <= 64 KB) to stay in LDS for the whole kernel.  One 512-thread workgroup per CU
// walks over 128/256-row tiles of A: the weights are staged once, A arrives in 32-wide k-chunks through a
// double-buffered LDS ring whose next chunk (possibly the next tile's first) is already in flight in registers while
// the current one is multiplied, so HBM latency and the epilogue stores hide under the MFMA stream and there is one
// barrier per chunk.  A is k-contiguous (activations or their gradients); Bm is either W[N,K] (forward, B_KC) or
// W[K,N] (input gradient).
template <int N, int BM, int WM, int WN, bool B_KC, bool GATE>
__global__ __launch_bounds__(512) void tsgemm_kernel(GemmArgs g) {
  constexpr int BK = 32;
  constexpr int TM = BM / (32 * WM), TN = N / (32 * WN);
  constexpr int LDA = BM + 1, LDB = N + 1;
  constexpr int PASSES = BM / 64;                  // 8 threads x float4 cover one 32-wide row chunk, 64 rows per pass
  static_assert(WM * WN == 8, "8 waves");
  extern __shared__ __align__(16) float ts_smem[];
  float* Bs = ts_smem;                             // [K][LDB]
  float* As = Bs + g.K * LDB;                      // [2][BK][LDA]
  float* keepS = As + 2 * BK * LDA;                // [2][BM]  row-mask flags of the tile (epilogue)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps index arithmetic on the scalar unit
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int K = g.K;
  const int cpt = K / BK;                          // chunks per tile
  const int ntiles = g.M / BM;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * cpt;
  const int mask_mode = g.a_mask_mode | g.c_mask_mode;
  const bool prologue = GATE || g.a_mask_mode != MBVAE_MASK_NONE || g.a_gate_scale != 1.f;

  // ---- weights -> LDS (k-major), once
  if (B_KC) {
    for (int e = tid; e < N * K; e += 512) { const int n = e / K, k = e - n * K; Bs[k * LDB + n] = g.B[(int64_t)n * g.ldb + k]; }
  } else {
    for (int e = tid; e < N * K; e += 512) { const int k = e / N, n = e - k * N; Bs[k * LDB + n] = g.B[(int64_t)k * g.ldb + n]; }
  }

  float ra[PASSES * 4], rg[PASSES * 4];
  unsigned ma[PASSES], mb[PASSES], ta = 1, tb = 1;     // raw mask bytes: rows of the staged chunk / this thread's tile row
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  auto fetch = [&](int c) {
    const int lt = c / cpt, kc = (c - lt * cpt) * BK;
    const int m0 = ((int)blockIdx.x + lt * (int)gridDim.x) * BM;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int row = m0 + (tid >> 3) + 64 * p;
      const int64_t off = (int64_t)row * g.lda + kc + (tid & 7) * 4;
      const float4 t = *reinterpret_cast<const float4*>(g.A + off);
      ra[p * 4] = t.x; ra[p * 4 + 1] = t.y; ra[p * 4 + 2] = t.z; ra[p * 4 + 3] = t.w;
      if (GATE) {
        const float4 q = *reinterpret_cast<const float4*>(g.a_gate + off);
        rg[p * 4] = q.x; rg[p * 4 + 1] = q.y; rg[p * 4 + 2] = q.z; rg[p * 4 + 3] = q.w;
      }
      mbvae_row_mask_bytes(g.mask, g.a_mask_mode, row, g.n_nodes, g.A, ma[p], mb[p]);
    }
    mbvae_row_mask_bytes(g.mask, mask_mode, m0 + min(tid, BM - 1), g.n_nodes, g.A, ta, tb);   // this thread's tile row
  };
  auto commit = [&](int c) {
    float* dst = As + (c & 1) * BK * LDA;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const float rk = mbvae_mask_keep(g.a_mask_mode, ma[p], mb[p]) ? g.a_gate_scale : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = ra[p * 4 + q];
        if (GATE && !(rg[p * 4 + q] > 0.f)) v = 0.f;
        if (prologue) v = rk != 0.f ? v * rk : 0.f;
        dst[((tid & 7) * 4 + q) * LDA + (tid >> 3) + 64 * p] = v;
      }
    }
    const int lt = c / cpt;
    if (c == lt * cpt && g.c_mask_mode != MBVAE_MASK_NONE && tid < BM) keepS[(lt & 1) * BM + tid] = mbvae_mask_keep(mask_mode, ta, tb) ? 1.f : 0.f;
  };

  float bj[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bj[j] = g.bias ? g.bias[(wn * TN + j) * 32 + r] : 0.f;

  if (total > 0) fetch(0);
  for (int c = 0; c < total; ++c) {
    commit(c);
    __syncthreads();
    fetch(min(c + 1, total - 1));                  // unconditional: a branch here would make the loads wait at its join
    const float* Ab = As + (c & 1) * BK * LDA;
    const int lt = c / cpt, kc = (c - lt * cpt) * BK;
#pragma unroll 4
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ab[(2 * kk + h) * LDA + (wm * TM + i) * 32 + r];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kc + 2 * kk + h) * LDB + (wn * TN + j) * 32 + r];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kc + BK == K) {                            // tile finished: epilogue, then restart the accumulators
      const int m0 = ((int)blockIdx.x + lt * (int)gridDim.x) * BM;
      const float* ks = keepS + (lt & 1) * BM;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float keep[16];
#pragma unroll
        for (int v = 0; v < 16; ++v)
          keep[v] = g.c_mask_mode == MBVAE_MASK_NONE ? 1.f : ks[(wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = (wn * TN + j) * 32 + r;
            float o = acc[i][j][v] + bj[j];
            if (g.act == 1) o = fmaxf(o, 0.f);
            if (g.drop_p > 0.f) o *= mbvae_drop_scale(g.drop_p, g.seed, (uint64_t)row * (uint64_t)g.ldc + col);
            o = keep[v] != 0.f ? o : 0.f;
            if (g.accumulate) o += g.C[(int64_t)row * g.ldc + col];
            g.C[(int64_t)row * g.ldc + col] = o;
            acc[i][j][v] = 0.f;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wave-autonomous variant of the tall-skinny product.  The kernel above moves A through LDS, which costs a scatter,
// a barrier per chunk and - worse - keeps all eight waves of the CU in the same phase: they all multiply, then all
// store, so the output stores (the larger half of the HBM traffic when N > K) never overlap the MFMA stream.  Here
// only the weights live in LDS and nothing synchronises after they are staged.  Every wave owns whole tasks of
// 32 rows x 32*TN columns: lane (m, h) reads 32 consecutive reduction indices of row m (128 B, eight 16-byte loads)
// straight into the registers that feed the MFMA A operand, and the reduction runs in the permuted order this
// register layout implies (step s multiplies indices s and 32+s of a 64-wide group; the matching weight rows come
// from LDS).  Two rules keep the memory pipeline out of the MFMAs' way:
//  * The hardware counter that orders vector memory operations tops out at 63 and counts stores as well.  A task
//    therefore stores at most 32*TN/... = 16*TN <= 64 dwords per lane (wider outputs are split into column blocks
//    handled by neighbouring waves, which re-read the same A rows through L1/L2), so a whole epilogue can be in
//    flight while the wave already multiplies its next task.
//  * The wait for a group's operand loads is pinned right after the previous group's MFMAs and BEFORE the epilogue
//    stores; placed after them it would have to drain the stores first.
//
// SPLIT: the same schedule on the bf16 matrix pipeline (see mbvae_split8: six bf16 MFMAs per fp32 product).  The
// weights are split ONCE, while they are staged, into three bf16 planes laid out [plane][n][K + 8] (the 16-byte pad
// makes the MFMA B-operand fetch - lane (n, h) reads the 8 consecutive reduction indices of its octet, one
// ds_read_b128 per plane - conflict-free); the A rows are split in registers right before their MFMAs, one split per
// 16 reduction indices, shared by the task's TN column tiles.  6 bytes of LDS per weight: K * N <= 25600.
// SPLIT == 2 (the 32768-weight shapes, whose planes would not fit): the weights stay fp32 in LDS and each lane splits
// its B octets as well, right after reading them - more vector work per MFMA, still ahead of the fp32 MFMA.
// ABF / CBF: A is read / C is written as bf16 (the opt-in bf16 storage of the large edge activations).  A bf16 operand
// IS its own hi plane, so its products need three MFMAs instead of six.
// LNE: residual + dropout + LayerNorm in the epilogue (TN == 2 and N == 64: a wave's task is 32 whole rows).  In the
// accumulator layout a row's 64 values sit in the 32 lanes of one half-wave (two per lane), 16 rows per lane: the two
// row statistics are half-wave sums (mean first, then the centred squares - the two-pass form of the stand-alone
// LayerNorm kernel).  The pre-LayerNorm sum z is stored next to the output for the backward pass.
template <int TN, bool B_KC, bool GATE, int WAVES, bool ADD = false, int SPLIT = 0, bool ABF = false, bool CBF = false, bool CGATE = false,
          bool LNE = false>
__global__ __launch_bounds__(64 * WAVES) void tsgemm2_kernel(GemmArgs g) {
  static_assert(!LNE || (TN == 2 && ADD && !CBF && !CGATE), "the LayerNorm epilogue: 64-wide fp32 output rows with a residual operand");
  static_assert(!(ABF && (GATE || SPLIT == 0)), "a bf16 A operand runs on the bf16 pipeline and has no gate");
  static_assert(!(ADD && CGATE) && !(CGATE && CBF), "an output gate comes without an addend and with fp32 output");
  constexpr int NP = ABF ? 4 : 8;                  // 16-byte words of A per lane and 64-wide reduction group
  constexpr unsigned CS = CBF ? 2u : 4u;           // bytes per output element
  constexpr int NT = 64 * WAVES;
  // Column groups (weights too large for one workgroup's LDS, e.g. the 256 x 256 node-path layers): workgroup column
  // blockIdx.y keeps only its `col_group` columns of the weights and produces only those output columns; everything
  // below then sees an N = col_group problem whose B, C, bias and addend start at the group's first column.
  const int col0 = g.col_group ? (int)blockIdx.y * g.col_group : 0;
  if (g.col_group) {
    g.N = g.col_group;
    g.B += B_KC ? (int64_t)col0 * g.ldb : (int64_t)col0;
    g.C = reinterpret_cast<float*>(reinterpret_cast<char*>(g.C) + (size_t)col0 * CS);
    if (g.bias) g.bias += col0;
    if (g.addend) g.addend += col0;
  }
  const int N = g.N, LDB = N + 1;
  extern __shared__ __align__(16) float ts2_smem[];
  float* Bs = ts2_smem;                            // [K][LDB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps index arithmetic on the scalar unit
  const int r = lane & 31, h = lane >> 5;
  const int K = g.K;
  const int LDK = K + 8;                           // SPLIT: bf16 elements per weight column
  unsigned short* Bp = reinterpret_cast<unsigned short*>(ts2_smem);   // SPLIT: [3][N][LDK]
  if (SPLIT == 1) {
    for (int e = tid; e < N * K; e += NT) {
      int n, k;
      float w;
      if (B_KC) { n = e / K; k = e - n * K; w = g.B[(int64_t)n * g.ldb + k]; }
      else { k = e / N; n = e - k * N; w = g.B[(int64_t)k * g.ldb + n]; }
      const unsigned u = __float_as_uint(w);
      const float r1 = w - __uint_as_float(u & 0xffff0000u);
      const unsigned v = __float_as_uint(r1);
      const float r2 = r1 - __uint_as_float(v & 0xffff0000u);
      Bp[n * LDK + k] = (unsigned short)(u >> 16);
      Bp[(N + n) * LDK + k] = (unsigned short)(v >> 16);
      Bp[(2 * N + n) * LDK + k] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
  } else if (B_KC) {
    for (int e = tid; e < N * K; e += NT) { const int n = e / K, k = e - n * K; Bs[k * LDB + n] = g.B[(int64_t)n * g.ldb + k]; }
  } else {
    for (int e = tid; e < N * K; e += NT) { const int k = e / N, n = e - k * N; Bs[k * LDB + n] = g.B[(int64_t)k * g.ldb + n]; }
  }
  __syncthreads();

  const int gpt = K / 64;                           // 64-wide reduction groups per task
  const int cbs = N / (32 * TN);                    // column blocks; consecutive waves share the rows of one tile
  const int ntasks = (g.M / 32) * cbs;
  const int wid = (int)blockIdx.x * WAVES + wave, nw = (int)gridDim.x * WAVES;
  const int my_tasks = (ntasks - wid + nw - 1) / nw;
  const int total = my_tasks * gpt;                 // groups this wave multiplies
  if (total <= 0) return;
  const int mask_mode = g.a_mask_mode | g.c_mask_mode;
  const bool relu = g.act == 1;
  const bool drop = g.drop_p > 0.f;
  const unsigned ldc = (unsigned)g.ldc;
  const float drop_keep = 1.f / (1.f - g.drop_p);

  float4 ra[2][8], rg[2][8];
  unsigned ma[2], mb[2];
  float keepf[2];                                   // row-mask flag of this lane's row in the staged tile
  float bjr[2][TN];                                 // bias of the task's columns: loaded with the operands, a load in the
                                                    // epilogue would stall the stores behind a full memory latency
  const float* biasp = g.bias ? g.bias : g.A;       // any readable address when there is no bias (the value is never used:
                                                    // selected away, not multiplied by zero - it may be non-finite)
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  // On gfx950 the f32 MFMA shares the vector ALU's cycles (measured: every ordinary VALU instruction next to the MFMA
  // stream costs ~3 of its 4 cycles of MFMA time), so the epilogue keeps its per-element VALU work minimal: the
  // accumulators START at the bias, row-mask / ReLU code runs only when needed, and the stores are buffer stores
  // whose row/column offset sits in a scalar register (no per-element address arithmetic).
  const __amdgpu_buffer_rsrc_t rsrcC =
      __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)((int64_t)g.M * g.ldc * CS), 0x00020000);
  const int voffC = (int)((4u * h * ldc + r) * CS);      // byte offset of (row 4h, column r) inside a task's tile
  const __amdgpu_buffer_rsrc_t rsrcZ =
      __builtin_amdgcn_make_buffer_rsrc(LNE ? g.ln_z : g.C, 0, (int)(unsigned)((int64_t)g.M * g.ldc * CS), 0x00020000);
  float lng[2] = {1.f, 1.f}, lnb[2] = {0.f, 0.f};        // LayerNorm scale / shift of this lane's two columns
  if (LNE) { lng[0] = g.ln_gamma[r]; lng[1] = g.ln_gamma[32 + r]; lnb[0] = g.ln_beta[r]; lnb[1] = g.ln_beta[32 + r]; }

#define TS2_TASK(qq, lt, gi, tile, cb)                                                                \
    const int lt = (qq) / gpt, gi = (qq) - lt * gpt;                                                   \
    const int task = wid + lt * nw, tile = task / cbs, cb = task - tile * cbs;
  // group q of this wave -> task and reduction offset; past the end the last group is re-read and multiplied by zero
#define TS2_FETCH(q, S)                                                                               \
  {                                                                                                   \
    TS2_TASK(min((q), total - 1), lt, gi, tile, cb)                                                    \
    const int64_t row = (int64_t)tile * 32 + r;                                                        \
    const float* src = ABF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.A) + row * g.lda + gi * 64 + 32 * h) \
                           : g.A + row * g.lda + gi * 64 + 32 * h;                                     \
    mbvae_row_mask_bytes(g.mask, mask_mode, row, g.n_nodes, g.A, ma[S], mb[S]);                        \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) bjr[S][j] = biasp[g.bias ? cb * (32 * TN) + j * 32 + r : 0]; \
    _Pragma("unroll") for (int p = 0; p < NP; ++p) ra[S][p] = *reinterpret_cast<const float4*>(src + 4 * p); \
    if (GATE) {                                                                                        \
      const float* gs = g.a_gate + row * g.lda + gi * 64 + 32 * h;                                     \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) rg[S][p] = *reinterpret_cast<const float4*>(gs + 4 * p); \
    }                                                                                                  \
    asm volatile("" : : : "memory");   /* the loads are issued HERE (the scheduler would sink them to their use) */ \
  }
  // operand prologue applied in place; the asm pins the wait for the loads at this point
#define TS2_PREP(q, S)                                                                                \
  {                                                                                                   \
    unsigned late;   /* a zero the compiler cannot see through: keeps the mask test (and its wait) down here */ \
    asm volatile("v_mov_b32 %0, 0" : "=v"(late) : : "memory");                                         \
    keepf[S] = mbvae_mask_keep(mask_mode, ma[S] | late, mb[S] | late) ? 1.f : 0.f;                     \
    float rk = g.a_mask_mode != MBVAE_MASK_NONE ? keepf[S] * g.a_gate_scale : g.a_gate_scale;          \
    if ((q) >= total) rk = 0.f;                                                                        \
    if (!GATE && __ballot(rk != 1.f) == 0ull) {        /* nothing to apply in this wave: only pin the loads */ \
      _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                 \
        float4 v = ra[S][p];                                                                           \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : : "memory");                    \
        ra[S][p] = v;                                                                                  \
      }                                                                                                \
    } else if (ABF) {                                  /* packed bf16 words: a masked row becomes zeros (no scale) */ \
      _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                 \
        float4 v = ra[S][p];                                                                           \
        v.x = rk != 0.f ? v.x : 0.f; v.y = rk != 0.f ? v.y : 0.f;                                      \
        v.z = rk != 0.f ? v.z : 0.f; v.w = rk != 0.f ? v.w : 0.f;                                      \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : : "memory");                    \
        ra[S][p] = v;                                                                                  \
      }                                                                                                \
    } else {                                                                                           \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                  \
        float4 v = ra[S][p];                                                                           \
        if (GATE) {                                                                                    \
          if (!(rg[S][p].x > 0.f)) v.x = 0.f;                                                          \
          if (!(rg[S][p].y > 0.f)) v.y = 0.f;                                                          \
          if (!(rg[S][p].z > 0.f)) v.z = 0.f;                                                          \
          if (!(rg[S][p].w > 0.f)) v.w = 0.f;                                                          \
        }                                                                                              \
        /* selects: a masked row may hold non-finite values */                                         \
        v.x = rk != 0.f ? v.x * rk : 0.f; v.y = rk != 0.f ? v.y * rk : 0.f;                            \
        v.z = rk != 0.f ? v.z * rk : 0.f; v.w = rk != 0.f ? v.w * rk : 0.f;                            \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : : "memory");                    \
        ra[S][p] = v;                                                                                  \
      }                                                                                                \
    }                                                                                                  \
    if ((q) == 0) {                                    /* first task: accumulators start at its bias */ \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) acc[j][v] = g.bias ? bjr[S][j] : 0.f;           \
    }                                                                                                  \
  }
#define TS2_MMA(q, S)                                                                                 \
  if (SPLIT == 2) {                                                                                   \
    TS2_TASK(min((q), total - 1), lt, gi, tile, cb)                                                    \
    const float* Bg = Bs + (gi * 64 + 32 * h) * LDB + cb * (32 * TN) + r;                              \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                    \
      MbvaeSplit3 a;                                                                                   \
      if (ABF) a.hi = a.mid = a.lo = __builtin_bit_cast(mbvae_bf16x8, ra[S][c]);                       \
      else a = mbvae_split8(ra[S][(2 * c) % NP].x, ra[S][(2 * c) % NP].y, ra[S][(2 * c) % NP].z, ra[S][(2 * c) % NP].w, \
                            ra[S][(2 * c + 1) % NP].x, ra[S][(2 * c + 1) % NP].y, ra[S][(2 * c + 1) % NP].z, ra[S][(2 * c + 1) % NP].w); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
        const float* bp = Bg + 8 * c * LDB + j * 32;                                                   \
        const MbvaeSplit3 b = mbvae_split8(bp[0], bp[LDB], bp[2 * LDB], bp[3 * LDB], bp[4 * LDB], bp[5 * LDB], bp[6 * LDB], bp[7 * LDB]); \
        if (!ABF) acc[j] = MBVAE_MFMA_BF16(a.lo, b.hi, acc[j]);                                        \
        acc[j] = MBVAE_MFMA_BF16(a.hi, b.lo, acc[j]);                                                  \
        if (!ABF) acc[j] = MBVAE_MFMA_BF16(a.mid, b.mid, acc[j]);                                      \
        if (!ABF) acc[j] = MBVAE_MFMA_BF16(a.mid, b.hi, acc[j]);                                       \
        acc[j] = MBVAE_MFMA_BF16(a.hi, b.mid, acc[j]);                                                 \
        acc[j] = MBVAE_MFMA_BF16(a.hi, b.hi, acc[j]);                                                  \
      }                                                                                                \
    }                                                                                                  \
  } else if (SPLIT == 1) {                                                                            \
    TS2_TASK(min((q), total - 1), lt, gi, tile, cb)                                                    \
    const unsigned short* Bg = Bp + (cb * (32 * TN) + r) * LDK + gi * 64 + 32 * h;                     \
    const int plane = N * LDK;                                                                         \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {          /* 16 reduction indices per step: octet c of each half */ \
      MbvaeSplit3 a;                                                                                   \
      if (ABF) a.hi = a.mid = a.lo = __builtin_bit_cast(mbvae_bf16x8, ra[S][c]);                       \
      else a = mbvae_split8(ra[S][(2 * c) % NP].x, ra[S][(2 * c) % NP].y, ra[S][(2 * c) % NP].z, ra[S][(2 * c) % NP].w, \
                            ra[S][(2 * c + 1) % NP].x, ra[S][(2 * c + 1) % NP].y, ra[S][(2 * c + 1) % NP].z, ra[S][(2 * c + 1) % NP].w); \
      mbvae_bf16x8 bh[TN], bm[TN], bl[TN];                                                             \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
        const unsigned short* bp = Bg + j * 32 * LDK + 8 * c;                                          \
        bh[j] = *reinterpret_cast<const mbvae_bf16x8*>(bp);                                            \
        bm[j] = *reinterpret_cast<const mbvae_bf16x8*>(bp + plane);                                    \
        bl[j] = *reinterpret_cast<const mbvae_bf16x8*>(bp + 2 * plane);                                \
      }                                                                                                \
      if (!ABF) { _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = MBVAE_MFMA_BF16(a.lo, bh[j], acc[j]); }  \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = MBVAE_MFMA_BF16(a.hi, bl[j], acc[j]);    \
      if (!ABF) { _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = MBVAE_MFMA_BF16(a.mid, bm[j], acc[j]); } \
      if (!ABF) { _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = MBVAE_MFMA_BF16(a.mid, bh[j], acc[j]); } \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = MBVAE_MFMA_BF16(a.hi, bm[j], acc[j]);    \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = MBVAE_MFMA_BF16(a.hi, bh[j], acc[j]);    \
    }                                                                                                  \
  } else {                                                                                            \
    TS2_TASK(min((q), total - 1), lt, gi, tile, cb)                                                    \
    const float* Bg = Bs + (gi * 64 + 32 * h) * LDB + cb * (32 * TN) + r;                              \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                    \
      const float av[4] = {ra[S][p].x, ra[S][p].y, ra[S][p].z, ra[S][p].w};                            \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], Bg[(4 * p + e) * LDB + j * 32], acc[j], 0, 0, 0); \
      }                                                                                                \
    }                                                                                                  \
  }
  // task finished: (bias is already in the accumulators) ReLU / dropout / row mask, stores; the accumulators restart
  // at the NEXT task's bias, whose operands - bias included - sit in the other buffer and have been waited for
#ifndef TS2_AUX
#define TS2_AUX 2   /* non-temporal: the output is not read again by this kernel and must not displace the A rows */
#endif
#define TS2_ST(val, rr, j, sbase)                                                                     \
    do {                                                                                              \
      const int so_ = (int)((sbase) + ((unsigned)(rr) * ldc + (j) * 32) * CS);                         \
      if (CBF) __builtin_amdgcn_raw_buffer_store_b16(mbvae_f2bf(val), rsrcC, voffC, so_, TS2_AUX);     \
      else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rsrcC, voffC, so_, TS2_AUX);    \
    } while (0)
#define TS2_EPI(q, S)                                                                                 \
  {                                                                                                   \
    TS2_TASK((q), lt, gi, tile, cb)                                                                    \
    if ((q) < total && gi == gpt - 1) {                                                                \
      const unsigned long long kb = g.c_mask_mode == MBVAE_MASK_NONE ? ~0ull : __ballot(keepf[S] != 0.f); \
      const unsigned kh = (unsigned)(kb >> (4 * h));   /* bit i: row 4h+i of the tile survives */      \
      const unsigned sbase = (((unsigned)tile * 32u) * ldc + cb * (32 * TN)) * CS;   /* wave-uniform byte offset */ \
      float bn[TN];                                                                                    \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) bn[j] = g.bias ? bjr[1 - S][j] : 0.f;             \
      if (LNE) {                                       /* residual + dropout + LayerNorm over the task's 32 whole rows */ \
        const unsigned off0 = ((unsigned)tile * 32u + 4u * h) * ldc + r;                               \
        float zz[2][16], mu[16], rs[16];                                                               \
        _Pragma("unroll") for (int v = 0; v < 16; ++v)                                                 \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
            zz[j][v] = g.addend[off0 + (unsigned)((v & 3) + 8 * (v >> 2)) * ldc + j * 32];             \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          const bool kr = (kh >> rr) & 1u;                                                             \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
            float o = kr ? acc[j % TN][v] : 0.f;                                                       \
            if (drop) o *= mbvae_u01(g.seed, off0 + (unsigned)rr * ldc + j * 32) < g.drop_p ? 0.f : drop_keep; \
            zz[j][v] += o;                                                                             \
          }                                                                                            \
          mu[v] = zz[0][v] + zz[1][v];                                                                 \
        }                                                                                              \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) mu[v] = half_wave_sum(mu[v]) * (1.f / 64.f);    \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const float d0 = zz[0][v] - mu[v], d1 = zz[1][v] - mu[v];                                    \
          rs[v] = d0 * d0 + d1 * d1;                                                                   \
        }                                                                                              \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) rs[v] = rsqrtf(half_wave_sum(rs[v]) * (1.f / 64.f) + g.ln_eps); \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
            const int so_ = (int)(sbase + ((unsigned)rr * ldc + j * 32) * 4u);                         \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(zz[j][v]), rsrcZ, voffC, so_, TS2_AUX); \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((zz[j][v] - mu[v]) * rs[v] * lng[j] + lnb[j]), rsrcC, voffC, so_, 0); \
          }                                                                                            \
        }                                                                                              \
      } else if (ADD || CGATE) {                       /* all addend / gate loads first, then one wait */ \
        const unsigned off0 = ((unsigned)tile * 32u + 4u * h) * ldc + cb * (32 * TN) + r;              \
        const float* adp = CGATE ? g.c_gate : g.addend;                                                \
        float ad[TN][16];                                                                              \
        _Pragma("unroll") for (int v = 0; v < 16; ++v)                                                 \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
            ad[j][v] = adp[off0 + (unsigned)((v & 3) + 8 * (v >> 2)) * ldc + j * 32];                  \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          const bool kr = (kh >> rr) & 1u;                                                             \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                             \
            float o = acc[j][v];                                                                       \
            if (relu) o = fmaxf(o, 0.f);                                                               \
            if (CGATE) TS2_ST((kr && ad[j][v] > 0.f) ? o * g.c_gate_scale : 0.f, rr, j, sbase);        \
            else TS2_ST((kr ? o : 0.f) + ad[j][v], rr, j, sbase);                                      \
          }                                                                                            \
        }                                                                                              \
      } else if (!drop && !relu && kb == ~0ull) {      /* nothing but the stores */                    \
        _Pragma("unroll") for (int v = 0; v < 16; ++v)                                                 \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                             \
            TS2_ST(acc[j][v], (v & 3) + 8 * (v >> 2), j, sbase);                                       \
          }                                                                                            \
      } else if (!drop) {                                                                              \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          const bool kr = (kh >> rr) & 1u;                                                             \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                             \
            float o = acc[j][v];                                                                       \
            if (relu) o = fmaxf(o, 0.f);                                                               \
            TS2_ST(kr ? o : 0.f, rr, j, sbase);                                                        \
          }                                                                                            \
        }                                                                                              \
      } else {                                                                                         \
        const unsigned off0 = ((unsigned)tile * 32u + 4u * h) * ldc + cb * (32 * TN) + r + (unsigned)col0;   \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) {                                               \
          const int rr = (v & 3) + 8 * (v >> 2);                                                       \
          const bool kr = (kh >> rr) & 1u;                                                             \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                             \
            const unsigned off = off0 + (unsigned)rr * ldc + j * 32;                                   \
            float o = acc[j][v];                                                                       \
            if (relu) o = fmaxf(o, 0.f);                                                               \
            o *= mbvae_u01(g.seed, off) < g.drop_p ? 0.f : drop_keep;                                  \
            TS2_ST(kr ? o : 0.f, rr, j, sbase);                                                        \
          }                                                                                            \
        }                                                                                              \
      }                                                                                                \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
        _Pragma("unroll") for (int v = 0; v < 16; ++v) acc[j][v] = bn[j];                              \
    }                                                                                                  \
  }

  TS2_FETCH(0, 0)
  TS2_PREP(0, 0)
  for (int q = 0; q < total; q += 2) {
    TS2_FETCH(q + 1, 1)
    TS2_MMA(q, 0)
    TS2_PREP(q + 1, 1)
    TS2_EPI(q, 0)
    TS2_FETCH(q + 2, 0)
    TS2_MMA(q + 1, 1)
    TS2_PREP(q + 2, 0)
    TS2_EPI(q + 1, 1)
  }
#undef TS2_TASK
#undef TS2_FETCH
#undef TS2_PREP
#undef TS2_MMA
#undef TS2_EPI
#undef TS2_ST
}

static int ts_num_cus() { return mbvae_num_cus(); }

// returns MBVAE_EUNSUPPORTED when the shape does not qualify (the caller then uses the generic kernel)
template <bool B_KC>
static int tsgemm_try(const GemmArgs& g, hipStream_t stream) {
  if (mbvae_opt(MBVAE_OPT_NO_TSGEMM)) return MBVAE_EUNSUPPORTED;
  const bool aligned = (((uintptr_t)g.A | (uintptr_t)g.a_gate) & 15) == 0 && (g.lda & (g.a_bf16 ? 7 : 3)) == 0;
  const bool bf16 = g.a_bf16 || g.c_bf16;
  if (g.b_bf16 || (bf16 && (g.a_gate || g.act || g.drop_p > 0.f || (g.bias && g.c_bf16) || g.a_gate_scale != 1.f)))
    return MBVAE_EUNSUPPORTED;
  if (g.c_gate && (bf16 || g.a_gate || g.addend || g.act || g.drop_p > 0.f || g.accumulate || ((uintptr_t)g.c_gate & 15) ||
                   (int64_t)g.K * g.N > 25600))
    return MBVAE_EUNSUPPORTED;
  // Node-path shapes (a few thousand rows, wide weights): the weights do not fit one workgroup's LDS, and there are too
  // few rows to amortise staging all of them anyway.  Workgroup COLUMN groups of 64 output columns each stage only
  // their slice (gridDim.y groups x row partitions).  Inside the train step this beats the generic LDS-tiled kernel
  // for the 768-wide q|k|v projection (66 -> 53 us) and loses slightly at 256 wide (36 vs 40 us, although it wins
  // in isolation: 24 vs 33 us), hence N >= 512.
  int col_group = 0;
  if (aligned && !bf16 && !mbvae_opt(MBVAE_OPT_NO_TSGEMM2) && !mbvae_opt(MBVAE_OPT_NO_COLGROUPS) && g.M >= 4096 && g.M % 32 == 0 &&
      g.K % 64 == 0 && g.K <= 384 && g.N % 64 == 0 && g.N >= 512 && ((int64_t)g.K * g.N > 32768 || g.M < 32768) &&
      !g.addend && !g.c_gate && !g.accumulate && (int64_t)g.M * g.ldc < 0x3fffffff)
    col_group = 64;
  if (!aligned || (!col_group && (g.M < 32768 || (int64_t)g.K * g.N > 32768)) || g.K % 32 || g.K > 512) return MBVAE_EUNSUPPORTED;
  // the wave-autonomous kernel keeps only the weights in LDS (up to 128 KB of them)
  if ((bf16 || !mbvae_opt(MBVAE_OPT_NO_TSGEMM2)) && g.K % 64 == 0 && g.M % 32 == 0 && g.N % 64 == 0 && !g.accumulate &&
      !(g.addend && g.drop_p > 0.f) && (int64_t)g.M * g.ldc < 0x3fffffff) {
    const int ngroups = col_group ? g.N / col_group : 1, neff = col_group ? col_group : g.N;
    const int grid2 = std::min(std::max(g.M / 256, 1), std::max(ts_num_cus() / ngroups, 1));
    // 4 column tiles per wave (64 stores per lane and task) unless the gate operand needs the registers
    const bool wide = neff % 128 == 0 && !g.a_gate && !g.addend && !g.c_gate;
    const int waves = 8;
    // bf16-pipeline variants: weight planes pre-split in LDS where they fit, else split by the lanes as they are read
    const size_t smem3 = 2 * 3 * (size_t)neff * (g.K + 8), smemf = sizeof(float) * (size_t)g.K * (neff + 1);
    const bool split = bf16 || !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA);
    const bool split1 = split && smem3 <= MBVAE_LDS_LIMIT, split2 = split && !split1;
    const size_t smem2 = split1 ? smem3 : smemf;
    void (*k)(GemmArgs) = nullptr;
    bool lds_ok = false;
#define TS2_PICK(cond, ...)                                                                                         \
    if (k == nullptr && (cond)) { k = __VA_ARGS__; lds_ok = MBVAE_ALLOW_BIG_LDS((__VA_ARGS__)); }
    if (bf16) {                                    // bf16 storage: the shapes of the edge block's large activations
      const bool ab = g.a_bf16 != 0, cb = g.c_bf16 != 0;
      TS2_PICK(ab && !cb && split2 && g.addend, tsgemm2_kernel<2, B_KC, false, 8, true, 2, true, false>)
      TS2_PICK(ab && !cb && split2 && !g.addend, tsgemm2_kernel<2, B_KC, false, 8, false, 2, true, false>)
      TS2_PICK(ab && !cb && split1 && !g.addend, tsgemm2_kernel<2, B_KC, false, 8, false, 1, true, false>)
      TS2_PICK(!ab && cb && split1 && wide, tsgemm2_kernel<4, B_KC, false, 8, false, 1, false, true>)
      if (k == nullptr) return MBVAE_EUNSUPPORTED;
    }
    TS2_PICK(split1 && g.c_gate, tsgemm2_kernel<2, B_KC, false, 8, false, 1, false, false, true>)
    TS2_PICK(g.c_gate && !split, tsgemm2_kernel<2, B_KC, false, 8, false, 0, false, false, true>)
    if (g.c_gate && k == nullptr) return MBVAE_EUNSUPPORTED;
    TS2_PICK(split1 && wide, tsgemm2_kernel<4, B_KC, false, 8, false, 1>)
    TS2_PICK(split1 && g.addend && g.a_gate, tsgemm2_kernel<2, B_KC, true, 8, true, 1>)
    TS2_PICK(split1 && g.addend, tsgemm2_kernel<2, B_KC, false, 8, true, 1>)
    TS2_PICK(split1 && g.a_gate, tsgemm2_kernel<2, B_KC, true, 8, false, 1>)
    TS2_PICK(split1, tsgemm2_kernel<2, B_KC, false, 8, false, 1>)
    TS2_PICK(split2 && wide, tsgemm2_kernel<4, B_KC, false, 8, false, 2>)
    TS2_PICK(split2 && g.addend && !g.a_gate, tsgemm2_kernel<2, B_KC, false, 8, true, 2>)
    TS2_PICK(split2 && !g.addend && !g.a_gate, tsgemm2_kernel<2, B_KC, false, 8, false, 2>)
    TS2_PICK(wide, tsgemm2_kernel<4, B_KC, false, 8>)
    TS2_PICK(g.addend && g.a_gate, tsgemm2_kernel<2, B_KC, true, 8, true>)
    TS2_PICK(g.addend, tsgemm2_kernel<2, B_KC, false, 8, true>)
    TS2_PICK(g.a_gate, tsgemm2_kernel<2, B_KC, true, 8>)
    TS2_PICK(true, tsgemm2_kernel<2, B_KC, false, 8>)
#undef TS2_PICK
    if (!lds_ok || smem2 > MBVAE_LDS_LIMIT) return MBVAE_ELAUNCH;
    GemmArgs gl = g;
    gl.col_group = col_group;
    hipLaunchKernelGGL(k, dim3(grid2, ngroups), dim3(64 * waves), smem2, stream, gl);
    return mbvae_launch_status();
  }
  if (col_group) return MBVAE_EUNSUPPORTED;
  if (bf16 || g.addend || g.c_gate || g.K > 256 || (int64_t)g.K * g.N > 16384) return MBVAE_EUNSUPPORTED;
  // 64-wide outputs are HBM-bound on the A stream; the generic kernel keeps more loads in flight per CU there (measured).
  if (g.N != 128 && g.N != 256 && !(g.N == 64 && mbvae_opt(MBVAE_OPT_TSGEMM_N64))) return MBVAE_EUNSUPPORTED;
  const int BM = g.N == 256 ? 128 : 256;
  if (g.M % BM) return MBVAE_EUNSUPPORTED;
  const size_t smem = sizeof(float) * ((size_t)g.K * (g.N + 1) + 2 * 32 * (BM + 1) + 2 * BM);
  const int grid = std::min(g.M / BM, ts_num_cus());
  const bool gate = g.a_gate != nullptr;
#define TS_LAUNCH(n, bm, wm, wn)                                                                                     \
  do {                                                                                                                \
    auto kg = tsgemm_kernel<n, bm, wm, wn, B_KC, true>;                                                               \
    auto kn = tsgemm_kernel<n, bm, wm, wn, B_KC, false>;                                                              \
    auto k = gate ? kg : kn;                                                                                          \
    if (!(gate ? MBVAE_ALLOW_BIG_LDS(kg) : MBVAE_ALLOW_BIG_LDS(kn)) || smem > MBVAE_LDS_LIMIT) return MBVAE_ELAUNCH;  \
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, g);                                                    \
  } while (0)
  if (g.N == 256) TS_LAUNCH(256, 128, 2, 4);
  else if (g.N == 128) TS_LAUNCH(128, 256, 4, 2);
  else TS_LAUNCH(64, 256, 4, 2);
#undef TS_LAUNCH
  return mbvae_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// Persistent weight gradient for the edge path: dW[P,Q] = pro(dY)[R,P]^T @ X[R,Q] with R ~ 10^6 rows and an output
// small enough (P*Q <= 16384) that ONE workgroup holds all of it in MFMA accumulators.  Every row of dY and X is then
// read from HBM exactly once (an output-tiled kernel re-reads each operand once per tile of the other).  One
// 512-thread workgroup per CU walks over 32-row chunks dealt round-robin; both operands are row-contiguous, so chunks
// are copied into LDS as they are (k-major for the MFMA fetch).  Two chunks are in flight in registers (40 KB per CU
// at P+Q = 320) underneath the MFMA stream, which is what keeps HBM busy with a single workgroup per CU.  Each
// workgroup writes its partial dW (and partial column sums of pro(dY), the bias gradient) to the workspace; a second
// kernel adds the partials in a fixed order, so results are bitwise reproducible.
template <int P, int Q, int WM, int WN, bool GATE>
__global__ __launch_bounds__(512) void tswgrad_kernel(GemmArgs g, int nchunks) {
  constexpr int BR = 32;
  constexpr int TM = P / (32 * WM), TN = Q / (32 * WN);
  constexpr int LDA = P + 4, LDB = Q + 4;
  constexpr int NA = P / 64, NB = Q / 64;          // float4 per thread per chunk
  constexpr int RGA = 2048 / P, RGB = 2048 / Q;    // rows covered by one pass of the 512 threads
  static_assert(WM * WN == 8 && TM >= 1 && TN >= 1 && NA >= 1 && NB >= 1, "8 waves, 64 <= P,Q");
  extern __shared__ __align__(16) float tw_smem[];
  float* As = tw_smem;                             // [2][BR][LDA]
  float* Bs = As + 2 * BR * LDA;                   // [2][BR][LDB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps index arithmetic on the scalar unit
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int G = (int)gridDim.x, bx = (int)blockIdx.x;
  const int my = (nchunks - bx + G - 1) / G;
  const int arow = tid / (P / 4), ac4 = (tid % (P / 4)) * 4;
  const int brow = tid / (Q / 4), bc4 = (tid % (Q / 4)) * 4;

  float4 ra[2][NA], rg[2][NA];
  float rb[2][NB * 4];     // scalars: a float4 array that is only copied (never modified) stays in scratch memory
  unsigned ma[2][NA], mb[2][NA];
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // Every step of the main loop is unconditional (no loads or LDS writes under a branch): a chunk index past the end
  // is clamped to the last chunk and its rows are scaled by zero, so the staged registers have one definition per
  // step and the compiler can leave the loads in flight until the commit one step later.
  float vld[2];
#define TW_FETCH(c, S)                                                                                \
  {                                                                                                   \
    const int cc = min((c), my - 1);                                                                   \
    vld[S] = (c) < my ? g.a_gate_scale : 0.f;                                                          \
    const int64_t m0 = (int64_t)(bx + cc * G) * BR;                                                    \
    _Pragma("unroll") for (int p = 0; p < NA; ++p) {                                                   \
      const int64_t row = m0 + arow + RGA * p;                                                         \
      ra[S][p] = ld4_stream(g.A + row * g.lda + ac4);                            \
      if (GATE) rg[S][p] = ld4_stream(g.a_gate + row * g.lda + ac4);             \
      mbvae_row_mask_bytes(g.mask, g.a_mask_mode, row, g.n_nodes, g.A, ma[S][p], mb[S][p]);             \
    }                                                                                                  \
    _Pragma("unroll") for (int p = 0; p < NB; ++p) {                                                   \
      const float4 t = ld4_stream(g.B + (m0 + brow + RGB * p) * g.ldb + bc4);    \
      rb[S][4 * p] = t.x; rb[S][4 * p + 1] = t.y; rb[S][4 * p + 2] = t.z; rb[S][4 * p + 3] = t.w;      \
    }                                                                                                  \
  }
#define TW_COMMIT(c, S)                                                                               \
  {                                                                                                   \
    float* da = As + ((c) & 1) * BR * LDA;                                                             \
    float* db = Bs + ((c) & 1) * BR * LDB;                                                             \
    _Pragma("unroll") for (int p = 0; p < NA; ++p) {                                                   \
      float4 v = ra[S][p];                                                                             \
      if (GATE) {                                                                                      \
        if (!(rg[S][p].x > 0.f)) v.x = 0.f;                                                            \
        if (!(rg[S][p].y > 0.f)) v.y = 0.f;                                                            \
        if (!(rg[S][p].z > 0.f)) v.z = 0.f;                                                            \
        if (!(rg[S][p].w > 0.f)) v.w = 0.f;                                                            \
      }                                                                                                \
      const float rk = mbvae_mask_keep(g.a_mask_mode, ma[S][p], mb[S][p]) ? vld[S] : 0.f;              \
      v.x = rk != 0.f ? v.x * rk : 0.f; v.y = rk != 0.f ? v.y * rk : 0.f;                              \
      v.z = rk != 0.f ? v.z * rk : 0.f; v.w = rk != 0.f ? v.w * rk : 0.f;                              \
      cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;                                              \
      *reinterpret_cast<float4*>(da + (arow + RGA * p) * LDA + ac4) = v;                               \
    }                                                                                                  \
    _Pragma("unroll") for (int p = 0; p < NB; ++p) {                                                   \
      *reinterpret_cast<float4*>(db + (brow + RGB * p) * LDB + bc4) =                                  \
          make_float4(rb[S][4 * p], rb[S][4 * p + 1], rb[S][4 * p + 2], rb[S][4 * p + 3]);             \
    }                                                                                                  \
  }
#define TW_STEP(c, S)                                                                                 \
  {                                                                                                   \
    TW_FETCH((c) + 2, S)                                                                               \
    const float* Ab = As + ((c) & 1) * BR * LDA;                                                       \
    const float* Bb = Bs + ((c) & 1) * BR * LDB;                                                       \
    _Pragma("unroll 4") for (int kk = 0; kk < BR / 2; ++kk) {                                          \
      float a[TM], b[TN];                                                                              \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) a[i] = Ab[(2 * kk + h) * LDA + (wm * TM + i) * 32 + r]; \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) b[j] = Bb[(2 * kk + h) * LDB + (wn * TN + j) * 32 + r]; \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);            \
    }                                                                                                  \
    TW_COMMIT((c) + 1, 1 - S)                                                                          \
    __syncthreads();                                                                                   \
  }

  TW_FETCH(0, 0)
  TW_FETCH(1, 1)
  TW_COMMIT(0, 0)
  __syncthreads();
  for (int c = 0; c < my; c += 2) {
    TW_STEP(c, 0)
    TW_STEP(c + 1, 1)
  }
#undef TW_FETCH
#undef TW_COMMIT
#undef TW_STEP

  // ---- partial dW of this workgroup
  float* part = g.partial + (int64_t)bx * P * Q;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
#pragma unroll
      for (int j = 0; j < TN; ++j) part[row * Q + (wn * TN + j) * 32 + r] = acc[i][j][v];
    }
  // ---- partial column sums of pro(dY): threads with equal column group combine through LDS in a fixed order
  if (g.partial_colsum != nullptr) {
    float* tmp = As;                                // [512 / (P/4)][P]; the last chunk has been consumed (barrier above)
    *reinterpret_cast<float4*>(tmp + arow * P + ac4) = cs;
    __syncthreads();
    if (tid < P) {
      float s = 0.f;
      for (int q = 0; q < RGA; ++q) s += tmp[q * P + tid];
      g.partial_colsum[(int64_t)bx * P + tid] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wave-autonomous weight gradient: no LDS, no barrier.  Both operands of dW = pro(dY)^T X are row-contiguous, and the
// MFMA A/B operand layouts want "lane = output column, lane half = reduction row", so a lane can read its operand
// words straight from global memory: for the k-step over rows (r, r+1) lane (c, h) loads dY[r + h][p0 + 2c .. +1] and
// X[r + h][q0 + 2c .. +1] (two 8-byte loads, each a pair of fully used 256-byte row segments per wave).  The two
// words of a load feed two different 32x32 output tiles, i.e. a wave's 64x64 block of dW interleaves its two row
// tiles (even / odd rows) and its two column tiles.  Eight waves cover WM x WN such blocks and share rows through
// L1.  Loads run two 8-step groups (32 rows) ahead in a three-slot register ring; the wait for a group is pinned
// right after the previous group's MFMAs, where the operand prologue (ReLU gate, row mask, scale) and the bias
// gradient's column sums are applied in place.  Per-workgroup partials, fixed-order reduction as above.
template <int P, int Q, int WM, int WN, bool GATE>
__global__ __launch_bounds__(512) void tswgrad2_kernel(GemmArgs g, int nchunks) {
  static_assert(WM * WN == 8 && P == 64 * WM && Q == 64 * WN, "8 waves, one 64x64 block each");
  constexpr int KS = 8, GPC = 4;                   // k-steps per group, groups per 64-row chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps index arithmetic on the scalar unit
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int G = (int)gridDim.x, bx = (int)blockIdx.x;
  const int my = (nchunks - bx + G - 1) / G;
  const int total = my * GPC;
  // addresses = wave-uniform row base (scalar registers) + a per-lane 32-bit offset fixed for the whole kernel
  const unsigned offA = (unsigned)(h * g.lda + wm * 64 + 2 * r), offB = (unsigned)(h * g.ldb + wn * 64 + 2 * r);

  float2 ra[3][KS], rb[3][KS], rg[3][KS];
  unsigned ma[3], mb[3];
  float cs0 = 0.f, cs1 = 0.f;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

#define W2_ROW0(q) ((int64_t)(bx + (min((q), total - 1) / GPC) * G) * 64)
#define W2_FETCH(q, S)                                                                                \
  {                                                                                                   \
    const int qq = min((q), total - 1);                                                                \
    const int64_t c0 = W2_ROW0(q);                                                                     \
    const int64_t row = c0 + (qq % GPC) * 16;                                                          \
    mbvae_row_mask_bytes(g.mask, g.a_mask_mode, c0 + lane, g.n_nodes, g.A, ma[S], mb[S]);              \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                   \
      const float* arow = g.A + (row + 2 * s) * g.lda;                                                 \
      const float* brow = g.B + (row + 2 * s) * g.ldb;                                                 \
      ra[S][s] = *reinterpret_cast<const float2*>(arow + offA);                                        \
      if (GATE) rg[S][s] = *reinterpret_cast<const float2*>(g.a_gate + (row + 2 * s) * g.lda + offA);  \
      rb[S][s] = *reinterpret_cast<const float2*>(brow + offB);                                        \
    }                                                                                                  \
    asm volatile("" : : : "memory");                                                                   \
  }
#define W2_PREP(q, S)                                                                                 \
  {                                                                                                   \
    unsigned late;                                                                                     \
    asm volatile("v_mov_b32 %0, 0" : "=v"(late) : : "memory");                                         \
    const unsigned long long kb = __ballot(mbvae_mask_keep(g.a_mask_mode, ma[S] | late, mb[S] | late)); \
    const float sc = (q) < total ? g.a_gate_scale : 0.f;                                               \
    if (!GATE && kb == ~0ull && sc == 1.f) {           /* nothing to apply: only pin the loads, sum columns */ \
      _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                 \
        float2 v = ra[S][s];                                                                           \
        float2 b = rb[S][s];                                                                           \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(b.x), "+v"(b.y) : : "memory");                    \
        cs0 += v.x; cs1 += v.y;                                                                        \
        ra[S][s] = v; rb[S][s] = b;                                                                    \
      }                                                                                                \
    } else {                                                                                           \
      const unsigned kbits = (unsigned)(kb >> ((min((q), total - 1) % GPC) * 16 + h));  /* bit 2s: row 2s + h */ \
      _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                 \
        float2 v = ra[S][s];                                                                           \
        float2 b = rb[S][s];                                                                           \
        if (GATE) {                                                                                    \
          if (!(rg[S][s].x > 0.f)) v.x = 0.f;                                                          \
          if (!(rg[S][s].y > 0.f)) v.y = 0.f;                                                          \
        }                                                                                              \
        const float rk = ((kbits >> (2 * s)) & 1u) ? sc : 0.f;                                         \
        v.x = rk != 0.f ? v.x * rk : 0.f; v.y = rk != 0.f ? v.y * rk : 0.f;                            \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(b.x), "+v"(b.y) : : "memory");                    \
        cs0 += v.x; cs1 += v.y;                                                                        \
        ra[S][s] = v; rb[S][s] = b;                                                                    \
      }                                                                                                \
    }                                                                                                  \
  }
#define W2_MMA(S)                                                                                     \
  {                                                                                                   \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                   \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[S][s].x, rb[S][s].x, acc[0][0], 0, 0, 0);    \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[S][s].x, rb[S][s].y, acc[0][1], 0, 0, 0);    \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[S][s].y, rb[S][s].x, acc[1][0], 0, 0, 0);    \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[S][s].y, rb[S][s].y, acc[1][1], 0, 0, 0);    \
    }                                                                                                  \
  }

  if (total > 0) {
    W2_FETCH(0, 0)
    W2_FETCH(1, 1)
    W2_PREP(0, 0)
    for (int q = 0; q < total; q += 3) {
      W2_FETCH(q + 2, 2)
      W2_MMA(0)
      W2_PREP(q + 1, 1)
      W2_FETCH(q + 3, 0)
      W2_MMA(1)                     /* groups past the end were scaled to zero */
      W2_PREP(q + 2, 2)
      W2_FETCH(q + 4, 1)
      W2_MMA(2)
      W2_PREP(q + 3, 0)
    }
  }
#undef W2_ROW0
#undef W2_FETCH
#undef W2_PREP
#undef W2_MMA

  // ---- partial dW of this workgroup: tile (i, j) holds rows p = 64 wm + 2 m + i, columns 64 wn + 2 c + j
  float* part = g.partial + (int64_t)bx * P * Q;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int m = (v & 3) + 8 * (v >> 2) + 4 * h;
      const int row = wm * 64 + 2 * m + i;
#pragma unroll
      for (int j = 0; j < 2; ++j) part[row * Q + wn * 64 + 2 * r + j] = acc[i][j][v];
    }
  // ---- partial column sums of pro(dY): this lane summed the rows of its half; waves with wn == 0 own the columns
  if (g.partial_colsum != nullptr) {
    cs0 += __shfl_xor(cs0, 32, 64);
    cs1 += __shfl_xor(cs1, 32, 64);
    if (wn == 0 && h == 0) {
      g.partial_colsum[(int64_t)bx * P + wm * 64 + 2 * r] = cs0;
      g.partial_colsum[(int64_t)bx * P + wm * 64 + 2 * r + 1] = cs1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The wave-autonomous weight gradient on the bf16 matrix pipeline (three-way split, six MFMAs per product: see
// mbvae_split8).  The native fp32 MFMA keeps the kernel above matrix-core bound (63 % MFMA-busy next to a saturated
// HBM stream); at 2.7x fewer MFMA cycles per product this one waits for HBM only.  The bf16 MFMA reduces over 16 rows
// per instruction and wants "lane (c, h) holds rows 8h .. 8h+7 of column c": lane (c, h) therefore loads
// dY[row + 8h + i][p0 + 2c .. +1] and X[row + 8h + i][q0 + 2c .. +1] for i = 0..7 (8-byte loads, 256-byte row
// segments per half-wave), which ARE the operand octets of the even / odd column tiles once split - no transposition,
// no LDS, no barrier.  A wave owns one 64 x 64 block of dW; outputs with fewer than eight blocks let the spare waves
// take every RS-th 16-row group, each (workgroup, row split) writing its own partial for the fixed-order reduction.
// ABF / BBF: dY / X is bf16 in memory (opt-in bf16 storage of the large edge activations): a lane's 4-byte load then
// holds its two columns, the octets are assembled with byte permutes, and a bf16 operand - being its own hi plane -
// drops the product terms of its (zero) mid and lo planes.
template <int P, int Q, bool GATE, bool ABF = false, bool BBF = false>
__global__ __launch_bounds__(512) void tswgrad3_kernel(GemmArgs g, int nchunks) {
  static_assert(!(ABF && GATE), "no gate on a bf16 operand");
  constexpr int NBQ = Q / 64, NBLK = (P / 64) * NBQ, RS = 8 / NBLK, GPC = 4;
  static_assert(P % 64 == 0 && Q % 64 == 0 && NBLK >= 1 && NBLK <= 8 && RS * NBLK == 8, "1, 2, 4 or 8 blocks of 64 x 64");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int blk = wave % NBLK, rs = wave / NBLK;
  const int wm = blk / NBQ, wn = blk % NBQ;
  const int G = (int)gridDim.x, bx = (int)blockIdx.x;
  const int my = (nchunks - bx + G - 1) / G;            // 64-row chunks of this workgroup
  const int total = (my * GPC - rs + RS - 1) / RS;      // 16-row groups of this wave: rs, rs + RS, ... of the workgroup's
  const unsigned offA = (unsigned)(8 * h * g.lda + wm * 64 + 2 * c), offB = (unsigned)(8 * h * g.ldb + wn * 64 + 2 * c);

  float2 ra[3][8], rb[3][8], rg[3][8];
  unsigned ma[3], mb[3];
  float cs0 = 0.f, cs1 = 0.f;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

#define W3_GROUP(q) (min((q), total - 1) * RS + rs)
#define W3_FETCH(q, S)                                                                                \
  {                                                                                                   \
    const int gq = W3_GROUP(q);                                                                        \
    const int64_t c0 = (int64_t)(bx + (gq / GPC) * G) * 64;                                            \
    const int64_t row = c0 + (gq % GPC) * 16;                                                          \
    mbvae_row_mask_bytes(g.mask, g.a_mask_mode, c0 + lane, g.n_nodes, g.A, ma[S], mb[S]);              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                    \
      if (ABF) ra[S][i].x = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.A) + (row + i) * g.lda + offA); \
      else ra[S][i] = *reinterpret_cast<const float2*>(g.A + (row + i) * g.lda + offA);                \
      if (GATE) rg[S][i] = *reinterpret_cast<const float2*>(g.a_gate + (row + i) * g.lda + offA);      \
      if (BBF) rb[S][i].x = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.B) + (row + i) * g.ldb + offB); \
      else rb[S][i] = *reinterpret_cast<const float2*>(g.B + (row + i) * g.ldb + offB);                \
    }                                                                                                  \
    asm volatile("" : : : "memory");                                                                   \
  }
#define W3_PREP(q, S)                                                                                 \
  {                                                                                                   \
    unsigned late;                                                                                     \
    asm volatile("v_mov_b32 %0, 0" : "=v"(late) : : "memory");                                         \
    const unsigned long long kb = __ballot(mbvae_mask_keep(g.a_mask_mode, ma[S] | late, mb[S] | late)); \
    const float sc = (q) < total ? g.a_gate_scale : 0.f;                                               \
    const bool plain = !GATE && kb == ~0ull && sc == 1.f;   /* nothing to apply: only pin the loads, sum columns */ \
    const unsigned kbits = (unsigned)(kb >> ((W3_GROUP(q) % GPC) * 16 + 8 * h));   /* bit i: row 8h + i */ \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                    \
      float2 v = ra[S][i];                                                                             \
      float2 b = rb[S][i];                                                                             \
      if (ABF) {                                                                                       \
        if (!plain) v.x = (((kbits >> i) & 1u) && sc != 0.f) ? v.x : 0.f;    /* both columns' bits at once */ \
        asm volatile("" : "+v"(v.x), "+v"(b.x) : : "memory");                                          \
        if (!BBF) asm volatile("" : "+v"(b.y) : : "memory");                                           \
        cs0 += __uint_as_float(__float_as_uint(v.x) << 16); cs1 += __uint_as_float(__float_as_uint(v.x) & 0xffff0000u); \
      } else {                                                                                         \
        if (!plain) {                                                                                  \
          if (GATE) {                                                                                  \
            if (!(rg[S][i].x > 0.f)) v.x = 0.f;                                                        \
            if (!(rg[S][i].y > 0.f)) v.y = 0.f;                                                        \
          }                                                                                            \
          const float rk = ((kbits >> i) & 1u) ? sc : 0.f;                                             \
          v.x = rk != 0.f ? v.x * rk : 0.f; v.y = rk != 0.f ? v.y * rk : 0.f;                          \
        }                                                                                              \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(b.x) : : "memory");                               \
        if (!BBF) asm volatile("" : "+v"(b.y) : : "memory");                                           \
        cs0 += v.x; cs1 += v.y;                                                                        \
      }                                                                                                \
      ra[S][i] = v; rb[S][i] = b;                                                                      \
    }                                                                                                  \
  }
  // smallest terms first; the four tiles alternate so that consecutive MFMAs never wait on each other's accumulator
#define W3_TERM(pa, pb, AH, BH)                                                                       \
    if ((AH || !ABF) && (BH || !BBF)) {                                                                \
      acc[0][0] = MBVAE_MFMA_BF16(a0.pa, b0.pb, acc[0][0]); acc[0][1] = MBVAE_MFMA_BF16(a0.pa, b1.pb, acc[0][1]); \
      acc[1][0] = MBVAE_MFMA_BF16(a1.pa, b0.pb, acc[1][0]); acc[1][1] = MBVAE_MFMA_BF16(a1.pa, b1.pb, acc[1][1]); \
    }
  // a lane's eight rows of one operand -> the split planes of its even (x0) and odd (x1) column
#define W3_OPERAND(BF, rr, x0, x1)                                                                    \
    MbvaeSplit3 x0, x1;                                                                                \
    if (BF) {                                                                                          \
      mbvae_u32x4 e_, o_;                                                                              \
      _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                  \
        const unsigned w0 = __float_as_uint(rr[2 * p].x), w1 = __float_as_uint(rr[2 * p + 1].x);       \
        e_[p] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);   /* low halves: the even column's rows 2p, 2p+1 */ \
        o_[p] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);   /* high halves: the odd column's */        \
      }                                                                                                \
      x0.hi = x0.mid = x0.lo = __builtin_bit_cast(mbvae_bf16x8, e_);                                   \
      x1.hi = x1.mid = x1.lo = __builtin_bit_cast(mbvae_bf16x8, o_);                                   \
    } else {                                                                                           \
      x0 = mbvae_split8(rr[0].x, rr[1].x, rr[2].x, rr[3].x, rr[4].x, rr[5].x, rr[6].x, rr[7].x);       \
      x1 = mbvae_split8(rr[0].y, rr[1].y, rr[2].y, rr[3].y, rr[4].y, rr[5].y, rr[6].y, rr[7].y);       \
    }
#define W3_MMA(S)                                                                                     \
  {                                                                                                   \
    W3_OPERAND(ABF, ra[S], a0, a1)                                                                     \
    W3_OPERAND(BBF, rb[S], b0, b1)                                                                     \
    W3_TERM(lo, hi, false, true) W3_TERM(hi, lo, true, false) W3_TERM(mid, mid, false, false)          \
    W3_TERM(mid, hi, false, true) W3_TERM(hi, mid, true, false) W3_TERM(hi, hi, true, true)            \
  }

  if (total > 0) {
    W3_FETCH(0, 0)
    W3_FETCH(1, 1)
    W3_PREP(0, 0)
    for (int q = 0; q < total; q += 3) {
      W3_FETCH(q + 2, 2)
      W3_MMA(0)
      W3_PREP(q + 1, 1)
      W3_FETCH(q + 3, 0)
      W3_MMA(1)                     /* groups past the end were scaled to zero */
      W3_PREP(q + 2, 2)
      W3_FETCH(q + 4, 1)
      W3_MMA(2)
      W3_PREP(q + 3, 0)
    }
  }
#undef W3_GROUP
#undef W3_FETCH
#undef W3_PREP
#undef W3_TERM
#undef W3_OPERAND
#undef W3_MMA

  // ---- partial dW of this (workgroup, row split): tile (i, j) holds rows 64 wm + 2 m + i, columns 64 wn + 2 c + j
  const int slot = bx * RS + rs;
  float* part = g.partial + (int64_t)slot * P * Q;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int m = (v & 3) + 8 * (v >> 2) + 4 * h;
      const int row = wm * 64 + 2 * m + i;
#pragma unroll
      for (int j = 0; j < 2; ++j) part[row * Q + wn * 64 + 2 * c + j] = acc[i][j][v];
    }
  if (g.partial_colsum != nullptr) {
    cs0 += __shfl_xor(cs0, 32, 64);
    cs1 += __shfl_xor(cs1, 32, 64);
    if (wn == 0 && h == 0) {
      g.partial_colsum[(int64_t)slot * P + wm * 64 + 2 * c] = cs0;
      g.partial_colsum[(int64_t)slot * P + wm * 64 + 2 * c + 1] = cs1;
    }
  }
}

// Sums `splits` partial [M,N] results (+ partial column sums) in a fixed order: 4 waves per 64 consecutive outputs,
// wave z adds the partials z, z+4, ..., the four sums are combined through LDS.
__global__ __launch_bounds__(256) void tswgrad_reduce_kernel(GemmArgs g, int splits) {
  __shared__ float part[4][64];
  const int64_t total = (int64_t)g.M * g.N;
  const int lane = threadIdx.x & 63, z0 = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < total) {
#pragma unroll 8
    for (int z = z0; z < splits; z += 4) s += g.partial[(int64_t)z * total + e];
  } else if (e < total + g.M && g.colsum) {
#pragma unroll 8
    for (int z = z0; z < splits; z += 4) s += g.partial_colsum[(int64_t)z * g.M + (e - total)];
  }
  part[z0][lane] = s;
  __syncthreads();
  if (z0 == 0) {
    s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (e < total) {
      const int row = (int)(e / g.N), col = (int)(e - (int64_t)row * g.N);
      if (g.accumulate) s += g.C[(int64_t)row * g.ldc + col];
      g.C[(int64_t)row * g.ldc + col] = s;
    } else if (e < total + g.M && g.colsum) {
      g.colsum[e - total] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Both gradients of the widest edge projection (64 -> 512: e_mul | e_add) from ONE pass over dY [M, 512]:
//     dX = addend + keep(dY) W        dW = keep(dY)^T X        db = column sums of keep(dY)
// As separate kernels each of them reads the 2 KB of a dY row from HBM; here the second reader finds it in the cache.
// Wave w owns the 64 columns [64w, 64w+64) of dY for BOTH products, 16 rows at a time:
//  * dW block [64 x 64] exactly as in the weight-gradient kernel above (lane (c, h) loads rows 8h .. 8h+7 of its two
//    columns: the octets of the 32x32x16 operand; four accumulator tiles per wave for the whole kernel),
//  * its share of dX [16 x 64] - the sum over its 64 columns - on the 16x16x32 shape, whose operand layout (lane
//    (row, q) holds columns 8q .. 8q+7 of a 32-wide step) is the same tile read again row-wise, right after the first
//    read brought it into L1/L2.  The weight slice is split once per kernel: the hi and six of the mid planes in
//    thread-private LDS slots (112 KB), the other planes in registers.  The X rows of a group, which all eight waves
//    multiply with, are split once per workgroup into bf16 planes in LDS, two groups ahead.
// The eight partial dX tiles meet in the remaining 32 KB of LDS (column blocks XOR-ed with the row quad: conflict-free
// both ways) and are added in wave order, so dX is reproducible; dW / db partials go to the workspace for the same
// fixed-order reduction kernel as above.
struct ProjBwdArgs {
  const float* dY; int lddy;
  const float* X; int ldx;
  const float* W;                       // [512][64]
  const float* addend; float* dX; int lddx;
  float* partial; float* partial_colsum;
  const uint8_t* mask; int mask_mode; int n_nodes;
  int nchunks;                          // M / 64
};
typedef float mbvae_f32x4 __attribute__((ext_vector_type(4)));
#define PB_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

template <bool HAS_ADD>
__global__ __launch_bounds__(512) void proj_bwd_kernel(ProjBwdArgs g) {
  extern __shared__ __align__(16) float pb_smem[];
  // LDS (156 KB): 14 thread-private 16-byte slots with the hi planes (8) and six of the mid planes of the lane's W octets,
  // the partial dX tiles [8 waves][16 rows][64], and two buffers with the bf16 planes of a group's X rows, which every
  // wave multiplies with: split once per workgroup, laid out [plane][column parity][row half][column pair][8 rows] so
  // that a lane's operand octet is one conflict-free 16-byte read
  mbvae_bf16x8* wl = reinterpret_cast<mbvae_bf16x8*>(pb_smem);
  float* part = pb_smem + 14 * 2048;
  unsigned char* xp = reinterpret_cast<unsigned char*>(pb_smem) + 14 * 8192 + 32768;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;          // 32x32x16 operand coordinates (weight gradient)
  const int m16 = lane & 15, q4 = lane >> 4;       // 16x16x32 operand coordinates (input gradient)
  const int n0 = wave * 64;
  const int G = (int)gridDim.x, bx = (int)blockIdx.x;
  const int my = (g.nchunks - bx + G - 1) / G;
  const int total = my * 4;                         // 16-row groups of this workgroup

  mbvae_bf16x8 wlo[8], wmr[2];                      // lo planes and the last two mid planes: registers
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      const float* wp = g.W + (int64_t)(n0 + 32 * t + 8 * q4) * 64 + 16 * et + m16;
      const MbvaeSplit3 sp = mbvae_split8(wp[0], wp[64], wp[128], wp[192], wp[256], wp[320], wp[384], wp[448]);
      wl[(4 * t + et) * 512 + tid] = sp.hi;
      if (4 * t + et < 6) wl[(8 + 4 * t + et) * 512 + tid] = sp.mid; else wmr[4 * t + et - 6] = sp.mid;
      wlo[4 * t + et] = sp.lo;
    }

  // all streams go through buffer descriptors rebuilt per 16-row group (scalar work): a lane keeps one 32-bit offset per
  // stream instead of 64-bit addresses
  typedef unsigned pb_u32x2 __attribute__((ext_vector_type(2)));
  const int voA = (8 * h * g.lddy + n0 + 2 * c) * 4;
  const int voD = (m16 * g.lddy + n0 + 8 * q4) * 4;
  const int rrow = tid >> 5, re2 = (tid & 31) * 2;                 // this thread's dX outputs - and its share of the X rows: row rrow, columns re2, re2 + 1
  const int rcol = re2 ^ (((rrow >> 2) & 3) << 4);
  const int voR = (rrow * g.lddx + re2) * 4, voX = (rrow * g.ldx + re2) * 4;
  const int szY = 16 * g.lddy * 4, szX = 16 * g.ldx * 4, szR = 16 * g.lddx * 4;
  const int xw = (rrow >> 3) * 512 + (re2 >> 1) * 16 + (rrow & 7) * 2;   // where this thread's two X values go (column parity 0; +1024: parity 1)
  const int xr = h * 512 + c * 16;                                       // this lane's operand octets
  float2 ra[2][8], ad[2], xq;
  unsigned ma[2], mb[2];
  float cs0 = 0.f, cs1 = 0.f;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // The weight gradient's view of a dY tile (with the addend and the mask bytes) is prefetched one group ahead, the X
  // rows two groups ahead; the input gradient's view is read when the group starts - the first read has brought the
  // tile into L1/L2 by then - and BEFORE the next prefetch is issued: the memory counter is in order, so waiting for
  // that read must not mean waiting for the prefetch.  (Letting the two waves of a SIMD take the products in opposite
  // order, one splitting operands while the other multiplies, measured 14 % slower.)
#define PB_ROW0(qq, c0, row)                                                                           \
    const int gq_ = min((qq), total - 1);                                                              \
    const int64_t c0 = (int64_t)(bx + (gq_ >> 2) * G) * 64;                                            \
    const int64_t row = c0 + (gq_ & 3) * 16;
#define PB_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, (bytes), 0x00020000)
#define PB_LOAD_COLS(dst, rsY)       /* weight-gradient view: rows 8h .. 8h+7 of the lane's two columns */ \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                      \
      dst[i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsY, voA, i * g.lddy * 4, 0));
#define PB_LOAD_ROWS(dst, rsY)       /* input-gradient view: columns 8q .. 8q+7 of both 32-wide steps of the lane's row */ \
    dst[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voD, 0, 0));        \
    dst[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voD + 16, 0, 0));   \
    dst[2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voD + 128, 0, 0));  \
    dst[3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voD + 144, 0, 0));
#define PB_LOAD_X(qq)                                                                                  \
  {                                                                                                   \
    PB_ROW0(qq, c0x, rowx)                                                                             \
    xq = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(PB_RSRC(g.X + rowx * g.ldx, szX), voX, 0, 0)); \
  }
  // this thread's two X values -> their bf16 planes in buffer `buf`
#define PB_STAGE_X(buf)                                                                               \
  {                                                                                                   \
    unsigned char* dst = xp + (buf) * 6144 + xw;                                                       \
    const float xv[2] = {xq.x, xq.y};                                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
      const unsigned u = __float_as_uint(xv[j]);                                                       \
      const float r1 = xv[j] - __uint_as_float(u & 0xffff0000u);                                       \
      const unsigned v = __float_as_uint(r1);                                                          \
      const float r2 = r1 - __uint_as_float(v & 0xffff0000u);                                          \
      *reinterpret_cast<unsigned short*>(dst + j * 1024) = (unsigned short)(u >> 16);                  \
      *reinterpret_cast<unsigned short*>(dst + 2048 + j * 1024) = (unsigned short)(v >> 16);           \
      *reinterpret_cast<unsigned short*>(dst + 4096 + j * 1024) = (unsigned short)(__float_as_uint(r2) >> 16); \
    }                                                                                                  \
  }
#define PB_FETCH(qq, S)                                                                               \
  {                                                                                                   \
    PB_ROW0(qq, c0, row)                                                                               \
    mbvae_row_mask_bytes(g.mask, g.mask_mode, c0 + lane, g.n_nodes, g.dY, ma[S], mb[S]);               \
    const __amdgpu_buffer_rsrc_t rsY = PB_RSRC(g.dY + row * g.lddy, szY);                              \
    PB_LOAD_COLS(ra[S], rsY)                                                                           \
    if (HAS_ADD) ad[S] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(PB_RSRC(g.addend + row * g.lddx, szR), voR, 0, 0)); \
    asm volatile("" : : : "memory");                                                                   \
  }
#define PB_TERM(pa, pb)                                                                               \
    acc[0][0] = MBVAE_MFMA_BF16(a0.pa, b0.pb, acc[0][0]); acc[0][1] = MBVAE_MFMA_BF16(a0.pa, b1.pb, acc[0][1]); \
    acc[1][0] = MBVAE_MFMA_BF16(a1.pa, b0.pb, acc[1][0]); acc[1][1] = MBVAE_MFMA_BF16(a1.pa, b1.pb, acc[1][1]);
  // masked rows of dY become zeros (selects: they may hold anything), column sums, then the four-tile product
#define PB_WGRAD(RA, buf)                                                                             \
    {                                                                                                  \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
        float2 v = RA[i];                                                                              \
        if (krows != 0xffffu) {                                                                        \
          const bool kr = (kbits >> i) & 1u;                                                           \
          v.x = kr ? v.x : 0.f; v.y = kr ? v.y : 0.f;                                                  \
        }                                                                                              \
        asm volatile("" : "+v"(v.x), "+v"(v.y) : : "memory");                                          \
        cs0 += v.x; cs1 += v.y;                                                                        \
        RA[i] = v;                                                                                     \
      }                                                                                                \
      const unsigned char* xb = xp + (buf) * 6144 + xr;                                                \
      MbvaeSplit3 b0, b1;                                                                              \
      b0.hi = *reinterpret_cast<const mbvae_bf16x8*>(xb);         b1.hi = *reinterpret_cast<const mbvae_bf16x8*>(xb + 1024);  \
      b0.mid = *reinterpret_cast<const mbvae_bf16x8*>(xb + 2048); b1.mid = *reinterpret_cast<const mbvae_bf16x8*>(xb + 3072); \
      b0.lo = *reinterpret_cast<const mbvae_bf16x8*>(xb + 4096);  b1.lo = *reinterpret_cast<const mbvae_bf16x8*>(xb + 5120);  \
      const MbvaeSplit3 a0 = mbvae_split8(RA[0].x, RA[1].x, RA[2].x, RA[3].x, RA[4].x, RA[5].x, RA[6].x, RA[7].x); \
      const MbvaeSplit3 a1 = mbvae_split8(RA[0].y, RA[1].y, RA[2].y, RA[3].y, RA[4].y, RA[5].y, RA[6].y, RA[7].y); \
      PB_TERM(lo, hi) PB_TERM(hi, lo) PB_TERM(mid, mid) PB_TERM(mid, hi) PB_TERM(hi, mid) PB_TERM(hi, hi) \
    }
#define PB_DGRAD(DA)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                    \
      float4 x0 = DA[2 * t], x1 = DA[2 * t + 1];                                                       \
      asm volatile("" : "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x0.w), "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x1.w) : : "memory"); \
      const MbvaeSplit3 a = mbvae_split8(x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w);              \
      _Pragma("unroll") for (int et = 0; et < 4; ++et) {                                               \
        const mbvae_bf16x8 wh = wl[(4 * t + et) * 512 + tid];                                          \
        const mbvae_bf16x8 wm = 4 * t + et < 6 ? wl[(8 + 4 * t + et) * 512 + tid] : wmr[(4 * t + et - 6) & 1]; \
        dxa[et] = PB_MFMA16(a.lo, wh, dxa[et]);  dxa[et] = PB_MFMA16(a.hi, wlo[4 * t + et], dxa[et]);  \
        dxa[et] = PB_MFMA16(a.mid, wm, dxa[et]); dxa[et] = PB_MFMA16(a.mid, wh, dxa[et]);              \
        dxa[et] = PB_MFMA16(a.hi, wm, dxa[et]);  dxa[et] = PB_MFMA16(a.hi, wh, dxa[et]);               \
      }                                                                                                \
    }
#define PB_STEP(qq, S, S1)                                                                            \
  {                                                                                                   \
    PB_ROW0(qq, c0, row)                                                                               \
    unsigned late;                                                                                     \
    asm volatile("v_mov_b32 %0, 0" : "=v"(late) : : "memory");                                         \
    const unsigned long long kb = __ballot(mbvae_mask_keep(g.mask_mode, ma[S] | late, mb[S] | late));  \
    const unsigned krows = (unsigned)(kb >> ((gq_ & 3) * 16)) & 0xffffu;   /* bit i: row i of the group */ \
    const unsigned kbits = krows >> (8 * h);                                                           \
    mbvae_f32x4 dxa[4];                                                                                \
    _Pragma("unroll") for (int et = 0; et < 4; ++et) dxa[et] = mbvae_f32x4{0.f, 0.f, 0.f, 0.f};        \
    const __amdgpu_buffer_rsrc_t rsY2 = PB_RSRC(g.dY + row * g.lddy, szY);                             \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)        /* the prefetched view has landed */          \
      asm volatile("" : "+v"(ra[S][i].x), "+v"(ra[S][i].y) : : "memory");                              \
    float4 dl[4];                                                                                      \
    PB_LOAD_ROWS(dl, rsY2)                                                                             \
    asm volatile("" : : : "memory");                                                                   \
    PB_FETCH((qq) + 1, S1)                                                                             \
    PB_LOAD_X((qq) + 2)                                                                                \
    asm volatile("" : : : "memory");                                                                   \
    PB_WGRAD(ra[S], S)                                                                                 \
    PB_DGRAD(dl)                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");   /* the previous group's partial tiles have been read; every wave is done with this group's X planes (its LDS reads have RETURNED: the products that consume them may be scheduled past this point) */ \
    _Pragma("unroll") for (int et = 0; et < 4; ++et)                                                   \
      _Pragma("unroll") for (int v = 0; v < 4; ++v)                                                    \
        part[wave * 1024 + (4 * q4 + v) * 64 + ((et ^ q4) << 4) + m16] = dxa[et][v];                   \
    PB_STAGE_X(S)                                      /* the group after next: read two barriers from here */ \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");                                  \
    {                                                                                                  \
      float2 s2 = *reinterpret_cast<const float2*>(part + rrow * 64 + rcol);                           \
      _Pragma("unroll") for (int w = 1; w < 8; ++w) {                                                  \
        const float2 p2 = *reinterpret_cast<const float2*>(part + w * 1024 + rrow * 64 + rcol);        \
        s2.x += p2.x; s2.y += p2.y;                                                                    \
      }                                                                                                \
      const bool kr = (krows >> rrow) & 1u;                                                            \
      s2.x = kr ? s2.x : 0.f; s2.y = kr ? s2.y : 0.f;                                                  \
      if (HAS_ADD) { s2.x += ad[S].x; s2.y += ad[S].y; }                                               \
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pb_u32x2, s2), PB_RSRC(g.dX + row * g.lddx, szR), voR, 0, 0); \
    }                                                                                                  \
  }

  PB_LOAD_X(0)
  PB_STAGE_X(0)
  PB_LOAD_X(1)
  PB_STAGE_X(1)
  PB_FETCH(0, 0)
  __syncthreads();
  for (int q = 0; q < total; q += 2) {
    PB_STEP(q, 0, 1)
    PB_STEP(q + 1, 1, 0)
  }
#undef PB_ROW0
#undef PB_RSRC
#undef PB_LOAD_COLS
#undef PB_LOAD_ROWS
#undef PB_LOAD_X
#undef PB_STAGE_X
#undef PB_FETCH
#undef PB_TERM
#undef PB_WGRAD
#undef PB_DGRAD
#undef PB_STEP

  // partial dW of this workgroup: tile (i, j) holds rows 64 wave + 2 m + i, columns 2 c + j
  float* pw = g.partial + (int64_t)bx * 512 * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int m = (v & 3) + 8 * (v >> 2) + 4 * h;
      const int row = wave * 64 + 2 * m + i;
#pragma unroll
      for (int j = 0; j < 2; ++j) pw[row * 64 + 2 * c + j] = acc[i][j][v];
    }
  if (g.partial_colsum != nullptr) {
    cs0 += __shfl_xor(cs0, 32, 64);
    cs1 += __shfl_xor(cs1, 32, 64);
    if (h == 0) {
      g.partial_colsum[(int64_t)bx * 512 + wave * 64 + 2 * c] = cs0;
      g.partial_colsum[(int64_t)bx * 512 + wave * 64 + 2 * c + 1] = cs1;
    }
  }
}

// returns MBVAE_EUNSUPPORTED when the shape does not qualify (the caller then uses the generic kernel)
static int tswgrad_try(GemmArgs g, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  if (mbvae_opt(MBVAE_OPT_NO_TSWGRAD)) return MBVAE_EUNSUPPORTED;
  const int P = g.M, Q = g.N, R = g.K;
  const bool aligned = (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.a_gate) & 15) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0;
  if (!aligned || R < 32768 || R % 32 || P * Q > 32768 || g.bias || g.act || g.drop_p > 0.f || g.c_mask_mode != MBVAE_MASK_NONE)
    return MBVAE_EUNSUPPORTED;
  // every edge-path shape: the bf16-pipeline kernel (one 64 x 64 block of dW per wave)
  if ((g.a_bf16 || g.b_bf16 || !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA)) && R % 64 == 0 && (g.lda & 1) == 0 && (g.ldb & 1) == 0 && P % 64 == 0 && Q % 64 == 0 &&
      (int64_t)8 * g.lda + P < 0x7fffffff && (int64_t)8 * g.ldb + Q < 0x7fffffff && workspace != nullptr) {
    const int nblk = (P / 64) * (Q / 64);
    if (nblk == 1 || nblk == 2 || nblk == 4 || nblk == 8) {
      const int nchunks3 = R / 64;
      const int grid3 = std::min(nchunks3, ts_num_cus());
      const int slots = grid3 * (8 / nblk);
      if (ws_bytes >= (int64_t)slots * ((int64_t)P * Q + P) * 4) {
        g.partial = (float*)workspace;
        g.partial_colsum = g.colsum ? g.partial + (int64_t)slots * P * Q : nullptr;
        const bool gate3 = g.a_gate != nullptr;
        void (*k3)(GemmArgs, int) = nullptr;
#define TW3_PICK(p, q) if (P == p && Q == q) k3 = gate3 ? tswgrad3_kernel<p, q, true> : tswgrad3_kernel<p, q, false>;
        if (g.a_bf16 || g.b_bf16) {                // bf16 storage: the edge block's two large-activation gradients
          if (!gate3 && g.a_gate_scale == 1.f && g.a_bf16 && !g.b_bf16 && P == 512 && Q == 64) k3 = tswgrad3_kernel<512, 64, false, true, false>;
          if (!gate3 && !g.a_bf16 && g.b_bf16 && P == 64 && Q == 256) k3 = tswgrad3_kernel<64, 256, false, false, true>;
        } else {
          TW3_PICK(512, 64) TW3_PICK(256, 64) TW3_PICK(64, 256) TW3_PICK(128, 64) TW3_PICK(64, 128) TW3_PICK(128, 128) TW3_PICK(64, 64)
          TW3_PICK(64, 512) TW3_PICK(256, 128) TW3_PICK(128, 256)
        }
#undef TW3_PICK
        if (k3 != nullptr) {
          hipLaunchKernelGGL(k3, dim3(grid3), dim3(512), 0, stream, g, nchunks3);
          const int64_t total3 = (int64_t)P * Q + P;
          hipLaunchKernelGGL(tswgrad_reduce_kernel, dim3((unsigned)((total3 + 63) / 64)), dim3(256), 0, stream, g, slots);
          return mbvae_launch_status();
        }
      }
    }
  }
  if (g.a_bf16 || g.b_bf16) return MBVAE_EUNSUPPORTED;   // bf16 storage exists for the kernel above only
  if (P * Q > 16384 && !(P == 512 && Q == 64 && !g.a_gate)) return MBVAE_EUNSUPPORTED;
  // the widest edge-path gradient (fused e_mul|e_add projection, 512 x 64) runs on the LDS-free kernel: 8 % faster there
  if (!mbvae_opt(MBVAE_OPT_NO_TSWGRAD2) && R % 64 == 0 && (g.lda & 1) == 0 && (g.ldb & 1) == 0 && P == 512 && Q == 64 && !g.a_gate &&
      (int64_t)2 * g.lda + P < 0x7fffffff) {
    const int nchunks2 = R / 64;
    const int grid2 = std::min(nchunks2, ts_num_cus());
    const int64_t need2 = (int64_t)grid2 * ((int64_t)P * Q + P) * 4;
    if (workspace != nullptr && ws_bytes >= need2) {
      g.partial = (float*)workspace;
      g.partial_colsum = g.colsum ? g.partial + (int64_t)grid2 * P * Q : nullptr;
      hipLaunchKernelGGL((tswgrad2_kernel<512, 64, 8, 1, false>), dim3(grid2), dim3(512), 0, stream, g, nchunks2);
      const int64_t total2 = (int64_t)P * Q + P;
      hipLaunchKernelGGL(tswgrad_reduce_kernel, dim3((unsigned)((total2 + 63) / 64)), dim3(256), 0, stream, g, grid2);
      return mbvae_launch_status();
    }
  }
  const int nchunks = R / 32;
  // two workgroups per CU where their chunk buffers fit (P + Q <= 192): 3-5 % faster than one (measured)
  const int per_cu = sizeof(float) * 2 * 32 * (size_t)(P + 4 + Q + 4) * 2 <= 150 * 1024 ? 2 : 1;
  const int grid = std::min(nchunks, per_cu * ts_num_cus());
  const int64_t need = (int64_t)grid * ((int64_t)P * Q + P) * 4;
  if (workspace == nullptr || ws_bytes < need) return MBVAE_EUNSUPPORTED;
  g.partial = (float*)workspace;
  g.partial_colsum = g.colsum ? g.partial + (int64_t)grid * P * Q : nullptr;
  const size_t smem = sizeof(float) * 2 * 32 * (size_t)(P + 4 + Q + 4);
  const bool gate = g.a_gate != nullptr;
#define TW_LAUNCH(p, q, wm, wn)                                                                                      \
  do {                                                                                                                \
    auto kg = tswgrad_kernel<p, q, wm, wn, true>;                                                                     \
    auto kn = tswgrad_kernel<p, q, wm, wn, false>;                                                                    \
    auto k = gate ? kg : kn;                                                                                          \
    if (!(gate ? MBVAE_ALLOW_BIG_LDS(kg) : MBVAE_ALLOW_BIG_LDS(kn)) || smem > MBVAE_LDS_LIMIT) return MBVAE_ELAUNCH;  \
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, g, nchunks);                                           \
  } while (0)
  if (P == 512 && Q == 64) {                       // fused e_mul|e_add projection: 64 accumulator registers per lane
    auto k = tswgrad_kernel<512, 64, 8, 1, false>;
    if (!MBVAE_ALLOW_BIG_LDS((tswgrad_kernel<512, 64, 8, 1, false>))) return MBVAE_ELAUNCH;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, g, nchunks);
  }
  else if (P == 256 && Q == 64) TW_LAUNCH(256, 64, 4, 2);
  else if (P == 64 && Q == 256) TW_LAUNCH(64, 256, 2, 4);
  else if (P == 128 && Q == 64) TW_LAUNCH(128, 64, 4, 2);
  else if (P == 64 && Q == 128) TW_LAUNCH(64, 128, 2, 4);
  else if (P == 128 && Q == 128) TW_LAUNCH(128, 128, 4, 2);
  else return MBVAE_EUNSUPPORTED;
#undef TW_LAUNCH
  const int64_t total = (int64_t)P * Q + P;
  hipLaunchKernelGGL(tswgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, g, grid);
  return mbvae_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// Small-output products with a long reduction (the global-feature path and the encoder MLPs: a few hundred rows,
// K ~ 1000): too few 64x64 tiles to fill the chip, so the generic kernel would split the reduction over workgroups
// and sum the partials in a second launch.  Here a workgroup owns ONE 32x32 output tile and its four waves split the
// reduction among themselves (64-wide groups, round robin); the four partial tiles meet in LDS and are added in wave
// order, so the result is reproducible, no workspace is needed and nothing else is launched.  Operands go straight
// from global memory into MFMA operand registers, whatever their layout:
//   reduction index contiguous (x [M,K], W [N,K]):   lane (row r, half h) reads 32 consecutive indices 32h .. 32h+31
//   output index contiguous (W [K,N], dy / x [R,.]): lane (col r, half h) reads one word of each of the rows 32h .. 32h+31
// and step s of a group multiplies indices s (lower half-wave) and 32+s (upper half-wave) for both operands.
// P8: 8-index steps per group of a wave (8: groups of 64 reduction indices, 4: of 32).  The operand double buffer is
// 2 x 2 x 4 P8 registers: with P8 = 8 a wave needs ~220 registers (two per SIMD), with P8 = 4 ~125 (four per SIMD),
// which measured 10-20 % faster at the global-feature shapes (256 x 1024 x 1024: 12.5 against 15.1 us) and is what the
// dispatch uses.
template <bool A_KC, bool B_KC, bool GATE, int P8 = 8>
__global__ __launch_bounds__(256) void smallm_gemm_kernel(GemmArgs g) {
  MBVAE_RAISE_PRIO();
  constexpr int GK = 8 * P8, NS = 4 * P8;               // reduction indices and register slots per group
  __shared__ float red[4][16][64];
  __shared__ float csum[4][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = (g.N + 31) / 32;
  const int m0 = ((int)blockIdx.x / tiles_n) * 32, n0 = ((int)blockIdx.x % tiles_n) * 32;
  const int K = g.K;
  const int groups = (K + GK - 1) / GK;
  const int mine = (groups - wave + 3) / 4;               // groups wave, wave + 4, ...
  const int am = min(m0 + r, g.M - 1), bn = min(n0 + r, g.N - 1);     // clamped: surplus rows / columns are never stored
  const bool want_colsum = g.colsum != nullptr && n0 == 0;

  float a[2][NS], b[2][NS], gt[2][NS];
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  float cs = 0.f;

  // Register slot 4 p + e of a group holds reduction index 8 p + 4 h + e (p < 8, e < 4): step (p, e) multiplies index
  // 8p + e in the lower half-wave with 8p + 4 + e in the upper one, for both operands.  With this pairing the two
  // half-waves of a reduction-contiguous operand read ADJACENT 16-byte pieces of a row (one cache line per row and
  // instruction instead of two), and an output-contiguous operand is addressed as a wave-uniform row pointer plus a
  // per-lane 32-bit offset.  K % 8 == 0 (checked by the host): a step is valid or invalid as a whole.
  const unsigned a_lane = A_KC ? (unsigned)(4 * h) : (unsigned)(4 * h) * (unsigned)g.lda + (unsigned)am;
  const unsigned b_lane = B_KC ? (unsigned)(4 * h) : (unsigned)(4 * h) * (unsigned)g.ldb + (unsigned)bn;
  const float* a_row = A_KC ? g.A + (int64_t)am * g.lda : g.A;
  const float* g_row = A_KC ? g.a_gate + (int64_t)am * g.lda : g.a_gate;
  const float* b_row = B_KC ? g.B + (int64_t)bn * g.ldb : g.B;
  // requests group q's operand words (no waits); steps past K re-read the last valid one and are zeroed in prep()
  auto fetch = [&](int q, int S) {
    const int k0 = (wave + 4 * min(q, mine - 1)) * GK;      // wave-uniform
#pragma unroll
    for (int p = 0; p < P8; ++p) {
      const int kp = min(k0 + 8 * p, K - 8);                // wave-uniform
      if (A_KC) {
        const float4 t = *reinterpret_cast<const float4*>(a_row + kp + a_lane);
        a[S][4 * p] = t.x; a[S][4 * p + 1] = t.y; a[S][4 * p + 2] = t.z; a[S][4 * p + 3] = t.w;
        if (GATE) {
          const float4 u = *reinterpret_cast<const float4*>(g_row + kp + a_lane);
          gt[S][4 * p] = u.x; gt[S][4 * p + 1] = u.y; gt[S][4 * p + 2] = u.z; gt[S][4 * p + 3] = u.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* rowp = a_row + (int64_t)(kp + e) * g.lda;      // wave-uniform row base
          a[S][4 * p + e] = rowp[a_lane];
          if (GATE) gt[S][4 * p + e] = (g_row + (int64_t)(kp + e) * g.lda)[a_lane];
        }
      }
      if (B_KC) {
        const float4 t = *reinterpret_cast<const float4*>(b_row + kp + b_lane);
        b[S][4 * p] = t.x; b[S][4 * p + 1] = t.y; b[S][4 * p + 2] = t.z; b[S][4 * p + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[S][4 * p + e] = (b_row + (int64_t)(kp + e) * g.ldb)[b_lane];
      }
    }
    asm volatile("" : : : "memory");
  };
  // operand prologue in place (gate, scale, zero past the end of the reduction) and the bias gradient's column sums
  auto prep = [&](int q, int S) {
    const int k0 = (wave + 4 * q) * GK;
    const bool live = q < mine;
    const bool full = live && k0 + GK <= K;
    if (full && !GATE && g.a_gate_scale == 1.f) {
      if (want_colsum) {
#pragma unroll
        for (int s = 0; s < NS; ++s) cs += a[S][s];
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bool in = live && k0 + 8 * (s >> 2) < K;        // wave-uniform
      float v = a[S][s];
      if (GATE && !(gt[S][s] > 0.f)) v = 0.f;
      v = in ? v * g.a_gate_scale : 0.f;                  // selects: a re-read word must not count
      a[S][s] = v;
      if (!in) b[S][s] = 0.f;
      cs += v;
    }
  };
  auto mma = [&](int S) {
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][s], b[S][s], acc, 0, 0, 0);
  };

  if (mine > 0) {
    fetch(0, 0);
    for (int q = 0; q < mine; q += 2) {
      fetch(q + 1, 1);
      prep(q, 0);
      mma(0);
      fetch(q + 2, 0);
      prep(q + 1, 1);                                     // (a group past the end is zeroed)
      mma(1);
    }
  }
#pragma unroll
  for (int v = 0; v < 16; ++v) red[wave][v][lane] = acc[v];
  cs += __shfl_xor(cs, 32, 64);
  if (h == 0) csum[wave][r] = cs;
  __syncthreads();
  // the four partial tiles are added in wave order; C/D layout: col = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = wave * 4 + u;
    const float sum = ((red[0][v][lane] + red[1][v][lane]) + red[2][v][lane]) + red[3][v][lane];
    const int row = m0 + (v & 3) + 8 * (v >> 2) + 4 * h, col = n0 + r;
    if (row < g.M && col < g.N) {
      float o = sum;
      if (g.bias) o += g.bias[col];
      if (g.act == 1) o = fmaxf(o, 0.f);
      if (g.drop_p > 0.f) o *= mbvae_drop_scale(g.drop_p, g.seed, (uint64_t)row * (uint64_t)g.ldc + col);
      g.C[(int64_t)row * g.ldc + col] = o;
    }
  }
  if (want_colsum && tid < 32 && m0 + tid < g.M)
    g.colsum[m0 + tid] = ((csum[0][tid] + csum[1][tid]) + csum[2][tid]) + csum[3][tid];
}

// MBVAE_EUNSUPPORTED when the shape is not one of these (the caller then uses the generic kernel).
template <bool A_KC, bool B_KC>
static int smallm_try(const GemmArgs& g, hipStream_t stream) {
  if (mbvae_opt(MBVAE_OPT_NO_SMALLM)) return MBVAE_EUNSUPPORTED;
  if (g.a_mask_mode != MBVAE_MASK_NONE || g.c_mask_mode != MBVAE_MASK_NONE || g.accumulate || g.addend) return MBVAE_EUNSUPPORTED;
  const int64_t tiles64 = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64);
  const int64_t tiles32 = (int64_t)((g.M + 31) / 32) * ((g.N + 31) / 32);
  // The regime where the generic kernel would split the reduction (few big tiles, a reduction worth splitting), and of
  // that the part where this kernel measured faster (tools/smallm_probe.py): at most one workgroup per compute unit,
  // and either enough of them to fill the chip or a short reduction.  Operand reuse of a 32x32 tile is low (8 flops per
  // byte from L2), so with two or more rounds of workgroups the 64x64 tiles of the generic kernel win.
  if (tiles64 >= 256 || g.K < 128 || (g.K & 7)) return MBVAE_EUNSUPPORTED;
  if (tiles32 > mbvae_num_cus() || g.K > 2048 || !(tiles32 >= mbvae_num_cus() / 2 || g.K <= 512)) return MBVAE_EUNSUPPORTED;
  if ((int64_t)4 * g.lda + g.M >= 0x7fffffff || (int64_t)4 * g.ldb + g.N >= 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (A_KC && ((((uintptr_t)g.A | (uintptr_t)g.a_gate) & 15) || (g.lda & 3) || (g.K & 3))) return MBVAE_EUNSUPPORTED;
  if (B_KC && (((uintptr_t)g.B & 15) || (g.ldb & 3) || (g.K & 3))) return MBVAE_EUNSUPPORTED;
  const dim3 grid((unsigned)tiles32), block(256);
  if (g.a_gate) hipLaunchKernelGGL((smallm_gemm_kernel<A_KC, B_KC, true, 4>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((smallm_gemm_kernel<A_KC, B_KC, false, 4>), grid, block, 0, stream, g);
  return mbvae_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// Thin products: Linear layers over the pair rows with a handful of input or output features (the 5 bond classes at
// either end of the edge MLPs: 5 -> 128 and 128 -> 5).  They are pure streams - 0.5 GB of the wide operand per call
// against a few multiply-adds per element - that the tiled matrix-core kernels run at a tenth of the memory rate
// (their tiles are 32 wide in every dimension).  Here a group of W/4 lanes owns a row of the wide operand (16 bytes per
// lane), the thin operand's few values per row are broadcast loads, everything is plain fp32 FMA in k order, and two
// rows per lane are in flight.
template <int LPR>                                        // sum over aligned groups of LPR = 16, 32 or 64 lanes (DPP, see half_wave_sum)
__device__ __forceinline__ float thin_group_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  if (LPR >= 32) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));      // lane ^ 16
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// out[m, n] = act(bias[n] + sum_{k < KT} A[m, k] * B(k, n)),  B(k, n) = B_KC ? B[n * ldb + k] : B[k * ldb + n];  N = 4 * LPR
template <int KT, bool B_KC>
__global__ __launch_bounds__(256) void thin_in_kernel(GemmArgs g) {
  const int lpr = g.N >> 2, sub = threadIdx.x % lpr, grp = threadIdx.x / lpr, groups = 256 / lpr;
  float4 w[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (B_KC) w[k] = make_float4(g.B[(int64_t)(4 * sub) * g.ldb + k], g.B[(int64_t)(4 * sub + 1) * g.ldb + k],
                                 g.B[(int64_t)(4 * sub + 2) * g.ldb + k], g.B[(int64_t)(4 * sub + 3) * g.ldb + k]);
    else w[k] = *reinterpret_cast<const float4*>(g.B + (int64_t)k * g.ldb + 4 * sub);
  }
  const float4 b4 = g.bias ? *reinterpret_cast<const float4*>(g.bias + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool relu = g.act == 1;
  const int64_t stride = (int64_t)gridDim.x * groups;
  for (int64_t row0 = (int64_t)blockIdx.x * groups + grp; row0 < g.M; row0 += 2 * stride) {
    float a[2][KT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = min(row0 + u * stride, (int64_t)g.M - 1);
#pragma unroll
      for (int k = 0; k < KT; ++k) a[u][k] = g.A[row * g.lda + k];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = row0 + u * stride;
      if (row < g.M) {
        float4 o = b4;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          o.x = fmaf(a[u][k], w[k].x, o.x); o.y = fmaf(a[u][k], w[k].y, o.y);
          o.z = fmaf(a[u][k], w[k].z, o.z); o.w = fmaf(a[u][k], w[k].w, o.w);
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(g.C + row * g.ldc + 4 * sub) = o;
      }
    }
  }
}

// out[m, j] = act(bias[j] + sum_k A[m, k] * B[j * ldb + k]),  j < N <= 8,  K = 4 * LPR
template <int LPR>
__global__ __launch_bounds__(256) void thin_out_kernel(GemmArgs g) {
  constexpr int GROUPS = 256 / LPR;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  float4 w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    w[j] = j < g.N ? *reinterpret_cast<const float4*>(g.B + (int64_t)j * g.ldb + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float bj = (g.bias && sub < g.N) ? g.bias[sub] : 0.f;
  const bool relu = g.act == 1;
  const int64_t stride = (int64_t)gridDim.x * GROUPS;
  for (int64_t row0 = (int64_t)blockIdx.x * GROUPS + grp; row0 < g.M; row0 += 2 * stride) {
    float4 x[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
      x[u] = *reinterpret_cast<const float4*>(g.A + min(row0 + u * stride, (int64_t)g.M - 1) * g.lda + 4 * sub);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < g.N) {                                     // (uniform)
          const float sj = thin_group_sum<LPR>((x[u].x * w[j].x + x[u].y * w[j].y) + (x[u].z * w[j].z + x[u].w * w[j].w));
          o = sub == j ? sj : o;
        }
      const int64_t row = row0 + u * stride;
      if (row < g.M && sub < g.N) {
        o += bj;
        g.C[row * g.ldc + sub] = relu ? fmaxf(o, 0.f) : o;
      }
    }
  }
}

// Weight gradient of a thin layer: part[c][t] = sum_r wide[r, c] * thin[r, t] over this workgroup's rows, W = 4 * LPR wide
// columns, T <= 8 thin ones; the wide operand optionally gated (ReLU gate of the layer's output).  WIDE_IS_A: the wide
// operand is dY (A), the result is dW [W][T] and the column sums are the wide ones; otherwise the thin operand is dY,
// the result dW [T][W] and the column sums the thin ones.  Partials per workgroup, summed by tswgrad_reduce_kernel.
template <int LPR, bool WIDE_IS_A, int U>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(GemmArgs g) {
  constexpr int GROUPS = 256 / LPR, W = 4 * LPR;
  __shared__ float4 red[256];
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const float* wide = WIDE_IS_A ? g.A : g.B;
  const float* thin = WIDE_IS_A ? g.B : g.A;
  const int64_t ldw = WIDE_IS_A ? g.lda : g.ldb, ldt = WIDE_IS_A ? g.ldb : g.lda;
  const int T = WIDE_IS_A ? g.N : g.M;
  const float* gate = WIDE_IS_A ? g.a_gate : nullptr;
  float4 acc[8], csw = make_float4(0.f, 0.f, 0.f, 0.f);
  float cst[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j] = make_float4(0.f, 0.f, 0.f, 0.f); cst[j] = 0.f; }
  const int64_t R = g.K, stride = (int64_t)gridDim.x * GROUPS;
  for (int64_t row0 = (int64_t)blockIdx.x * GROUPS + grp; row0 < R; row0 += U * stride) {
    float4 wv[U], gv[U];
    float t[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = min(row0 + u * stride, R - 1);
      wv[u] = *reinterpret_cast<const float4*>(wide + row * ldw + 4 * sub);
      if (gate) gv[u] = *reinterpret_cast<const float4*>(gate + row * ldw + 4 * sub);
#pragma unroll
      for (int j = 0; j < 8; ++j) t[u][j] = j < T ? thin[row * ldt + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 v = wv[u];
      if (gate) {
        v.x = gv[u].x > 0.f ? v.x * g.a_gate_scale : 0.f; v.y = gv[u].y > 0.f ? v.y * g.a_gate_scale : 0.f;
        v.z = gv[u].z > 0.f ? v.z * g.a_gate_scale : 0.f; v.w = gv[u].w > 0.f ? v.w * g.a_gate_scale : 0.f;
      }
      if (row0 + u * stride >= R) v = make_float4(0.f, 0.f, 0.f, 0.f);     // a row past the end contributes nothing
      const float live = row0 + u * stride < R ? 1.f : 0.f;
      csw.x += v.x; csw.y += v.y; csw.z += v.z; csw.w += v.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j].x = fmaf(v.x, t[u][j], acc[j].x); acc[j].y = fmaf(v.y, t[u][j], acc[j].y);
        acc[j].z = fmaf(v.z, t[u][j], acc[j].z); acc[j].w = fmaf(v.w, t[u][j], acc[j].w);
        cst[j] += live * t[u][j];
      }
    }
  }
  // the GROUPS row groups of the workgroup meet in LDS, one thin column at a time, and are added in group order
  const int64_t PQ = (int64_t)g.M * g.N;
  float* part = g.partial + (int64_t)blockIdx.x * PQ;
  float* pcs = g.partial_colsum ? g.partial_colsum + (int64_t)blockIdx.x * g.M : nullptr;
  for (int j = 0; j <= T; ++j) {
    float4 mine = csw;                                      // (the pass after the last thin column: the wide column sums)
    if (j < T) {
#pragma unroll
      for (int q = 0; q < 8; ++q) mine = j == q ? acc[q] : mine;
    }
    __syncthreads();
    red[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x < W) {
      const int c = threadIdx.x;
      float sgm = 0.f;
      for (int gi = 0; gi < GROUPS; ++gi) sgm += reinterpret_cast<const float*>(&red[gi * LPR + (c >> 2)])[c & 3];
      if (j < T) part[WIDE_IS_A ? (int64_t)c * T + j : (int64_t)j * W + c] = sgm;
      else if (WIDE_IS_A && pcs) pcs[c] = sgm;
    }
  }
  if (!WIDE_IS_A && pcs) {                                  // thin column sums: every lane of a group holds the same ones
    __syncthreads();
    float* rf = reinterpret_cast<float*>(red);
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rf[grp * 8 + j] = cst[j];
    }
    __syncthreads();
    if (threadIdx.x < T) {
      float sgm = 0.f;
      for (int gi = 0; gi < GROUPS; ++gi) sgm += rf[gi * 8 + threadIdx.x];
      pcs[threadIdx.x] = sgm;
    }
  }
}

// The thin kernels take plain layers only (no mask, gate only on a wide dY, no dropout, no accumulation); everything
// else, and the NO_THIN option, stays on the tiled kernels.
template <bool A_KC, bool B_KC>
static int thin_try(GemmArgs g, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  if (mbvae_opt(MBVAE_OPT_NO_THIN)) return MBVAE_EUNSUPPORTED;
  if (g.a_bf16 || g.b_bf16 || g.c_bf16 || g.a_mask_mode != MBVAE_MASK_NONE || g.c_mask_mode != MBVAE_MASK_NONE || g.drop_p > 0.f ||
      g.accumulate || g.addend || g.c_gate)
    return MBVAE_EUNSUPPORTED;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const int blocks = 8 * ts_num_cus();
  if (A_KC) {                                              // forward / input gradient: rows = g.M
    if (g.M < 32768 || g.a_gate) return MBVAE_EUNSUPPORTED;
    if (g.K <= 8 && g.N % 4 == 0 && pow2(g.N / 4) && g.N <= 1024 && (g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 &&
        (B_KC || ((g.ldb & 3) == 0 && ((uintptr_t)g.B & 15) == 0)) && (!g.bias || ((uintptr_t)g.bias & 15) == 0)) {
#define THIN_IN(kt) case kt: hipLaunchKernelGGL((thin_in_kernel<kt, B_KC>), dim3(blocks), dim3(256), 0, stream, g); break;
      switch (g.K) { THIN_IN(1) THIN_IN(2) THIN_IN(3) THIN_IN(4) THIN_IN(5) THIN_IN(6) THIN_IN(7) THIN_IN(8) }
#undef THIN_IN
      return mbvae_launch_status();
    }
    if (B_KC && g.N <= 8 && g.K % 4 == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0) {
      if (g.K == 64) hipLaunchKernelGGL(thin_out_kernel<16>, dim3(blocks), dim3(256), 0, stream, g);
      else if (g.K == 128) hipLaunchKernelGGL(thin_out_kernel<32>, dim3(blocks), dim3(256), 0, stream, g);
      else if (g.K == 256) hipLaunchKernelGGL(thin_out_kernel<64>, dim3(blocks), dim3(256), 0, stream, g);
      else return MBVAE_EUNSUPPORTED;
      return mbvae_launch_status();
    }
    return MBVAE_EUNSUPPORTED;
  }
  if (B_KC) return MBVAE_EUNSUPPORTED;
  // weight gradient: A = dY [R, P], B = X [R, Q], C = dW [P, Q] (contiguous), R = g.K rows
  const int P = g.M, Q = g.N;
  if (g.K < 32768 || g.ldc != Q || workspace == nullptr) return MBVAE_EUNSUPPORTED;
  const bool wide_a = Q <= 8 && (P == 64 || P == 128 || P == 256);
  const bool wide_b = P <= 8 && (Q == 64 || Q == 128 || Q == 256) && !g.a_gate;
  if (!wide_a && !wide_b) return MBVAE_EUNSUPPORTED;
  const float* wide = wide_a ? g.A : g.B;
  const int64_t ldw = wide_a ? g.lda : g.ldb;
  if ((ldw & 3) || ((uintptr_t)wide & 15) || (wide_a && g.a_gate && ((uintptr_t)g.a_gate & 15))) return MBVAE_EUNSUPPORTED;
  const int wblocks = 2 * ts_num_cus();       // (1024-thread workgroups, one per CU, measured 15 % slower)
  const int64_t need = (int64_t)wblocks * ((int64_t)P * Q + P) * 4;
  if (ws_bytes < need) return MBVAE_EUNSUPPORTED;
  g.partial = (float*)workspace;
  g.partial_colsum = g.colsum ? g.partial + (int64_t)wblocks * P * Q : nullptr;
  const int Wd = wide_a ? P : Q;
  // rows per trip: four where the wide operand is the only wide stream, two next to a gate operand (same bytes in flight)
#define THIN_WG(lpr)                                                                                                  \
  do {                                                                                                                 \
    if (wide_a && g.a_gate) hipLaunchKernelGGL((thin_wgrad_kernel<lpr, true, 2>), dim3(wblocks), dim3(256), 0, stream, g);  \
    else if (wide_a) hipLaunchKernelGGL((thin_wgrad_kernel<lpr, true, 4>), dim3(wblocks), dim3(256), 0, stream, g);   \
    else hipLaunchKernelGGL((thin_wgrad_kernel<lpr, false, 4>), dim3(wblocks), dim3(256), 0, stream, g);              \
  } while (0)
  if (Wd == 64) THIN_WG(16); else if (Wd == 128) THIN_WG(32); else THIN_WG(64);
#undef THIN_WG
  const int64_t total = (int64_t)P * Q + P;
  hipLaunchKernelGGL(tswgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, g, wblocks);
  return mbvae_launch_status();
}

#ifndef GEMM_MID_SPLIT
#define GEMM_MID_SPLIT 1     // tuning macro: 0 keeps products of 256..767 tiles unsplit
#endif
template <bool A_KC, bool B_KC>
static int gemm_dispatch(GemmArgs g, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return MBVAE_EINVAL;
  if (!g.A || !g.B || !g.C) return MBVAE_EINVAL;
  const bool bf16 = g.a_bf16 || g.b_bf16 || g.c_bf16;
  {
    const int st = thin_try<A_KC, B_KC>(g, workspace, ws_bytes, stream);
    if (st != MBVAE_EUNSUPPORTED) return st;
  }
  if (A_KC) {
    const int st = tsgemm_try<B_KC>(g, stream);
    if (st != MBVAE_EUNSUPPORTED) return st;
    // An addend outside the tall-skinny shapes: the tiled kernel adds it in its epilogue (or in the split reduction's).
    // Outputs of few rows stay with the small-output kernel, which has no addend: the caller adds it in a separate pass.
    if (g.addend && g.M < 1024) return MBVAE_EUNSUPPORTED;
  } else if (!B_KC) {
    const int st = tswgrad_try(g, workspace, ws_bytes, stream);
    if (st != MBVAE_EUNSUPPORTED) return st;
  }
  if (bf16) return MBVAE_EUNSUPPORTED;             // only the edge-path kernels read or write bf16 storage
  {
    const int st = smallm_try<A_KC, B_KC>(g, stream);
    if (st != MBVAE_EUNSUPPORTED) return st;
  }
  int BM, BN;
  if (g.M <= 64 || g.N <= 32 || (int64_t)g.M * g.N <= 512 * 1024) { BM = 64; BN = 64; }
  else if (g.N <= 256) { BM = 128; BN = 64; }     // (256-wide node layers over packed rows: 2.5 % faster sampling step than 128x128)
  else { BM = 128; BN = 128; }
  // fewer than two big tiles per CU leave the chip under-filled and every block latency bound: use small tiles
  if (BM == 128 && (int64_t)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) < 2 * ts_num_cus()) { BM = 64; BN = 64; }
  if (const int f = mbvae_opt(MBVAE_OPT_GEMM_TILE); f >= 0) {           // tuning aid: 0 = 64x64, 1 = 128x64, 2 = 128x128, 3 = 32x128, 4 = 64x128
    BM = f == 0 || f == 4 ? 64 : (f == 3 ? 32 : 128); BN = f >= 2 ? 128 : 64;
  }
  const int64_t tiles = (int64_t)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (tiles > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  // split the reduction until ~2 blocks per CU are in flight
  int splits = 1;
  // (second case: one to three workgroups per CU with a long reduction each - the global-feature products of a sampling
  //  chunk, ~1250 rows x 1024 - are bound by the latency of their 32+ dependent k-steps: more, shorter workgroups)
  const bool mid = GEMM_MID_SPLIT && tiles >= 256 && tiles < 768 && g.K >= 512;
  if ((tiles < 256 && g.K >= 256) || mid) {
    splits = mid ? (int)std::min<int64_t>((1024 + tiles - 1) / tiles, g.K / 256) : (int)std::min<int64_t>(512 / tiles, g.K / 128);
    const int64_t per = ((int64_t)g.M * g.N + g.M) * 4;
    if (workspace == nullptr || ws_bytes < per * 2) splits = 1;
    else splits = (int)std::min<int64_t>(splits, ws_bytes / per);
    if (splits < 1) splits = 1;
  }
  g.k_per_split = ((g.K + splits - 1) / splits + 31) / 32 * 32;
  splits = (g.K + g.k_per_split - 1) / g.k_per_split;
  if (splits > 1) {
    g.partial = (float*)workspace;
    g.partial_colsum = g.partial + (int64_t)splits * g.M * g.N;
  } else {
    g.partial = nullptr; g.partial_colsum = nullptr;
  }
  dim3 grid((unsigned)tiles, 1, (unsigned)splits), block(256);
  const bool aligned = (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.a_gate) & 15) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0;
  // branch-free operand loads: whole N and K tiles; M may be ragged when A is [M,K] (rows are clamped, see fetch)
  const bool fast = aligned && (A_KC || g.M % BM == 0) && g.N % BN == 0 && g.K % 32 == 0;
  const bool gate = g.a_gate != nullptr;
  // whole aligned tiles run on the bf16 matrix pipeline (fp32-exact three-way operand splits, planes staged in LDS)
  const bool split = fast && !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA) && !mbvae_opt(MBVAE_OPT_NO_GEMM_SPLIT);
#define GEMM_LAUNCH_F32(bm, bn, wm, wn)                                                                                 \
  do {                                                                                                                   \
    if (fast && gate) hipLaunchKernelGGL((gemm_f32_kernel<bm, bn, wm, wn, A_KC, B_KC, true, true>), grid, block, 0, stream, g);        \
    else if (fast) hipLaunchKernelGGL((gemm_f32_kernel<bm, bn, wm, wn, A_KC, B_KC, true, false>), grid, block, 0, stream, g);          \
    else if (gate) hipLaunchKernelGGL((gemm_f32_kernel<bm, bn, wm, wn, A_KC, B_KC, false, true>), grid, block, 0, stream, g);          \
    else hipLaunchKernelGGL((gemm_f32_kernel<bm, bn, wm, wn, A_KC, B_KC, false, false>), grid, block, 0, stream, g);                   \
  } while (0)
#define GEMM_LAUNCH(bm, bn, wm, wn)                                                                                     \
  do {                                                                                                                   \
    if (split && gate) hipLaunchKernelGGL((gemm_f32_kernel<bm, bn, wm, wn, A_KC, B_KC, true, true, true>), grid, block, 0, stream, g); \
    else if (split) hipLaunchKernelGGL((gemm_f32_kernel<bm, bn, wm, wn, A_KC, B_KC, true, false, true>), grid, block, 0, stream, g);   \
    else GEMM_LAUNCH_F32(bm, bn, wm, wn);                                                                                \
  } while (0)
  if (BM == 32) GEMM_LAUNCH_F32(32, 128, 1, 4);
  else if (BM == 64 && BN == 128) GEMM_LAUNCH(64, 128, 2, 2);
  else if (BM == 64) GEMM_LAUNCH(64, 64, 2, 2);
  else if (BN == 64) GEMM_LAUNCH(128, 64, 4, 1);
  else GEMM_LAUNCH(128, 128, 2, 2);
#undef GEMM_LAUNCH_F32
#undef GEMM_LAUNCH
  if (splits > 1) {
    const int64_t total = (int64_t)g.M * g.N + g.M;
    int blocks = (int)std::min<int64_t>(2048, (total + 255) / 256);
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g, splits);
  }
  return mbvae_launch_status();
}

// x = gate > 0 ? x * scale : 0, element-wise (the fallback of mbvae_linear_bwd_input_outgate)
__global__ __launch_bounds__(256) void out_gate_kernel(float* __restrict__ x, const float* __restrict__ gate, float scale, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    x[e] = gate[e] > 0.f ? x[e] * scale : 0.f;
}

extern "C" {

// Process-wide tuning switch `name` (without the MBVAE_ prefix, e.g. "NO_EDGE_FUSED"); see MbvaeOption.
int mbvae_set_option(const char* name, int value) {
  if (!name) return MBVAE_EINVAL;
  for (int i = 0; i < MBVAE_OPT_COUNT; ++i)
    if (strcmp(name, mbvae_option_names()[i]) == 0) {
      mbvae_options().v[i].store(value, std::memory_order_relaxed);
      return MBVAE_OK;
    }
  return MBVAE_EINVAL;
}

// Current value of a tuning switch, or MBVAE_EINVAL (-1) for an unknown name (GEMM_TILE's "automatic" is also -1).
int mbvae_get_option(const char* name) {
  if (!name) return MBVAE_EINVAL;
  for (int i = 0; i < MBVAE_OPT_COUNT; ++i)
    if (strcmp(name, mbvae_option_names()[i]) == 0) return mbvae_opt(i);
  return MBVAE_EINVAL;
}

// y[M,N] = mask_rows( dropout( act( x[M,K] @ w[N,K]^T + bias ) ) )  (+= y if accumulate)
int mbvae_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                     int64_t M, int64_t K, int64_t N, int act, const uint8_t* mask, int mask_mode, int n_nodes,
                     float drop_p, uint64_t seed, int accumulate, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  GemmArgs g = {};
  g.A = x; g.B = w; g.C = y; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = ldx; g.ldb = K; g.ldc = ldy;
  g.a_gate_scale = 1.f; g.mask = mask; g.c_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.bias = bias; g.act = act; g.drop_p = drop_p; g.seed = seed; g.accumulate = accumulate;
  return gemm_dispatch<true, true>(g, workspace, ws_bytes, stream);
}

// dx[M,K] (+)= pro(dy)[M,N] @ w[N,K];  pro(dy) = (gate > 0 ? dy : 0) * gate_scale * keep(row)
int mbvae_linear_bwd_input(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx,
                           int64_t M, int64_t K, int64_t N, const float* gate, float gate_scale,
                           const uint8_t* mask, int mask_mode, int n_nodes, int accumulate,
                           void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if (gate && lddy != N) return MBVAE_EINVAL;     // gate shares dy's layout
  GemmArgs g = {};
  g.A = dy; g.B = w; g.C = dx; g.M = (int)M; g.N = (int)K; g.K = (int)N;
  g.lda = lddy; g.ldb = K; g.ldc = lddx;
  g.a_gate = gate; g.a_gate_scale = gate_scale; g.mask = mask; g.a_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.accumulate = accumulate;
  return gemm_dispatch<true, false>(g, workspace, ws_bytes, stream);
}

// y[M,64] = LayerNorm(z) * gamma + beta,  z[M,64] = resid + dropout_p(mask_rows(x[M,K] @ w[64,K]^T + bias)): a Linear whose
// 64-wide output feeds a residual LayerNorm (e_out -> normE1, linE2 -> normE2) with the normalisation in the product's
// epilogue, so the Linear's output never travels through memory; z is kept for mbvae_ln_sum_bwd.  The dropout mask is
// the one mbvae_ln_residual_fwd would draw for (seed, element index).  MBVAE_EUNSUPPORTED for shapes the wave-autonomous
// tall-skinny kernel does not take (run the Linear and mbvae_ln_residual_fwd separately then).
static int linear_ln_fwd_impl(const void* x, int x_bf16, int64_t ldx, const float* w, const float* bias, const float* resid,
                              const float* gamma, const float* beta, float* y, float* z, int64_t M, int64_t K, int64_t N,
                              const uint8_t* mask, int mask_mode, int n_nodes, float eps, float drop_p, uint64_t seed,
                              hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !w || !resid || !gamma || !beta || !y || !z || M <= 0) return MBVAE_EINVAL;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if (mbvae_opt(MBVAE_OPT_NO_TSGEMM) || mbvae_opt(MBVAE_OPT_NO_TSGEMM2) || mbvae_opt(MBVAE_OPT_NO_LN_EPILOGUE)) return MBVAE_EUNSUPPORTED;
  if (N != 64 || K % 64 || K <= 0 || K > 512 || M % 32 || M < 8192 || M * 64 >= 0x3fffffff) return MBVAE_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)resid | (uintptr_t)y | (uintptr_t)z) & 15) || (ldx & (x_bf16 ? 7 : 3))) return MBVAE_EUNSUPPORTED;
  GemmArgs g = {};
  g.A = reinterpret_cast<const float*>(x); g.B = w; g.C = y; g.M = (int)M; g.N = 64; g.K = (int)K;
  g.lda = ldx; g.ldb = K; g.ldc = 64; g.a_bf16 = x_bf16;
  g.a_gate_scale = 1.f; g.mask = mask; g.c_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.bias = bias; g.drop_p = drop_p; g.seed = seed;
  g.addend = resid; g.ln_gamma = gamma; g.ln_beta = beta; g.ln_z = z; g.ln_eps = eps;
  const size_t smem3 = 2 * 3 * (size_t)64 * (K + 8), smemf = sizeof(float) * (size_t)K * 65;
  // (a bf16 operand always runs on the bf16 pipeline)
  const bool split1 = (x_bf16 || !mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA)) && smem3 <= MBVAE_LDS_LIMIT;
  if (x_bf16 && !split1) return MBVAE_EUNSUPPORTED;
  const size_t smem = split1 ? smem3 : smemf;
  if (smem > MBVAE_LDS_LIMIT) return MBVAE_EUNSUPPORTED;
  auto kb = tsgemm2_kernel<2, true, false, 8, true, 1, true, false, false, true>;
  auto ks = tsgemm2_kernel<2, true, false, 8, true, 1, false, false, false, true>;
  auto kf = tsgemm2_kernel<2, true, false, 8, true, 0, false, false, false, true>;
  if (!(x_bf16 ? MBVAE_ALLOW_BIG_LDS(kb) : split1 ? MBVAE_ALLOW_BIG_LDS(ks) : MBVAE_ALLOW_BIG_LDS(kf))) return MBVAE_ELAUNCH;
  const int grid = std::min(std::max((int)(M / 256), 1), ts_num_cus());
  hipLaunchKernelGGL(x_bf16 ? kb : split1 ? ks : kf, dim3(grid), dim3(512), smem, stream, g);
  return mbvae_launch_status();
}

int mbvae_linear_ln_fwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* resid,
                        const float* gamma, const float* beta, float* y, float* z, int64_t M, int64_t K, int64_t N,
                        const uint8_t* mask, int mask_mode, int n_nodes, float eps, float drop_p, uint64_t seed,
                        hipStream_t stream) {
  return linear_ln_fwd_impl(x, 0, ldx, w, bias, resid, gamma, beta, y, z, M, K, N, mask, mask_mode, n_nodes, eps, drop_p, seed, stream);
}

// The same with x stored as bf16 (the attention output P of the bf16 storage mode feeding e_out).
int mbvae_linear_ln_fwd_bf16(const uint16_t* x, int64_t ldx, const float* w, const float* bias, const float* resid,
                             const float* gamma, const float* beta, float* y, float* z, int64_t M, int64_t K, int64_t N,
                             const uint8_t* mask, int mask_mode, int n_nodes, float eps, float drop_p, uint64_t seed,
                             hipStream_t stream) {
  return linear_ln_fwd_impl(x, 1, ldx, w, bias, resid, gamma, beta, y, z, M, K, N, mask, mask_mode, n_nodes, eps, drop_p, seed, stream);
}

// dx[M,K] = pro(dy)[M,N] @ w[N,K] + addend[M,K] (addend laid out like dx).  Only shapes served by the persistent
// tall-skinny kernel are supported (MBVAE_EUNSUPPORTED otherwise: add in a separate pass).
int mbvae_linear_bwd_input_add(const float* dy, int64_t lddy, const float* w, const float* addend, float* dx, int64_t lddx,
                               int64_t M, int64_t K, int64_t N, const float* gate, float gate_scale,
                               const uint8_t* mask, int mask_mode, int n_nodes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if (!addend || (gate && lddy != N)) return MBVAE_EINVAL;
  GemmArgs g = {};
  g.A = dy; g.B = w; g.C = dx; g.M = (int)M; g.N = (int)K; g.K = (int)N;
  g.lda = lddy; g.ldb = K; g.ldc = lddx;
  g.a_gate = gate; g.a_gate_scale = gate_scale; g.mask = mask; g.a_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.addend = addend;
  return gemm_dispatch<true, false>(g, nullptr, 0, stream);
}

// dw[N,K] = pro(dy)[M,N]^T @ x[M,K];  db[N] = column sums of pro(dy)  (db may be null)
int mbvae_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db,
                            int64_t M, int64_t K, int64_t N, const float* gate, float gate_scale,
                            const uint8_t* mask, int mask_mode, int n_nodes, int accumulate,
                            void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if (gate && lddy != N) return MBVAE_EINVAL;
  GemmArgs g = {};
  g.A = dy; g.B = x; g.C = dw; g.M = (int)N; g.N = (int)K; g.K = (int)M;
  g.lda = lddy; g.ldb = ldx; g.ldc = K;
  g.a_gate = gate; g.a_gate_scale = gate_scale; g.mask = mask; g.a_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.accumulate = accumulate; g.colsum = db;
  return gemm_dispatch<false, false>(g, workspace, ws_bytes, stream);
}

// dx[M,K] = out_gate[M,K] > 0 ? (rowmask(dy)[M,N] @ w[N,K]) * out_gate_scale : 0   (out_gate laid out like dx)
// The input gradient of a layer whose INPUT came out of ReLU (+ dropout): out_gate is that input, so the gate of the
// layer below is applied here, where the gradient is produced, and that layer's own gradient kernels read an already
// gated dy (mbvae_linear_bwd_input* / _bwd_weight with gate = NULL).  Edge-path shapes apply the gate in the product's
// epilogue; any other shape runs the plain product followed by one element-wise pass.
int mbvae_linear_bwd_input_outgate(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx, int64_t M, int64_t K,
                                   int64_t N, const float* out_gate, float out_gate_scale, const uint8_t* mask, int mask_mode,
                                   int n_nodes, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if (!out_gate || lddx != K) return MBVAE_EINVAL;
  GemmArgs g = {};
  g.A = dy; g.B = w; g.C = dx; g.M = (int)M; g.N = (int)K; g.K = (int)N;
  g.lda = lddy; g.ldb = K; g.ldc = lddx;
  g.a_gate_scale = 1.f; g.mask = mask; g.a_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.c_gate = out_gate; g.c_gate_scale = out_gate_scale;
  if (dy && w && dx && M > 0 && K > 0 && N > 0) {
    const int st = tsgemm_try<false>(g, stream);
    if (st != MBVAE_EUNSUPPORTED) return st;
  }
  g.c_gate = nullptr;
  const int st = gemm_dispatch<true, false>(g, workspace, ws_bytes, stream);
  if (st != MBVAE_OK) return st;
  const int64_t total = M * K;
  hipLaunchKernelGGL(out_gate_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, stream, dx, out_gate,
                     out_gate_scale, total);
  return mbvae_launch_status();
}

// ---- bf16 storage of the edge block's large activations (opt-in; fp32 weights, fp32 accumulation) ----------------
// y[M,N] (fp32) = rowmask(x[M,K] (bf16) @ w[N,K]^T + bias).  Edge-path shapes only (M >= 32768, M % 32 == 0,
// K % 64 == 0, N % 64 == 0, K*N <= 32768), MBVAE_EUNSUPPORTED otherwise.
int mbvae_linear_fwd_bf16(const uint16_t* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                          int64_t M, int64_t K, int64_t N, const uint8_t* mask, int mask_mode, int n_nodes,
                          hipStream_t stream) {
  (void)hipGetLastError();
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  GemmArgs g = {};
  g.A = (const float*)x; g.B = w; g.C = y; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = ldx; g.ldb = K; g.ldc = ldy;
  g.a_gate_scale = 1.f; g.mask = mask; g.c_mask_mode = mask_mode; g.n_nodes = n_nodes; g.bias = bias;
  g.a_bf16 = 1;
  return gemm_dispatch<true, true>(g, nullptr, 0, stream);
}

// dx[M,K] = rowmask(dy)[M,N] @ w[N,K] (+ addend[M,K], fp32, may be NULL).  dy_bf16 / dx_bf16 say which of the two
// activations is bf16 in memory (exactly one of them; the other is fp32).  Edge-path shapes only.
int mbvae_linear_bwd_input_bf16(const void* dy, int dy_bf16, int64_t lddy, const float* w, const float* addend, void* dx,
                                int dx_bf16, int64_t lddx, int64_t M, int64_t K, int64_t N, const uint8_t* mask,
                                int mask_mode, int n_nodes, hipStream_t stream) {
  (void)hipGetLastError();
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if ((dy_bf16 != 0) == (dx_bf16 != 0) || (addend && dx_bf16)) return MBVAE_EUNSUPPORTED;
  GemmArgs g = {};
  g.A = (const float*)dy; g.B = w; g.C = (float*)dx; g.M = (int)M; g.N = (int)K; g.K = (int)N;
  g.lda = lddy; g.ldb = K; g.ldc = lddx;
  g.a_gate_scale = 1.f; g.mask = mask; g.a_mask_mode = mask_mode; g.n_nodes = n_nodes;
  g.addend = addend; g.a_bf16 = dy_bf16 != 0; g.c_bf16 = dx_bf16 != 0;
  return gemm_dispatch<true, false>(g, nullptr, 0, stream);
}

// dw[N,K] = rowmask(dy)[M,N]^T @ x[M,K];  db[N] = column sums of rowmask(dy) (may be NULL).  dy_bf16 / x_bf16: which
// operand is bf16 in memory (exactly one).  Shapes: (N, K) = (512, 64) with dy bf16, (64, 256) with x bf16.
int mbvae_linear_bwd_weight_bf16(const void* dy, int dy_bf16, int64_t lddy, const void* x, int x_bf16, int64_t ldx,
                                 float* dw, float* db, int64_t M, int64_t K, int64_t N, const uint8_t* mask,
                                 int mask_mode, int n_nodes, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (M > 0x7fffffff || K > 0x7fffffff || N > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if ((dy_bf16 != 0) == (x_bf16 != 0)) return MBVAE_EUNSUPPORTED;
  GemmArgs g = {};
  g.A = (const float*)dy; g.B = (const float*)x; g.C = dw; g.M = (int)N; g.N = (int)K; g.K = (int)M;
  g.lda = lddy; g.ldb = ldx; g.ldc = K;
  g.a_gate_scale = 1.f; g.mask = mask; g.a_mask_mode = mask_mode; g.n_nodes = n_nodes; g.colsum = db;
  g.a_bf16 = dy_bf16 != 0; g.b_bf16 = x_bf16 != 0;
  return gemm_dispatch<false, false>(g, workspace, ws_bytes, stream);
}

// Both gradients of a Linear layer from one pass over dy:  dx = (addend +) rowmask(dy) @ w,  dw = rowmask(dy)^T @ x,
// db = column sums (may be NULL).  Only the widest edge projection qualifies (K = 64 inputs, N = 512 outputs,
// M >= 32768 rows, M % 64 == 0); everything else returns "unsupported" and the caller issues the two separate calls.
// workspace: at least CUs * (N * K + N) * 4 bytes.
int mbvae_linear_bwd_both(const float* dy, int64_t lddy, const float* w, const float* x, int64_t ldx, const float* addend,
                          float* dx, int64_t lddx, float* dw, float* db, int64_t M, int64_t K, int64_t N,
                          const uint8_t* mask, int mask_mode, int n_nodes, void* workspace, int64_t ws_bytes,
                          hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !w || !x || !dx || !dw || M <= 0) return MBVAE_EINVAL;
  if (mask_mode != MBVAE_MASK_NONE && (!mask || n_nodes <= 0)) return MBVAE_EINVAL;
  if (mbvae_opt(MBVAE_OPT_NO_BWD_BOTH) || mbvae_opt(MBVAE_OPT_NO_SPLIT_MFMA)) return MBVAE_EUNSUPPORTED;
  const bool aligned = (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)addend | (uintptr_t)w) & 15) == 0 &&
                       (lddy & 3) == 0 && (ldx & 1) == 0 && (lddx & 1) == 0;
  if (K != 64 || N != 512 || M % 64 || M < 32768 || !aligned || (M + 16) * lddy >= 0x7fffffff || workspace == nullptr)
    return MBVAE_EUNSUPPORTED;
  const int nchunks = (int)(M / 64);
  const int grid = std::min(nchunks, mbvae_num_cus());
  if (ws_bytes < (int64_t)grid * (512 * 64 + 512) * 4) return MBVAE_EUNSUPPORTED;
  constexpr int kLds = 14 * 8192 + 32768 + 2 * 6144;
  {
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MBVAE_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      if (hipFuncSetAttribute((const void*)proj_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess ||
          hipFuncSetAttribute((const void*)proj_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess)
        return MBVAE_ELAUNCH;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  ProjBwdArgs a;
  a.dY = dy; a.lddy = (int)lddy; a.X = x; a.ldx = (int)ldx; a.W = w; a.addend = addend; a.dX = dx; a.lddx = (int)lddx;
  a.partial = (float*)workspace;
  a.partial_colsum = db ? a.partial + (int64_t)grid * 512 * 64 : nullptr;
  a.mask = mask; a.mask_mode = mask_mode; a.n_nodes = n_nodes; a.nchunks = nchunks;
  if (addend) hipLaunchKernelGGL(proj_bwd_kernel<true>, dim3(grid), dim3(512), kLds, stream, a);
  else hipLaunchKernelGGL(proj_bwd_kernel<false>, dim3(grid), dim3(512), kLds, stream, a);
  GemmArgs r{};
  r.M = 512; r.N = 64; r.C = dw; r.ldc = 64; r.partial = a.partial; r.partial_colsum = a.partial_colsum; r.colsum = db; r.accumulate = 0;
  hipLaunchKernelGGL(tswgrad_reduce_kernel, dim3((unsigned)((512 * 64 + 512 + 63) / 64)), dim3(256), 0, stream, r, grid);
  return mbvae_launch_status();
}

}  // extern "C"
