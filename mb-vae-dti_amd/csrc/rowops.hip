// Row-wise streaming kernels of the transformer layer (all HBM-bound):
//   * residual + dropout + LayerNorm, forward and backward          (XEyTransformerLayer norm*1 / norm*2)
//   * mean|min|max|std pooling over nodes / node pairs, fwd + bwd   (Xtoy / Etoy)
//   * symmetrise + mask of an edge tensor                           (GraphTransformer input / output)
// Layout everywhere: feature dimension contiguous, one lane owns 4 consecutive features
// (16-byte accesses); a group of LPR = D/4 lanes (a power of two, <= 64) owns a row so that a
// 64-feature edge row uses 16 lanes and a wave streams 4 rows per instruction.
#include "common.h"
#include <algorithm>

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm( x + dropout(r) ) * gamma + beta.   D = LPR*4*VPL features per row.
// grid-stride over row groups; block 256.
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void ln_res_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, int64_t rows, float eps,
                                                         float drop_p, uint64_t seed) {
  constexpr int D = LPR * 4 * VPL;
  const int sub = threadIdx.x % LPR;
  const int64_t rows_per_block = 256 / LPR;
  float4 gm[VPL], bt[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) { gm[v] = ld4(gamma + (v * LPR + sub) * 4); bt[v] = ld4(beta + (v * LPR + sub) * 4); }
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / LPR; row < rows;
       row += (int64_t)gridDim.x * rows_per_block) {
    float4 z[VPL];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int64_t off = row * D + (v * LPR + sub) * 4;
      z[v] = ld4(x + off);
      if (r) {
        const float4 t = ld4(r + off);
        if (drop_p > 0.f) {
          z[v].x += t.x * mbvae_drop_scale(drop_p, seed, off + 0);
          z[v].y += t.y * mbvae_drop_scale(drop_p, seed, off + 1);
          z[v].z += t.z * mbvae_drop_scale(drop_p, seed, off + 2);
          z[v].w += t.w * mbvae_drop_scale(drop_p, seed, off + 3);
        } else { z[v].x += t.x; z[v].y += t.y; z[v].z += t.z; z[v].w += t.w; }
      }
      s += (z[v].x + z[v].y) + (z[v].z + z[v].w);
    }
    const float mean = group_sum<LPR>(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float a = z[v].x - mean, b = z[v].y - mean, c = z[v].z - mean, d = z[v].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(group_sum<LPR>(q) * (1.f / D) + eps);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float4 o;
      o.x = (z[v].x - mean) * rstd * gm[v].x + bt[v].x;
      o.y = (z[v].y - mean) * rstd * gm[v].y + bt[v].y;
      o.z = (z[v].z - mean) * rstd * gm[v].z + bt[v].z;
      o.w = (z[v].w - mean) * rstd * gm[v].w + bt[v].w;
      st4(y + row * D + (v * LPR + sub) * 4, o);
    }
    if (sub == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// Backward: dz = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma;  dx = dz;  dr = dz * dropscale.
// Per-block partial sums of dgamma/dbeta go to part[blockIdx.x][2][D] (summed by colsum_partials_kernel).
#ifndef LN_BWD_UNR
#define LN_BWD_UNR 2      // rows per trip of the narrow-row form (tuning macro; 3 and 4 measured the same / slightly slower)
#endif
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void ln_res_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                         const float* __restrict__ rstd_in, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ dr,
                                                         float* __restrict__ part, int64_t rows, float drop_p, uint64_t seed,
                                                         float eps) {
  // mean_in == null: x already IS the pre-LayerNorm sum z (r is null then) and the row statistics are recomputed
  // from it, in the forward's two-pass form; dr = dz * dropout scale is still written for the branch that was added.
  constexpr int D = LPR * 4 * VPL;
  constexpr int GROUPS = 256 / LPR;
  __shared__ float red[2 * 256 * 4];
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  float4 gm[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    gm[v] = ld4(gamma + (v * LPR + sub) * 4);
    dg[v] = make_float4(0.f, 0.f, 0.f, 0.f); db[v] = dg[v];
  }
  // Narrow rows (one 16-byte word per lane) take two rows per trip, every load of both issued before the first use:
  // with two or three streams of 16 bytes per lane the kernel is otherwise short of bytes in flight.
  constexpr int UNR = VPL == 1 ? LN_BWD_UNR : 1;
  const int64_t stride = (int64_t)gridDim.x * GROUPS;
  for (int64_t row0 = (int64_t)blockIdx.x * GROUPS + grp; row0 < rows; row0 += stride * UNR) {
    float4 zz[UNR][VPL], dd[UNR][VPL];
    float mean[UNR], rstd[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      ok[u] = row0 + u * stride < rows;
      const int64_t row = ok[u] ? row0 + u * stride : row0;      // (a row past the end re-reads row0; nothing of it is kept)
      if (mean_in) { mean[u] = mean_in[row]; rstd[u] = rstd_in[row]; }
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int64_t off = row * D + (v * LPR + sub) * 4;
        zz[u][v] = ld4(x + off);
        dd[u][v] = ld4(dy + off);
        if (r) {
          const float4 t = ld4(r + off);
          if (drop_p > 0.f) {
            zz[u][v].x += t.x * mbvae_drop_scale(drop_p, seed, off + 0);
            zz[u][v].y += t.y * mbvae_drop_scale(drop_p, seed, off + 1);
            zz[u][v].z += t.z * mbvae_drop_scale(drop_p, seed, off + 2);
            zz[u][v].w += t.w * mbvae_drop_scale(drop_p, seed, off + 3);
          } else { zz[u][v].x += t.x; zz[u][v].y += t.y; zz[u][v].z += t.z; zz[u][v].w += t.w; }
        }
      }
    }
    if (!mean_in) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) s += (zz[u][v].x + zz[u][v].y) + (zz[u][v].z + zz[u][v].w);
        mean[u] = group_sum<LPR>(s) * (1.f / D);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const float a = zz[u][v].x - mean[u], b = zz[u][v].y - mean[u], c = zz[u][v].z - mean[u], d = zz[u][v].w - mean[u];
          q += (a * a + b * b) + (c * c + d * d);
        }
        rstd[u] = rsqrtf(group_sum<LPR>(q) * (1.f / D) + eps);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = ok[u] ? row0 + u * stride : row0;
      const float keep = ok[u] ? 1.f : 0.f;
      float4 xh[VPL], g[VPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float4 z = zz[u][v], d = dd[u][v];
        xh[v] = make_float4((z.x - mean[u]) * rstd[u], (z.y - mean[u]) * rstd[u], (z.z - mean[u]) * rstd[u], (z.w - mean[u]) * rstd[u]);
        g[v] = make_float4(d.x * gm[v].x, d.y * gm[v].y, d.z * gm[v].z, d.w * gm[v].w);
        s1 += (g[v].x + g[v].y) + (g[v].z + g[v].w);
        s2 += (g[v].x * xh[v].x + g[v].y * xh[v].y) + (g[v].z * xh[v].z + g[v].w * xh[v].w);
        if (UNR == 1) {
          dg[v].x += d.x * xh[v].x; dg[v].y += d.y * xh[v].y; dg[v].z += d.z * xh[v].z; dg[v].w += d.w * xh[v].w;
          db[v].x += d.x; db[v].y += d.y; db[v].z += d.z; db[v].w += d.w;
        } else {                                   // a row past the end contributes nothing
          dg[v].x += keep * d.x * xh[v].x; dg[v].y += keep * d.y * xh[v].y; dg[v].z += keep * d.z * xh[v].z; dg[v].w += keep * d.w * xh[v].w;
          db[v].x += keep * d.x; db[v].y += keep * d.y; db[v].z += keep * d.z; db[v].w += keep * d.w;
        }
      }
      const float m1 = group_sum<LPR>(s1) * (1.f / D), m2 = group_sum<LPR>(s2) * (1.f / D);
      if (ok[u]) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int64_t off = row * D + (v * LPR + sub) * 4;
          float4 dz;
          dz.x = rstd[u] * (g[v].x - m1 - xh[v].x * m2);
          dz.y = rstd[u] * (g[v].y - m1 - xh[v].y * m2);
          dz.z = rstd[u] * (g[v].z - m1 - xh[v].z * m2);
          dz.w = rstd[u] * (g[v].w - m1 - xh[v].w * m2);
          st4(dx + off, dz);
          if (dr) {
            if (drop_p > 0.f) {
              dz.x *= mbvae_drop_scale(drop_p, seed, off + 0);
              dz.y *= mbvae_drop_scale(drop_p, seed, off + 1);
              dz.z *= mbvae_drop_scale(drop_p, seed, off + 2);
              dz.w *= mbvae_drop_scale(drop_p, seed, off + 3);
            }
            st4(dr + off, dz);
          }
        }
      }
    }
  }
  // reduce dgamma/dbeta over the GROUPS row groups of this block (fixed order -> deterministic)
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    __syncthreads();
    float* a = red + threadIdx.x * 4;
    float* b = red + 1024 + threadIdx.x * 4;
    a[0] = dg[v].x; a[1] = dg[v].y; a[2] = dg[v].z; a[3] = dg[v].w;
    b[0] = db[v].x; b[1] = db[v].y; b[2] = db[v].z; b[3] = db[v].w;
    __syncthreads();
    if (threadIdx.x < LPR * 4) {
      const int lanei = threadIdx.x / 4, comp = threadIdx.x % 4;
      float sg = 0.f, sb = 0.f;
      for (int gidx = 0; gidx < GROUPS; ++gidx) {
        sg += red[(gidx * LPR + lanei) * 4 + comp];
        sb += red[1024 + (gidx * LPR + lanei) * 4 + comp];
      }
      const int col = (v * LPR + lanei) * 4 + comp;
      part[((int64_t)blockIdx.x * 2 + 0) * D + col] = sg;
      part[((int64_t)blockIdx.x * 2 + 1) * D + col] = sb;
    }
  }
}

// (c < half ? out0[c] : out1[c - half]) = sum_p part[p][c], c < width  (second stage of the per-block partial sums; the
// two halves are dgamma and dbeta, written straight to their destinations).
// block = 32 columns x 8 partial groups; fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                              float* __restrict__ out1, int nparts, int width, int half) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float s = 0.f;
  if (c < width) {
    // four independent chains: the loads of a chain are issued back to back instead of one per memory round trip
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = py;
    for (; p + 24 < nparts; p += 32) {
      s0 += part[(int64_t)p * width + c];        s1 += part[(int64_t)(p + 8) * width + c];
      s2 += part[(int64_t)(p + 16) * width + c]; s3 += part[(int64_t)(p + 24) * width + c];
    }
    for (; p < nparts; p += 8) s0 += part[(int64_t)p * width + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[py][cx] = s;
  __syncthreads();
  if (py == 0 && c < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    if (c < half) out0[c] = t; else out1[c - half] = t;
  }
}

template <int LPR, int VPL>
static int ln_launch(bool fwd, const float* x, const float* r, const float* gamma, const float* beta, float* y,
                     float* mean, float* rstd, const float* dy, float* dx, float* dr, float* dgamma, float* dbeta,
                     float* part, int64_t part_floats, int64_t rows, float eps, float drop_p, uint64_t seed,
                     hipStream_t stream) {
  constexpr int D = LPR * 4 * VPL;
  const int64_t rpb = 256 / LPR;
  int64_t blocks = (rows + rpb - 1) / rpb;
  if (fwd) {
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((ln_res_fwd_kernel<LPR, VPL>), dim3((unsigned)blocks), dim3(256), 0, stream, x, r, gamma, beta, y,
                       mean, rstd, rows, eps, drop_p, seed);
  } else {
    if (blocks > 512) blocks = 512;
    if (part_floats < blocks * 2 * D) blocks = part_floats / (2 * D);
    if (blocks < 1) return MBVAE_EINVAL;
    hipLaunchKernelGGL((ln_res_bwd_kernel<LPR, VPL>), dim3((unsigned)blocks), dim3(256), 0, stream, x, r, gamma, mean,
                       rstd, dy, dx, dr, part, rows, drop_p, seed, eps);
    // part is [blocks][2][D]: view as [blocks][2D] and sum -> dgamma | dbeta
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((2 * D + 31) / 32), dim3(256), 0, stream, part, dgamma, dbeta, (int)blocks, 2 * D, D);
  }
  return mbvae_launch_status();
}

static int ln_dispatch(bool fwd, int64_t D, const float* x, const float* r, const float* gamma, const float* beta,
                       float* y, float* mean, float* rstd, const float* dy, float* dx, float* dr, float* dgamma,
                       float* dbeta, float* part, int64_t part_floats, int64_t rows, float eps, float drop_p,
                       uint64_t seed, hipStream_t stream) {
#define LN_CASE(dd, lpr, vpl)                                                                                        \
  if (D == dd)                                                                                                       \
    return ln_launch<lpr, vpl>(fwd, x, r, gamma, beta, y, mean, rstd, dy, dx, dr, dgamma, dbeta, part, part_floats, \
                               rows, eps, drop_p, seed, stream);
  LN_CASE(16, 4, 1) LN_CASE(32, 8, 1) LN_CASE(64, 16, 1) LN_CASE(128, 32, 1) LN_CASE(256, 64, 1)
  LN_CASE(512, 64, 2) LN_CASE(768, 64, 3) LN_CASE(1024, 64, 4) LN_CASE(1536, 64, 6) LN_CASE(2048, 64, 8)
#undef LN_CASE
  return MBVAE_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Pooling: x [bs, L, D] -> stats [bs, 4D] = mean | min | max | unbiased std over L, arg [bs, 2, D] (int32)
// One 512- or 1024-thread block per (graph, 256-feature chunk); lane owns 4 features, row groups stride over L.
// Mean/variance use Welford updates per thread and Chan's pairwise merge across threads (stable in fp32).
struct Moments { float n, mean, m2; };
__device__ __forceinline__ void mom_add(Moments& a, float x) {
  a.n += 1.f; const float d = x - a.mean; a.mean += d / a.n; a.m2 += d * (x - a.mean);
}
__device__ __forceinline__ void mom_merge(Moments& a, const Moments& b) {
  if (b.n == 0.f) return;
  const float n = a.n + b.n, d = b.mean - a.mean;
  a.mean += d * (b.n / n); a.m2 += b.m2 + d * d * (a.n * b.n / n); a.n = n;
}

template <int NT>
__global__ __launch_bounds__(NT) void pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                       int* __restrict__ arg, int L, int D) {
  extern __shared__ float sm[];
  const int lpr = min(64, D / 4);                 // lanes per row (power of two)
  const int groups = NT / lpr;
  const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  const int b = blockIdx.y, c = blockIdx.x * 256 + sub * 4;
  const bool active = c < D;
  Moments mo[4]; float mn[4], mx[4]; int amn[4], amx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { mo[q] = {0.f, 0.f, 0.f}; mn[q] = INFINITY; mx[q] = -INFINITY; amn[q] = 0x7fffffff; amx[q] = 0x7fffffff; }
  if (active) {
    // Welford update with the row count and its reciprocal shared by the lane's four features (one division per
    // row instead of four; the kernel is bound by this arithmetic, not by the 256 B it reads per row)
    float cnt = 0.f;
    for (int l = grp; l < L; l += groups) {
      const float4 v = ld4(x + ((int64_t)b * L + l) * D + c);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      cnt += 1.f;
      const float inv = 1.f / cnt;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float d = vv[q] - mo[q].mean;
        mo[q].mean += d * inv;
        mo[q].m2 += d * (vv[q] - mo[q].mean);
        if (vv[q] < mn[q]) { mn[q] = vv[q]; amn[q] = l; }
        if (vv[q] > mx[q]) { mx[q] = vv[q]; amx[q] = l; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) mo[q].n = cnt;
  }
  // tree merge over row groups through LDS: layout [grp][sub][q][7]
  float* slot = sm + ((grp * lpr + sub) * 4) * 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float* s = slot + q * 7;
    s[0] = mo[q].n; s[1] = mo[q].mean; s[2] = mo[q].m2; s[3] = mn[q]; s[4] = mx[q];
    s[5] = __int_as_float(amn[q]); s[6] = __int_as_float(amx[q]);
  }
  for (int stride = groups / 2; stride > 0; stride >>= 1) {
    __syncthreads();
    if (grp < stride) {
      const float* o = sm + (((grp + stride) * lpr + sub) * 4) * 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* s = slot + q * 7; const float* t = o + q * 7;
        Moments a = {s[0], s[1], s[2]}; const Moments bb = {t[0], t[1], t[2]};
        mom_merge(a, bb); s[0] = a.n; s[1] = a.mean; s[2] = a.m2;
        const int ia = __float_as_int(s[5]), ib = __float_as_int(t[5]);
        if (t[3] < s[3] || (t[3] == s[3] && ib < ia)) { s[3] = t[3]; s[5] = t[5]; }
        const int ja = __float_as_int(s[6]), jb = __float_as_int(t[6]);
        if (t[4] > s[4] || (t[4] == s[4] && jb < ja)) { s[4] = t[4]; s[6] = t[6]; }
      }
    }
  }
  if (grp == 0 && active) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* s = slot + q * 7;
      float* o = stats + (int64_t)b * 4 * D;
      o[c + q] = s[1]; o[D + c + q] = s[3]; o[2 * D + c + q] = s[4];
      o[3 * D + c + q] = sqrtf(s[2] / (s[0] - 1.f));
      arg[((int64_t)b * 2 + 0) * D + c + q] = __float_as_int(s[5]);
      arg[((int64_t)b * 2 + 1) * D + c + q] = __float_as_int(s[6]);
    }
  }
}

// Inference form of the pair pooling for ragged batches whose padding pair rows all hold ONE row of values (the
// denoiser's edge tensor: see mbvae_edge_ffn_fwd): only the valid pairs (i, j) of a graph are read; the padding
// rows enter as `count x the constant row`, merged with Chan's formula (exact arithmetic, different rounding than
// walking over them).  The constant row is x[row_idx[32 * ceil(nvalid / 32)]], the first padding row of the packed
// row order; the padding rows themselves need not have been written.  No arg-min / arg-max output (forward only).
// One NT-thread block per graph; D = 64.  NT = 512: with 40 KB of LDS several graphs share a CU, so one block's load phase
// runs under another's merge tree (1024 threads and 80 KB meant one graph per CU at a time, phases in series: 146 us per
// call on 1250 molecules; 512: 102 us, 256: 110 us, 128: 196 us).
#ifndef POOL_PAIRS_NT
#define POOL_PAIRS_NT 512     // threads per graph (tuning macro)
#endif
template <int NT>
__global__ __launch_bounds__(NT) void pool_pairs_fwd_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                              const uint8_t* __restrict__ mask,
                                                              const int32_t* __restrict__ row_idx,
                                                              const int32_t* __restrict__ nvalid, int n, int64_t M) {
  extern __shared__ float sm[];                    // [NT / 16 groups][16 lanes][4][5]
  __shared__ uint8_t mk[1024];
  constexpr int D = 64, LPR = 16, GROUPS = NT / LPR;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int b = blockIdx.x, c = sub * 4;
  for (int t = threadIdx.x; t < n; t += NT) mk[t] = mask[(int64_t)b * n + t];
  __syncthreads();
  Moments mo[4]; float mn[4], mx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { mo[q] = {0.f, 0.f, 0.f}; mn[q] = INFINITY; mx[q] = -INFINITY; }
  const int L = n * n;
  float cnt = 0.f, skipped = 0.f;
  int i = grp / n, j = grp % n;
  // Four pairs per trip: their loads are issued together (the running moments make the updates one dependent chain, so a
  // lane with one load in flight waits a full memory latency per pair), then consumed in pair order as before.
  constexpr int U = 4;
  for (int l = grp; l < L; l += GROUPS * U) {
    float4 v[U];
    int kind[U];                                   // 0: past the end, 1: valid pair, 2: padding pair
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int lu = l + u * GROUPS;
      kind[u] = 0;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lu < L) {
        kind[u] = (mk[i] && mk[j]) ? 1 : 2;
        if (kind[u] == 1) v[u] = ld4(x + ((int64_t)b * L + lu) * D + c);
        j += GROUPS;
        while (j >= n) { j -= n; ++i; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kind[u] == 1) {
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        cnt += 1.f;
        const float inv = 1.f / cnt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d = vv[q] - mo[q].mean;
          mo[q].mean += d * inv;
          mo[q].m2 += d * (vv[q] - mo[q].mean);
          mn[q] = fminf(mn[q], vv[q]); mx[q] = fmaxf(mx[q], vv[q]);
        }
      } else if (kind[u] == 2) {
        skipped += 1.f;
      }
    }
  }
  const int64_t nv = nvalid[0];
  if (skipped > 0.f && nv < M) {
    const float4 cv = ld4(x + (int64_t)row_idx[(nv + 31) / 32 * 32] * D + c);
    const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mo[q].n = cnt;
      const Moments pad = {skipped, cc[q], 0.f};
      mom_merge(mo[q], pad);
      mn[q] = fminf(mn[q], cc[q]); mx[q] = fmaxf(mx[q], cc[q]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) mo[q].n = cnt;
  }
  float* slot = sm + ((grp * LPR + sub) * 4) * 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float* s = slot + q * 5;
    s[0] = mo[q].n; s[1] = mo[q].mean; s[2] = mo[q].m2; s[3] = mn[q]; s[4] = mx[q];
  }
  for (int stride = GROUPS / 2; stride > 0; stride >>= 1) {
    __syncthreads();
    if (grp < stride) {
      const float* o = sm + (((grp + stride) * LPR + sub) * 4) * 5;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* s = slot + q * 5; const float* t = o + q * 5;
        Moments a = {s[0], s[1], s[2]}; const Moments bb = {t[0], t[1], t[2]};
        mom_merge(a, bb); s[0] = a.n; s[1] = a.mean; s[2] = a.m2;
        s[3] = fminf(s[3], t[3]); s[4] = fmaxf(s[4], t[4]);
      }
    }
  }
  if (grp == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* s = slot + q * 5;
      float* o = stats + (int64_t)b * 4 * D;
      o[c + q] = s[1]; o[D + c + q] = s[3]; o[2 * D + c + q] = s[4];
      o[3 * D + c + q] = sqrtf(s[2] / (s[0] - 1.f));
    }
  }
}

// dx[b,l,c] (+)= dmean/L + dstd*(x-mean)/((L-1)*std) [0 if std==0] + [l==argmin]*dmin + [l==argmax]*dmax
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                       const int* __restrict__ arg, const float* __restrict__ dstats,
                                                       float* __restrict__ dx, int L, int D, int accumulate) {
  const int d4 = D / 4;
  const int64_t total = (int64_t)gridDim.y * L * d4;    // gridDim.y = bs
  const int b = blockIdx.y;
  const float invL = 1.f / L;
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x, e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gstride % d4 == 0) {
    // The lane keeps its four channels for the whole walk over l: the statistics and their gradients are loaded once
    // and folded into dx = base + slope * x (+ dmin / dmax at the arg rows); four rows per trip, loads first.
    const int c = (int)(e0 % d4) * 4, lstep = (int)(gstride / d4);
    const float* st = stats + (int64_t)b * 4 * D; const float* ds = dstats + (int64_t)b * 4 * D;
    const float4 mean = ld4(st + c), sd = ld4(st + 3 * D + c);
    const float4 dmean = ld4(ds + c), dmin = ld4(ds + D + c), dmax = ld4(ds + 2 * D + c), dsd = ld4(ds + 3 * D + c);
    const int4 amn = *reinterpret_cast<const int4*>(arg + ((int64_t)b * 2 + 0) * D + c);
    const int4 amx = *reinterpret_cast<const int4*>(arg + ((int64_t)b * 2 + 1) * D + c);
    const float k = 1.f / (L - 1.f);
    for (int l0 = (int)(e0 / d4); l0 < L; l0 += 4 * lstep) {
      float4 v[4], p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = min(l0 + u * lstep, L - 1);
        const int64_t off = ((int64_t)b * L + l) * D + c;
        v[u] = ld4(x + off);
        p[u] = accumulate ? ld4(dx + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = l0 + u * lstep;
        if (l < L) {
          float4 o;      // same expression as the general loop below (bit-identical results)
          o.x = dmean.x * invL + (sd.x > 0.f ? dsd.x * (v[u].x - mean.x) * k / sd.x : 0.f) + (l == amn.x ? dmin.x : 0.f) + (l == amx.x ? dmax.x : 0.f);
          o.y = dmean.y * invL + (sd.y > 0.f ? dsd.y * (v[u].y - mean.y) * k / sd.y : 0.f) + (l == amn.y ? dmin.y : 0.f) + (l == amx.y ? dmax.y : 0.f);
          o.z = dmean.z * invL + (sd.z > 0.f ? dsd.z * (v[u].z - mean.z) * k / sd.z : 0.f) + (l == amn.z ? dmin.z : 0.f) + (l == amx.z ? dmax.z : 0.f);
          o.w = dmean.w * invL + (sd.w > 0.f ? dsd.w * (v[u].w - mean.w) * k / sd.w : 0.f) + (l == amn.w ? dmin.w : 0.f) + (l == amx.w ? dmax.w : 0.f);
          if (accumulate) { o.x += p[u].x; o.y += p[u].y; o.z += p[u].z; o.w += p[u].w; }
          st4(dx + ((int64_t)b * L + l) * D + c, o);
        }
      }
    }
    return;
  }
  for (int64_t e = e0; e < (int64_t)L * d4; e += gstride) {
    const int l = (int)(e / d4), c = (int)(e % d4) * 4;
    const int64_t off = ((int64_t)b * L + l) * D + c;
    const float4 v = ld4(x + off);
    const float* st = stats + (int64_t)b * 4 * D; const float* ds = dstats + (int64_t)b * 4 * D;
    const float4 mean = ld4(st + c), sd = ld4(st + 3 * D + c);
    const float4 dmean = ld4(ds + c), dmin = ld4(ds + D + c), dmax = ld4(ds + 2 * D + c), dsd = ld4(ds + 3 * D + c);
    const int4 amn = *reinterpret_cast<const int4*>(arg + ((int64_t)b * 2 + 0) * D + c);
    const int4 amx = *reinterpret_cast<const int4*>(arg + ((int64_t)b * 2 + 1) * D + c);
    float4 o;
    const float k = 1.f / (L - 1.f);
    o.x = dmean.x * invL + (sd.x > 0.f ? dsd.x * (v.x - mean.x) * k / sd.x : 0.f) + (l == amn.x ? dmin.x : 0.f) + (l == amx.x ? dmax.x : 0.f);
    o.y = dmean.y * invL + (sd.y > 0.f ? dsd.y * (v.y - mean.y) * k / sd.y : 0.f) + (l == amn.y ? dmin.y : 0.f) + (l == amx.y ? dmax.y : 0.f);
    o.z = dmean.z * invL + (sd.z > 0.f ? dsd.z * (v.z - mean.z) * k / sd.z : 0.f) + (l == amn.z ? dmin.z : 0.f) + (l == amx.z ? dmax.z : 0.f);
    o.w = dmean.w * invL + (sd.w > 0.f ? dsd.w * (v.w - mean.w) * k / sd.w : 0.f) + (l == amn.w ? dmin.w : 0.f) + (l == amx.w ? dmax.w : 0.f);
    if (accumulate) { const float4 p = ld4(dx + off); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    st4(dx + off, o);
  }
  (void)total;
}

// ------------------------------------------------------------------------------------------------
// out[b,i,j,:] = 0.5 * (x[b,i,j,:] + x[b,j,i,:] (+ s[b,i,j,:D] + s[b,j,i,:D])) * em(b,i,j) * (zero_diag ? i != j : 1)
// `s` (optional) has row stride lds and contributes its first D features.  One block per (b,i) row of pairs.
template <int VEC>
__global__ __launch_bounds__(256) void sym_mask_kernel(const float* __restrict__ x, const float* __restrict__ s, int lds,
                                                       const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                       int n, int D, int zero_diag) {
  const int64_t bi = blockIdx.x;
  const int64_t b = bi / n; const int i = (int)(bi - b * n);
  const bool row_ok = mask[bi] != 0;
  const int dv = D / VEC;
  for (int t = threadIdx.x; t < n * dv; t += 256) {
    const int j = t / dv, c = (t - j * dv) * VEC;
    const int64_t p = bi * n + j, pt = (b * n + j) * n + i;
    const bool keep = row_ok && mask[b * n + j] && !(zero_diag && i == j);
    if (VEC == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (keep) {
        const float4 a = ld4(x + p * D + c), bb = ld4(x + pt * D + c);
        v = make_float4(a.x + bb.x, a.y + bb.y, a.z + bb.z, a.w + bb.w);
        if (s) {
          const float* s1 = s + p * lds + c; const float* s2 = s + pt * lds + c;
          v.x += s1[0] + s2[0]; v.y += s1[1] + s2[1]; v.z += s1[2] + s2[2]; v.w += s1[3] + s2[3];
        }
        v.x *= 0.5f; v.y *= 0.5f; v.z *= 0.5f; v.w *= 0.5f;
      }
      st4(out + p * D + c, v);
    } else {
      float v = 0.f;
      if (keep) {
        v = x[p * D + c] + x[pt * D + c];
        if (s) v += s[p * lds + c] + s[pt * lds + c];
        v *= 0.5f;
      }
      out[p * D + c] = v;
    }
  }
}

// Generic (any D) residual + LayerNorm: one wave per row, scalar accesses.  Slow path for odd widths.
__global__ __launch_bounds__(256) void ln_res_fwd_generic_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ y, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out, int64_t rows, int D, float eps,
                                                                 float drop_p, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    auto zval = [&](int c) {
      const int64_t off = row * D + c;
      float z = x[off];
      if (r) z += r[off] * mbvae_drop_scale(drop_p, seed, off);
      return z;
    };
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += zval(c);
    const float mean = wave_sum(s) / D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = zval(c) - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    for (int c = lane; c < D; c += 64) y[row * D + c] = (zval(c) - mean) * rstd * gamma[c] + beta[c];
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// Generic backward (any width): one wave per row writes dx / dr; dgamma and dbeta come from a second kernel in which a
// thread owns a column and walks a block of rows in order, followed by the fixed-order sum of the row blocks' partials
// (colsum_partials_kernel) - no atomics, bitwise reproducible like the vectorised kernels.
__global__ __launch_bounds__(256) void ln_res_bwd_generic_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                                 const float* __restrict__ rstd_in, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ dr,
                                                                 int64_t rows, int D, float drop_p, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    auto xhat = [&](int c) {
      const int64_t off = row * D + c;
      float z = x[off];
      if (r) z += r[off] * mbvae_drop_scale(drop_p, seed, off);
      return (z - mean) * rstd;
    };
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < D; c += 64) { const float g = dy[row * D + c] * gamma[c]; s1 += g; s2 += g * xhat(c); }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
    for (int c = lane; c < D; c += 64) {
      const int64_t off = row * D + c;
      const float dz = rstd * (dy[off] * gamma[c] - m1 - xhat(c) * m2);
      dx[off] = dz;
      if (dr) dr[off] = dz * mbvae_drop_scale(drop_p, seed, off);
    }
  }
}

// part[blockIdx.y][0][c] = sum over the block's rows of dy * xhat, part[blockIdx.y][1][c] = sum of dy (rows in order)
__global__ __launch_bounds__(256) void ln_res_bwd_generic_colsum_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                        const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                        const float* __restrict__ dy, float* __restrict__ part,
                                                                        int64_t rows, int D, int64_t rows_per_block, float drop_p,
                                                                        uint64_t seed) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float sg = 0.f, sb = 0.f;
  for (int64_t row = r0; row < r1; ++row) {
    const int64_t off = row * D + c;
    float z = x[off];
    if (r) z += r[off] * mbvae_drop_scale(drop_p, seed, off);
    const float d = dy[off];
    sg += d * ((z - mean_in[row]) * rstd_in[row]);
    sb += d;
  }
  part[((int64_t)blockIdx.y * 2 + 0) * D + c] = sg;
  part[((int64_t)blockIdx.y * 2 + 1) * D + c] = sb;
}

// ------------------------------------------------------------------------------------------------
// Node update modulated by the global features:  out[b,i,c] = add[b,c] + (mul[b,c] + 1) * x[b,i,c]   (x, out [bs,n,D]).
// Forward: element-wise, 16 bytes per lane.  Backward: one workgroup per graph, a lane owns channels and walks the n
// nodes: dx = dout * (mul + 1), dadd = sum_i dout, dmul = sum_i dout * x in node order (no atomics: reproducible).
__global__ __launch_bounds__(256) void film_fwd_kernel(const float* __restrict__ add, const float* __restrict__ mul,
                                                       const float* __restrict__ x, float* __restrict__ out, int64_t total4,
                                                       int nD4, int D4) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / nD4;
    const int c4 = (int)(e % D4);
    const float4 a = ld4(add + (b * D4 + c4) * 4), m = ld4(mul + (b * D4 + c4) * 4), v = ld4(x + e * 4);
    st4(out + e * 4, make_float4(a.x + (m.x + 1.f) * v.x, a.y + (m.y + 1.f) * v.y, a.z + (m.z + 1.f) * v.z, a.w + (m.w + 1.f) * v.w));
  }
}

__global__ __launch_bounds__(256) void film_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mul,
                                                       const float* __restrict__ x, float* __restrict__ dx,
                                                       float* __restrict__ dadd, float* __restrict__ dmul, int n, int D) {
  const int64_t b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) {
    const float m1 = mul[b * D + c] + 1.f;
    float sa = 0.f, sm = 0.f;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const int64_t off = (b * n + i) * D + c;
      const float g = dout[off];
      dx[off] = g * m1;
      sa += g; sm += g * x[off];
    }
    dadd[b * D + c] = sa; dmul[b * D + c] = sm;
  }
}

// ------------------------------------------------------------------------------------------------
// Sigmoid-gate update of the encoders:  y = x + u * g,  g = sigmoid(z) * keep * scale   (all [n] element-wise; keep is the
// dropout keep mask of the gate, one byte per element, null = no dropout).  Backward from (dy, u, z, keep): g recomputed;
// du = dy * g (or, where u IS x, dx = dy + dy * g in one tensor), dz = dy * u * keep * scale * s * (1 - s).
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                       const float* __restrict__ z, const uint8_t* __restrict__ keep, float scale,
                                                       float* __restrict__ y, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float s = 1.f / (1.f + expf(-z[e]));
    const float g = keep ? (keep[e] ? s * scale : 0.f) : s;
    y[e] = x[e] + u[e] * g;
  }
}

template <bool U_IS_X>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                                       const float* __restrict__ z, const uint8_t* __restrict__ keep, float scale,
                                                       float* __restrict__ du, float* __restrict__ dz, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float s = 1.f / (1.f + expf(-z[e]));
    const float k = keep ? (keep[e] ? scale : 0.f) : 1.f;
    const float g = dy[e];
    du[e] = U_IS_X ? g + g * (s * k) : g * (s * k);
    dz[e] = g * u[e] * k * (s * (1.f - s));
  }
}

extern "C" {

// y = LayerNorm(x + dropout_p(r)) * gamma + beta over the last dim D; r may be null.  Saves mean/rstd [rows].
int mbvae_ln_residual_fwd(const float* x, const float* r, const float* gamma, const float* beta, float* y,
                          float* mean, float* rstd, int64_t rows, int64_t D, float eps, float drop_p, uint64_t seed,
                          hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return MBVAE_EINVAL;
  int st = ln_dispatch(true, D, x, r, gamma, beta, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                       rows, eps, drop_p, seed, stream);
  if (st != MBVAE_EUNSUPPORTED) return st;
  if (D > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int64_t blocks = std::min<int64_t>((rows + 3) / 4, 4096);
  hipLaunchKernelGGL(ln_res_fwd_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, r, gamma, beta, y, mean,
                     rstd, rows, (int)D, eps, drop_p, seed);
  return mbvae_launch_status();
}

// Gradients of the above: dx (wrt x), dr (wrt r, may be null), dgamma, dbeta [D].
// workspace: at least mbvae_ln_residual_bwd_workspace(D) bytes.
int64_t mbvae_ln_residual_bwd_workspace(int64_t D) { return (int64_t)sizeof(float) * (1024 * 2 * D + 2 * D); }

int mbvae_ln_residual_bwd(const float* x, const float* r, const float* gamma, const float* mean, const float* rstd,
                          const float* dy, float* dx, float* dr, float* dgamma, float* dbeta, int64_t rows, int64_t D,
                          float drop_p, uint64_t seed, void* workspace, int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !gamma || !mean || !rstd || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0) return MBVAE_EINVAL;
  const int64_t floats = ws_bytes / (int64_t)sizeof(float) - 2 * D;
  if (floats < 2 * D) return MBVAE_EINVAL;
  int st = ln_dispatch(false, D, x, r, gamma, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dy, dx,
                       dr, dgamma, dbeta, (float*)workspace, floats, rows, 0.f, drop_p, seed, stream);
  if (st != MBVAE_EUNSUPPORTED) return st;
  if (D > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int64_t blocks = std::min<int64_t>((rows + 3) / 4, 1024);
  hipLaunchKernelGGL(ln_res_bwd_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, r, gamma, mean, rstd, dy,
                     dx, dr, rows, (int)D, drop_p, seed);
  int64_t row_blocks = std::min<int64_t>(std::min<int64_t>(256, rows), floats / (2 * D));
  const int64_t rpb = (rows + row_blocks - 1) / row_blocks;
  row_blocks = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(ln_res_bwd_generic_colsum_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)row_blocks), dim3(256), 0, stream, x, r,
                     mean, rstd, dy, (float*)workspace, rows, (int)D, rpb, drop_p, seed);
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)((2 * D + 31) / 32)), dim3(256), 0, stream, (const float*)workspace, dgamma, dbeta,
                     (int)row_blocks, (int)(2 * D), (int)D);
  return mbvae_launch_status();
}

// Backward of y = LayerNorm(z) * gamma + beta where z = x + dropout_p(r) was formed and stored by the producer
// (mbvae_linear_ln_fwd): dx = dz (gradient of the residual operand), dr = dz * dropout scale of (seed, element index)
// (gradient of the Linear's output; may be null), dgamma, dbeta.  The row statistics are recomputed from z.
// Widths of the vectorised kernel only (16 ... 2048, see ln_dispatch).
int mbvae_ln_sum_bwd(const float* z, const float* gamma, const float* dy, float* dx, float* dr, float* dgamma,
                     float* dbeta, int64_t rows, int64_t D, float eps, float drop_p, uint64_t seed, void* workspace,
                     int64_t ws_bytes, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!z || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0) return MBVAE_EINVAL;
  const int64_t floats = ws_bytes / (int64_t)sizeof(float) - 2 * D;
  if (floats < 2 * D) return MBVAE_EINVAL;
  return ln_dispatch(false, D, z, nullptr, gamma, nullptr, nullptr, nullptr, nullptr, dy, dx, dr, dgamma, dbeta,
                     (float*)workspace, floats, rows, eps, drop_p, seed, stream);
}

// x [bs,L,D] -> stats [bs,4D] (mean|min|max|std), arg [bs,2,D] int32 (first argmin / argmax along L)
int mbvae_pool_stats_fwd(const float* x, float* stats, int32_t* arg, int64_t bs, int64_t L, int64_t D, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !stats || !arg || bs <= 0 || L <= 0 || D <= 0) return MBVAE_EINVAL;
  if ((D & 3) || bs > 65535 || L > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int lpr = (int)(D / 4 < 64 ? D / 4 : 64);
  if (lpr & (lpr - 1)) return MBVAE_EUNSUPPORTED;     // D/4 must be a power of two (or D >= 256 and D % 256 == 0)
  if (D > 256 && (D % 256)) return MBVAE_EUNSUPPORTED;
  // every thread walks its rows one after the other (each Welford step needs the previous mean), so long pair
  // dimensions get 1024 threads per graph
  const dim3 grid((unsigned)((D + 255) / 256), (unsigned)bs);
  if (L >= 2048) {
    const size_t smem = (size_t)1024 * 4 * 7 * sizeof(float);
    if (!MBVAE_ALLOW_BIG_LDS(pool_fwd_kernel<1024>)) return MBVAE_ELAUNCH;
    hipLaunchKernelGGL(pool_fwd_kernel<1024>, grid, dim3(1024), smem, stream, x, stats, (int*)arg, (int)L, (int)D);
  } else {
    const size_t smem = (size_t)512 * 4 * 7 * sizeof(float);
    hipLaunchKernelGGL(pool_fwd_kernel<512>, grid, dim3(512), smem, stream, x, stats, (int*)arg, (int)L, (int)D);
  }
  return mbvae_launch_status();
}

// Node pooling over packed rows: graph b's valid nodes are rows off[b] .. off[b+1]-1 of x [rows, D]; its n - nb padding
// nodes all hold row pad_row and are merged analytically.  One 256-thread block per (graph, 256-feature chunk):
// 64 lanes per row (or D/4), the rest of the block strides over the rows.
__global__ __launch_bounds__(256) void pool_nodes_fwd_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                             const int32_t* __restrict__ off, int64_t pad_row, int n, int D) {
  __shared__ float sm[256 * 4 * 5];
  const int lpr = min(64, D / 4), groups = 256 / lpr;
  const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  const int b = blockIdx.y, c = blockIdx.x * 256 + sub * 4;
  const bool active = c < D;
  const int r0 = off[b], nb = off[b + 1] - r0;
  Moments mo[4]; float mn[4], mx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { mo[q] = {0.f, 0.f, 0.f}; mn[q] = INFINITY; mx[q] = -INFINITY; }
  if (active) {
    float cnt = 0.f;
    for (int l = grp; l < nb; l += groups) {
      const float4 v = ld4(x + (int64_t)(r0 + l) * D + c);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      cnt += 1.f;
      const float inv = 1.f / cnt;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float d = vv[q] - mo[q].mean;
        mo[q].mean += d * inv;
        mo[q].m2 += d * (vv[q] - mo[q].mean);
        mn[q] = fminf(mn[q], vv[q]); mx[q] = fmaxf(mx[q], vv[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) mo[q].n = cnt;
    if (grp == 0 && nb < n) {                      // the padding nodes: (n - nb) copies of one row
      const float4 cv = ld4(x + pad_row * D + c);
      const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const Moments pad = {(float)(n - nb), cc[q], 0.f};
        mom_merge(mo[q], pad);
        mn[q] = fminf(mn[q], cc[q]); mx[q] = fmaxf(mx[q], cc[q]);
      }
    }
  }
  float* slot = sm + ((grp * lpr + sub) * 4) * 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float* s = slot + q * 5;
    s[0] = mo[q].n; s[1] = mo[q].mean; s[2] = mo[q].m2; s[3] = mn[q]; s[4] = mx[q];
  }
  for (int stride = groups / 2; stride > 0; stride >>= 1) {
    __syncthreads();
    if (grp < stride) {
      const float* o = sm + (((grp + stride) * lpr + sub) * 4) * 5;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* s = slot + q * 5; const float* t = o + q * 5;
        Moments a = {s[0], s[1], s[2]}; const Moments bb = {t[0], t[1], t[2]};
        mom_merge(a, bb); s[0] = a.n; s[1] = a.mean; s[2] = a.m2;
        s[3] = fminf(s[3], t[3]); s[4] = fmaxf(s[4], t[4]);
      }
    }
  }
  if (grp == 0 && active) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* s = slot + q * 5;
      float* o = stats + (int64_t)b * 4 * D;
      o[c + q] = s[1]; o[D + c + q] = s[3]; o[2 * D + c + q] = s[4];
      o[3 * D + c + q] = sqrtf(s[2] / (s[0] - 1.f));
    }
  }
}

// x [bs,n,n,64] -> stats [bs,256] as above, reading only the pairs whose two nodes are unmasked; see the kernel.
int mbvae_pool_stats_pairs_fwd(const float* x, float* stats, const uint8_t* mask, const int32_t* row_idx,
                               const int32_t* nvalid, int64_t bs, int64_t n, int64_t D, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !stats || !mask || !row_idx || !nvalid || bs <= 0 || n <= 0) return MBVAE_EINVAL;
  if (D != 64 || n > 1024 || bs > 0x7fffffff || bs * n * n > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  constexpr int NT = POOL_PAIRS_NT;
  const size_t smem = (size_t)NT * 4 * 5 * sizeof(float);
  if (!MBVAE_ALLOW_BIG_LDS(pool_pairs_fwd_kernel<NT>)) return MBVAE_ELAUNCH;
  hipLaunchKernelGGL(pool_pairs_fwd_kernel<NT>, dim3((unsigned)bs), dim3(NT), smem, stream, x, stats, mask, row_idx, nvalid,
                     (int)n, bs * n * n);
  return mbvae_launch_status();
}

int mbvae_pool_stats_nodes_fwd(const float* x, float* stats, const int32_t* node_off, int64_t pad_row, int64_t bs,
                               int64_t n, int64_t D, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !stats || !node_off || bs <= 0 || n <= 0 || D <= 0 || pad_row < 0) return MBVAE_EINVAL;
  if ((D & 3) || bs > 65535 || n > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int lpr = (int)(D / 4 < 64 ? D / 4 : 64);
  if ((lpr & (lpr - 1)) || (D > 256 && (D % 256))) return MBVAE_EUNSUPPORTED;
  hipLaunchKernelGGL(pool_nodes_fwd_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)bs), dim3(256), 0, stream, x, stats,
                     node_off, pad_row, (int)n, (int)D);
  return mbvae_launch_status();
}

int mbvae_pool_stats_bwd(const float* x, const float* stats, const int32_t* arg, const float* dstats, float* dx,
                         int64_t bs, int64_t L, int64_t D, int accumulate, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !stats || !arg || !dstats || !dx || bs <= 0 || L <= 0 || D <= 0) return MBVAE_EINVAL;
  if ((D & 3) || bs > 65535) return MBVAE_EUNSUPPORTED;
  int64_t blocks = (L * (D / 4) + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)blocks, (unsigned)bs), dim3(256), 0, stream, x, stats, (const int*)arg,
                     dstats, dx, (int)L, (int)D, accumulate);
  return mbvae_launch_status();
}

// x [bs,n,n,D] (+ optional skip s [bs,n,n,lds] using its first D features) -> out [bs,n,n,D]
int mbvae_sym_mask(const float* x, const float* s, int64_t lds, const uint8_t* mask, float* out, int64_t bs, int64_t n,
                   int64_t D, int zero_diag, hipStream_t stream) {
  (void)hipGetLastError();  // drop stale (non-sticky) errors left by other runtime users
  if (!x || !mask || !out || bs <= 0 || n <= 0 || D <= 0 || x == out) return MBVAE_EINVAL;
  if (bs * n > 0x7fffffff || n * D > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const bool vec = (D % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % 16 == 0);
  if (vec) hipLaunchKernelGGL(sym_mask_kernel<4>, dim3((unsigned)(bs * n)), dim3(256), 0, stream, x, s, (int)lds, mask, out,
                              (int)n, (int)D, zero_diag);
  else hipLaunchKernelGGL(sym_mask_kernel<1>, dim3((unsigned)(bs * n)), dim3(256), 0, stream, x, s, (int)lds, mask, out,
                          (int)n, (int)D, zero_diag);
  return mbvae_launch_status();
}

// out[b,i,:] = add[b,:] + (mul[b,:] + 1) * x[b,i,:]   (x, out [bs,n,D]; add, mul [bs,D]; D % 4 == 0)
int mbvae_film_fwd(const float* add, const float* mul, const float* x, float* out, int64_t bs, int64_t n, int64_t D,
                   hipStream_t stream) {
  (void)hipGetLastError();
  if (!add || !mul || !x || !out || bs <= 0 || n <= 0 || D <= 0) return MBVAE_EINVAL;
  if (D % 4 || (((uintptr_t)add | (uintptr_t)mul | (uintptr_t)x | (uintptr_t)out) & 15) || n * D > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  const int64_t total4 = bs * n * D / 4;
  hipLaunchKernelGGL(film_fwd_kernel, dim3((unsigned)std::min<int64_t>((total4 + 255) / 256, 8192)), dim3(256), 0, stream, add, mul, x,
                     out, total4, (int)(n * D / 4), (int)(D / 4));
  return mbvae_launch_status();
}

// dx = dout * (mul + 1);  dadd[b,:] = sum_i dout[b,i,:];  dmul[b,:] = sum_i dout[b,i,:] * x[b,i,:]
int mbvae_film_bwd(const float* dout, const float* mul, const float* x, float* dx, float* dadd, float* dmul, int64_t bs,
                   int64_t n, int64_t D, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dout || !mul || !x || !dx || !dadd || !dmul || bs <= 0 || n <= 0 || D <= 0) return MBVAE_EINVAL;
  if (bs > 0x7fffffff || n > 0x7fffffff || D > 0x7fffffff) return MBVAE_EUNSUPPORTED;
  hipLaunchKernelGGL(film_bwd_kernel, dim3((unsigned)bs), dim3(256), 0, stream, dout, mul, x, dx, dadd, dmul, (int)n, (int)D);
  return mbvae_launch_status();
}

// y = x + u * sigmoid(z) * keep * scale, element-wise over n floats (u may be x itself; keep null: no dropout)
int mbvae_gate_fwd(const float* x, const float* u, const float* z, const uint8_t* keep, float scale, float* y, int64_t n,
                   hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !u || !z || !y || n <= 0) return MBVAE_EINVAL;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, x, u, z, keep,
                     scale, y, n);
  return mbvae_launch_status();
}

// du = dy * g (u_is_x: dy + dy * g, the whole gradient of x), dz = dy * u * keep * scale * s * (1 - s); g, s from z and keep
int mbvae_gate_bwd(const float* dy, const float* u, const float* z, const uint8_t* keep, float scale, float* du, float* dz,
                   int64_t n, int u_is_x, hipStream_t stream) {
  (void)hipGetLastError();
  if (!dy || !u || !z || !du || !dz || n <= 0) return MBVAE_EINVAL;
  const dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 8192));
  if (u_is_x) hipLaunchKernelGGL(gate_bwd_kernel<true>, grid, dim3(256), 0, stream, dy, u, z, keep, scale, du, dz, n);
  else hipLaunchKernelGGL(gate_bwd_kernel<false>, grid, dim3(256), 0, stream, dy, u, z, keep, scale, du, dz, n);
  return mbvae_launch_status();
}

}  // extern "C"
