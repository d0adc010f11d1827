/* libmbvae_hip.so -- C ABI of the MI355X (gfx950) kernels behind the MB-VAE-DTI hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (fp32 unless noted; masks are uint8, class ids int64);
 *     tensors are dense row-major with the feature dimension contiguous;
 *   - nothing is allocated, freed or synchronised: outputs and workspaces are caller-owned, work is
 *     enqueued on `stream` and the call returns immediately;
 *   - return value: 0 ok, -1 invalid argument, -2 unsupported shape, -3 launch failure; no exceptions;
 *   - threads: entry points may be called from several host threads at once as long as concurrent calls do not share
 *     a workspace, an output or a work counter (calls on ONE stream are serialised by the stream; the Python host keeps
 *     one workspace per stream).  Not safe to overlap on one GPU, and said so where they are declared: the entry points
 *     that hand out tasks through a caller-owned device counter (work_counter arguments) when they share that counter,
 *     and mbvae_masked_ce_fwd_static, whose partial sums live in one static device array per GPU.  Process-wide the
 *     library holds (a) tables filled once and read-only afterwards (compute-unit count and kernel LDS limits per
 *     device) and (b) the tuning switches of mbvae_set_option below (relaxed atomics).
 * The reference has no native interface; each entry point replaces the sequence of PyTorch ops cited
 * next to it (paths relative to the reference repository root).
 */
#ifndef MBVAE_HIP_H
#define MBVAE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mbvae_stream_t; /* == hipStream_t */

#define MBVAE_MASK_NONE 0
#define MBVAE_MASK_NODE 1 /* row r = node (b,i):   keep iff mask[r] */
#define MBVAE_MASK_EDGE 2 /* row r = pair (b,i,j): keep iff mask[b*n+i] && mask[b*n+j] */

/* ---- tuning switches ------------------------------------------------------------------------------------
 * No counterpart in the reference.  Process-wide switches that choose between equivalent HIP routes (kernel A/B
 * timing, parity tests of the fallback kernels; never a CPU path).  Each starts from the environment variable
 * MBVAE_<NAME>, read once.  The 26 names (0 / 1 unless a range is given):
 *   kernel families   NO_TSGEMM, NO_TSGEMM2 (tall-skinny products over the pair rows), TSGEMM_N64, NO_TSWGRAD, NO_TSWGRAD2
 *                     (their weight gradients), NO_SMALLM (few-tile products of the global-feature path), NO_COLGROUPS
 *                     (node-path column groups), NO_THIN (the 5-wide ends of the edge MLPs), GEMM_TILE (-1 automatic,
 *                     0 = 64x64, 1 = 128x64, 2 = 128x128, 3 = 32x128, 4 = 64x128 tiles of the generic kernel)
 *   arithmetic        NO_SPLIT_MFMA (all products on the native fp32 matrix instructions instead of six bf16 ones), NO_GEMM_SPLIT
 *                     (the same choice for the generic LDS-tiled kernel only)
 *   attention block   NO_EDGE_FUSED, NO_EDGE_TRAIN_FUSED, NO_E12_RECOMPUTE (store e_mul|e_add instead of recomputing it in
 *                     the backward), NO_EF_COOP / NO_EFB_COOP (operand tile split per wave instead of once per
 *                     workgroup, forward / backward), EF_STAGGER (0..3: how the waves of a workgroup are phased),
 *                     EFB_SYNC (rendezvous period of the recomputing backward in query rows, power of two; 0 = 8),
 *                     EA_KEYS (8, else 16 keys per wave in the storing attention backward), NO_EA_PREFETCH
 *   fusions           NO_EDGE_FFN (inference feed-forward block), NO_EDGE_FFN_TRAIN (its training form), NO_LN_EPILOGUE
 *                     (Linear + LayerNorm), NO_BWD_BOTH (both gradients of e_mul|e_add from one pass), NO_FFN_HANDOVER
 *                     and NO_FILM (read by the host side)
 * mbvae_set_option: 0, or -1 for an unknown name.  mbvae_get_option: the value (-1 for an unknown name; GEMM_TILE's
 * "automatic" is -1 as well).  A change takes effect at the next entry-point call. */
int mbvae_set_option(const char* name, int value);
int mbvae_get_option(const char* name);

/* ---- Linear layers on the matrix cores ------------------------------------------------------------------
 * fp32 in, fp32 out, fp32 accumulation.  Products over the pair rows ([bs*n*n, d] shapes) and all other shapes made of
 * whole aligned tiles run on the bf16 matrix instructions with every fp32 operand cut exactly into three bf16 words (six
 * products per fp32 product, error as the native fp32 instruction's); ragged or unaligned shapes and the few-tile
 * products of the global-feature path run on the native fp32 matrix instruction.
 * Replaces nn.Linear forward/backward plus the `* x_mask`, `* e_mask1 * e_mask2`, ReLU and Dropout that
 * surround it in mb_vae_dti/training/models/graph_transformer.py:201-202,217-221,233,240,257 (NodeEdgeBlock),
 * :134-144 (feed-forward blocks), :283-308,321-335 (in/out MLPs) and the encoder / head MLPs of
 * mb_vae_dti/training/models/blocks.py:36-66,87-129 and heads.py:31-41,206-207.
 *   y[M,N] = rowmask( dropout_p( act( x[M,K] @ w[N,K]^T + bias[N] ) ) )      act: 0 none, 1 ReLU
 * workspace (optional, may be NULL) enables split-K for small M*N; any size is accepted. */
int mbvae_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                     int64_t M, int64_t K, int64_t N, int act, const uint8_t* mask, int mask_mode, int n_nodes,
                     float drop_p, uint64_t seed, int accumulate, void* workspace, int64_t ws_bytes,
                     mbvae_stream_t stream);
/* dx[M,K] (+)= pro(dy)[M,N] @ w[N,K], pro(dy) = (gate > 0 ? dy : 0) * gate_scale * rowmask   (gate may be NULL) */
int mbvae_linear_bwd_input(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx, int64_t M,
                           int64_t K, int64_t N, const float* gate, float gate_scale, const uint8_t* mask,
                           int mask_mode, int n_nodes, int accumulate, void* workspace, int64_t ws_bytes,
                           mbvae_stream_t stream);
/* dx[M,K] = pro(dy)[M,N] @ w[N,K] + addend[M,K]: the input gradient of a Linear layer whose input also feeds other
 * consumers (graph_transformer.py:114-146, 187-267: the edge tensor E goes to e_mul/e_add, Etoy and the residual),
 * with the other consumers' gradient folded in instead of a separate accumulation pass.  The [bs*n*n, d] shapes
 * (M >= 32768, M % 32 == 0, N % 64 == 0, K % 64 == 0, K*N <= 32768) add it inside the wave-autonomous kernel, any other
 * shape of at least 1024 rows (the node tensor X, which goes to q|k|v, Xtoy and the residual) in the tiled kernel's
 * epilogue; MBVAE_EUNSUPPORTED below 1024 rows (the caller then adds it in a separate pass). */
int mbvae_linear_bwd_input_add(const float* dy, int64_t lddy, const float* w, const float* addend, float* dx,
                               int64_t lddx, int64_t M, int64_t K, int64_t N, const float* gate, float gate_scale,
                               const uint8_t* mask, int mask_mode, int n_nodes, mbvae_stream_t stream);
/* y[M,64] = LayerNorm(z) * gamma + beta with z[M,64] = resid + dropout_p(rowmask(x[M,K] @ w[64,K]^T + bias)): a Linear with a
 * 64-wide output and the residual LayerNorm that consumes it in ONE kernel - `E = normE1(E + dropoutE1(newE))` after
 * e_out (graph_transformer.py:123-130 with the masked e_out of :255-258) and `E = normE2(E + dropoutE3(linE2(.)))`
 * (graph_transformer.py:139-146).  z is kept for mbvae_ln_sum_bwd; the Linear's own gradients then come from the usual
 * mbvae_linear_bwd_* calls on that kernel's dr.  Dropout mask: that of mbvae_ln_residual_fwd for (seed, element index).
 * Only N == 64, K % 64 == 0, K <= 384, M >= 8192, M % 32 == 0; MBVAE_EUNSUPPORTED otherwise (run the two separately). */
int mbvae_linear_ln_fwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* resid,
                        const float* gamma, const float* beta, float* y, float* z, int64_t M, int64_t K, int64_t N,
                        const uint8_t* mask, int mask_mode, int n_nodes, float eps, float drop_p, uint64_t seed,
                        mbvae_stream_t stream);
/* The same with x stored as bf16: e_out on the attention output P of the bf16 storage mode (see the *_bf16 entry points). */
int mbvae_linear_ln_fwd_bf16(const uint16_t* x, int64_t ldx, const float* w, const float* bias, const float* resid,
                             const float* gamma, const float* beta, float* y, float* z, int64_t M, int64_t K, int64_t N,
                             const uint8_t* mask, int mask_mode, int n_nodes, float eps, float drop_p, uint64_t seed,
                             mbvae_stream_t stream);
/* dx[M,K] = out_gate[M,K] > 0 ? (rowmask(dy)[M,N] @ w[N,K]) * out_gate_scale : 0.  out_gate is this layer's INPUT when that
 * input came out of ReLU (+ dropout) - linE2 / linX2 / lin_y2 of the feed-forward blocks, graph_transformer.py:139-146 -:
 * the gate of the layer below is applied where the gradient is produced, and that layer's gradient calls take gate = NULL. */
int mbvae_linear_bwd_input_outgate(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx, int64_t M, int64_t K,
                                   int64_t N, const float* out_gate, float out_gate_scale, const uint8_t* mask, int mask_mode,
                                   int n_nodes, void* workspace, int64_t ws_bytes, mbvae_stream_t stream);
/* dw[N,K] (+)= pro(dy)[M,N]^T @ x[M,K];  db[N] = column sums of pro(dy) (db may be NULL) */
int mbvae_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db,
                            int64_t M, int64_t K, int64_t N, const float* gate, float gate_scale,
                            const uint8_t* mask, int mask_mode, int n_nodes, int accumulate, void* workspace,
                            int64_t ws_bytes, mbvae_stream_t stream);

/* Both gradients of one Linear layer from a single pass over dy (the backward of e_mul | e_add,
 * graph_transformer.py:196-204, whose dy rows are the largest stream of a train step):
 *   dx[M,K] = (addend +) rowmask(dy)[M,N] @ w[N,K];  dw[N,K] = rowmask(dy)^T @ x[M,K];  db[N] = column sums (may be NULL).
 * Supported: K = 64, N = 512, M % 64 == 0, M >= 32768; otherwise -2 and the caller issues mbvae_linear_bwd_input(_add) and
 * mbvae_linear_bwd_weight.  workspace: at least CUs * (N*K + N) * 4 bytes.  Bitwise reproducible. */
int mbvae_linear_bwd_both(const float* dy, int64_t lddy, const float* w, const float* x, int64_t ldx, const float* addend,
                          float* dx, int64_t lddx, float* dw, float* db, int64_t M, int64_t K, int64_t N,
                          const uint8_t* mask, int mask_mode, int n_nodes, void* workspace, int64_t ws_bytes,
                          mbvae_stream_t stream);

/* Node update modulated by the global features (graph_transformer.py:254: newX = yx1 + (yx2 + 1) * weighted_V):
 * out[b,i,:] = add[b,:] + (mul[b,:] + 1) * x[b,i,:];  x, out [bs,n,D], add, mul [bs,D], D % 4 == 0, 16-byte aligned.
 * Backward: dx = dout * (mul + 1), dadd[b,:] = sum_i dout[b,i,:], dmul[b,:] = sum_i dout[b,i,:] * x[b,i,:] (node order,
 * bitwise reproducible). */
int mbvae_film_fwd(const float* add, const float* mul, const float* x, float* out, int64_t bs, int64_t n, int64_t D,
                   mbvae_stream_t stream);
int mbvae_film_bwd(const float* dout, const float* mul, const float* x, float* dx, float* dadd, float* dmul, int64_t bs,
                   int64_t n, int64_t D, mbvae_stream_t stream);

/* Sigmoid-gate update of the feature encoders and of the drug / target fusion (models/blocks.py:125 `x + x * att(x)`,
 * :318 `a + gate_a * align_proj(b)`, :321-322): y = x + u * g with g = sigmoid(z) * keep * scale, element-wise over n
 * floats; z is the output of the gate's second Linear, keep the dropout keep mask of the gate (nn.Dropout after the
 * Sigmoid, :92-99: one byte per element, NULL = no dropout, scale = 1 / (1 - p)), u may be x itself.
 * Backward: g is recomputed from z and keep; du = dy * g, or with u_is_x the whole gradient of x, dy + dy * g;
 * dz = dy * u * keep * scale * s * (1 - s), s = sigmoid(z).  The gradient of x in the other case is dy itself. */
int mbvae_gate_fwd(const float* x, const float* u, const float* z, const uint8_t* keep, float scale, float* y, int64_t n,
                   mbvae_stream_t stream);
int mbvae_gate_bwd(const float* dy, const float* u, const float* z, const uint8_t* keep, float scale, float* du, float* dz,
                   int64_t n, int u_is_x, mbvae_stream_t stream);

/* ---- Edge-feature attention ------------------------------------------------------------------------
 * Replaces graph_transformer.py:206-249 (Y = Q*K/sqrt(df); Y*(E1+1)+E2; FiLM by y; per-channel masked
 * softmax over keys, masked_softmax :22-27; weighted V).  Q,K,V [bs,n,dx] with rows ldq floats apart and
 * E1,E2 [bs,n,n,dx] with rows lde apart, already masked (the strides let them be column blocks of one fused
 * q|k|v resp. e_mul|e_add projection output); ye_add,ye_mul [bs,dx]; mask [bs,n].  Dense outputs: P [bs,n,n,dx]
 * is the input of e_out, wV [bs,n,dx], lse [bs,n,dx] (log-sum-exp per query row and channel, kept for backward).
 * Backward writes dE1,dE2 / dQ,dK,dV with the same strides as their forward counterparts.
 * Forward: dx, ldq, lde % 4 == 0. */
int mbvae_edge_attn_fwd(const float* Q, const float* K, const float* V, int64_t ldq, const float* E1, const float* E2,
                        int64_t lde, const float* ye_add, const float* ye_mul, const uint8_t* mask, float* P,
                        float* wV, float* lse, int64_t bs, int64_t n, int64_t dx, float scale, mbvae_stream_t stream);
/* Backward workspace: the per-key-group partial sums of dQ / dye_add / dye_mul (16 keys per group), added in a fixed
 * order by a second kernel, so the gradients are bitwise reproducible.  mbvae_edge_attn_bwd_workspace bytes run the
 * whole batch in one launch; with less the call walks over the batch in chunks of graphs (-1 if not even one graph
 * fits; nothing is needed when n <= 16). */
int64_t mbvae_edge_attn_bwd_workspace(int64_t bs, int64_t n, int64_t dx);
int mbvae_edge_attn_bwd(const float* Q, const float* K, const float* V, int64_t ldq, const float* E1, const float* E2,
                        int64_t lde, const float* ye_mul, const uint8_t* mask, const float* wV, const float* lse,
                        const float* dP, const float* dwV, float* dE1, float* dE2, float* dQ, float* dK, float* dV,
                        float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t dx, float scale,
                        void* workspace, int64_t ws_bytes, mbvae_stream_t stream);

/* ---- e_mul|e_add projection + edge attention in one forward kernel --------------------------------------
 * Replaces graph_transformer.py:206-249 together with the e_mul / e_add Linear layers of :196-204 that feed it:
 * E1|E2 = mask(E @ W12^T + b12) is computed on the matrix cores inside the attention kernel instead of making a
 * round trip through memory.  E [bs,n,n,de] dense; W12 [2dx,de] = e_mul.weight over e_add.weight, b12 [2dx] or
 * NULL; the other arguments as for mbvae_edge_attn_fwd.  E12 [bs,n,n,2dx] receives the projection output when it
 * is not NULL (the backward entry points read it); with NULL nothing of that size is written, P is only written
 * for valid (query, key) pairs (the rest of the buffer keeps whatever it held) and tiles without a valid pair cost
 * nothing: the inference form, whose P feeds mbvae_edge_out_ln_fwd.
 * work_counter (NULL, or one int32 in device memory that the call zeroes on `stream` and then owns until the kernel
 * ends): the workgroups draw their (graph, block of query rows) tasks from it first come first served, which evens
 * out ragged batches; with NULL the tasks are dealt round robin.  Calls that overlap on one device need separate
 * counters.  The results do not depend on the order.
 * node_off (NULL, or int32 [bs+1]; inference form only): packed node rows.  Q, K, V, wV and lse then hold only the
 * valid nodes, graph b's nodes 0 .. nb-1 at rows node_off[b] .. node_off[b+1]-1 (the mask must keep exactly the first
 * nb nodes of graph b); rows of masked nodes do not exist and are neither read nor written.  E and P stay dense.
 * graph_order (NULL, or int32 [bs], a permutation of 0..bs-1): the order in which the graphs' tasks are handed out;
 * largest graphs first evens out a ragged batch without reordering it.  The results do not depend on it.
 * MBVAE_EUNSUPPORTED unless de == 64, dx == 256 and n <= 64: callers then use the two separate entry points. */
int mbvae_edge_fused_fwd(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                         const float* V, int64_t ldq, const float* ye_add, const float* ye_mul, const uint8_t* mask,
                         float* P, float* wV, float* lse, float* E12, int32_t* work_counter, const int32_t* node_off,
                         const int32_t* graph_order, int64_t bs, int64_t n, int64_t de, int64_t dx, float scale,
                         mbvae_stream_t stream);

/* ---- edge feed-forward block, inference form ----------------------------------------------------------
 * Replaces `normE2(E + linE2(relu(linE1(E))))` of XEyTransformerLayer.forward (graph_transformer.py:139-142) when
 * no gradient is needed and dropout is off (sampling, validation): one pass over the rows, nothing but x is read
 * and nothing but out is written.  x, out [M,d]; W1 [dff,d], b1 [dff], W2 [d,dff], b2 [d] (torch Linear layouts,
 * biases may be NULL); gamma, beta [d].  MBVAE_EUNSUPPORTED unless d == 64, dff == 128: callers then
 * use mbvae_linear_fwd twice and mbvae_ln_residual_fwd.
 * tile_ok (NULL, or ceil(M/32) bytes): a 0 marks a tile of 32 consecutive rows that are ALL padding (pairs with a masked
 * node).  The caller guarantees that every row of every such tile holds the same values -- true inside the denoiser,
 * where padding pairs enter as zeros and only pass through row-wise operations -- so the kernel evaluates one of them
 * per wave and stores that result to the others without reading them.  The output is bit-identical to tile_ok = NULL. */
int mbvae_edge_ffn_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* gamma, const float* beta, float* out, const uint8_t* tile_ok,
                       const int32_t* row_idx, const int32_t* nvalid, int64_t M, int64_t d, int64_t dff, float eps,
                       mbvae_stream_t stream);

/* Training form of the same block (graph_transformer.py:139-142 with dropoutE2 / dropoutE3 active), one pass over the rows:
 *   hidden [M,dff] = dropout_p1(relu(x W1^T + b1)),  z [M,d] = x + dropout_p2(hidden W2^T + b2),  out = LayerNorm(z) * gamma + beta.
 * hidden and z are what the backward reads (mbvae_ln_sum_bwd on z, then mbvae_linear_bwd_input_outgate / _bwd_weight /
 * _bwd_input_add on hidden and x): dropout keep-scales are hashed from (seed, row * width + column) exactly as
 * mbvae_linear_fwd (seed1, width dff) and mbvae_linear_ln_fwd (seed2, width d) hash them.  Per row 256 B are read and
 * 1 KB written; the two separate kernels read 1 KB and write 1 KB.  MBVAE_EUNSUPPORTED unless d == 64, dff == 128, M a
 * multiple of 32 and M * dff < 2^32: callers then use mbvae_linear_fwd and mbvae_linear_ln_fwd. */
int mbvae_edge_ffn_fwd_train(const float* x, const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* gamma, const float* beta, float* hidden, float* z, float* out, int64_t M,
                             int64_t d, int64_t dff, float eps, float p1, uint64_t seed1, float p2, uint64_t seed2,
                             mbvae_stream_t stream);

/* Backward data path of the block (autograd of graph_transformer.py:139-142 with respect to the activations), one pass:
 *   dh [M,dff] = (h > 0 ? dr W2 : 0) * gate_scale,   dx [M,d] = dh W1 + dres
 * dr / dres [M,d]: the gradient entering linE2's output and the residual branch's (what mbvae_ln_sum_bwd returns; the same
 * pointer when dropoutE3 is off); h: the hidden activation stored by mbvae_edge_ffn_fwd_train, whose sign pattern is the
 * ReLU-and-dropout gate (gate_scale = 1 / (1 - p1)).  The weight gradients come from mbvae_linear_bwd_weight on (dr, h) and
 * (dh, x).  MBVAE_EUNSUPPORTED unless d == 64, dff == 128, M a multiple of 32: callers then use
 * mbvae_linear_bwd_input_outgate and mbvae_linear_bwd_input_add. */
int mbvae_edge_ffn_bwd_data(const float* dr, const float* dres, const float* h, const float* W1, const float* W2,
                            float* dh, float* dx, int64_t M, int64_t d, int64_t dff, float gate_scale, mbvae_stream_t stream);

/* Inference form of `normE1(E + e_out(newE))` (graph_transformer.py:251-252 followed by :133-134): out [M,de] =
 * LayerNorm(E + keep(row) * (P W^T + b)) * gamma + beta, P [M,dx], W [de,dx] (torch Linear layout), b [de] or NULL,
 * keep(row) = edge mask of row (b,i,j) from mask [bs,n] with M = bs*n*n (NULL: all rows kept).
 * MBVAE_EUNSUPPORTED unless dx == 256, de == 64: callers then use mbvae_linear_fwd + mbvae_ln_residual_fwd.
 * P is only read where keep(row) holds: rows of masked pairs may be uninitialised (mbvae_edge_fused_fwd without E12
 * does not write them).  tile_ok (NULL, or ceil(M/32) bytes): a 0 marks a tile of 32 consecutive rows without any kept
 * row; its P rows are not even fetched.  uniform_padding != 0 adds the caller's guarantee that every row of E in such
 * tiles holds the same values (as for mbvae_edge_ffn_fwd): one of them is normalised per wave and the result stored
 * to the others without reading them.  The output is bit-identical to tile_ok = NULL either way. */
int mbvae_edge_out_ln_fwd(const float* P, const float* W, const float* b, const float* E, const uint8_t* mask,
                          int64_t n, const float* gamma, const float* beta, float* out, const uint8_t* tile_ok,
                          const int32_t* row_idx, const int32_t* nvalid, int uniform_padding, int64_t M, int64_t dx,
                          int64_t de, float eps, mbvae_stream_t stream);

/* Packed row order, the alternative to tile_ok for both entry points above (pass tile_ok = NULL with it; M >= 32):
 * nvalid (two device int32, so that no host synchronisation is needed to build them): nvalid[0] = number of valid
 * rows; nvalid[1] != 0 = evaluate and store only the FIRST padding tile (rows row_idx[32*ceil(nvalid[0]/32) ..+31],
 * which carry the padding rows' common value from one kernel to the next) and leave every other padding row
 * untouched -- for callers that read no other padding row (mbvae_pool_stats_pairs_fwd does not);
 * row_idx (int32, at least M + 64 entries) = the valid rows in ascending order, padded to a multiple of 32 by
 * repeating the last one, followed by the padding rows, padded likewise.  The kernels then form their 32-row tiles
 * from row_idx, so products are only spent on valid rows; the padding rows are treated as under tile_ok = 0.
 * mbvae_amd.ops.edge_row_order(node_mask) builds both.  Same bits as the natural order. */

/* ---- residual + dropout + LayerNorm ------------------------------------------------------------------
 * Replaces `norm(X + dropout(newX))` of graph_transformer.py:125-144 (and nn.LayerNorm elsewhere with r = NULL). */
int mbvae_ln_residual_fwd(const float* x, const float* r, const float* gamma, const float* beta, float* y,
                          float* mean, float* rstd, int64_t rows, int64_t D, float eps, float drop_p,
                          uint64_t seed, mbvae_stream_t stream);
int64_t mbvae_ln_residual_bwd_workspace(int64_t D);
int mbvae_ln_residual_bwd(const float* x, const float* r, const float* gamma, const float* mean, const float* rstd,
                          const float* dy, float* dx, float* dr, float* dgamma, float* dbeta, int64_t rows,
                          int64_t D, float drop_p, uint64_t seed, void* workspace, int64_t ws_bytes,
                          mbvae_stream_t stream);

/* Backward of the LayerNorm inside mbvae_linear_ln_fwd (torch.nn.LayerNorm backward of graph_transformer.py:130,146 plus
 * the dropout of :128,144): z is the stored pre-LayerNorm sum; dx = gradient of the residual operand, dr = dx * dropout
 * scale = gradient of the Linear's output (null: not wanted), dgamma, dbeta.  Row statistics are recomputed from z. */
int mbvae_ln_sum_bwd(const float* z, const float* gamma, const float* dy, float* dx, float* dr, float* dgamma,
                     float* dbeta, int64_t rows, int64_t D, float eps, float drop_p, uint64_t seed, void* workspace,
                     int64_t ws_bytes, mbvae_stream_t stream);

/* ---- X->y / E->y pooling ---------------------------------------------------------------------------
 * Replaces Xtoy / Etoy statistics, graph_transformer.py:40-46,57-65: x [bs,L,D] -> [bs,4D] = mean|min|max|
 * unbiased std over L (L = n or n*n; padded entries included).  arg [bs,2,D] int32 = first argmin/argmax. */
int mbvae_pool_stats_fwd(const float* x, float* stats, int32_t* arg, int64_t bs, int64_t L, int64_t D,
                         mbvae_stream_t stream);
int mbvae_pool_stats_bwd(const float* x, const float* stats, const int32_t* arg, const float* dstats, float* dx,
                         int64_t bs, int64_t L, int64_t D, int accumulate, mbvae_stream_t stream);
/* Forward-only form of the Etoy statistics (graph_transformer.py:57-65) for ragged batches in the packed row order of
 * mbvae_edge_ffn_fwd: x [bs,n,n,D], mask [bs,n].  Only pairs of two unmasked nodes are read; the padding pairs, which
 * all hold the row x[row_idx[32*ceil(nvalid/32)]] by the caller's guarantee (and need not be written anywhere else),
 * enter as `count x that row`.  Same mathematics as mbvae_pool_stats_fwd over all n*n rows, different rounding.
 * MBVAE_EUNSUPPORTED unless D == 64. */
/* Forward-only Xtoy statistics (graph_transformer.py:40-46) over packed node rows: x [rows,D] holds graph b's valid
 * nodes at rows node_off[b] .. node_off[b+1]-1; its n - nb padding nodes all hold the row x[pad_row] and enter as
 * `count x that row`.  stats [bs,4D].  D % 256 == 0 or D in {64, 128}. */
int mbvae_pool_stats_nodes_fwd(const float* x, float* stats, const int32_t* node_off, int64_t pad_row, int64_t bs,
                               int64_t n, int64_t D, mbvae_stream_t stream);
int mbvae_pool_stats_pairs_fwd(const float* x, float* stats, const uint8_t* mask, const int32_t* row_idx,
                               const int32_t* nvalid, int64_t bs, int64_t n, int64_t D, mbvae_stream_t stream);

/* ---- symmetrise + mask -----------------------------------------------------------------------------
 * Replaces graph_transformer.py:322,327 (input) and :338,341,343 (output):
 * out[b,i,j,:] = 0.5*(x[b,i,j,:] + x[b,j,i,:] (+ s[b,i,j,:D] + s[b,j,i,:D])) * pairmask * (zero_diag ? i!=j : 1).
 * The op is self-adjoint, so the same call (s = NULL) maps output gradients to input gradients. */
int mbvae_sym_mask(const float* x, const float* s, int64_t lds, const uint8_t* mask, float* out, int64_t bs,
                   int64_t n, int64_t D, int zero_diag, mbvae_stream_t stream);

/* ---- masked cross-entropy + per-class BCE trackers ---------------------------------------------------
 * Replaces ReconstructionHead.forward, mb_vae_dti/training/models/heads.py:270-296, and CEPerClass.update,
 * mb_vae_dti/training/metrics/molecular_metrics.py:77-94, in one pass over the logits.
 * out[0] = sum CE over kept rows, out[1] = #kept rows, out[2+c] = sum BCEWithLogits(logit_c, target_c). */
int64_t mbvae_masked_ce_workspace(void);
int mbvae_masked_ce_fwd(const float* logits, const float* target, float* out, int64_t rows, int64_t C,
                        void* workspace, int64_t ws_bytes, mbvae_stream_t stream);
/* mbvae_masked_ce_fwd for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc): no workspace argument, no allocation of any kind inside.  No counterpart in the reference.
 * The partial sums live in a 72 KB array inside the library (one copy per GPU, held until the process ends).  Same
 * result, bit for bit, as mbvae_masked_ce_fwd.  rows <= 2^27, else -2 (and a message on stderr).  Calls that overlap
 * on one GPU share the array and race: issue them on one stream, or serialise them. */
int mbvae_masked_ce_fwd_static(const float* logits, const float* target, float* out, int64_t rows, int64_t C,
                               mbvae_stream_t stream);
int mbvae_masked_ce_bwd(const float* logits, const float* target, const float* coef, float* dlogits, int64_t rows,
                        int64_t C, mbvae_stream_t stream);

/* ---- reparameterisation + KL -------------------------------------------------------------------------
 * Replaces KLVariationalHead.reparameterize / kl_divergence, heads.py:209-228 (eps supplied by the caller). */
int64_t mbvae_kl_workspace(void);
int mbvae_kl_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, float* kl, int64_t numel,
                         void* workspace, int64_t ws_bytes, mbvae_stream_t stream);
int mbvae_kl_reparam_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, const float* dkl,
                         float* dmu, float* dlogvar, int64_t numel, mbvae_stream_t stream);

/* ---- discrete diffusion ------------------------------------------------------------------------------
 * mbvae_apply_noise replaces FullDTIModel.apply_noise's probability construction and sampling,
 * mb_vae_dti/training/modules/full.py:485-500 + mb_vae_dti/training/diffusion/utils.py:124-162
 * (get_Qt_bar: diffusion/discrete_noise.py:82-99).  expX [bs*n,cx] / expE [bs*n*n,ce] are Exp(1) draws.
 * mbvae_reverse_sample replaces full.py:1120-1171 after the network call
 * (compute_batched_over0_posterior_distribution: diffusion/utils.py:319-349). */
int mbvae_apply_noise(const float* X0, const float* E0, const uint8_t* mask, const float* abar_t, const float* mx,
                      const float* me, const float* expX, const float* expE, int64_t* Xc, int64_t* Ec, float* Xt,
                      float* Et, int64_t bs, int64_t n, int64_t cx, int64_t ce, mbvae_stream_t stream);
int mbvae_reverse_sample(const float* logitsX, const float* logitsE, const float* Xt, const float* Et,
                         const uint8_t* mask, const float* beta_t, const float* abar_s, const float* abar_t,
                         const float* mx, const float* me, const float* expX, const float* expE, int64_t* Xc,
                         int64_t* Ec, float* Xs, float* Es, float* probX, float* probE, int64_t bs, int64_t n,
                         int64_t cx, int64_t ce, mbvae_stream_t stream);

/* ---- fused edge block, training form with the projection recomputed in the backward ------------------------
 * Same mathematics as mbvae_edge_fused_fwd + mbvae_edge_attn_bwd (graph_transformer.py:206-267 and its autograd
 * backward); the pair exists to keep E1|E2 [bs,n,n,2dx] out of memory: the forward writes P (for every pair), wV and
 * lse only, the backward forms E1|E2 again from E on the matrix cores.  de == 64, dx == 256, n <= 64; the forward
 * returns MBVAE_EUNSUPPORTED for batches of fewer than one graph per two compute units (the backward runs one
 * workgroup per graph) - use the storing pair then.  Bitwise reproducible. */
int mbvae_edge_fused_fwd_recompute(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                                   const float* V, int64_t ldq, const float* ye_add, const float* ye_mul,
                                   const uint8_t* mask, float* P, float* wV, float* lse, int64_t bs, int64_t n,
                                   int64_t de, int64_t dx, float scale, mbvae_stream_t stream);
int mbvae_edge_fused_bwd(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                         const float* V, int64_t ldq, const float* ye_mul, const uint8_t* mask, const float* wV,
                         const float* lse, const float* dP, const float* dwV, float* dE12, float* dQ, float* dK,
                         float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t de, int64_t dx,
                         float scale, mbvae_stream_t stream);

/* ---- opt-in bf16 STORAGE of the edge block's large activations ------------------------------------------
 * No counterpart in the reference (which trains in fp32 throughout; its own mixed-precision switch is Lightning's
 * `precision` flag, mb_vae_dti/training/run.py:642).  The projection output E1|E2 [bs,n,n,2dx], the attention output
 * P [bs,n,n,dx] (graph_transformer.py:206-267) and the gradients of both are bf16 IN MEMORY (uint16_t = bf16 bits,
 * round to nearest even on store); weights, biases, every other tensor, all arithmetic and all sums stay fp32.
 * Same argument meaning as the fp32 entry points they mirror; edge-path shapes only (MBVAE_EUNSUPPORTED otherwise).
 * Parity bound of the mode: 2e-2 against the fp32 path. */
int mbvae_edge_fused_fwd_bf16(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                              const float* V, int64_t ldq, const float* ye_add, const float* ye_mul,
                              const uint8_t* mask, uint16_t* P, float* wV, float* lse, uint16_t* E12, int64_t bs,
                              int64_t n, int64_t de, int64_t dx, float scale, mbvae_stream_t stream);
int mbvae_edge_attn_bwd_bf16(const float* Q, const float* K, const float* V, int64_t ldq, const uint16_t* E1,
                             const uint16_t* E2, int64_t lde, const float* ye_mul, const uint8_t* mask,
                             const float* wV, const float* lse, const uint16_t* dP, const float* dwV, uint16_t* dE1,
                             uint16_t* dE2, float* dQ, float* dK, float* dV, float* dye_add, float* dye_mul,
                             int64_t bs, int64_t n, int64_t dx, float scale, void* workspace, int64_t ws_bytes,
                             mbvae_stream_t stream);
/* the recomputing pair in this mode: P, dP and dE1|dE2 are bf16, E1|E2 is never stored */
int mbvae_edge_fused_fwd_recompute_bf16(const float* E, const float* W12, const float* b12, const float* Q,
                                        const float* K, const float* V, int64_t ldq, const float* ye_add,
                                        const float* ye_mul, const uint8_t* mask, uint16_t* P, float* wV, float* lse,
                                        int64_t bs, int64_t n, int64_t de, int64_t dx, float scale,
                                        mbvae_stream_t stream);
int mbvae_edge_fused_bwd_bf16(const float* E, const float* W12, const float* b12, const float* Q, const float* K,
                              const float* V, int64_t ldq, const float* ye_mul, const uint8_t* mask, const float* wV,
                              const float* lse, const uint16_t* dP, const float* dwV, uint16_t* dE12, float* dQ,
                              float* dK, float* dV, float* dye_add, float* dye_mul, int64_t bs, int64_t n, int64_t de,
                              int64_t dx, float scale, mbvae_stream_t stream);
/* y (fp32) = rowmask(x (bf16) @ w^T + bias): the e_out projection reading P */
int mbvae_linear_fwd_bf16(const uint16_t* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                          int64_t M, int64_t K, int64_t N, const uint8_t* mask, int mask_mode, int n_nodes,
                          mbvae_stream_t stream);
/* dx = rowmask(dy) @ w (+ addend, fp32, may be NULL): exactly one of dy / dx is bf16 (dy_bf16 / dx_bf16 != 0) */
int mbvae_linear_bwd_input_bf16(const void* dy, int dy_bf16, int64_t lddy, const float* w, const float* addend,
                                void* dx, int dx_bf16, int64_t lddx, int64_t M, int64_t K, int64_t N,
                                const uint8_t* mask, int mask_mode, int n_nodes, mbvae_stream_t stream);
/* dw = rowmask(dy)^T @ x, db = column sums of rowmask(dy): exactly one of dy / x is bf16 */
int mbvae_linear_bwd_weight_bf16(const void* dy, int dy_bf16, int64_t lddy, const void* x, int x_bf16, int64_t ldx,
                                 float* dw, float* db, int64_t M, int64_t K, int64_t N, const uint8_t* mask,
                                 int mask_mode, int n_nodes, void* workspace, int64_t ws_bytes,
                                 mbvae_stream_t stream);

/* ---- class-id wire format ------------------------------------------------------------------------------
 * Dense clean graphs from int8 class ids (atoms [bs,n], < 0 = padding; bonds [bs,n,n], 0 = none): the output
 * convention of to_dense + encode_no_edge (mb_vae_dti/training/diffusion/utils.py:90-120, called from
 * modules/utils.py:391-426): padded nodes are all-zero rows of X, the diagonal of E is all-zero, every other
 * pair without a bond - padded pairs included - is class 0, nothing is masked.  Lets a data loader ship
 * bs*n*(n+1) bytes instead of the one-hot floats. */
int mbvae_dense_from_ids(const int8_t* atoms, const int8_t* bonds, float* X, float* E, uint8_t* mask, int64_t bs,
                         int64_t n, int64_t cx, int64_t ce, mbvae_stream_t stream);

/* ---- k-cycle counts (n <= 64) ----------------------------------------------------------------------------
 * Replaces KNodeCycles.k_cycles (k3/k4/k5/k6_cycle), mb_vae_dti/training/diffusion/augmentation.py:177-250: five
 * batched matrix products and ~30 element-wise launches become one kernel.  A [bs,n,n] adjacency ->
 * xc [bs,n,3] (3-, 4-, 5-cycles through each node), yc [bs,4] (3-, 4-, 5-, 6-cycles per graph), before the
 * reference's /10 and clamp.  MBVAE_EUNSUPPORTED for n > 64. */
int mbvae_cycle_counts(const float* A, float* xc, float* yc, int64_t bs, int64_t n, mbvae_stream_t stream);

/* ---- batched symmetric eigendecomposition (n <= 64) ----------------------------------------------------
 * Replaces torch.linalg.eigh on the graph Laplacians, mb_vae_dti/training/diffusion/augmentation.py:68.
 * A [bs,n,n] symmetric -> W [bs,n] ascending, V [bs,n,n] eigenvectors in columns (signs / degenerate
 * subspaces are a valid but not LAPACK-identical choice). */
int mbvae_sym_eig(const float* A, float* W, float* V, int64_t bs, int64_t n, mbvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
