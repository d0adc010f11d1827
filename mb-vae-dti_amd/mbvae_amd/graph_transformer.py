"""Graph-transformer denoiser on the HIP kernels; drop-in for the reference module.

Same constructor, parameter names (state_dict keys) and forward contract as
mb_vae_dti/training/models/graph_transformer.py:70-343 (Xtoy, Etoy, XEyTransformerLayer, NodeEdgeBlock,
GraphTransformer).  Parameters live in ordinary nn.Linear / nn.LayerNorm containers and are read in place
by the kernels, so checkpoints load unchanged.  The reference's host-synchronising assert_* checks are dropped.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .containers import PlaceHolder


# Diagnostic (tools/critical_lane_probe.py): ("E" | "N", cycles) puts a one-thread spin kernel of that many clock cycles into
# every layer's forward on the edge-tensor stream or on the node / global stream: the lane whose delay shows up in the
# step time is the one the step waits for.  None in normal use.
LANE_DELAY = None

class _SeedStream:
    """Per-forward stream of dropout seeds derived from torch's CPU generator (no device sync)."""

    def __init__(self, active):
        self.base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if active else 0
        self.k = 0

    def next(self):
        self.k += 1
        return self.base * 4096 + self.k


def _lin(layer, x, **kw):
    return ops.linear(x, layer.weight, layer.bias, **kw)


class Xtoy(nn.Module):
    def __init__(self, dx, dy):
        super().__init__()
        self.lin = nn.Linear(4 * dx, dy)

    def forward(self, X):
        return _lin(self.lin, ops.pool_stats(X))


class Etoy(nn.Module):
    def __init__(self, d, dy):
        super().__init__()
        self.lin = nn.Linear(4 * d, dy)

    def forward(self, E):
        return _lin(self.lin, ops.pool_stats(E))


_NO_STACK_CACHE = os.environ.get("MBVAE_NO_STACK_CACHE", "0") not in ("", "0")


class _StackedParams(torch.autograd.Function):
    """(stacked weight, stacked bias) of several Linear layers as a function of their parameters: the forward hands out the
    up-to-date copies it is given (no kernel), the backward returns row slices of the stacked gradients."""

    @staticmethod
    def forward(ctx, w, b, *params):
        ctx.rows = [p.shape[0] for p in params[0::2]]
        return w.view_as(w), b.view_as(b)

    @staticmethod
    def backward(ctx, dw, db):
        out = [None, None]
        r0 = 0
        for i, rows in enumerate(ctx.rows):
            out.append(dw[r0:r0 + rows] if (dw is not None and ctx.needs_input_grad[2 + 2 * i]) else None)
            out.append(db[r0:r0 + rows] if (db is not None and ctx.needs_input_grad[3 + 2 * i]) else None)
            r0 += rows
        return tuple(out)


def refresh_stacked_weights(model):
    """Renews every stacked weight / bias copy of the model's attention blocks from the layers' parameters, all of them in
    one multi-tensor copy (call it on the stream that ran the optimiser step, right after it).  Blocks whose copies do
    not exist yet are left to build them at their next forward."""
    dst, src, hits = [], [], []
    for m in model.modules():
        cache = m.__dict__.get("_stack_cache") if isinstance(m, NodeEdgeBlock) else None
        for hit in (cache or {}).values():
            r0 = 0
            for l in hit[3]:
                rows = l.weight.shape[0]
                dst += [hit[1][r0:r0 + rows], hit[2][r0:r0 + rows]]; src += [l.weight, l.bias]
                r0 += rows
            hits.append(hit)
    if dst:
        with torch.no_grad():
            torch._foreach_copy_(dst, [p.detach() for p in src])
        for hit in hits:
            hit[0] = tuple((p.data_ptr(), p._version) for l in hit[3] for p in (l.weight, l.bias))


class NodeEdgeBlock(nn.Module):
    """Self-attention block that also updates the edge and global representations."""

    def __init__(self, dx, de, dy, n_head, **kwargs):
        super().__init__()
        assert dx % n_head == 0, f"dx: {dx} -- nhead: {n_head}"
        self.dx, self.de, self.dy, self.df, self.n_head = dx, de, dy, dx // n_head, n_head
        self.q, self.k, self.v = nn.Linear(dx, dx), nn.Linear(dx, dx), nn.Linear(dx, dx)
        self.e_add, self.e_mul = nn.Linear(de, dx), nn.Linear(de, dx)
        self.y_e_mul, self.y_e_add = nn.Linear(dy, dx), nn.Linear(dy, dx)
        self.y_x_mul, self.y_x_add = nn.Linear(dy, dx), nn.Linear(dy, dx)
        self.y_y = nn.Linear(dy, dy)
        self.x_y, self.e_y = Xtoy(dx, dy), Etoy(de, dy)
        self.x_out, self.e_out = nn.Linear(dx, dx), nn.Linear(dx, de)
        self.y_out = nn.Sequential(nn.Linear(dy, dy), nn.ReLU(), nn.Linear(dy, dy))

    def _stacked(self, key, layers):
        """Weights and biases of several Linear layers side by side.  The copy is kept until a parameter changes
        (in-place updates and load_state_dict bump the tensors' version counters; .data swaps change data_ptr).  With
        gradients it enters autograd through _StackedParams, whose backward hands each layer its slice of the stacked
        gradient; refresh_stacked_weights() renews all copies of a model in a few launches after an optimiser step, so
        that no concatenation sits in front of the products."""
        if _NO_STACK_CACHE and torch.is_grad_enabled():      # A/B aid: concatenate at every call, as autograd nodes
            return torch.cat([l.weight for l in layers], 0), torch.cat([l.bias for l in layers], 0)
        params = [p for l in layers for p in (l.weight, l.bias)]
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.setdefault("_stack_cache", {})
        hit = cache.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                if hit is None:
                    hit = [stamp, torch.cat([l.weight for l in layers], 0), torch.cat([l.bias for l in layers], 0), layers]
                else:                       # renew in place: the buffers may be referenced by a refresh list
                    torch.cat([l.weight for l in layers], 0, out=hit[1]); torch.cat([l.bias for l in layers], 0, out=hit[2])
                    hit[0] = stamp
            cache[key] = hit
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _StackedParams.apply(hit[1], hit[2], *params)
        return hit[1], hit[2]

    def _fused(self, key, x, layers, **kw):
        """One product for several Linear layers that read the same input: outputs side by side in the last dim."""
        w, b = self._stacked(key, layers)
        return ops.linear(x, w, b, **kw)

    def forward(self, X, E, y, node_mask, return_edge_alias=False, edge_norm=None, tile_map=None, order=None, pack=None,
                edge_norm_train=None, side=None):
        """(newX, newE, new_y) as in the reference; with return_edge_alias also an alias of E that the caller should
        use for the residual branch, so that all gradients of E meet inside one autograd node (ops.edge_input).
        pack (ops.NodePack, inference only): X and newX are packed node rows [rows, dx] instead of [bs, n, dx]."""
        n = E.shape[1]
        dx = self.dx
        node = dict(mask=node_mask, mask_mode=ops.MASK_NODE, n_nodes=n)
        if pack is not None:            # every packed row is a valid node except the one padding row
            node = dict(mask=pack.keep, mask_mode=ops.MASK_NODE, n_nodes=pack.keep.shape[1])
        edge = dict(mask=node_mask, mask_mode=ops.MASK_EDGE, n_nodes=n)
        if side is not None:
            # the two small projections that open the block also run on the second stream: in the forward pass they are
            # issued behind the previous layer's node / global work (done long before the main stream gets here), in
            # the backward pass their gradient kernels run next to the edge projection's
            main = torch.cuda.current_stream()
            if not (getattr(X, "_mbvae_stream", None) is side and getattr(y, "_mbvae_stream", None) is side):
                side.wait_stream(main)               # X, y were not produced on the second stream
                X.record_stream(side); y.record_stream(side)
            with torch.cuda.stream(side):
                QKV = self._fused("qkv", X, (self.q, self.k, self.v), **node)
                yp = self._fused("y5", y, (self.y_e_add, self.y_e_mul, self.y_x_add, self.y_x_mul, self.y_y))
            main.wait_stream(side)
            QKV.record_stream(main); yp.record_stream(main)
        else:
            QKV = self._fused("qkv", X, (self.q, self.k, self.v), **node)            # [bs,n,3dx]
            yp = self._fused("y5", y, (self.y_e_add, self.y_e_mul, self.y_x_add, self.y_x_mul, self.y_y))
        w12, b12 = self._stacked("e12", (self.e_mul, self.e_add))
        ye_add, ye_mul, yx_add, yx_mul, yy = yp.split((dx, dx, dx, dx, self.dy), dim=-1)
        # e_mul|e_add projection + attention (+ Etoy statistics and the residual alias of E) as one node
        P, wV, e_stats, E_alias = ops.edge_block(E, w12, b12, QKV, ye_add, ye_mul, node_mask, self.n_head,
                                                 order=order if edge_norm is not None else None, pack=pack)
        def node_and_global_updates():
            if pack is not None:
                film = torch.addcmul(yx_add[pack.graph], yx_mul[pack.graph] + 1, wV)
            elif torch.is_grad_enabled():
                film = (ops.film(yx_add, yx_mul, wV) if dx % 4 == 0 and not ops.option_set("NO_FILM")
                        else yx_add.unsqueeze(1) + (yx_mul.unsqueeze(1) + 1) * wV)
            else:                                  # same arithmetic in two element-wise kernels instead of three
                film = torch.addcmul(yx_add.unsqueeze(1), yx_mul.unsqueeze(1) + 1, wV)
            nX = _lin(self.x_out, film, **node)
            x_stats = _lin(self.x_y.lin, ops.pool_stats_nodes(X, pack)) if pack is not None else self.x_y(X)
            ny = yy + x_stats + _lin(self.e_y.lin, e_stats)
            return nX, _lin(self.y_out[2], _lin(self.y_out[0], ny, act=1))

        if side is not None:
            # second stream, forked BEFORE the long e_out kernel is issued: these small kernels only need the attention
            # block's outputs; the caller joins the streams at the end of the layer.  Tensors of this stream that the other
            # one reads are marked for the allocator.
            side.wait_stream(torch.cuda.current_stream())
            for t in (X, yp, wV, e_stats):
                t.record_stream(side)
            with torch.cuda.stream(side):
                newX, new_y = node_and_global_updates()
        if edge_norm is not None:       # inference: e_out, the residual add and the layer's first edge LayerNorm in one pass
            ln, eps = edge_norm
            newE = ops.edge_out_ln_inference(P, self.e_out.weight, self.e_out.bias, E, node_mask, n, ln.weight, ln.bias, eps,
                                             tile_map=tile_map, order=order,
                                             uniform_padding=tile_map is not None or order is not None)
        else:
            if not P.requires_grad:         # exactly when the no-gradient attention forward ran: it skips padding pairs
                ops.zero_masked_pairs_(P, node_mask)
            if edge_norm_train is not None and ops.linear_ln_supported(P, self.e_out.weight, E_alias):
                # training: e_out with the layer's first edge LayerNorm (residual + dropout) in its epilogue; the caller
                # sees the flag on the block and skips its own normE1
                ln, eps, p, seed = edge_norm_train
                newE = ops.linear_ln(P, self.e_out.weight, self.e_out.bias, E_alias, ln.weight, ln.bias, eps, p, seed, **edge)
                self.__dict__["_edge_normed"] = True
            else:
                newE = _lin(self.e_out, P, **edge)
                self.__dict__["_edge_normed"] = False
        if side is None:
            newX, new_y = node_and_global_updates()
        return (newX, newE, new_y, E_alias) if return_edge_alias else (newX, newE, new_y)


class XEyTransformerLayer(nn.Module):
    """Updates node, edge and global features (attention block + three residual feed-forward blocks)."""

    def __init__(self, dx, de, dy, n_head, dim_ffX=2048, dim_ffE=128, dim_ffy=2048, dropout=0.1,
                 layer_norm_eps=1e-5, device=None, dtype=None):
        super().__init__()
        kw = {"device": device, "dtype": dtype}
        self.self_attn = NodeEdgeBlock(dx, de, dy, n_head)
        self.linX1, self.linX2 = nn.Linear(dx, dim_ffX, **kw), nn.Linear(dim_ffX, dx, **kw)
        self.normX1, self.normX2 = nn.LayerNorm(dx, eps=layer_norm_eps, **kw), nn.LayerNorm(dx, eps=layer_norm_eps, **kw)
        self.linE1, self.linE2 = nn.Linear(de, dim_ffE, **kw), nn.Linear(dim_ffE, de, **kw)
        self.normE1, self.normE2 = nn.LayerNorm(de, eps=layer_norm_eps, **kw), nn.LayerNorm(de, eps=layer_norm_eps, **kw)
        self.lin_y1, self.lin_y2 = nn.Linear(dy, dim_ffy, **kw), nn.Linear(dim_ffy, dy, **kw)
        self.norm_y1, self.norm_y2 = nn.LayerNorm(dy, eps=layer_norm_eps, **kw), nn.LayerNorm(dy, eps=layer_norm_eps, **kw)
        self.dropout_p = dropout
        self.eps = layer_norm_eps

    def forward(self, X, E, y, node_mask, seeds=None, tile_map=None, order=None, pack=None):
        """tile_map / order (ops.edge_tile_map / ops.edge_row_order of node_mask, inference only): let the fused edge
        kernels skip or pack away the padding rows.  Both rely on every padding pair row of E holding the same values,
        which holds for E as GraphTransformer feeds it (padding rows enter as zeros and only pass through row-wise
        operations)."""
        p = self.dropout_p if self.training else 0.0
        seeds = seeds or _SeedStream(p > 0)
        inference = p == 0.0 and not torch.is_grad_enabled()      # sampling / validation
        # dropout seeds in the order the three first norms use them (X, y, E), drawn before the attention block because
        # the edge norm may run inside it
        sX1, sy1, sE1 = seeds.next(), seeds.next(), seeds.next()
        # training: the node / global remainder of the layer on a second stream next to the long edge-tensor kernels
        side = ops.side_stream(X.device) if (torch.is_grad_enabled() and not inference and X.is_cuda and pack is None) else None
        if side is not None:
            y.record_stream(side)
        newX, newE, new_y, E = self.self_attn(X, E, y, node_mask, return_edge_alias=True,
                                              edge_norm=(self.normE1, self.eps) if inference else None,
                                              tile_map=tile_map if inference else None,
                                              order=order if inference else None,
                                              pack=pack if inference else None,
                                              edge_norm_train=None if inference else (self.normE1, self.eps, p, sE1),
                                              side=side)

        def norm(ln, x, r, seed=None):
            return ops.ln_residual(x, r, ln.weight, ln.bias, self.eps, p, seeds.next() if seed is None else seed)

        # (inference, or training with the fused epilogue: newE already is normE1(E + dropout(e_out(.))))
        def edge_norm1():
            return newE if (inference or self.self_attn.__dict__.get("_edge_normed")) else norm(self.normE1, E, newE, sE1)

        def ff_block(ln, l1, l2, x, s_ffn, s_norm):
            """norm(x + dropout(l2(dropout(relu(l1(x)))))); the residual branch uses l1's alias of x so that both
            gradients of x meet inside l1's input-gradient product.  Where the second Linear takes the LayerNorm into
            its epilogue (the edge tensor) ffn returns the normalised result itself."""
            out, x = ops.ffn(x, l1.weight, l1.bias, l2.weight, l2.bias, drop_p=p, seed=s_ffn,
                             ln=(ln.weight, ln.bias, self.eps, p, s_norm))
            return out if x is None else norm(ln, x, out, s_norm)

        sXf, sEf, syf = [(seeds.next(), seeds.next()) for _ in range(3)]      # seeds in the order X, E, y use them
        if side is not None:
            if LANE_DELAY is not None and LANE_DELAY[0] == "E":
                torch.cuda._sleep(LANE_DELAY[1])
            with torch.cuda.stream(side):
                if LANE_DELAY is not None and LANE_DELAY[0] == "N":
                    torch.cuda._sleep(LANE_DELAY[1])
                X, y = norm(self.normX1, X, newX, sX1), norm(self.norm_y1, y, new_y, sy1)
                X = ff_block(self.normX2, self.linX1, self.linX2, X, *sXf)
                y = ff_block(self.norm_y2, self.lin_y1, self.lin_y2, y, *syf)
            E = ff_block(self.normE2, self.linE1, self.linE2, edge_norm1(), *sEf)
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            X.record_stream(main); y.record_stream(main)
            X._mbvae_stream = y._mbvae_stream = side      # (the next layer's opening projections need not wait for `main`)
            return X, E, y
        X, y = norm(self.normX1, X, newX, sX1), norm(self.norm_y1, y, new_y, sy1)
        E = edge_norm1()
        X = ff_block(self.normX2, self.linX1, self.linX2, X, *sXf)
        if inference:                                     # the edge feed-forward block in one pass
            E = ops.ffn_ln_inference(E, self.linE1.weight, self.linE1.bias, self.linE2.weight, self.linE2.bias,
                                     self.normE2.weight, self.normE2.bias, self.eps, tile_map=tile_map, order=order)
        else:
            E = ff_block(self.normE2, self.linE1, self.linE2, E, *sEf)
        y = ff_block(self.norm_y2, self.lin_y1, self.lin_y2, y, *syf)
        return X, E, y


class GraphTransformer(nn.Module):
    """n_layers x XEyTransformerLayer between input and output MLPs; returns a masked PlaceHolder(X, E, y)."""

    def __init__(self, n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, act_fn_in=None, act_fn_out=None):
        super().__init__()
        for fn in (act_fn_in, act_fn_out):
            if fn is not None and not isinstance(fn, nn.ReLU):
                raise ValueError("only ReLU activations are implemented in the HIP path (the reference uses ReLU)")
        self.n_layers = n_layers
        self.out_dim_X, self.out_dim_E, self.out_dim_y = output_dims["X"], output_dims["E"], output_dims["y"]
        h, hm = hidden_dims, hidden_mlp_dims
        relu = nn.ReLU
        self.mlp_in_X = nn.Sequential(nn.Linear(input_dims["X"], hm["X"]), relu(), nn.Linear(hm["X"], h["dx"]), relu())
        self.mlp_in_E = nn.Sequential(nn.Linear(input_dims["E"], hm["E"]), relu(), nn.Linear(hm["E"], h["de"]), relu())
        self.mlp_in_y = nn.Sequential(nn.Linear(input_dims["y"], hm["y"]), relu(), nn.Linear(hm["y"], h["dy"]), relu())
        self.tf_layers = nn.ModuleList([
            XEyTransformerLayer(dx=h["dx"], de=h["de"], dy=h["dy"], n_head=h["n_head"], dim_ffX=h["dim_ffX"],
                                dim_ffE=h["dim_ffE"], dim_ffy=h["dim_ffy"]) for _ in range(n_layers)])
        self.mlp_out_X = nn.Sequential(nn.Linear(h["dx"], hm["X"]), relu(), nn.Linear(hm["X"], output_dims["X"]))
        self.mlp_out_E = nn.Sequential(nn.Linear(h["de"], hm["E"]), relu(), nn.Linear(hm["E"], output_dims["E"]))
        # the widths the fused inference edge kernels (and with them the packed row order) are written for
        self._fused_edge_widths = (h["dx"], h["de"], h["dim_ffE"]) == (256, 64, 128)
        self.mlp_out_y = nn.Sequential(nn.Linear(h["dy"], hm["y"]), relu(), nn.Linear(hm["y"], output_dims["y"]))

    def _padding_shortcuts(self, node_mask):
        """(tile map, row order) of a node mask.  The tile map is cheap and built on first sight; a mask that comes back
        (reverse diffusion passes the same mask tensor 500 times) also gets the packed row order, which costs a sort.
        The entry keeps a reference to the mask tensor, so its address cannot be handed to another tensor, and the
        version counter catches in-place changes."""
        hit = self.__dict__.get("_pad_cache")
        if hit is not None and hit["mask"] is node_mask and hit["version"] == node_mask._version:
            # (never while a HIP graph is being captured: building the node pack synchronises with the host)
            if (hit["order"] is None and self._fused_edge_widths and os.environ.get("MBVAE_NO_ROW_ORDER") is None
                    and not torch.cuda.is_current_stream_capturing()):
                # The sparse form leaves most padding rows of the edge tensor unwritten; only the fused attention
                # kernel (at most 64 nodes per graph) knows to skip them.  Larger graphs take the dense order, which
                # writes every row.
                fused_attention = node_mask.shape[1] <= 64
                hit["order"] = ops.edge_row_order(node_mask, sparse=fused_attention)
                if hit["order"] is not None and fused_attention and os.environ.get("MBVAE_NO_NODE_PACK") is None:
                    hit["pack"] = ops.NodePack.build(node_mask)      # (one host synchronisation)
                    if hit["pack"] is not None:                      # the valid pair rows, for the edge in / out MLPs
                        hit["pack"].pair_rows = hit["order"].idx[:int(hit["order"].nvalid[0])].long()
            return (None, hit["order"]) if hit["order"] is not None else (hit["tile_map"], None)
        hit = {"mask": node_mask, "version": node_mask._version, "tile_map": ops.edge_tile_map(node_mask), "order": None,
               "pack": None}
        self.__dict__["_pad_cache"] = hit
        return hit["tile_map"], None

    def prepare_padding_shortcuts(self, node_mask):
        """Builds the packed row order of `node_mask` right away (callers that know the same mask tensor is about to be
        used many times, e.g. the reverse-diffusion loop; must run outside HIP graph capture)."""
        self.__dict__["_pad_cache"] = {"mask": node_mask, "version": node_mask._version,
                                       "tile_map": ops.edge_tile_map(node_mask), "order": None, "pack": None}
        return self._padding_shortcuts(node_mask)

    def forward(self, X, E, y, node_mask):
        n = X.shape[1]
        X, E, y = X.float(), E.float(), y.float()
        xm = node_mask.unsqueeze(-1).to(X.dtype)
        seeds = _SeedStream(self.training)

        def mlp(seq, t, last_act, **kw):
            return _lin(seq[2], _lin(seq[0], t, act=1), act=1 if last_act else 0, **kw)

        # ragged batch without gradients: the fused inference kernels skip the padding rows of the edge tensor, and
        # the node path runs on the valid nodes only (plus one padding node that stands for all the others)
        tile_map = order = pack = None
        if not torch.is_grad_enabled() and not self.training:
            tile_map, order = self._padding_shortcuts(node_mask)
            if order is not None and order.sparse and self.n_layers > 0:
                pack = self._pad_cache.get("pack")
        if pack is not None:            # the edge input MLP on the valid pairs only (sym_mask zeroes the others anyway)
            Eflat = E.reshape(-1, E.shape[-1])
            hv = mlp(self.mlp_in_E, Eflat[pack.pair_rows], True)
            hE = hv.new_zeros((Eflat.shape[0], hv.shape[-1])).index_copy_(0, pack.pair_rows, hv)
            hE = ops.sym_mask(hE.reshape(*E.shape[:-1], -1), node_mask)
        # training with a second stream: the node / global input and output MLPs (small) next to the edge ones (long)
        side = ops.side_stream(X.device) if (torch.is_grad_enabled() and self.training and X.is_cuda and pack is None) else None
        main = torch.cuda.current_stream() if side is not None else None
        if side is not None:
            side.wait_stream(main)
            for t in (X, y, xm):
                t.record_stream(side)
            with torch.cuda.stream(side):
                hX = mlp(self.mlp_in_X, X, True, mask=node_mask, mask_mode=ops.MASK_NODE, n_nodes=n)
                hy = mlp(self.mlp_in_y, y, True)
            hE = ops.sym_mask(mlp(self.mlp_in_E, E, True), node_mask)
            main.wait_stream(side)
            hX.record_stream(main); hy.record_stream(main)
            hX._mbvae_stream = hy._mbvae_stream = side
        elif pack is None:
            hE = ops.sym_mask(mlp(self.mlp_in_E, E, True), node_mask)
        if side is not None:
            pass
        elif pack is not None:
            Xflat = X.reshape(-1, X.shape[-1])
            hX = mlp(self.mlp_in_X, Xflat[pack.rows], True, mask=pack.keep, mask_mode=ops.MASK_NODE, n_nodes=pack.keep.shape[1])
        else:
            hX = mlp(self.mlp_in_X, X, True, mask=node_mask, mask_mode=ops.MASK_NODE, n_nodes=n)
        if side is None:
            hy = mlp(self.mlp_in_y, y, True)
        for layer in self.tf_layers:
            hX, hE, hy = layer(hX, hE, hy, node_mask, seeds, tile_map=tile_map, order=order, pack=pack)
        if pack is not None:                         # the edge output MLP on the valid pairs only, likewise
            ov = mlp(self.mlp_out_E, hE.reshape(-1, hE.shape[-1])[pack.pair_rows], False)
            oE = ov.new_zeros((Eflat.shape[0], ov.shape[-1])).index_copy_(0, pack.pair_rows, ov)
            oE = ops.sym_mask(oE.reshape(*E.shape[:-1], -1), node_mask, skip=E, zero_diag=True)
        elif order is not None and order.sparse and self.n_layers > 0:
            ops.zero_masked_pairs_(hE, node_mask)    # the sparse order left most padding rows of hE unwritten
        if pack is not None:                         # back to [bs, n, .]: padding nodes are zero in the output
            valid = pack.rows[:pack.n_valid]
            oXp = mlp(self.mlp_out_X, hX, False)[:pack.n_valid] + Xflat[valid, :self.out_dim_X]
            oX = oXp.new_zeros((Xflat.shape[0], self.out_dim_X)).index_copy_(0, valid, oXp).reshape(*X.shape[:-1], -1)
        elif side is not None:                       # (hX, hy were produced on the second stream: no wait needed)
            with torch.cuda.stream(side):
                oX = (mlp(self.mlp_out_X, hX, False) + X[..., :self.out_dim_X]) * xm
                oy = mlp(self.mlp_out_y, hy, False) + y[..., :self.out_dim_y]
            oE = ops.sym_mask(mlp(self.mlp_out_E, hE, False), node_mask, skip=E, zero_diag=True)
            main.wait_stream(side)
            oX.record_stream(main); oy.record_stream(main)
            return PlaceHolder(X=oX, E=oE, y=oy)
        else:
            oX = (mlp(self.mlp_out_X, hX, False) + X[..., :self.out_dim_X]) * xm
        if pack is None:
            oE = ops.sym_mask(mlp(self.mlp_out_E, hE, False), node_mask, skip=E, zero_diag=True)
        oy = mlp(self.mlp_out_y, hy, False) + y[..., :self.out_dim_y]
        return PlaceHolder(X=oX, E=oE, y=oy)
