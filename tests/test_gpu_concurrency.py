"""Short kernels must return the same bits whether or not long kernels of ANOTHER stream share the GPU with them.

A library built with packed-fp32 instructions failed exactly this: the node-path LayerNorm backward (256 features, one
wave per row) returned a few wrong rows per call whenever tall-skinny product kernels (mbvae_linear_fwd on the pair rows,
its weight gradient, the Linear + LayerNorm forward) were running on the chip, and never on an idle one - which is what
made two-stream training runs differ in the last bits (mb-vae-dti_amd/build.py records the build flag that ended it).
Victims run on a second stream while the partner is issued in a loop on the first; every result is compared bit for bit
with the idle-GPU result of the same call.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

BS, N, DX, DE = 128, 64, 256, 64
REPS = 30


def _setup():
    from mbvae_amd import _lib, ops as O
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    r_ = lambda *s: torch.randn(*s, generator=g).to(dev)
    d = dict(x=r_(BS, N, DX), r=r_(BS, N, DX), dy=r_(BS, N, DX) * torch.rand(BS, N, 1, generator=g).to(dev), gamma=r_(DX), beta=r_(DX),
             mul=r_(BS, DX), y=r_(BS, 1024), ry=r_(BS, 1024), dyy=r_(BS, 1024), g1k=r_(1024), b1k=r_(1024),
             zE=r_(BS, N, N, DE), dyE=r_(BS, N, N, DE), g64=r_(DE), b64=r_(DE), w1=r_(128, DE) * 0.1, b1=r_(128), w2=r_(DE, 128) * 0.1, bo=r_(DE))
    d["mean"], d["rstd"], d["yln"] = torch.empty(BS * N, device=dev), torch.empty(BS * N, device=dev), torch.empty_like(d["x"])
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("mbvae_ln_residual_fwd", d["x"].data_ptr(), d["r"].data_ptr(), d["gamma"].data_ptr(), d["beta"].data_ptr(), d["yln"].data_ptr(),
              d["mean"].data_ptr(), d["rstd"].data_ptr(), BS * N, DX, 1e-5, 0.1, 1234, st)
    d["mean_y"], d["rstd_y"], d["yln_y"] = torch.empty(BS, device=dev), torch.empty(BS, device=dev), torch.empty_like(d["y"])
    _lib.call("mbvae_ln_residual_fwd", d["y"].data_ptr(), d["ry"].data_ptr(), d["g1k"].data_ptr(), d["b1k"].data_ptr(), d["yln_y"].data_ptr(),
              d["mean_y"].data_ptr(), d["rstd_y"].data_ptr(), BS, 1024, 1e-5, 0.1, 77, st)
    d["h128"] = torch.empty(BS, N, N, 128, device=dev)
    d["oE1"], d["oE2"] = torch.empty_like(d["zE"]), torch.empty_like(d["zE"])
    d["dw"], d["db"] = torch.empty(128, DE, device=dev), torch.empty(128, device=dev)
    return _lib, O, dev, d


def _victims(_lib, O, dev, d):
    cur = lambda: torch.cuda.current_stream().cuda_stream

    def ln_bwd(x, r, gamma, mean, rstd, dy, rows, D, seed):
        def run():
            o = [torch.empty_like(x), torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)]
            ws = O.workspace(dev)
            _lib.call("mbvae_ln_residual_bwd", x.data_ptr(), r.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
                      o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), rows, D, 0.1, seed, ws.data_ptr(), ws.numel(), cur())
            return o
        return run

    def ln_fwd():
        o = [torch.empty_like(d["x"]), torch.empty_like(d["mean"]), torch.empty_like(d["mean"])]
        _lib.call("mbvae_ln_residual_fwd", d["x"].data_ptr(), d["r"].data_ptr(), d["gamma"].data_ptr(), d["beta"].data_ptr(), o[0].data_ptr(),
                  o[1].data_ptr(), o[2].data_ptr(), BS * N, DX, 1e-5, 0.1, 1234, cur())
        return o

    def film_bwd():
        o = [torch.empty_like(d["x"]), torch.empty_like(d["mul"]), torch.empty_like(d["mul"])]
        _lib.call("mbvae_film_bwd", d["dy"].data_ptr(), d["mul"].data_ptr(), d["x"].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                  BS, N, DX, cur())
        return o

    def pool():
        with torch.no_grad():
            return [O.pool_stats(d["x"])]

    return {"LayerNorm backward, 256 features": ln_bwd(d["x"], d["r"], d["gamma"], d["mean"], d["rstd"], d["dy"], BS * N, DX, 1234),
            "LayerNorm backward, 1024 features": ln_bwd(d["y"], d["ry"], d["g1k"], d["mean_y"], d["rstd_y"], d["dyy"], BS, 1024, 77),
            "LayerNorm forward, 256 features": ln_fwd, "FiLM backward": film_bwd, "node pooling": pool}


def _partners(_lib, O, dev, d):
    M = BS * N * N
    cur = lambda: torch.cuda.current_stream().cuda_stream

    def linear_fwd():
        ws = O.workspace(dev)
        _lib.call("mbvae_linear_fwd", d["zE"].data_ptr(), DE, d["w1"].data_ptr(), d["b1"].data_ptr(), d["h128"].data_ptr(), 128, M, DE, 128, 1,
                  0, 0, 0, 0.1, 5, 0, ws.data_ptr(), ws.numel(), cur())

    def wgrad():
        ws = O.workspace(dev)
        _lib.call("mbvae_linear_bwd_weight", d["h128"].data_ptr(), 128, d["zE"].data_ptr(), DE, d["dw"].data_ptr(), d["db"].data_ptr(), M, DE, 128,
                  0, 1.0, 0, 0, 0, 0, ws.data_ptr(), ws.numel(), cur())

    def linear_ln():
        _lib.call("mbvae_linear_ln_fwd", d["h128"].data_ptr(), 128, d["w2"].data_ptr(), d["bo"].data_ptr(), d["zE"].data_ptr(), d["g64"].data_ptr(),
                  d["b64"].data_ptr(), d["oE1"].data_ptr(), d["oE2"].data_ptr(), M, 128, DE, 0, 0, 0, 1e-5, 0.1, 6, cur())

    return {"Linear 64->128 over the pair rows": linear_fwd, "its weight gradient": wgrad, "Linear + LayerNorm forward": linear_ln}


def test_short_kernels_are_bit_stable_next_to_long_kernels_of_another_stream():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib, O, dev, d = _setup()
    d["h128"].normal_()
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    bad = []
    for vname, victim in _victims(_lib, O, dev, d).items():
        ref = victim()
        torch.cuda.synchronize()
        for pname, partner in _partners(_lib, O, dev, d).items():
            side.wait_stream(main)
            results = []
            for _ in range(REPS):
                partner()
                with torch.cuda.stream(side):
                    results.append(victim())
            torch.cuda.synchronize()
            n_bad = sum(any(not torch.equal(a, b) for a, b in zip(ref, res)) for res in results)
            if n_bad:
                bad.append(f"{vname} next to {pname}: {n_bad} of {REPS} results differ")
    assert not bad, "; ".join(bad)
