"""The allocation-free masked-CE entry point (mbvae_masked_ce_fwd_static) against mbvae_masked_ce_fwd, through the C ABI:
two calls in a row on the same GPU, the second with a smaller input (the library-owned scratch array keeps the first
call's contents), and the size limit."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(rows, C, seed):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(rows, C, generator=g) * 3).cuda()
    cls = torch.randint(0, C, (rows,), generator=g)
    target = (F.one_hot(cls, C).float() * (torch.rand(rows, 1, generator=g) < 0.8)).cuda()
    return logits, target


def test_masked_ce_static_scratch_matches_workspace_entry_point():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd import _lib, ops
    st = torch.cuda.current_stream().cuda_stream
    ws = ops.workspace(torch.device("cuda", torch.cuda.current_device()))
    for rows, C, seed in ((300_000, 5, 1), (777, 8, 2), (64 * 64 * 40, 5, 3), (1, 3, 4)):    # large, then smaller, ...
        logits, target = _case(rows, C, seed)
        want = torch.full((18,), -7.0, device="cuda")
        got = torch.full((18,), -9.0, device="cuda")
        _lib.call("mbvae_masked_ce_fwd", logits.data_ptr(), target.data_ptr(), want.data_ptr(), rows, C, ws.data_ptr(), ws.numel(), st)
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()          # torch's allocator ...
        free_before = torch.cuda.mem_get_info()[0]       # ... and the device itself: a hipMalloc inside the library shows here
        _lib.call("mbvae_masked_ce_fwd_static", logits.data_ptr(), target.data_ptr(), got.data_ptr(), rows, C, st)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.cuda.mem_get_info()[0] == free_before, "device memory changed across the allocation-free entry point"
        assert torch.equal(got, want), f"rows={rows}: {got.tolist()} vs {want.tolist()}"
        ref = F.cross_entropy(logits[target.sum(1) > 0].double().cpu(), target[target.sum(1) > 0].argmax(1).cpu(), reduction="sum")
        assert abs(float(got[0]) - float(ref)) <= 1e-5 * max(abs(float(ref)), 1e-30) and int(got[1]) == int((target.sum(1) > 0).sum())
    # beyond the stated maximum: refused, nothing launched
    logits, target = _case(16, 5, 9)
    out = torch.zeros(18, device="cuda")
    with pytest.raises(RuntimeError, match="unsupported"):
        _lib.call("mbvae_masked_ce_fwd_static", logits.data_ptr(), target.data_ptr(), out.data_ptr(), (1 << 27) + 1, 5, st)
    with pytest.raises(RuntimeError, match="unsupported"):
        _lib.call("mbvae_masked_ce_fwd_static", logits.data_ptr(), target.data_ptr(), out.data_ptr(), 16, 17, st)
    torch.cuda.synchronize()
