# A/B of the fused edge forward on the bf16 pipeline against the native fp32 MFMA form (GPU box only)
cd "$(dirname "$0")/.."
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "edge_block_fused or edge_attention or padding_shortcuts or packed" --timeout 120 --timeout-method thread 2>&1 | tail -3
echo "== split"; timeout -k 10 200 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline'].get('edge_fused_fwd_GBps'))"
echo "== fp32"; MBVAE_NO_SPLIT_MFMA=1 timeout -k 10 200 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline'].get('edge_fused_fwd_GBps'))"
