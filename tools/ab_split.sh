# A/B of the bf16-pipeline (split) kernels against the native fp32 MFMA kernels on the edge-path shapes (GPU box only)
cd "$(dirname "$0")/.."
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "test_linear_fwd_bwd or nan_poison or dropout_masks or edge_input_fanout" --timeout 120 --timeout-method thread 2>&1 | tail -3
for s in ${SHAPES:-"64 256" "256 64" "64 128" "128 64" "64 512"}; do
  echo "== split $s"; timeout -k 10 120 python tools/gemm_probe.py $s | grep -v amdgpu.ids
  echo "== fp32  $s"; MBVAE_NO_SPLIT_MFMA=1 timeout -k 10 120 python tools/gemm_probe.py $s | grep -v amdgpu.ids
done
