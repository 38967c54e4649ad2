SGCLIP_GEMM_VARIANT=8 timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -m gpu -k "gemm" 2>&1 | tail -3 || exit 1
for v in 7 8; do
  echo "variant $v"
  SGCLIP_GEMM_VARIANT=$v M=262144 ONLY=qkv,oproj,fc1,fc2,sq4096 timeout -k 10 100 python tools/bench_gemm.py || exit 1
done
echo "variant 8 no epilogue"
SGCLIP_GEMM_VARIANT=8 SGCLIP_GEMM_DEBUG=1 M=262144 ONLY=qkv timeout -k 10 100 python tools/bench_gemm.py
