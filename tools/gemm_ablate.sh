for d in 0 1 2; do
  echo "qkvattn debug $d"
  SGCLIP_QKVATTN_DEBUG=$d M=262144 ONLY=qkvattn timeout -k 10 100 python tools/bench_gemm.py || exit 1
done
