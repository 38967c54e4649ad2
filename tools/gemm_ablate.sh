for d in 0 1 9; do echo "debug $d"; SGCLIP_GEMM_DEBUG=$d M=262144 ONLY=qkv,oproj,fc2 timeout -k 10 100 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids || exit 1; done
