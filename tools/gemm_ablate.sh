for t in 64 128 256; do echo "TN target $t"; SGCLIP_TN_WGS=$t timeout -k 10 100 python tools/bench_gemm_tn.py 2>&1 | grep -v amdgpu.ids || exit 1; done
echo "KD 12800 target 256"; KD=12800 timeout -k 10 100 python tools/bench_gemm_tn.py 2>&1 | grep -v amdgpu.ids
