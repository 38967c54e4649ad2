# A/B of the tower GEMM shapes between the working tree's library and laser_amd/libsgclip_alt.so (tools/build_alt.sh),
# interleaved, on the GPU box from the repo root.  M=262144 is the towers' chunk size.
export M=${M:-262144}
for i in 1 2; do
  echo "== new"; timeout -k 10 200 python tools/bench_gemm.py || exit 1
  echo "== alt"; SGCLIP_LIB=$PWD/laser_amd/libsgclip_alt.so timeout -k 10 200 python tools/bench_gemm.py || exit 1
done
