S="nt:3072:768:0,nt:768:3072:0,nt:768:2304:0,nt:2304:768:0"
for pad in "0 0" "64 0" "0 64" "64 64" "128 128" "32 32"; do
  set -- $pad
  echo "== pad A=$1 B=$2 : full / probe"
  PAD_A=$1 PAD_B=$2 TNR_GEMM_BM=256 ONLY=$S python tools/gemm_bench.py | grep NT
  PAD_A=$1 PAD_B=$2 TNR_GEMM_BM=256 TNR_GEMM_PROBE=1 ONLY=$S python tools/gemm_bench.py | grep NT
done
