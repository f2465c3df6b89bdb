set -uo pipefail
for L in m8_eo_late2 m7_eo_late2 m6_eo_late2 m5_eo_late2 m4_eo_late2 m3_eo_late2 m6_eo_late; do
  FDTDX_LIB=$PWD/build/$L/libfdtdx.so timeout -k 10 200 python -u -m pytest tests/test_gpu_parity_small.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -1 | sed "s/^/$L parity: /"
done
echo "== fdtd8"; bash tools/ab_libs.sh fdtd8 100 m8_eo_late m8_eo_late2 m8_eo_late m8_eo_late2
echo "== fdtd7"; bash tools/ab_libs.sh fdtd7 100 m7_eo_late2 m7_eo m7_eo_late2
echo "== fdtd6"; bash tools/ab_libs.sh fdtd6 100 m6_eo_late2 m6_eo_late m6_eo_late2
echo "== fdtd5"; bash tools/ab_libs.sh fdtd5 100 m5_eo_late2 
echo "== fdtd4"; bash tools/ab_libs.sh fdtd4 100 m4_eo_late2 
echo "== fdtd3"; bash tools/ab_libs.sh fdtd3 100 m3_eo_late2 
